#!/bin/bash
# One GPU-box visit: parity tests, the bench line, rocprofv3 kernel stats and PMC passes (counters in their own runs,
# never combined with a trace domain other than --kernel-trace).
# Usage (from the repo root on the GPU box): bash scripts/gpu_round.sh <tag> [skip-tests]
TAG=${1:-r02}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ -z "$2" ]; then python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log; fi
python bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json; cat $OUT/bench.json
REPO=$PWD
cd /tmp
BENCH="python $REPO/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-residual-roofline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $BENCH > $OUT/stats.log 2>&1
for adj in 1 0; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_proj$adj -o proj -- python $REPO/scripts/proj_bench.py 262144 5 $adj > $OUT/stats_proj$adj.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_proj$adj -o proj -- python $REPO/scripts/proj_bench.py 262144 3 $adj > $OUT/pmc_fetch_proj$adj.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_proj$adj -o proj -- python $REPO/scripts/proj_bench.py 262144 3 $adj > $OUT/pmc_write_proj$adj.log 2>&1
done
# PMC passes over the bench (FETCH_SIZE and WRITE_SIZE do not fit one pass)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -o bench -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VALU --output-format csv -d $OUT/pmc_sq2 -o bench -- $BENCH > $OUT/pmc_sq2.log 2>&1
# the same passes over the two-kernel path (HPV_FUSE=b: forward + projection-fused reverse), the ablation of the whole-iteration kernel
HPV_FUSE=b rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b -o bench -- $BENCH > $OUT/stats_b.log 2>&1
HPV_FUSE=b rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_b -o bench -- $BENCH > $OUT/pmc_fetch_b.log 2>&1
HPV_FUSE=b rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_b -o bench -- $BENCH > $OUT/pmc_write_b.log 2>&1
HPV_FUSE=b rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq_b -o bench -- $BENCH > $OUT/pmc_sq_b.log 2>&1
cd $REPO
python scripts/summarize_profiles.py $OUT > $OUT/summary.md 2>&1
echo >> $OUT/summary.md; echo "### iterations/sec of the five BASELINE configs (1 GPU)" >> $OUT/summary.md; echo >> $OUT/summary.md
python scripts/config_bench.py 2>/dev/null | grep "^|" >> $OUT/summary.md
echo >> $OUT/summary.md; echo "### config 4 through the three launch structures (HPV_FUSE)" >> $OUT/summary.md; echo >> $OUT/summary.md
for f in i b n; do echo "HPV_FUSE=$f: $(HPV_FUSE=$f python scripts/quick_step.py 2000 2>/dev/null | tail -1)" >> $OUT/summary.md; done
cat $OUT/summary.md
# keep only the small summaries (the traces are large)
find $OUT -name "*kernel_trace.csv" -size +2M -delete
