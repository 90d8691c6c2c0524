#!/bin/bash
# One GPU-box visit: parity tests, the bench line, rocprofv3 kernel stats and PMC passes.
# Usage (from the repo root on the GPU box): bash scripts/gpu_round.sh <tag>
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
python bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json; cat $OUT/bench.json
REPO=$PWD
cd /tmp
BENCH="python $REPO/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-residual-roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $BENCH > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_proj -o proj -- python $REPO/scripts/proj_bench.py 262144 5 > $OUT/stats_proj.log 2>&1
# PMC passes (counters only, separate runs; FETCH_SIZE and WRITE_SIZE do not fit one pass)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -o bench -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_proj -o proj -- python $REPO/scripts/proj_bench.py 262144 3 > $OUT/pmc_fetch_proj.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_proj -o proj -- python $REPO/scripts/proj_bench.py 262144 3 > $OUT/pmc_write_proj.log 2>&1
cd $REPO
python scripts/summarize_profiles.py $OUT > $OUT/summary.md 2>&1
echo >> $OUT/summary.md; echo "### iterations/sec of the five BASELINE configs (1 GPU)" >> $OUT/summary.md; echo >> $OUT/summary.md
python scripts/config_bench.py 2>/dev/null | grep "^|" >> $OUT/summary.md
echo >> $OUT/summary.md; echo "### fp64 ubench ceilings on this box" >> $OUT/summary.md; echo >> $OUT/summary.md
echo '```' >> $OUT/summary.md; ./scripts/mfma_f64_peak.bin 2>/dev/null | grep -E "waves/SIMD=2 nacc=8|v_fma_f64 waves/SIMD=2" >> $OUT/summary.md; ./scripts/mix_f64.bin 2>/dev/null >> $OUT/summary.md; echo '```' >> $OUT/summary.md
cat $OUT/summary.md
# keep only the small summaries (the traces are large)
find $OUT -name "*kernel_trace.csv" -size +2M -delete
