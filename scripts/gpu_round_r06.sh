#!/bin/bash
# Round-6 GPU-box visit (round 5's + the general forms of k_iter_fused and the ragged grids): the bench line (default and driver-style), rocprofv3 kernel stats and PMC passes (counters in their own
# runs, never combined with a trace domain other than --kernel-trace) for config 4 (k_iter_fused), config 5 (k_iter_tall), the
# config-4 grid with a 32-wide network (k_fwd_wide / k_bwd_wide) and the stand-alone residual kernel; per-config timings, the
# wide-network and element-shape tables, shard timings.
# Usage (from the repo root on the GPU box): bash scripts/gpu_round_r06.sh <tag> [tests]
TAG=${1:-r06}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
if [ -n "$2" ]; then python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log; fi
python bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json; cut -c1-300 $OUT/bench.json
python bench.py --steps 20 --warmup 5 2> $OUT/bench_driver_style.err | tail -1 > $OUT/bench_driver_style.json; cut -c1-200 $OUT/bench_driver_style.json
cd /tmp
BENCH="python $REPO/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-residual-roofline --no-extras --no-pmc"
C5="python $REPO/scripts/cfg5_quick.py 200"
W32="python $REPO/scripts/wide_step.py 32 200"
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $BENCH > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $OUT/pmc_sq -o bench -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -o c5 -- $C5 > $OUT/stats_c5.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_c5 -o c5 -- $C5 > $OUT/pmc_fetch_c5.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_c5 -o c5 -- $C5 > $OUT/pmc_write_c5.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $OUT/pmc_sq_c5 -o c5 -- $C5 > $OUT/pmc_sq_c5.log 2>&1
# the general forms on the whole-iteration kernel: AdvDiff var_form 0 (four channels) and var_form 1 on 16x16 elements of 16x16 points
for gp in advf0 advf1; do
  G="python $REPO/scripts/gen_step.py $gp 16 200"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$gp -o $gp -- $G > $OUT/stats_$gp.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$gp -o $gp -- $G > $OUT/pmc_fetch_$gp.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$gp -o $gp -- $G > $OUT/pmc_write_$gp.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $OUT/pmc_sq_$gp -o $gp -- $G > $OUT/pmc_sq_$gp.log 2>&1
done
# the width-generic kernels: config-4 grid, [2,32,32,32,1]
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_w32 -o w32 -- $W32 > $OUT/stats_w32.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_w32 -o w32 -- $W32 > $OUT/pmc_fetch_w32.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_w32 -o w32 -- $W32 > $OUT/pmc_write_w32.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $OUT/pmc_sq_w32 -o w32 -- $W32 > $OUT/pmc_sq_w32.log 2>&1
# the stand-alone residual kernel on the scaled batch (the HBM-roofline measurement of SURVEY.md 8d)
for adj in 1 0; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_proj$adj -o proj -- python $REPO/scripts/proj_bench.py 262144 5 $adj > $OUT/stats_proj$adj.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_proj$adj -o proj -- python $REPO/scripts/proj_bench.py 262144 3 $adj > $OUT/pmc_fetch_proj$adj.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_proj$adj -o proj -- python $REPO/scripts/proj_bench.py 262144 3 $adj > $OUT/pmc_write_proj$adj.log 2>&1
done
cd $REPO
python scripts/summarize_profiles.py $OUT > $OUT/summary.md 2>&1
echo >> $OUT/summary.md; echo "### iterations/sec of the five BASELINE configs (1 GPU)" >> $OUT/summary.md; echo >> $OUT/summary.md
python scripts/config_bench.py 2>/dev/null | grep "^|" >> $OUT/summary.md
echo >> $OUT/summary.md; echo "### networks of other widths / depths (scripts/wide_bench.py)" >> $OUT/summary.md; echo >> $OUT/summary.md
python scripts/wide_bench.py 2>/dev/null | grep "^|" >> $OUT/summary.md
echo >> $OUT/summary.md; echo "### other element shapes / variational forms: default dispatch vs the generic element-resident kernel (HPV_FUSE=e) vs the separate launches (scripts/elem_bench.py)" >> $OUT/summary.md; echo >> $OUT/summary.md
python scripts/elem_bench.py 2>/dev/null | grep "^|" >> $OUT/summary.md
echo >> $OUT/summary.md; echo "### grids larger than the chip with a ragged last round (scripts/ragged_bench.py)" >> $OUT/summary.md; echo >> $OUT/summary.md
python scripts/ragged_bench.py 2>/dev/null | grep "^|" >> $OUT/summary.md
echo >> $OUT/summary.md; echo "### shards of config 4 one GPU of N owns (SPLIT mode), no communication" >> $OUT/summary.md; echo >> $OUT/summary.md
python scripts/shard_bench.py 2>/dev/null | sed 's/^/    /' >> $OUT/summary.md
echo >> $OUT/summary.md; echo "### iteration tail of the multi-GPU launches on one GPU (scripts/exchange_overhead.py)" >> $OUT/summary.md; echo >> $OUT/summary.md
python scripts/exchange_overhead.py 2>/dev/null | grep "^|" | sed 's/^/    /' >> $OUT/summary.md
echo >> $OUT/summary.md; echo "### the scaled batch: what one rank of N owns, single-GPU and 1-rank-RCCL tails (scripts/large_shards.py)" >> $OUT/summary.md; echo >> $OUT/summary.md
python scripts/large_shards.py 2>/dev/null | grep "^|" >> $OUT/summary.md
cat $OUT/summary.md
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
find $OUT -name "*.db" -delete
