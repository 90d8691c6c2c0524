#!/bin/bash
# deferred TF1-Adam: per-kernel picture of the tail (rocprofv3 kernel stats, one run per mode and shard)
OUT=$PWD/gpurun_out/r05l
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for mode in none rccl rccl-k_adam; do for n in 1 8; do
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${mode}_$n -o p -- python $GRAFT_REPO_ROOT/scripts/rccl_tail_profile.py $mode $n 2>&1 < /dev/null | grep "iteration"
  f=$(find $OUT/prof_${mode}_$n -name "*kernel_stats.csv" | head -1)
  echo "== $mode 1/$n"
  if [ -n "$f" ]; then head -7 "$f" | cut -d, -f1-8 | cut -c1-220; cp "$f" $OUT/stats_${mode}_$n.csv; fi
  rm -rf $OUT/prof_${mode}_$n
done; done
