#!/bin/bash
# round 5, visit C: the whole GPU suite on the pruned library, the wide kernels (forward-only launches alias one tile of the store),
# phase tables of k_iter_tile / k_iter_small / k_iter_fused from the timing build (build_alt/timing)
OUT=$PWD/gpurun_out/r05c
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q --durations=15 > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
python scripts/wide_bench.py 2>/dev/null | grep "^|" > $OUT/wide.md; cat $OUT/wide.md
T=$PWD/build_alt/timing/hp_vpinns_amd/libhpvpinn.so
for m in t1 t2 3 t5b 4; do echo "== fz_timing $m"; HPV_LIBRARY=$T python scripts/fz_timing.py $m 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | head -24; done > $OUT/phases.txt 2>&1; cat $OUT/phases.txt
