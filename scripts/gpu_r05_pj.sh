#!/bin/bash
OUT=$PWD/gpurun_out/r05pj
mkdir -p $OUT
for v in default pjw2 pjw3 default pjw2 pjw3; do
  if [ "$v" = default ]; then unset HPV_LIBRARY; else export HPV_LIBRARY=$PWD/build_alt/$v/hp_vpinns_amd/libhpvpinn.so; fi
  for adj in 0 1; do echo "$v: $(python scripts/proj_bench.py 262144 10 $adj 2>/dev/null | tail -1)"; done
done | tee $OUT/pj.txt
