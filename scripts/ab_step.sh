#!/bin/bash
# A/B of library builds on the GPU box: scripts/ab_step.sh <iterations> <name>...   (name "default" = the in-tree library)
N=$1; shift
for v in "$@"; do
  if [ "$v" = default ]; then unset HPV_LIBRARY; else export HPV_LIBRARY=$PWD/build_alt/$v/hp_vpinns_amd/libhpvpinn.so; fi
  for rep in 1 2; do echo "$v: $(python scripts/quick_step.py $N 2>/dev/null | tail -1)"; done
done
