#!/bin/bash
# A/B on ONE box: the in-tree library against build_alt/<variant> (scripts/build_variant.sh), three alternating runs of config 4 each.
# Usage: bash scripts/ab_step.sh <variant> [iterations]
V=$1; N=${2:-4000}
for i in 1 2 3; do
  echo -n "in-tree: "; timeout 300 python scripts/quick_step.py $N 2>&1 | tail -1
  echo -n "$V: "; HPV_LIBRARY=$PWD/build_alt/$V/hp_vpinns_amd/libhpvpinn.so timeout 300 python scripts/quick_step.py $N 2>&1 | tail -1
done
