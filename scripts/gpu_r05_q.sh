#!/bin/bash
# A/B on one box: the tanh tail with v_cmp_class + one select (default) against the integer test + two selects (build_alt/tailA)
for i in 1 2 3; do
  echo -n "class+1 select: "; timeout 300 python scripts/quick_step.py 4000 2>&1 | tail -1
  echo -n "int test+2 selects: "; HPV_LIBRARY=$PWD/build_alt/tailA/hp_vpinns_amd/libhpvpinn.so timeout 300 python scripts/quick_step.py 4000 2>&1 | tail -1
done
