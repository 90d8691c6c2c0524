#!/bin/bash
# A/B of build variants (scripts/build_variant.sh <name> ...) against the product library, alternating bench.py runs on one box: ab_variant.sh <name>...
B="python bench.py --no-cpu-baseline --no-extras --no-pmc --no-residual-roofline"
for rep in 1 2 3; do
  for v in product "$@"; do
    if [ $v = product ]; then unset HPV_LIBRARY; else export HPV_LIBRARY=$PWD/build_alt/$v/hp_vpinns_amd/libhpvpinn.so; fi
    $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'it/s %.0f' % d['value'], 'us/iter %.3f' % (d['ms_per_step']*1e3), 'kernel us %.3f' % (d['roofline']['avg_ms']*1e3))"
  done
done
