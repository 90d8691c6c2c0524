#!/bin/bash
OUT=$PWD/gpurun_out/r05u
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "sincos or tanh" > $OUT/pytest_a.log 2>&1; tail -3 $OUT/pytest_a.log
for i in 1 2; do
  echo "== trimmed sincos (in-tree)"; timeout 300 python scripts/config_bench.py 2>/dev/null | grep "^| [12]:"
  echo "== round-4 sincos"; HPV_LIBRARY=$PWD/build_alt/sincosR4/hp_vpinns_amd/libhpvpinn.so timeout 300 python scripts/config_bench.py 2>/dev/null | grep "^| [12]:"
done
timeout 900 python -m pytest tests/test_gpu_headline.py tests/test_gpu_parity.py -m gpu -q -x -k "1d or poisson1d or config1 or config2 or sin" > $OUT/pytest_b.log 2>&1; tail -3 $OUT/pytest_b.log
