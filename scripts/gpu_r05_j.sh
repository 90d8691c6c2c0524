#!/bin/bash
OUT=$PWD/gpurun_out/r05n
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "tanh" > $OUT/pytest_a.log 2>&1; tail -8 $OUT/pytest_a.log
for i in 1 2; do timeout 300 python scripts/quick_step.py 4000 2>&1 | tail -1; done | tee $OUT/quick.log
timeout 600 python scripts/config_bench.py 2>/dev/null | tail -7 | tee $OUT/configs.log
