#!/bin/bash
OUT=$PWD/gpurun_out/r05t
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "tanh" > $OUT/pytest_a.log 2>&1; tail -2 $OUT/pytest_a.log
for i in 1 2 3; do timeout 300 python scripts/quick_step.py 4000 2>&1 | tail -1; done | tee $OUT/quick.log
