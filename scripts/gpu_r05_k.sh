#!/bin/bash
OUT=$PWD/gpurun_out/r05o
mkdir -p $OUT
export TMPDIR=/tmp
for i in 1 2 3; do timeout 300 python scripts/quick_step.py 4000 2>&1 | tail -1; done | tee $OUT/quick.log
timeout 900 python -m pytest tests/test_gpu_headline.py -m gpu -q -x -k "config4 or 2d or poisson2d" > $OUT/pytest_b.log 2>&1; tail -3 $OUT/pytest_b.log
