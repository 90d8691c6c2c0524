#!/bin/bash
OUT=$PWD/gpurun_out/r05q
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_wide.py -m gpu -q --durations=5 > $OUT/pytest_sel.log 2>&1; tail -9 $OUT/pytest_sel.log
timeout 600 python scripts/soak_deferred.py 400000 2>/dev/null | grep "^|" | tee $OUT/soak_deferred.md
