#!/bin/bash
# deferred TF1-Adam (two launches + one collective): parity + cost
OUT=$PWD/gpurun_out/r05k
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_deferred.py -m gpu -q > $OUT/pytest_deferred.log 2>&1; tail -15 $OUT/pytest_deferred.log
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rccl_faults.py tests/test_gpu_split.py tests/test_gpu_headline.py -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -6 $OUT/pytest_gpu.log
timeout 600 python scripts/exchange_overhead.py 2>/dev/null | grep "^|" > $OUT/exchange.md; cat $OUT/exchange.md
timeout 300 python scripts/quick_step.py 4000 2>&1 | tail -3 | tee $OUT/quick.log
