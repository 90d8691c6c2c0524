#!/bin/bash
OUT=$PWD/gpurun_out/r05y
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_deferred.py tests/test_gpu_bench.py tests/test_gpu_headline.py -m gpu -q -x > $OUT/pytest_sel.log 2>&1; tail -4 $OUT/pytest_sel.log
bash scripts/gpu_round_r05.sh r05y > $OUT/visit.log 2>&1; head -8 $OUT/visit.log | cut -c1-250; tail -6 $OUT/visit.log
