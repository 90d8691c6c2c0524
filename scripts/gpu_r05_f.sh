#!/bin/bash
OUT=$PWD/gpurun_out/r05i
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_elem.py tests/test_gpu_headline.py -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
python scripts/multi_bench.py 2>/dev/null | grep "^|" > $OUT/multi.md; cat $OUT/multi.md
