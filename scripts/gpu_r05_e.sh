#!/bin/bash
OUT=$PWD/gpurun_out/r05e
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --durations=12 > $OUT/pytest_gpu.log 2>&1; tail -8 $OUT/pytest_gpu.log
python scripts/multi_bench.py 2>/dev/null | grep "^|" > $OUT/multi.md; cat $OUT/multi.md
python scripts/exchange_overhead.py 2>/dev/null | grep "^|" > $OUT/exchange_overhead.md; cat $OUT/exchange_overhead.md
