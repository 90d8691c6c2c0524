#!/bin/bash
OUT=$PWD/gpurun_out/r05d
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --durations=15 > $OUT/pytest_gpu.log 2>&1; tail -8 $OUT/pytest_gpu.log
python scripts/exchange_overhead.py 2>/dev/null | grep "^|" > $OUT/exchange_overhead.md; cat $OUT/exchange_overhead.md
python scripts/advdiff_eps_probe.py 2>/dev/null | grep "^|" > $OUT/advdiff_eps.md; cat $OUT/advdiff_eps.md
