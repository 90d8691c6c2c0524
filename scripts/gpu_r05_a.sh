#!/bin/bash
# round 5, visit A: the suite with per-test durations, the bench line both ways, the scaled-batch shards, the 1-D rule sweep
OUT=$PWD/gpurun_out/r05a
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q --durations=80 > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
python bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json; cut -c1-300 $OUT/bench.json
( time python bench.py --steps 20 --warmup 5 2> $OUT/bench_driver_style.err | tail -1 > $OUT/bench_driver_style.json ) 2> $OUT/bench_driver_style.time; cut -c1-200 $OUT/bench_driver_style.json; cat $OUT/bench_driver_style.time
python scripts/large_shards.py > $OUT/large_shards.md 2>&1; cat $OUT/large_shards.md
python scripts/exchange_overhead.py > $OUT/exchange_overhead.md 2>&1; cat $OUT/exchange_overhead.md
python scripts/rule1d_sweep.py 600 > $OUT/rule1d_sweep.md 2>&1; cat $OUT/rule1d_sweep.md
