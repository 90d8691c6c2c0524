#!/bin/bash
# round 5, visit B: the pruned library -- the whole GPU suite, the bench line (with the in-run traffic measurement), the wide
# kernels after their save flag became a run-time value, the five configs
OUT=$PWD/gpurun_out/r05b
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q --durations=25 > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
( time python bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json ) 2> $OUT/bench.time; cut -c1-300 $OUT/bench.json; cat $OUT/bench.time
python -c "
import json; d=json.load(open('$OUT/bench.json')); print(json.dumps(d['roofline'], indent=1)); print(d.get('scaled_strong_64x64')); print(d.get('weak_scaling_probe'))"
python scripts/wide_bench.py 2>/dev/null | grep "^|" > $OUT/wide.md; cat $OUT/wide.md
python scripts/config_bench.py 2>/dev/null | grep "^|" > $OUT/configs.md; cat $OUT/configs.md
