#!/bin/bash
# cheaper tanh: accuracy, parity on the headline configs, iteration times
OUT=$PWD/gpurun_out/r05m
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tanh or sincos or loss_grad or traj" > $OUT/pytest_a.log 2>&1; tail -3 $OUT/pytest_a.log
timeout 900 python -m pytest tests/test_gpu_headline.py -m gpu -q -x > $OUT/pytest_b.log 2>&1; tail -3 $OUT/pytest_b.log
for i in 1 2; do timeout 300 python scripts/quick_step.py 4000 2>&1 | tail -1; done | tee $OUT/quick.log
timeout 600 python bench.py --no-pmc --no-extras --cpu-iters 1 2>/dev/null | tail -1 > $OUT/bench.json; python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['roofline'])"
timeout 600 python scripts/config_bench.py 2>/dev/null | tail -12 | tee $OUT/configs.log
