#!/bin/bash
OUT=$PWD/gpurun_out/r05p
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_deferred.py tests/test_gpu_split.py tests/test_gpu_rccl_faults.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "zero_copy" > $OUT/pytest2.log 2>&1; tail -3 $OUT/pytest2.log
