#!/bin/bash
# A/B of a build_alt variant against the in-tree library: config 4 per-iteration time + parity of the variant (headline + elem tests)
V=${1:-epilds}
OUT=$PWD/gpurun_out/r05ab_$V
mkdir -p $OUT
export TMPDIR=/tmp
bash scripts/ab_step.sh 4000 default $V default $V | tee $OUT/ab.txt
HPV_LIBRARY=$PWD/build_alt/$V/hp_vpinns_amd/libhpvpinn.so python -m pytest tests/test_gpu_headline.py tests/test_gpu_elem.py -m gpu -q -x -k "not saved_values" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
