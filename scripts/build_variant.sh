#!/bin/bash
# A/B builds of libhpvpinn.so: scripts/build_variant.sh <name> "<extra hipcc flags>" -> build_alt/<name>/hp_vpinns_amd/libhpvpinn.so
# (select it at run time with HPV_LIBRARY=<that path>; build_alt/ is git-ignored but travels to the GPU box)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
FUSED_ONLY=0
if [ "$1" = "--fused-only" ]; then FUSED_ONLY=1; shift; fi   # the flags go to kernels_fused.hip only; every other object is reused
D=$ROOT/build_alt/$NAME
mkdir -p $D/hp_vpinns_amd/csrc $D/include $D/scripts
cp -u $ROOT/hp_vpinns_amd/csrc/*.hip $ROOT/hp_vpinns_amd/csrc/*.h $ROOT/hp_vpinns_amd/csrc/build.sh $D/hp_vpinns_amd/csrc/
cp -u $ROOT/include/hpvpinn.h $D/include/
cp -u $ROOT/scripts/check_agpr.py $D/scripts/
if [ $FUSED_ONLY = 1 ]; then
  cp -p $ROOT/hp_vpinns_amd/csrc/*.o $ROOT/hp_vpinns_amd/csrc/.flags $D/hp_vpinns_amd/csrc/
  touch $D/hp_vpinns_amd/csrc/*.o; rm -f $D/hp_vpinns_amd/csrc/kernels_fused.o
  HPV_FUSED_EXTRA="$*" bash $D/hp_vpinns_amd/csrc/build.sh
else
  HPV_EXTRA_FLAGS="$*" bash $D/hp_vpinns_amd/csrc/build.sh
fi
