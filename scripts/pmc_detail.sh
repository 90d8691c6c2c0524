#!/bin/bash
# Detailed SQ counter passes over the bench (stall breakdown of the forward / reverse kernels).
OUT=$PWD/gpurun_out/${1:-pmc_detail}
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
BENCH="python $REPO/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-residual-roofline"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/a -o bench -- $BENCH > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INST_CYCLES_VALU SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/b -o bench -- $BENCH > $OUT/b.log 2>&1
cd $REPO
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in ("a", "b"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key); cnt[k] += 1
    for k, v in agg.items():
        print(d, k, cnt[k], {c: f"{x / cnt[k]:.3g}" for c, x in sorted(v.items())})
PY
