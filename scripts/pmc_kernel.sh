#!/bin/bash
# SQ counter passes (stall breakdown) + kernel stats over scripts/quick_step.py.  Usage: bash scripts/pmc_kernel.sh <tag>
OUT=$PWD/gpurun_out/${1:-pmck}
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
python $REPO/scripts/quick_step.py 2000
cd /tmp
CMD="python $REPO/scripts/quick_step.py 96"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o q -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/a -o q -- $CMD > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VALU --output-format csv -d $OUT/b -o q -- $CMD > $OUT/b.log 2>&1
cd $REPO
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(f"{out}/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 1: print("stats", r["Name"][:70], r["Calls"], "avg us %.2f" % (float(r["AverageNs"]) / 1e3))
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
        if v.get("SQ_WAVE_CYCLES", 1e9) > 1e5 or d == "b":
            print(d, k, cnt[k], {c: f"{x / cnt[k]:.4g}" for c, x in sorted(v.items())})
PY
find $OUT -name "*kernel_trace.csv" -size +2M -delete
