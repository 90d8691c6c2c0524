export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $REPO/gpurun_out/clk -o bench -- python $REPO/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-residual-roofline --no-extras > $REPO/gpurun_out/clk.log 2>&1
cd $REPO
python - <<'PY'
import csv, glob, collections
cc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/clk/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cc[r["Kernel_Name"].split("(")[0][:40]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k, v in cc.items():
    v = v[len(v)//2:]
    c = sum(a for a, b in v) / len(v); t = sum(b for a, b in v) / len(v)
    print(f"{k}: GRBM_GUI_ACTIVE {c:.0f} cycles / {t:.0f} ns = {c/t:.2f} GHz (per-dispatch avg over {len(v)})")
PY
