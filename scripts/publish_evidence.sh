#!/bin/bash
# Copy what a scripts/gpu_round_<round>.sh visit left under gpurun_out/<tag>/ into profiles/<round>_* (tracked) and regenerate DESIGN.md section 6.
# Usage: bash scripts/publish_evidence.sh <tag> [round = r05]
set -e
cd "$(dirname "$0")/.."
R=gpurun_out/$1
P=${2:-r06}
cp $R/summary.md profiles/${P}_rocprof_summary.md
cp $R/bench.json profiles/${P}_bench.json
cp $R/bench_driver_style.json profiles/${P}_bench_driver_style.json
cp $R/stats/bench_kernel_stats.csv profiles/${P}_bench_kernel_stats.csv
cp $R/stats_c5/c5_kernel_stats.csv profiles/${P}_config5_kernel_stats.csv
cp $R/stats_w32/w32_kernel_stats.csv profiles/${P}_wide32_kernel_stats.csv
for gp in advf0 advf1; do [ -f $R/stats_$gp/${gp}_kernel_stats.csv ] && cp $R/stats_$gp/${gp}_kernel_stats.csv profiles/${P}_${gp}_kernel_stats.csv; done
cp $R/stats_proj1/proj_kernel_stats.csv profiles/${P}_proj_kernel_stats.csv
cp $R/stats_proj0/proj_kernel_stats.csv profiles/${P}_proj_residual_only_kernel_stats.csv
python3 - "$R" "$P" <<'PY'
import json, sys
a = json.load(open(sys.argv[1] + "/traffic.json")); b = json.load(open("profiles/traffic.json"))
b.update(a)
b["_measured_at"] = "round %s (%s: scripts/gpu_round_%s.sh, separate --pmc passes, FETCH_SIZE x 2 correction; profiles/%s_rocprof_summary.md)" % (sys.argv[2][1:].lstrip("0"), sys.argv[1], sys.argv[2], sys.argv[2])
json.dump(b, open("profiles/traffic.json", "w"), indent=1, sort_keys=True)
PY
awk '/^### other element shapes/{f=1} /^### shards of config 4/{f=0} f' $R/summary.md > profiles/${P}_element_shapes.md
awk '/^### networks of other widths/{f=1} /^### other element shapes/{f=0} f' $R/summary.md > profiles/${P}_wide_networks.md
python3 scripts/design_numbers.py $P > /dev/null
echo "profiles/${P}_* <- $R"
