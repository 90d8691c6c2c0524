#!/bin/bash
# Copy what a scripts/gpu_round_r04.sh visit left under gpurun_out/<tag>/ into profiles/r04_* (tracked) and regenerate DESIGN.md section 6.
# Usage: bash scripts/publish_evidence.sh <tag>
set -e
cd "$(dirname "$0")/.."
R=gpurun_out/$1
cp $R/summary.md profiles/r04_rocprof_summary.md
cp $R/bench.json profiles/r04_bench.json
cp $R/bench_driver_style.json profiles/r04_bench_driver_style.json
cp $R/stats/bench_kernel_stats.csv profiles/r04_bench_kernel_stats.csv
cp $R/stats_c5/c5_kernel_stats.csv profiles/r04_config5_kernel_stats.csv
cp $R/stats_w32/w32_kernel_stats.csv profiles/r04_wide32_kernel_stats.csv
cp $R/stats_proj1/proj_kernel_stats.csv profiles/r04_proj_kernel_stats.csv
cp $R/stats_proj0/proj_kernel_stats.csv profiles/r04_proj_residual_only_kernel_stats.csv
python3 - "$R" <<'PY'
import json, sys
a = json.load(open(sys.argv[1] + "/traffic.json")); b = json.load(open("profiles/traffic.json"))
b.update(a); json.dump(b, open("profiles/traffic.json", "w"), indent=1, sort_keys=True)
PY
awk '/^### other element shapes/{f=1} /^### shards of config 4/{f=0} f' $R/summary.md > profiles/r04_element_shapes.md
awk '/^### networks of other widths/{f=1} /^### other element shapes/{f=0} f' $R/summary.md > profiles/r04_wide_networks.md
python3 scripts/design_numbers.py r04 > /dev/null
echo "profiles/r04_* <- $R"
