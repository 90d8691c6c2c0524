#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of scripts/gpu_round_r0N.sh into a markdown summary."""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]


def short(name):
    return name.split("(")[0].replace("void ", "")[:60]


def stats(d):
    f = glob.glob(os.path.join(out, d, "*kernel_stats.csv"))
    if not f:
        return
    print(f"\n### rocprofv3 --kernel-trace --stats ({d})\n\n| kernel | calls | avg us | total % |\n|---|---|---|---|")
    for r in csv.DictReader(open(f[0])):
        print(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |")


def pmc(d):
    f = glob.glob(os.path.join(out, d, "*counter_collection.csv"))
    if not f:
        return {}
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return {k: {c: v / max(len(disp[k]), 1) for c, v in cs.items()} for k, cs in acc.items()}


stats("stats")
stats("stats_c5")
stats("stats_c5n")
stats("stats_w32")
stats("stats_advf0")
stats("stats_advf1")
stats("stats_b")
stats("stats_proj1")
stats("stats_proj0")
TITLES = (("", "bench (config 4), default path: whole-iteration kernel"),
          ("_c5", "config 5 (AdvDiff 8 x 80x80 points), default path: tall-element whole-iteration kernel"),
          ("_c5n", "config 5, HPV_FUSE=n: round 2's launches (forward -> activation store -> row-split projection -> reverse)"), ("_b", "bench (config 4), HPV_FUSE=b: forward + projection-fused reverse kernel"),
          ("_w32", "config-4 grid with [2,32,32,32,1]: the width-generic kernels k_fwd_wide / k_bwd_wide + k_project_wg"),
          ("_advf0", "AdvDiff var_form 0, 16x16 elements of 16x16 points: k_iter_fused<.., NT2 = 1, GEN>"),
          ("_advf1", "AdvDiff var_form 1, 16x16 elements of 16x16 points: k_iter_fused<.., GEN>"),
          ("_proj1", "projection kernel, residual + adjoint, 2^18-element batch"), ("_proj0", "projection kernel, residual only, 2^18-element batch"))
for tagp, title in TITLES:
    fe, wr = pmc("pmc_fetch" + tagp), pmc("pmc_write" + tagp)
    if fe or wr:
        print(f"\n### HBM traffic per launch from PMC ({title})\n")
        print("FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced "
              "reads (MI355X_MICROARCH.md section HBM), so the corrected read bytes are 2 x FETCH_SIZE x 1024.\n")
        print("| kernel | FETCH_SIZE KB | WRITE_SIZE KB | corrected HBM bytes (2*F+W)*1024 |\n|---|---|---|---|")
        for k in sorted(set(fe) | set(wr)):
            f_, w_ = fe.get(k, {}).get("FETCH_SIZE", 0.0), wr.get(k, {}).get("WRITE_SIZE", 0.0)
            if f_ + w_ > 1.0:
                print(f"| `{k}` | {f_:.0f} | {w_:.0f} | {(2 * f_ + w_) * 1024:.3e} |")
for tagp, title in (("", "default path"), ("_c5", "config 5, tall-element kernel"), ("_w32", "config-4 grid, 32-wide network"), ("_advf0", "AdvDiff var_form 0 on k_iter_fused<NT2=1,GEN>"), ("_advf1", "AdvDiff var_form 1 on k_iter_fused<GEN>"), ("2", "default path, second pass"), ("_b", "HPV_FUSE=b")):
    sq = pmc("pmc_sq" + tagp)
    if sq:
        print(f"\n### SQ counters per launch (bench, {title})\n")
        cols = sorted({c for v in sq.values() for c in v})
        print("| kernel | " + " | ".join(cols) + " |\n|---|" + "---|" * len(cols))
        for k, v in sq.items():
            if max(v.values()) > 1e5 and "rocclr" not in k:
                print(f"| `{k}` | " + " | ".join(f"{v.get(c, 0):.3g}" for c in cols) + " |")

# machine-readable HBM bytes per launch (corrected) for bench.py's roofline.traffic
import json
tj = {}


def corrected(fe, wr, k):
    return (2 * fe.get(k, {}).get("FETCH_SIZE", 0.0) + wr.get(k, {}).get("WRITE_SIZE", 0.0)) * 1024


fe, wr = pmc("pmc_fetch"), pmc("pmc_write")
for k in set(fe) | set(wr):
    if "k_iter_fused" in k: tj["iter_fused"] = corrected(fe, wr, k)
    if "k_finalize" in k: tj["finalize"] = corrected(fe, wr, k)
fe, wr = pmc("pmc_fetch_c5"), pmc("pmc_write_c5")
for k in set(fe) | set(wr):
    if "k_iter_tall" in k: tj["iter_tall"] = corrected(fe, wr, k)
for gp in ("advf0", "advf1"):
    fe, wr = pmc("pmc_fetch_" + gp), pmc("pmc_write_" + gp)
    for k in set(fe) | set(wr):
        if "k_iter_fused" in k: tj["iter_fused_" + gp] = corrected(fe, wr, k)
fe, wr = pmc("pmc_fetch_c5n"), pmc("pmc_write_c5n")
c5n = sum(corrected(fe, wr, k) for k in set(fe) | set(wr) if "rocclr" not in k)
if c5n: tj["config5_round2_launches_total"] = c5n
fe, wr = pmc("pmc_fetch_b"), pmc("pmc_write_b")
for k in set(fe) | set(wr):
    # (several instantiations may appear -- e.g. the forward kernel without activation store for the loss read-back:
    #  the training iteration's kernel is the one that moves the most bytes)
    if "k_bwd_mfma" in k: tj["mlp_bwd"] = max(corrected(fe, wr, k), tj.get("mlp_bwd", 0.0))
    if "k_fwd_mfma" in k: tj["mlp_fwd"] = max(corrected(fe, wr, k), tj.get("mlp_fwd", 0.0))
for tagp, key in (("_proj1", "project_scaled"), ("_proj0", "project_scaled_residual_only")):
    fe, wr = pmc("pmc_fetch" + tagp), pmc("pmc_write" + tagp)
    for k in set(fe) | set(wr):
        if "k_project" in k:
            tj[key] = corrected(fe, wr, k)
if tj:
    with open(os.path.join(out, "traffic.json"), "w") as f:
        json.dump(tj, f, indent=1)
