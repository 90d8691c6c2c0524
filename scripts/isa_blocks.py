#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a device assembly file: isa_blocks.py file.s <mangled substring> [min_instrs]
Lists blocks in program order with their label, size, MFMA / fp64 VALU / other VALU / AGPR-move / LDS / wait counts and the
branch that ends them (loops show as backward branches)."""
import re
import sys
from collections import Counter

src, key = sys.argv[1], sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lines = open(src).read().split("\n")
st = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and key in l)
end = next(i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end"))   # (a kernel may hold several s_endpgm: early returns)


def kind(op):
    if op.startswith("v_mfma_f64_16x16"): return "mfma16"
    if op.startswith("v_mfma_f64_4x4"): return "mfma4"
    if op.startswith("v_accvgpr"): return "acc_mov"
    if op.startswith("v_mov") or op.startswith("v_pk_mov"): return "v_mov"
    if op.startswith("v_") and "f64" in op: return "valu_f64"
    if op.startswith("v_"): return "valu_other"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    return "other"


blocks, cur, label = [], [], "entry"
pos = {}
for i in range(st + 1, end + 1):
    l = lines[i]
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append((label, cur)); label, cur = m.group(1), []
        continue
    if l.startswith("\t") and not l.strip().startswith((";", ".")):
        cur.append((i + 1, l.strip()))
blocks.append((label, cur))
order = {b[0]: k for k, b in enumerate(blocks)}
tot = Counter()
for k, (lab, ins) in enumerate(blocks):
    c = Counter(kind(x[1].split()[0]) for x in ins)
    tot.update(c)
    br = [x[1] for x in ins if x[1].startswith(("s_cbranch", "s_branch"))]
    back = [b for b in br if b.split()[-1] in order and order[b.split()[-1]] <= k]
    if len(ins) >= minsz or back:
        cyc = c["mfma16"] * 64 + c["mfma4"] * 16 + c["valu_f64"] * 4 + (c["valu_other"] + c["acc_mov"] + c["v_mov"]) * 2
        print(f"{lab:12s} line {ins[0][0] if ins else 0:6d} n={len(ins):5d} mfma16={c['mfma16']:3d} mfma4={c['mfma4']:3d} f64={c['valu_f64']:4d} "
              f"valu={c['valu_other']:4d} vmov={c['v_mov']:4d} acc={c['acc_mov']:4d} lds={c['lds']:3d} vmem={c['vmem']:3d} wait={c['waitcnt']:3d} "
              f"nop={c['nop']:3d} salu={c['salu']:3d}  ~issue={cyc:6d}  {'BACK:' + ','.join(b.split()[-1] for b in back) if back else ''}")
print("total", dict(tot))
