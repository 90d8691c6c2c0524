#!/usr/bin/env python3
"""Instruction mix of one kernel in a device assembly file (hipcc -S --cuda-device-only): isa_stats.py file.s <mangled substring>"""
import re
import sys
from collections import Counter

src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and key in l]
for st in starts:
    name = lines[st].split(":")[0]
    end = next(i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end"))   # (a kernel may hold several s_endpgm: early returns)
    body = [l.strip() for l in lines[st + 1:end + 1] if l.startswith("\t") and not l.strip().startswith((";", "."))]
    c = Counter()
    for l in body:
        op = l.split()[0]
        if op.startswith("v_mfma_f64_16x16"): c["mfma16"] += 1
        elif op.startswith("v_mfma_f64_4x4"): c["mfma4"] += 1
        elif op.startswith("v_") and "f64" in op: c["valu_f64"] += 1
        elif op.startswith("v_accvgpr"): c["accvgpr_mov"] += 1
        elif op.startswith("v_"): c["valu_other"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith(("global_load", "flat_load", "buffer_load")): c["gload"] += 1
        elif op.startswith(("global_store", "flat_store", "buffer_store")): c["gstore"] += 1
        elif op.startswith("scratch_"): c["scratch"] += 1
        elif op.startswith("s_waitcnt"): c["waitcnt"] += 1
        elif op.startswith("s_nop"): c["nop"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        else: c["other"] += 1
    meta = {}
    for l in lines[end:end + 80]:
        m = re.match(r"\s*; (NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|LDSByteSize|codeLenInByte): (\d+)", l)
        if m: meta[m.group(1)] = int(m.group(2))
    print(name)
    print("   ", dict(c), "total", len(body))
    print("   ", meta)
