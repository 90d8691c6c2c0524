#!/usr/bin/env python3
"""Build guard for kernels_fused.hip: the kernel parks live values in the AGPRs a[base..255] by hand (inline asm, printed
as `a[0x..]`); compiler-generated code (printed as `aN` / `a[N:M]`) must stay below `base`.
usage: check_agpr.py file.s kernel-substring base"""
import re
import sys

src, key, base = sys.argv[1], sys.argv[2], int(sys.argv[3])
lines = open(src).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and key in l]
assert starts, f"kernel {key} not found in {src}"
worst = -1
for st in starts:
    end = next(i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end"))   # (a kernel may hold several s_endpgm: early returns)
    for l in lines[st:end]:
        code = l.split(";")[0]
        for m in re.finditer(r"\ba(\d+)\b", code):
            worst = max(worst, int(m.group(1)))
        for m in re.finditer(r"\ba\[(\d+):(\d+)\]", code):
            worst = max(worst, int(m.group(2)))
print(f"check_agpr: {key}: highest compiler-allocated AGPR a{worst}, hand-managed range starts at a{base}")
if worst >= base:
    sys.exit(f"check_agpr: FAILED -- the compiler uses a{worst}, which overlaps the hand-managed AGPR stash of {key}")
