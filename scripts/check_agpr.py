#!/usr/bin/env python3
"""Build guard for kernels_fused.hip: the kernel parks live values in the AGPRs a[base..255] by hand (inline asm, printed
as `a[0x..]`); compiler-generated code (printed as `aN` / `a[N:M]`) must stay below `base`.
usage: check_agpr.py file.s kernel-substring base
exit codes: 0 = clear, 1 = register overlap (the guard TRIPPED: build.sh then builds without that kernel / instantiation),
            3 = no overlap, but the kernel spills registers to scratch memory (build.sh FAILS: a silent 2-3x slow-down; --spills-ok: report only),
            2 = the check itself could not run (no assembly file, kernel symbol not found after a rename / mangling change,
                no function end marker): build.sh FAILS on it -- a silently compiled-out kernel would be a large, quiet
                performance regression"""
import re
import sys

src, key, base = sys.argv[1], sys.argv[2], int(sys.argv[3])
try:
    lines = open(src).read().split("\n")
except OSError as e:
    print(f"check_agpr: ERROR -- cannot read {src}: {e}", file=sys.stderr)
    sys.exit(2)
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and key in l]
if not starts:
    print(f"check_agpr: ERROR -- kernel {key} not found in {src} (renamed? template arguments changed?)", file=sys.stderr)
    sys.exit(2)
worst = -1
spills = 0          # scratch accesses = register spills: a hand-scheduled kernel that spills runs at a fraction of its speed (round 6: a third
                    # parking pointer kept alive across the phases -> 420 - 540 scratch accesses, 148 instead of 58 us per iteration)
for st in starts:
    end = next((i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end")), None)   # (a kernel may hold several s_endpgm: early returns)
    if end is None:
        print(f"check_agpr: ERROR -- no function end marker behind {key} in {src}", file=sys.stderr)
        sys.exit(2)
    for l in lines[st:end]:
        code = l.split(";")[0]
        if re.search(r"\bscratch_(load|store)_", code):
            spills += 1
        for m in re.finditer(r"\ba(\d+)\b", code):
            worst = max(worst, int(m.group(1)))
        for m in re.finditer(r"\ba\[(\d+):(\d+)\]", code):
            worst = max(worst, int(m.group(2)))
print(f"check_agpr: {key}: highest compiler-allocated AGPR a{worst}, hand-managed range starts at a{base}" + (f", {spills} SCRATCH ACCESSES (spills)" if spills else ""))
if spills:
    print(f"check_agpr: WARNING -- {key} spills registers to scratch memory ({spills} accesses): expect a large slow-down", file=sys.stderr)
    if "--spills-ok" not in sys.argv:
        sys.exit(3)
if worst >= base:
    print(f"check_agpr: FAILED -- the compiler uses a{worst}, which overlaps the hand-managed AGPR stash of {key}", file=sys.stderr)
    sys.exit(1)
