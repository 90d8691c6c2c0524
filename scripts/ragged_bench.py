#!/usr/bin/env python3
"""Grids larger than the chip with a ragged last round (verdict round 5, item 5): the default dispatch -- full rounds with one
workgroup per element + the tail's elements shared by 2 - 8 workgroups each in a second launch -- against one workgroup per element
on every round (HPV_FUSE=i: ceil(n / CUs) rounds) and the separate launches (HPV_FUSE=n).  Reference point: a full round (256
elements) and the ideal n / 256 rounds."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd.drivers import advdiff, poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402


def run(build, fuse):
    if fuse:
        os.environ["HPV_FUSE"] = fuse
    try:
        m = build()
    finally:
        os.environ.pop("HPV_FUSE", None)
    m.h.step(20, False)
    n = 200
    t0 = time.perf_counter()
    m.h.step(n, False)
    return (time.perf_counter() - t0) / n * 1e6, m.h.kernel_variant()


print("| problem | elements | default: us / iteration | = rounds of the 256-element time | one workgroup per element (HPV_FUSE=i) | separate launches | kernels |\n|---|---|---|---|---|---|---|")
for (name, q, nt, L, vf, grids) in [("Poisson-2D var_form 1", 20, 10, [2, 20, 20, 20, 1], 1, [(16, 16), (17, 17), (20, 20), (40, 40), (36, 36)]),
                                    ("Poisson-2D var_form 1", 16, 8, [2, 20, 20, 20, 1], 1, [(16, 16), (17, 17), (24, 23)]),
                                    ("Poisson-2D var_form 0", 16, 8, [2, 20, 20, 20, 1], 0, [(16, 16), (25, 22)])]:
    base = None
    for (nex, ney) in grids:
        s = poisson2d.setup(N_el_x=nex, N_el_y=ney, N_test_x=nt, N_test_y=nt, N_quad=q, with_test_grid=False, assemble="device")
        build = lambda: poisson2d.build_model(s, L, var_form=vf, init_params=xavier_init(L, 1234))
        a, va = run(build, None)
        b, _ = run(build, "i")
        c, _ = run(build, "n")
        if base is None:
            base = a
        print(f"| {name}, {q}x{q} points, {L} | {nex * ney} | {a:.1f} | {a / base:.2f} (ideal {nex * ney / 256:.2f}) | {b:.1f} | {c:.1f} | `{va}` |")
