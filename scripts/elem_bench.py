#!/usr/bin/env python3
"""Iteration time of element shapes / variational forms the hand-tuned whole-iteration kernels do not take: the generic
element-resident kernel (csrc/kernels_elem.hip, one launch + finalize) against the separate launches (HPV_FUSE=n: forward ->
activation store -> projection -> reverse -> finalize).  Prints a markdown table (profiles/r04_element_shapes.md)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd.drivers import advdiff, poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402


def timeit(m, n=400):
    m._step(50, False)
    m.h.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        m._step(n, False)
        m.h.sync()
        best = min(best, (time.perf_counter() - t0) / n)
    return 1e6 * best


def both(build):
    m = build()
    us, v, ps = timeit(m), m.h.kernel_variant(), m.h.pass_structure()
    del m
    prev = os.environ.get("HPV_FUSE")
    os.environ["HPV_FUSE"] = "n"
    try:
        m = build()
        us2, v2 = timeit(m), m.h.kernel_variant()
    finally:
        if prev is None:
            del os.environ["HPV_FUSE"]
        else:
            os.environ["HPV_FUSE"] = prev
    return us, v, ps, us2, v2


rows = []
for (q, nt, ne, L, vf) in [(16, 8, 16, [2, 20, 20, 20, 1], 1), (16, 8, 16, [2, 20, 20, 20, 1], 0), (16, 8, 16, [2, 20, 20, 1], 1),
                           (12, 6, 16, [2, 20, 20, 20, 1], 1), (20, 10, 16, [2, 20, 20, 20, 1], 0), (20, 10, 16, [2, 20, 20, 20, 1], 2),
                           (16, 8, 16, [2, 32, 32, 32, 1], 1), (16, 8, 32, [2, 20, 20, 20, 1], 1)]:
    s = poisson2d.setup(N_el_x=ne, N_el_y=ne, N_test_x=nt, N_test_y=nt, N_quad=q, with_test_grid=False)
    r = both(lambda: poisson2d.build_model(s, L, var_form=vf, init_params=xavier_init(L, 1234)))
    rows.append((f"Poisson-2D var_form {vf}, {ne}x{ne} elements, {q}x{q} points, {nt}x{nt} test fcns, {L}", ne * ne * q * q) + r)
L = [2, 20, 20, 20, 1]
s = advdiff.setup(N_el_x=16, N_el_t=16, N_test_x=8, N_test_t=8, N_quad=16, with_test_grid=False)
for vf in (0, 1):
    r = both(lambda: advdiff.build_model(s, L, var_form=vf, init_params=xavier_init(L, 1234, extra=[1.0])))
    rows.append((f"AdvDiff var_form {vf} (trainable epsilon), 16x16 elements, 16x16 points, 8x8 test fcns, {L}", 256 * 256) + r)
print("| problem | points | us / iteration | separate launches (HPV_FUSE=n) | ratio | kernel |\n|---|---|---|---|---|---|")
for name, npt, us, v, ps, us2, v2 in rows:
    print(f"| {name} | {npt} | {us:.1f} ({ps}) | {us2:.1f} | {us2 / us:.2f} | `{v}` |")
