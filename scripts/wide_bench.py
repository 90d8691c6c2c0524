#!/usr/bin/env python3
"""Iteration time of the config-4 grid (16x16 elements, 20x20 points, 10x10 test functions, var_form 1) and of a 1-D grid for
networks of other hidden widths / depths than the hand-tuned [.,20,20,20,1]: what the width-generic MFMA kernels
(csrc/kernels_wide.hip) buy over the generic VALU kernels, next to the flop-scaled 20-wide time (verdict round 3, next 3:
"within 2x of the flop-scaled 20-wide time").  Prints a markdown table (profiles/r04_wide_networks.md)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402,F401

from hp_vpinns_amd.drivers import poisson1d, poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402
from hp_vpinns_amd.vpinn import VPINN1D  # noqa: E402


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


def gemm_flops(L):
    return 2 * sum(L[i] * L[i + 1] for i in range(len(L) - 1))


s4 = poisson2d.setup(N_el_x=16, N_el_y=16, N_test_x=10, N_test_y=10, N_quad=20, with_test_grid=False)
rows = []


def run2d(L, backend="auto", env=None, n=400):
    for k, v in (env or {}).items():
        os.environ[k] = v
    try:
        m = poisson2d.build_model(s4, L, init_params=xavier_init(L, 1234), backend=backend)
        us = timeit(m, n)
        return us, m.backend(), m.h.kernel_variant()
    finally:
        for k in (env or {}):
            del os.environ[k]


L20 = [2, 20, 20, 20, 1]
base_us, _, v = run2d(L20)
rows.append(("config-4 grid", L20, "default (whole-iteration kernel)", base_us, 1.0, v))
sep_us, _, v = run2d(L20, env={"HPV_FUSE": "n"})
rows.append(("config-4 grid", L20, "HPV_FUSE=n (three launches: the structure the wide kernels use)", sep_us, 1.0, v))
for H in (24, 32, 40, 48, 64):
    L = [2, H, H, H, 1]
    us, bk, v = run2d(L)
    rows.append(("config-4 grid", L, bk, us, gemm_flops(L) / gemm_flops(L20), v))
for L in ([2, 32, 32, 1], [2, 32, 32, 32, 32, 1], [2, 20, 20, 20, 20, 1], [2] + [20] * 5 + [1], [2] + [20] * 6 + [1]):
    us, bk, v = run2d(L)
    rows.append(("config-4 grid", L, bk, us, gemm_flops(L) / gemm_flops(L20), v))
us, bk, v = run2d([2, 32, 32, 32, 1], backend="generic", n=20)
rows.append(("config-4 grid", [2, 32, 32, 32, 1], "generic (what every width != 20 ran on until round 3)", us, gemm_flops([2, 32, 32, 32, 1]) / gemm_flops(L20), v))

s1 = poisson1d.setup(N_Element=16)


def run1d(L):
    m = VPINN1D(s1["X_u_train"], s1["u_train"], s1["X_quad_train"], s1["W_quad_train"], s1["F_ext_total"], s1["grid"],
                s1["X_test"], s1["u_test"], L, s1["X_f_train"], s1["f_train"], init_params=xavier_init(L, 1234))
    return timeit(m), m.backend(), m.h.kernel_variant()


L1 = [1, 20, 20, 20, 1]
b1, _, v = run1d(L1)
rows.append(("Poisson-1D 16 elements (config 2)", L1, "default", b1, 1.0, v))
for L in ([1, 20, 20, 20, 20, 1], [1, 32, 32, 32, 32, 1], [1, 40, 40, 40, 1]):
    us, bk, v = run1d(L)
    rows.append(("Poisson-1D 16 elements (config 2)", L, bk, us, gemm_flops(L) / gemm_flops(L1), v))

print("| grid | Net_layer | path | us / iteration | layer-product flops vs 20-wide | us / (flop-scaled 20-wide us) | kernels |\n|---|---|---|---|---|---|---|")
for grid, L, path, us, fr, v in rows:
    ref = base_us if grid.startswith("config-4") else b1
    print(f"| {grid} | {L} | {path} | {us:.1f} | {fr:.2f} | {us / (ref * fr):.2f} | `{v}` |")
