#!/usr/bin/env python3
"""Iterations/sec of the five BASELINE.json configurations on one GPU (hpv_step, whole iteration on device)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from hp_vpinns_amd.drivers import advdiff, poisson1d, poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402
from hp_vpinns_amd.vpinn import VPINN1D  # noqa: E402


def timeit(m, n=2000):
    m._step(200, False)
    m.h.sync()
    t0 = time.perf_counter()
    m._step(n, False)
    m.h.sync()
    dt = time.perf_counter() - t0
    return n / dt, 1e6 * dt / n


def p1(ne):
    s = poisson1d.setup(N_Element=ne)
    L = [1, 20, 20, 20, 1]
    return VPINN1D(s["X_u_train"], s["u_train"], s["X_quad_train"], s["W_quad_train"], s["F_ext_total"], s["grid"],
                   s["X_test"], s["u_test"], L, s["X_f_train"], s["f_train"], init_params=xavier_init(L, 1234))


L2 = [2, 20, 20, 20, 1]
rows = []
rows.append(("1: Poisson-1D, 1 element, Q=80, 60 test fcns", p1(1)))
rows.append(("2: Poisson-1D, 16 elements", p1(16)))
s = poisson2d.setup(N_el_x=8, N_el_y=8, with_test_grid=False)
rows.append(("3: Poisson-2D 8x8 el, 10x10 quad, 5x5 test", poisson2d.build_model(s, L2, init_params=xavier_init(L2, 1234))))
s = poisson2d.setup(N_el_x=16, N_el_y=16, N_test_x=10, N_test_y=10, N_quad=20, with_test_grid=False)
rows.append(("4: Poisson-2D 16x16 el, 20x20 quad, 10x10 test", poisson2d.build_model(s, L2, init_params=xavier_init(L2, 1234))))
s = advdiff.setup(N_el_x=8, N_quad=80, with_test_grid=False)
rows.append(("5: AdvDiff 8x1 el, 80x80 quad, 5x5 test (51 200 pts)", advdiff.build_model(s, L2, init_params=xavier_init(L2, 1234, extra=[1.0]))))
s = advdiff.setup(N_el_x=8, N_quad=10, with_test_grid=False)
rows.append(("5b: AdvDiff 8x1 el, 10x10 quad (reference rule)", advdiff.build_model(s, L2, init_params=xavier_init(L2, 1234, extra=[1.0]))))
print("| config | backend | it/s | us/iter |\n|---|---|---|---|")
for name, m in rows:
    its, us = timeit(m)
    print(f"| {name} | {m.backend()} | {its:.0f} | {us:.1f} |")
