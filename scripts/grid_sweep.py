#!/usr/bin/env python3
"""Two-term Poisson-2D form, reference rule (10x10 / 5x5) and the config-4 rule (20x20 / 10x10), on growing grids: the default
dispatch against the separate launches (HPV_FUSE=n) -- is one workgroup per element still the right structure for thousands of
elements?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd.drivers import poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

L = [2, 20, 20, 20, 1]


def run(s, fuse):
    if fuse:
        os.environ["HPV_FUSE"] = fuse
    try:
        m = poisson2d.build_model(s, L, var_form=1, init_params=xavier_init(L, 1234))
    finally:
        os.environ.pop("HPV_FUSE", None)
    m.h.step(30, False)
    n = 300
    t0 = time.perf_counter()
    m.h.step(n, False)
    return (time.perf_counter() - t0) / n * 1e6, m.h.kernel_variant()


print("| rule | elements | points | default: us / iteration | separate launches | kernel |\n|---|---|---|---|---|---|")
for q, nt in ((10, 5), (20, 10)):
    for ne in (8, 16, 32, 64):
        s = poisson2d.setup(N_el_x=ne, N_el_y=ne, N_test_x=nt, N_test_y=nt, N_quad=q, with_test_grid=False, assemble="device")
        a, va = run(s, None)
        b, _ = run(s, "n")
        print(f"| {q}x{q} / {nt}x{nt} | {ne * ne} | {ne * ne * q * q} | {a:.1f} | {b:.1f} | `{va}` |")
