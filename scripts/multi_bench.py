#!/usr/bin/env python3
"""Grids with more elements than CUs: k_iter_fused walking several elements per workgroup (MULTI, the default there) against one
workgroup per element (HPV_FUSE=1) and the separate launches (HPV_FUSE=n).   multi_bench.py [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd.drivers import poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300


def run(s, L, fuse):
    os.environ.pop("HPV_FUSE", None)
    if fuse:
        os.environ["HPV_FUSE"] = fuse
    m = poisson2d.build_model(s, L, init_params=xavier_init(L, 1234))
    m.h.step(30, False)
    m.h.sync()
    t0 = time.perf_counter()
    m.h.step(iters, False)
    m.h.sync()
    us = (time.perf_counter() - t0) / iters * 1e6
    v = m.h.kernel_variant()
    del m
    os.environ.pop("HPV_FUSE", None)
    return us, v


print("| grid | points / test fcns | network | default: us / iteration | kernel | element loop forced (HPV_FUSE=m) | one workgroup per element (HPV_FUSE=1) | separate launches (HPV_FUSE=n) |\n|---|---|---|---|---|---|---|---|")
for (ne, q, nt, nh) in ((17, 16, 8, 3), (32, 16, 8, 3), (40, 16, 8, 3), (48, 16, 8, 3), (32, 12, 6, 3), (40, 12, 6, 3), (64, 12, 6, 3), (32, 16, 8, 2), (48, 16, 8, 2), (32, 20, 10, 2), (40, 20, 10, 2), (64, 20, 10, 2), (17, 20, 10, 3), (32, 20, 10, 3), (48, 20, 10, 3), (64, 20, 10, 3)):
    L = [2] + [20] * nh + [1]
    s = poisson2d.setup(N_el_x=ne, N_el_y=ne, N_test_x=nt, N_test_y=nt, N_quad=q, with_test_grid=False, assemble="device")
    a, va = run(s, L, None)
    f, vf = run(s, L, "m")
    b, vb = run(s, L, "1")
    c, vc = run(s, L, "n")
    short = lambda v: ("k_iter_fused" + (" loop" if "elements-per" in v else "")) if "k_iter_fused" in v else v.split("<")[0]   # noqa: E731
    print("| %dx%d | %dx%d / %dx%d | %s | **%.1f** | `%s` | %.1f `%s` | %.1f `%s` | %.1f |" % (ne, ne, q, q, nt, nt, L, a, va, f, short(vf), b, short(vb), c), flush=True)
