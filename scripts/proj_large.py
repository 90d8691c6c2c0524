#!/usr/bin/env python3
"""Separate launches (HPV_FUSE=n) on a large grid: forward / projection / reverse split.  proj_large.py ne q [HPV_PJ_WG_ONLY]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["HPV_FUSE"] = "n"
from hp_vpinns_amd.drivers import poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

ne, q = int(sys.argv[1]), int(sys.argv[2])
L = [2, 20, 20, 20, 1]
s = poisson2d.setup(N_el_x=ne, N_el_y=ne, N_test_x=q // 2, N_test_y=q // 2, N_quad=q, with_test_grid=False, assemble="device")
m = poisson2d.build_model(s, L, var_form=1, init_params=xavier_init(L, 1234))
h = m.h
h.step(20, False)
t0 = time.perf_counter()
h.step(200, False)
dt = (time.perf_counter() - t0) / 200 * 1e6
h.enable_timing(True)
for _ in range(20):
    h.forward_backward()
h.sync()
t = [h.kernel_time_ms(i)[0] * 1e3 for i in range(3)]
print(f"{ne}x{ne} elements of {q}x{q} points: {dt:.1f} us/iter (fwd {t[0]:.1f}, project {t[1]:.1f}, reverse {t[2]:.1f})  {h.kernel_variant()}")
