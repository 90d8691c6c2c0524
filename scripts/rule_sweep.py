#!/usr/bin/env python3
"""Two-term Poisson-2D form on a 16x16-element grid for a sweep of quadrature rules (N_test = N_quad / 2): iteration time and the
forward / projection / reverse split where the launches are separate -- where are the cliffs?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd.drivers import poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

L = [2, 20, 20, 20, 1]
print("| N_quad | N_test | points | us / iteration | us / 1000 points | forward / projection / reverse (us, separate launches only) | kernels |\n|---|---|---|---|---|---|---|")
for q in [int(a) for a in sys.argv[1:]] or (6, 8, 10, 12, 14, 16, 18, 20, 24, 28, 32, 40):
    nt = q // 2
    s = poisson2d.setup(N_el_x=16, N_el_y=16, N_test_x=nt, N_test_y=nt, N_quad=q, with_test_grid=False)
    m = poisson2d.build_model(s, L, var_form=1, init_params=xavier_init(L, 1234))
    h = m.h
    h.step(100, False)
    t0 = time.perf_counter()
    h.step(1000, False)
    dt = (time.perf_counter() - t0) / 1000 * 1e6
    split = ""
    if h.pass_structure() in ("separate", "fused-reverse"):
        h.enable_timing(True)
        for _ in range(30):
            h.forward_backward()
        h.sync()
        split = " / ".join("%.1f" % (h.kernel_time_ms(i)[0] * 1e3) for i in range(3))
    npt = 256 * q * q
    print(f"| {q} | {nt} | {npt} | {dt:.1f} | {dt / npt * 1e3:.3f} | {split} | `{h.kernel_variant()}` |")
    del m, h
