#!/usr/bin/env python3
"""Large grids on the separate launches: full activation store against the s-only store + tangent recompute (HPV_WIDE_RC=1), width 24.
rc_large.py [elements per direction] [points per direction]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd.drivers import poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

ne = int(sys.argv[1]) if len(sys.argv) > 1 else 64
q = int(sys.argv[2]) if len(sys.argv) > 2 else 10
L = [2, 24, 24, 24, 1]
s = poisson2d.setup(N_el_x=ne, N_el_y=ne, N_test_x=q // 2, N_test_y=q // 2, N_quad=q, with_test_grid=False, assemble="device")
for rc in ("0", "1"):
    os.environ["HPV_WIDE_RC"] = rc
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
    print(f"{ne}x{ne} elements of {q}x{q} points, HPV_WIDE_RC={rc}: {dt:.1f} us/iter (fwd {t[0]:.1f}, project {t[1]:.1f}, reverse {t[2]:.1f})  {h.kernel_variant()}")
    del m, h
