#!/usr/bin/env python3
"""Config-4 element shape on larger grids (32x32, 64x64 elements = 0.4M / 1.6M quadrature points): known-answer check
at theta = 0 and iterations/s -- the regime where per-GPU work dwarfs launch and all-reduce latency."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd.drivers import poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

L = [2, 20, 20, 20, 1]
for ne in (16, 32, 64):
    s = poisson2d.setup(N_el_x=ne, N_el_y=ne, N_test_x=10, N_test_y=10, N_quad=20, with_test_grid=False, assemble="device")
    m0 = poisson2d.build_model(s, L, init_params=np.zeros(921))
    z = m0.loss()
    F = s["F_ext_total"]
    ok = abs(z[2] - (F ** 2).mean(axis=(2, 3)).sum()) < 1e-10 * z[2]
    del m0
    m = poisson2d.build_model(s, L, init_params=xavier_init(L, 1234))
    m._step(50, False)
    t0 = time.perf_counter()
    n = 400 if ne < 64 else 100
    m._step(n, False)
    dt = (time.perf_counter() - t0) / n
    pts = ne * ne * 400
    print(f"{ne}x{ne} elements ({pts} points): zero-network loss identity {'OK' if ok else 'FAILED'}; "
          f"{dt * 1e6:.1f} us/iter = {1 / dt:.0f} it/s = {pts / dt / 1e9:.2f} Gpoint-iterations/s")
