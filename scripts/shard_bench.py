#!/usr/bin/env python3
"""Per-iteration time of the config-4 kernels on the element shards one GPU owns at N = 1, 2, 4, 8
(16 x 16/N elements of the same shape; no communication): what strong scaling can at best deliver."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd.drivers import poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

LAYERS = [2, 20, 20, 20, 1]
for n in (1, 2, 4, 8):
    s = poisson2d.setup(N_el_x=16, N_el_y=16 // n, N_test_x=10, N_test_y=10, N_quad=20, with_test_grid=False)
    m = poisson2d.build_model(s, LAYERS, var_form=1, init_params=xavier_init(LAYERS, 1234))
    h = m.h
    h.step(200, False)
    t0 = time.perf_counter()
    h.step(1000, False)
    dt = (time.perf_counter() - t0) / 1000 * 1e6
    h.enable_timing(True)
    for _ in range(50):
        h.forward_backward()
    h.sync()
    t = [h.kernel_time_ms(i)[0] * 1e3 for i in range(3)]
    print(f"shard of N={n}: {256 // n} elements: {dt:.1f} us/iter (fwd {t[0]:.1f}, project {t[1]:.1f}, reverse {t[2]:.1f})")
