#!/usr/bin/env python3
"""Config-4 grid with a wider network, training iterations only (profiling target): wide_step.py [H] [iterations]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd.drivers import poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
L = [2, H, H, H, 1]
s = poisson2d.setup(N_el_x=16, N_el_y=16, N_test_x=10, N_test_y=10, N_quad=20, with_test_grid=False)
m = poisson2d.build_model(s, L, var_form=1, init_params=xavier_init(L, 1234))
m.h.step(16, False)
t0 = time.perf_counter()
m.h.step(n, False)
print("H=%d step(%d): %.2f us/iter  %s" % (H, n, (time.perf_counter() - t0) / n * 1e6, m.h.kernel_variant()))
