#!/usr/bin/env python3
"""Training iterations of the two-term form on a 16x16-element grid of q x q points / nt x nt test functions: shape_step.py q nt [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd.drivers import poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

q, nt = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
L = [2, 20, 20, 20, 1]
s = poisson2d.setup(N_el_x=16, N_el_y=16, N_test_x=nt, N_test_y=nt, N_quad=q, with_test_grid=False)
m = poisson2d.build_model(s, L, var_form=1, init_params=xavier_init(L, 1234))
m.h.step(16, False)
t0 = time.perf_counter()
m.h.step(n, False)
print("%dx%d / %dx%d step(%d): %.2f us/iter  %s" % (q, q, nt, nt, n, (time.perf_counter() - t0) / n * 1e6, m.h.kernel_variant()))
