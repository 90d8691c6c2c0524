#!/usr/bin/env python3
"""Strong-form PINN branch of the Poisson-2D driver (scheme='PINNs', P2:128-129): iterations on N collocation points (profiling target).
pinn_step.py [n_points = 102400] [iterations]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd.drivers import poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

npts = int(sys.argv[1]) if len(sys.argv) > 1 else 102400
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
L = [2, 20, 20, 20, 1]
s = poisson2d.setup(N_el_x=2, N_el_y=2, N_residual=npts, with_test_grid=False)
m = poisson2d.build_model(s, L, scheme="PINNs", init_params=xavier_init(L, 1234))
m.h.step(40, False)
t0 = time.perf_counter()
m.h.step(n, False)
print("PINNs branch, %d collocation points, step(%d): %.2f us/iter, backend %s" % (npts, n, (time.perf_counter() - t0) / n * 1e6, m.backend()))
