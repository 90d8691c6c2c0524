#!/usr/bin/env python3
"""Training iterations of one small BASELINE config (profiling target): small_cfg_step.py {1,2,3,5b} [iterations]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd.drivers import advdiff, poisson1d, poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402
from hp_vpinns_amd.vpinn import VPINN1D  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
L2 = [2, 20, 20, 20, 1]
if cfg in ("1", "2"):
    s = poisson1d.setup(N_Element=1 if cfg == "1" else 16)
    L = [1, 20, 20, 20, 1]
    m = VPINN1D(s["X_u_train"], s["u_train"], s["X_quad_train"], s["W_quad_train"], s["F_ext_total"], s["grid"],
                s["X_test"], s["u_test"], L, s["X_f_train"], s["f_train"], init_params=xavier_init(L, 1234))
elif cfg == "3":
    s = poisson2d.setup(N_el_x=8, N_el_y=8, with_test_grid=False)
    m = poisson2d.build_model(s, L2, init_params=xavier_init(L2, 1234))
else:
    s = advdiff.setup(N_el_x=8, N_quad=10, with_test_grid=False)
    m = advdiff.build_model(s, L2, init_params=xavier_init(L2, 1234, extra=[1.0]))
m.h.step(16, False)
t0 = time.perf_counter()
m.h.step(n, False)
print("config %s step(%d): %.2f us/iter" % (cfg, n, (time.perf_counter() - t0) / n * 1e6))
