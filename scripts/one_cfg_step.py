#!/usr/bin/env python3
"""Training iterations of ONE of the small BASELINE configurations (for rocprofv3): one_cfg_step.py 1|2|5b [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd.drivers import advdiff, poisson1d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402
from hp_vpinns_amd.vpinn import VPINN1D  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "1"
it = int(sys.argv[2]) if len(sys.argv) > 2 else 400
if which in ("1", "2"):
    s = poisson1d.setup(N_Element=1 if which == "1" else 16)
    L = [1, 20, 20, 20, 1]
    m = VPINN1D(s["X_u_train"], s["u_train"], s["X_quad_train"], s["W_quad_train"], s["F_ext_total"], s["grid"],
                s["X_test"], s["u_test"], L, s["X_f_train"], s["f_train"], init_params=xavier_init(L, 1234))
else:
    L = [2, 20, 20, 20, 1]
    s = advdiff.setup(N_el_x=8, N_quad=10, with_test_grid=False)
    m = advdiff.build_model(s, L, init_params=xavier_init(L, 1234, extra=[1.0]))
m._step(16, False)
m.h.sync()
t0 = time.perf_counter()
m._step(it, False)
m.h.sync()
print("config %s step(%d): %.2f us/iter" % (which, it, (time.perf_counter() - t0) / it * 1e6))
