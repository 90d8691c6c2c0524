#!/usr/bin/env python3
"""Per-kernel times (hipEvents) for configs 1 and 5."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd.drivers import advdiff, poisson1d
from hp_vpinns_amd.init import xavier_init
from hp_vpinns_amd.vpinn import VPINN1D
L2 = [2, 20, 20, 20, 1]
s = advdiff.setup(N_el_x=8, N_quad=80, with_test_grid=False)
m5 = advdiff.build_model(s, L2, init_params=xavier_init(L2, 1234, extra=[1.0]))
s = poisson1d.setup(N_Element=16)
L1 = [1, 20, 20, 20, 1]
m1 = VPINN1D(s["X_u_train"], s["u_train"], s["X_quad_train"], s["W_quad_train"], s["F_ext_total"], s["grid"], s["X_test"], s["u_test"], L1, s["X_f_train"], s["f_train"], init_params=xavier_init(L1, 1234))
for name, m in (("cfg5 advdiff 80x80", m5), ("cfg2 1D 16 el", m1)):
    h = m.h
    for _ in range(20): h.forward_backward()
    h.sync(); h.enable_timing(True)
    for _ in range(100): h.forward_backward()
    h.sync()
    print(name, {n: round(h.kernel_time_ms(i)[0] * 1e3, 1) for i, n in enumerate(("mlp_fwd", "project", "mlp_bwd"))})
    h.enable_timing(False)
