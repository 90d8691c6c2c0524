import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from hp_vpinns_amd.drivers import poisson2d
from hp_vpinns_amd.init import xavier_init
L = [2, 20, 20, 20, 1]
s = poisson2d.setup(N_el_x=16, N_el_y=16, N_test_x=10, N_test_y=10, N_quad=20)
for seed in (1234, 1, 2, 3):
    m = poisson2d.build_model(s, L, init_params=xavier_init(L, seed))
    l0 = m.loss()[0]
    out = []
    for k in range(10):
        m._step(5000, False)
        out.append("%.1e/%.1e" % (m.rel_l2_error(s["X_test"], s["u_test"]), m.loss()[0] / l0))
    print("seed", seed, "relL2/loss-ratio every 5k:", out, flush=True)
