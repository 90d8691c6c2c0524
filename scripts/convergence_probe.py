#!/usr/bin/env python3
"""Numbers behind tests/test_gpu_convergence.py: error levels vs iteration count and seed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hp_vpinns_amd.drivers import poisson1d, poisson2d
from hp_vpinns_amd.init import xavier_init

L = [2, 20, 20, 20, 1]
s = poisson2d.setup(N_el_x=16, N_el_y=16, N_test_x=10, N_test_y=10, N_quad=20)
for seed in (1234, 1, 2):
    m = poisson2d.build_model(s, L, init_params=xavier_init(L, seed))
    out = []
    for k in range(5):
        m._step(10000, False)
        out.append("%.2e" % m.rel_l2_error(s["X_test"], s["u_test"]))
    print("config 4 seed", seed, "rel L2 after 10k,20k,..50k:", out, flush=True)
L1 = [1, 20, 20, 20, 20, 1]
for seed in (1234, 1, 2, 3, 4):
    r = poisson1d.run(Opt_Niter=40000 + 1, N_Element=3, verbose=False, init_params=xavier_init(L1, seed))
    rec = np.array(r["total_record"])
    err = np.abs(r["setup"]["u_test"] - r["u_pred"]).max()
    print("1-D 3 elements seed", seed, "last loss %.2e min loss %.2e min(last 1000 its) %.2e max err %.2e" % (rec[-1, 1], rec[:, 1].min(), rec[-100:, 1].min(), err), flush=True)
L2 = [2, 5, 5, 5, 1]
errs = []
for seed in range(8):
    r = poisson2d.run(n_iter=10000 + 1, record_every=100, verbose=False, init_params=xavier_init(L2, seed))
    errs.append(float(np.abs(r["setup"]["u_test"] - r["u_pred"]).max()))
print("P2 defaults, seeds 0..7, max err:", ["%.3f" % e for e in errs])
