#!/usr/bin/env python3
"""End-to-end training on the round-6 kernels: Poisson-2D var_form 0 (four channels, k_iter_fused<.., NT2, GEN>) against var_form 1 on the
same 16x16-element grid of 16x16 points -- relative L2 error of u(x) on the 201x201 test grid after 10 k ... 40 k Adam iterations --
and the AdvDiff driver (var_form 0 and 1, trainable epsilon: 1.0 -> 0.1 / pi) on 16x16 elements of 16x16 points."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from hp_vpinns_amd.drivers import advdiff, poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

L = [2, 20, 20, 20, 1]
s = poisson2d.setup(N_el_x=16, N_el_y=16, N_test_x=8, N_test_y=8, N_quad=16)
for vf in (1, 0):
    for seed in (1234, 1):
        m = poisson2d.build_model(s, L, var_form=vf, init_params=xavier_init(L, seed))
        out = []
        for k in range(4):
            m._step(10000, False)
            out.append("%.2e" % m.rel_l2_error(s["X_test"], s["u_test"]))
        print("Poisson-2D var_form", vf, "seed", seed, m.h.kernel_variant(), "rel L2 after 10k,20k,30k,40k:", out, "loss", "%.3e" % m.loss()[0], flush=True)
s = advdiff.setup(N_el_x=16, N_el_t=16, N_test_x=8, N_test_t=8, N_quad=16)
for vf in (0, 1):
    for seed in (1234, 1):
        m = advdiff.build_model(s, L, var_form=vf, init_params=xavier_init(L, seed, extra=[1.0]))
        out = []
        for k in range(5):
            m._step(30000, False)
            out.append("%.4f" % m.get_params()[-1])
        print("AdvDiff var_form", vf, "seed", seed, m.h.kernel_variant(), "epsilon after 30k..150k:", out, "(0.1/pi = %.4f)" % (0.1 / math.pi), "loss %.3e" % m.loss()[0], flush=True)
