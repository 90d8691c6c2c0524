#!/usr/bin/env python3
"""Identified diffusion coefficient of the AdvDiff driver (P3:41-42, 63) over seeds, network widths and grids: what the band of
tests/test_gpu_convergence.py::test_advdiff_identifies_the_published_diffusion_coefficient is made of.   advdiff_eps_probe.py [iters]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd.drivers import advdiff  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
exact = 0.1 / np.pi
print("| network | grid / rule | seed | identified epsilon | vs 0.1/pi | final loss |\n|---|---|---|---|---|---|")
for width, nex, net, q, seeds in ((5, 1, 1, 10, range(8)), (20, 1, 1, 10, range(4)), (20, 4, 2, 10, range(2)), (20, 1, 1, 20, range(2))):
    L = [2] + [width] * 3 + [1]
    s = advdiff.setup(N_el_x=nex, N_el_t=net, N_quad=q, with_test_grid=False)
    for seed in seeds:
        m = advdiff.build_model(s, L, var_form=0, init_params=xavier_init(L, seed, extra=[1.0]))
        m._step(iters, False)
        l = float(m._step(1, True)[0])
        e = float(m.epsilon[0])
        print("| %s | %dx%d elements, %dx%d points | %d | %.6f | %+.1f %% | %.3e |" % (L, nex, net, q, q, seed, e, 100 * (e / exact - 1), l), flush=True)
