#!/usr/bin/env python3
"""How fast do the Adam trajectories of two launch structures part?  AdvDiff var_form 0 / 1 (trainable epsilon) and Poisson-2D var_form 0 on
16x16 elements: default (whole-iteration kernel), separate launches (n), generic element-resident kernel (e) -- gradient at step 0
and parameters after 10 / 100 / 300 / 1000 iterations, pairwise."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from hp_vpinns_amd.drivers import advdiff, poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

L = [2, 20, 20, 20, 1]


def build(mk, fuse):
    if fuse:
        os.environ["HPV_FUSE"] = fuse
    try:
        return mk()
    finally:
        os.environ.pop("HPV_FUSE", None)


def d(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


for name, mk in (("AdvDiff vf0", lambda: advdiff.build_model(advdiff.setup(N_el_x=16, N_el_t=16, N_test_x=8, N_test_t=8, N_quad=16, with_test_grid=False), L, var_form=0, init_params=xavier_init(L, 1234, extra=[1.0]))),
                 ("AdvDiff vf1", lambda: advdiff.build_model(advdiff.setup(N_el_x=16, N_el_t=16, N_test_x=8, N_test_t=8, N_quad=16, with_test_grid=False), L, var_form=1, init_params=xavier_init(L, 1234, extra=[1.0]))),
                 ("Poisson vf0", lambda: poisson2d.build_model(poisson2d.setup(N_el_x=16, N_el_y=16, N_test_x=8, N_test_y=8, N_quad=16, with_test_grid=False), L, var_form=0, init_params=xavier_init(L, 1234)))):
    ms = {"default": build(mk, None), "n": build(mk, "n"), "e": build(mk, "e")}
    g = {k: m.loss_and_grad()[1] for k, m in ms.items()}
    print(name, "| gradient at step 0: default-n %.1e, e-n %.1e; d/d eps (last entry) default %.15e n %.15e e %.15e"
          % (d(g["default"], g["n"]), d(g["e"], g["n"]), g["default"][-1], g["n"][-1], g["e"][-1]))
    done = 0
    for upto in (10, 100, 300, 1000, 3000):
        for m in ms.values():
            m._step(upto - done, False)
        done = upto
        p = {k: m.get_params() for k, m in ms.items()}
        print("   after %4d iterations: default-n %.1e, e-n %.1e, default-e %.1e" % (upto, d(p["default"], p["n"]), d(p["e"], p["n"]), d(p["default"], p["e"])), flush=True)
