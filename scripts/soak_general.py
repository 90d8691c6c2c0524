#!/usr/bin/env python3
"""Long runs of the round-6 launch structures next to the separate launches (HPV_FUSE=n) of the same problem: the general forms of the
whole-iteration kernel (AdvDiff var_form 0 / 1 with the trainable epsilon, Poisson-2D var_form 0) on a full grid and on SPLIT shards, and
a ragged grid (full rounds + the tail in split mode).  A missed exchange makes hpv_step raise (-7); the Adam trajectories agree to
round-off for ~1 000 iterations (checked: 1e-7) and then drift apart chaotically like any two summation orders do.
soak_general.py [iterations per case, default 100000]   (SOAK_TIGHT_ONLY=1: only the tight-plan cases at the end of the list)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from hp_vpinns_amd.drivers import advdiff, poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
L = [2, 20, 20, 20, 1]


def build(mk, fuse):
    if fuse:
        os.environ["HPV_FUSE"] = fuse
    try:
        return mk()
    finally:
        os.environ.pop("HPV_FUSE", None)


cases = []
for (nex, ney) in ((16, 16), (16, 4), (5, 3)):
    for vf in (0, 1):
        s = advdiff.setup(N_el_x=nex, N_el_t=ney, N_test_x=8, N_test_t=8, N_quad=16, with_test_grid=False)
        cases.append(("AdvDiff var_form %d, %dx%d elements of 16x16 points" % (vf, nex, ney),
                      (lambda s=s, vf=vf: advdiff.build_model(s, L, var_form=vf, init_params=xavier_init(L, 1234, extra=[1.0])))))
    s = poisson2d.setup(N_el_x=nex, N_el_y=ney, N_test_x=6, N_test_y=6, N_quad=12, with_test_grid=False)
    cases.append(("Poisson-2D var_form 0, %dx%d elements of 12x12 points" % (nex, ney),
                  (lambda s=s: poisson2d.build_model(s, L, var_form=0, init_params=xavier_init(L, 1234)))))
s = poisson2d.setup(N_el_x=24, N_el_y=23, N_test_x=10, N_test_y=10, N_quad=20, with_test_grid=False, assemble="device")
cases.append(("Poisson-2D var_form 1, 24x23 elements of 20x20 points (two rounds + a 40-element tail)",
              (lambda s=s: poisson2d.build_model(s, L, var_form=1, init_params=xavier_init(L, 1234)))))
# the tight plan (four channels, three hidden layers, 20x20 points: FzPlan of kernels_fused.hip): full grid, SPLIT shard, ragged grid
tight = []
for (nex, ney) in ((16, 16), (16, 4), (24, 23)):
    s = poisson2d.setup(N_el_x=nex, N_el_y=ney, N_test_x=10, N_test_y=10, N_quad=20, with_test_grid=False, assemble="device")
    tight.append(("Poisson-2D var_form 0, %dx%d elements of 20x20 points" % (nex, ney),
                  (lambda s=s: poisson2d.build_model(s, L, var_form=0, init_params=xavier_init(L, 1234)))))
s = advdiff.setup(N_el_x=16, N_el_t=16, N_test_x=10, N_test_t=10, N_quad=20, with_test_grid=False)
tight.append(("AdvDiff var_form 0, 16x16 elements of 20x20 points",
              (lambda s=s: advdiff.build_model(s, L, var_form=0, init_params=xavier_init(L, 1234, extra=[1.0])))))
cases = tight if os.environ.get("SOAK_TIGHT_ONLY") else cases + tight
for name, mk in cases:
    a, b = build(mk, None), build(mk, "n")
    t0 = time.perf_counter()
    done, early = 0, None
    while done < n:
        k = 1000 if done == 0 else min(20000, n - done)
        a._step(k, False)
        b._step(k, False)
        done += k
        pa, pb = a.get_params(), b.get_params()
        assert np.all(np.isfinite(pa)) and np.all(np.isfinite(pb)), "non-finite parameters"
        if early is None:
            early = float(np.abs(pa - pb).max() / np.abs(pb).max())
            assert early < 1e-6, (name, early)      # (AdvDiff: any two structures part at this rate, scripts/traj_probe.py)
    print("%s: %s | %d iterations x 2 structures in %.1f s; relative parameter difference to the separate launches after 1 000 iterations "
          "%.1e, at the end %.1e; loss %.3e against %.3e%s" % (name, a.h.kernel_variant(), n, time.perf_counter() - t0, early,
          float(np.abs(pa - pb).max() / np.abs(pb).max()), a.loss()[0], b.loss()[0],
          ("; epsilon %.5f against %.5f" % (pa[-1], pb[-1])) if "AdvDiff" in name else ""), flush=True)
