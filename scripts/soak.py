#!/usr/bin/env python3
"""Long back-to-back runs of the barrier-carrying kernel: the SPLIT mode of the whole-iteration kernel (2 / 4 / 8 workgroups per
element meet at a device-memory barrier in every launch) next to the same shard on two barrier-free structures (HPV_FUSE=s:
forward + split reverse kernels; HPV_FUSE=n: separate launches).  A barrier that is ever missed makes hpv_step raise (-7).  The
three Adam trajectories agree to round-off for ~1 000 iterations and then drift apart chaotically (1e-2 relative by 50 000
iterations) -- the SPLIT one no further from `s` than `s` is from `n` (the control), which is what a correct exchange looks like.
soak.py [iterations per shard, default 300000] [points per direction, default 20: 20, 16 or 12]
(measured: 1.2 M launches without a timeout, profiles/README.md)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from hp_vpinns_amd.drivers import poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 20
L = [2, 20, 20, 20, 1]


def build(s, th, fuse):
    if fuse:
        os.environ["HPV_FUSE"] = fuse
    try:
        return poisson2d.build_model(s, L, init_params=th)
    finally:
        os.environ.pop("HPV_FUSE", None)


for ney in (8, 4, 2):
    s = poisson2d.setup(N_el_x=16, N_el_y=ney, N_test_x=Q // 2, N_test_y=Q // 2, N_quad=Q, with_test_grid=False)
    th = xavier_init(L, 1234)
    ms = {"split": build(s, th, None), "s": build(s, th, "s"), "n": build(s, th, "n")}
    t0 = time.perf_counter()
    done, early = 0, None
    while done < n:
        k = 1000 if done == 0 else min(50000, n - done)
        for m in ms.values():
            m._step(k, False)
        done += k
        p = {a: m.get_params() for a, m in ms.items()}
        assert all(np.all(np.isfinite(v)) for v in p.values()), "non-finite parameters"
        d = lambda a, b: float(np.abs(p[a] - p[b]).max() / np.abs(p[b]).max())
        if early is None:
            early = (d("split", "s"), d("s", "n"))
    print("shard of %3d elements (%s): %d iterations x 3 structures in %.1f s; relative parameter difference after 1 000 iterations "
          "split-vs-s %.1e, s-vs-n %.1e; at the end %.1e, %.1e; loss %.3e"
          % (16 * ney, ms["split"].h.pass_structure(), n, time.perf_counter() - t0, early[0], early[1], d("split", "s"),
             d("s", "n"), ms["split"].loss()[0]))
