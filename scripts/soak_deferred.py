#!/usr/bin/env python3
"""Long run of the multi-GPU launch sequence on ONE GPU (in-library RCCL exchange on a 1-rank communicator, the TF1-Adam update deferred
into the next iteration's kernels): config 4 and its 8-GPU shard (32 elements, an element shared by 8 workgroups), n iterations each,
next to the same run with a k_adam launch per iteration (HPV_NO_DEFERRED_ADAM=1).  The first 500 iterations must agree to round-off
(the sequences are the same arithmetic); after that the trajectories of a chaotic optimisation drift apart -- the final losses are
printed side by side -- and the update counts must be exact.   soak_deferred.py [iterations, default 400000]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from hp_vpinns_amd.drivers import poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
L = [2, 20, 20, 20, 1]
print("| shard | iterations | deferred: it/s, final loss | k_adam per iteration: it/s, final loss | max rel. parameter difference after 500 iterations | updates applied |\n|---|---|---|---|---|---|")
for ney in (16, 2):
    s = poisson2d.setup(N_el_x=16, N_el_y=ney, N_test_x=10, N_test_y=10, N_quad=20, with_test_grid=False)
    res = []
    for mode in ("deferred", "k_adam"):
        if mode == "k_adam":
            os.environ["HPV_NO_DEFERRED_ADAM"] = "1"
        m = poisson2d.build_model(s, L, init_params=xavier_init(L, 1234))
        os.environ.pop("HPV_NO_DEFERRED_ADAM", None)
        m.h.rccl_connect(1, 0, m.h.rccl_unique_id())
        m.h.step(500, False)
        p500 = m.h.get_params()
        t0 = time.perf_counter()
        done = 500
        while done < n:
            k = min(50000 + 7, n - done)         # (8-iteration graphs + a remainder graph in every call)
            l3 = m.h.step(k, True)
            assert np.all(np.isfinite(l3)), (mode, done, l3)
            done += k
        dt = time.perf_counter() - t0
        res.append(((n - 500) / dt, float(l3[0]), p500, m.h.updates_applied()))
    d = np.max(np.abs(res[0][2] - res[1][2]) / np.maximum(np.abs(res[1][2]), 1e-300))
    assert res[0][3] == res[1][3] == n and d < 1e-9, (res[0][3], res[1][3], d)
    print("| %d elements | %d | %.0f, %.3e | %.0f, %.3e | %.1e | %d = %d |" % (16 * ney, n, res[0][0], res[0][1], res[1][0], res[1][1], d, res[0][3], res[1][3]))
