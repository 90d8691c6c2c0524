#!/usr/bin/env python3
"""One mode of the multi-GPU iteration tail on ONE GPU, for `rocprofv3 --kernel-trace --stats`: the shard one of N GPUs owns of
config 4 (256 / N elements), 2 000 iterations.   rccl_tail_profile.py <none|rccl|rccl-k_adam> <N>"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd.drivers import poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

mode, n = sys.argv[1], int(sys.argv[2])
L = [2, 20, 20, 20, 1]
if mode == "rccl-k_adam":
    os.environ["HPV_NO_DEFERRED_ADAM"] = "1"
s = poisson2d.setup(N_el_x=16, N_el_y=16 // n, N_test_x=10, N_test_y=10, N_quad=20, with_test_grid=False)
m = poisson2d.build_model(s, L, init_params=xavier_init(L, 1234))
if mode.startswith("rccl"):
    m.h.rccl_connect(1, 0, m.h.rccl_unique_id())
m.h.step(200, False)
m.h.sync()
t0 = time.perf_counter()
m.h.step(2000, False)
m.h.sync()
print("%s 1/%d: %.2f us / iteration" % (mode, n, (time.perf_counter() - t0) / 2000 * 1e6))
