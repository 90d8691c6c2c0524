#!/usr/bin/env python3
"""Training iterations on the config-4 shard one of N GPUs owns (256 / N elements), single GPU, no exchange: shard_step.py N [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd.drivers import poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

LAYERS = [2, 20, 20, 20, 1]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
it = int(sys.argv[2]) if len(sys.argv) > 2 else 400
s = poisson2d.setup(N_el_x=16, N_el_y=16 // n, N_test_x=10, N_test_y=10, N_quad=20, with_test_grid=False)
m = poisson2d.build_model(s, LAYERS, var_form=1, init_params=xavier_init(LAYERS, 1234))
m.h.step(16, False)
t0 = time.perf_counter()
m.h.step(it, False)
print("shard 1/%d (%d elements) step(%d): %.2f us/iter" % (n, 256 // n, it, (time.perf_counter() - t0) / it * 1e6))
