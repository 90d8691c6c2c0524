#!/usr/bin/env python3
"""What a 20-step timed window of bench.py costs beyond its 20 iterations: windows of K = 20 / 200 / 2000 steps bracketed by the handle's
sync + torch.cuda.synchronize(), median of 25, through the Python class and through the handle alone.
Measured (round 5): K = 2000 / 200 / 20: 59.78 / 59.86 / 60.95 us per step, the same through the handle alone (59.76 / 59.90 / 60.87): the
22 us a 20-step window costs beyond its iterations are the launch and synchronisation latency of the runtime, not Python or torch.
(hipSetDeviceFlags(hipDeviceScheduleSpin) through a second ctypes handle before torch's first HIP call HUNG the process: not an option.)
  window_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from hp_vpinns_amd.drivers import poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

L = [2, 20, 20, 20, 1]
s = poisson2d.setup(N_el_x=16, N_el_y=16, N_test_x=10, N_test_y=10, N_quad=20, with_test_grid=False)
m = poisson2d.build_model(s, L, init_params=xavier_init(L, 1234))
m._step(4000, False)
m.h.sync()
for K in (2000, 200, 20):
    w = []
    for _ in range(25 if K < 2000 else 5):
        m.h.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        m._step(K, False)
        m.h.sync(); torch.cuda.synchronize()
        w.append(time.perf_counter() - t0)
    w2 = []
    for _ in range(25 if K < 2000 else 5):      # the same through the handle alone (no Python class, no torch sync)
        m.h.sync()
        t0 = time.perf_counter()
        m.h.step(K, False)
        m.h.sync()
        w2.append(time.perf_counter() - t0)
    print("K = %4d: %.2f us / step (class + torch sync), %.2f us / step (handle only)" % (K, np.median(w) / K * 1e6, np.median(w2) / K * 1e6))
