#!/usr/bin/env python3
"""Kernel-level timings (hipEvents inside the library) of the config-4 iteration pieces."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from hp_vpinns_amd.drivers import poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

LAYERS = [2, 20, 20, 20, 1]
backend = sys.argv[1] if len(sys.argv) > 1 else "auto"
s = poisson2d.setup(N_el_x=16, N_el_y=16, N_test_x=10, N_test_y=10, N_quad=20, with_test_grid=False)
m = poisson2d.build_model(s, LAYERS, var_form=1, init_params=xavier_init(LAYERS, 1234), backend=backend)
h = m.h
names = ("mlp_fwd", "project", "mlp_bwd")
for label, fn in (("eval_loss (no act save, no adjoint)", h.eval_loss), ("forward_backward", h.forward_backward)):
    for _ in range(20):
        fn()
    h.sync()
    h.enable_timing(True)
    for _ in range(100):
        fn()
    h.sync()
    t = {n: h.kernel_time_ms(i)[0] * 1e3 for i, n in enumerate(names)}
    h.enable_timing(False)
    t0 = time.perf_counter()
    for _ in range(200):
        fn()
    h.sync()
    wall = (time.perf_counter() - t0) / 200 * 1e6
    print(f"{label}: " + "  ".join(f"{k}={v:.1f}us" for k, v in t.items()) + f"  wall/iter={wall:.1f}us")
t0 = time.perf_counter()
h.step(500, False)
print("step(500): %.1f us/iter" % ((time.perf_counter() - t0) / 500 * 1e6))
for ne in (1 << 14, 1 << 18):
    ms, by = h.bench_projection(ne, 5)
    print("projection bench n_elem=%d: %.3f ms  %.1f GB/s (%.1f%% of 8 TB/s)" % (ne, ms, by / ms / 1e6, by / ms / 1e6 / 80))
