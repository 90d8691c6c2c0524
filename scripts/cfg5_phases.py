#!/usr/bin/env python3
"""Config 5 (AdvDiff, 8 elements x 80x80 points) on the two-kernel path: forward kernel with and without the activation store
(eval pass vs training pass), projection, reverse -- hipEvent averages per launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd.drivers import advdiff  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

L = [2, 20, 20, 20, 1]
s = advdiff.setup(N_el_x=8, N_quad=80, with_test_grid=False)
m = advdiff.build_model(s, L, init_params=xavier_init(L, 1234, extra=[1.0]))
h = m.h
for name, fn in (("training pass (store)", h.forward_backward), ("eval pass (no store)", h.eval_loss)):
    for _ in range(20):
        fn()
    h.enable_timing(True)
    for _ in range(200):
        fn()
    h.sync()
    t = [h.kernel_time_ms(i)[0] * 1e3 for i in range(3)]
    h.enable_timing(False)
    print("%-24s forward %.1f us, projection %.1f us, reverse %.1f us" % (name, t[0], t[1], t[2]))
