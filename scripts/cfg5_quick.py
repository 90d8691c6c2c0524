#!/usr/bin/env python3
"""BASELINE config 5 (AdvDiff, 8 elements x 80x80 points) iterations only (profiling target): cfg5_quick.py [iterations]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd.drivers import advdiff  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

L = [2, 20, 20, 20, 1]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
s = advdiff.setup(N_el_x=8, N_quad=80, with_test_grid=False)
m = advdiff.build_model(s, L, init_params=xavier_init(L, 1234, extra=[1.0]))
m.h.step(64, False)
t0 = time.perf_counter()
m.h.step(n, False)
print("config 5 step(%d): %.2f us/iter, structure %s" % (n, (time.perf_counter() - t0) / n * 1e6, m.h.pass_structure()))
