#!/usr/bin/env python3
"""Iterations of a general variational form on the whole-iteration kernel (profiling target): gen_step.py <advf0|advf1|p2vf0> [q=16] [iterations]
16x16 elements of q x q points, q/2 x q/2 test functions, [2,20,20,20,1]."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd.drivers import advdiff, poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

prob = sys.argv[1] if len(sys.argv) > 1 else "advf0"
q = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
L = [2, 20, 20, 20, 1]
if prob == "p2vf0":
    s = poisson2d.setup(N_el_x=16, N_el_y=16, N_test_x=q // 2, N_test_y=q // 2, N_quad=q, with_test_grid=False)
    m = poisson2d.build_model(s, L, var_form=0, init_params=xavier_init(L, 1234))
else:
    s = advdiff.setup(N_el_x=16, N_el_t=16, N_test_x=q // 2, N_test_t=q // 2, N_quad=q, with_test_grid=False)
    m = advdiff.build_model(s, L, var_form=int(prob[-1]), init_params=xavier_init(L, 1234, extra=[1.0]))
m.h.step(64, False)
t0 = time.perf_counter()
m.h.step(n, False)
print("%s %dx%d step(%d): %.2f us/iter, %s, %s" % (prob, q, q, n, (time.perf_counter() - t0) / n * 1e6, m.h.pass_structure(), m.h.kernel_variant()))
