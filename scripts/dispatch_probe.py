#!/usr/bin/env python3
"""Which structure does a pass take, and (HPV_TRACE_DISPATCH=1 on libhpvpinn_testhooks.so) where does the whole-iteration kernel decline?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["HPV_TRACE_DISPATCH"] = "1"
from hp_vpinns_amd import _lib  # noqa: E402
from hp_vpinns_amd.drivers import advdiff, poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

with _lib.library(_lib.TEST_HOOKS_LIB_PATH):
    for (prob, q, nt, nex, ney, L) in [("p2vf0", 16, 8, 5, 3, [2, 20, 20, 1]), ("advf0", 16, 8, 5, 3, [2, 20, 20, 20, 1]), ("advf1", 16, 8, 5, 3, [2, 20, 20, 20, 1]),
                                        ("p2vf0", 16, 8, 16, 4, [2, 20, 20, 20, 1]), ("p2vf1", 20, 10, 17, 17, [2, 20, 20, 20, 1]), ("p2vf0", 16, 8, 25, 22, [2, 20, 20, 20, 1])]:
        if prob.startswith("p2"):
            s = poisson2d.setup(N_el_x=nex, N_el_y=ney, N_test_x=nt, N_test_y=nt, N_quad=q, N_bound=13, with_test_grid=False)
            m = poisson2d.build_model(s, L, var_form=int(prob[-1]), init_params=xavier_init(L, 1))
        else:
            s = advdiff.setup(N_el_x=nex, N_el_t=ney, N_test_x=nt, N_test_t=nt, N_quad=q, N_bound=11, with_test_grid=False)
            m = advdiff.build_model(s, L, var_form=int(prob[-1]), init_params=xavier_init(L, 1, extra=[0.9]))
        l3, g = m.loss_and_grad()
        print(prob, q, nex, ney, L, "->", m.h.pass_structure(), m.h.kernel_variant(), l3, flush=True)

# ---- timing probe: Poisson-2D var_form 0 on 16x16 elements of 16x16 points, per 100-iteration chunk, host- and device-assembled F ----
import time  # noqa: E402
for asm in ("host", "device"):
    s = poisson2d.setup(N_el_x=16, N_el_y=16, N_test_x=8, N_test_y=8, N_quad=16, with_test_grid=False, assemble=asm)
    L = [2, 20, 20, 20, 1]
    m = poisson2d.build_model(s, L, var_form=0, init_params=xavier_init(L, 1234))
    ts = []
    for _ in range(8):
        t0 = time.perf_counter()
        l3 = m.h.step(100, True)
        ts.append((time.perf_counter() - t0) / 100 * 1e6)
    print(asm, m.h.kernel_variant(), ["%.1f" % t for t in ts], l3, flush=True)
