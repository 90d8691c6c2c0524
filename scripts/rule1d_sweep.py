#!/usr/bin/env python3
"""1-D h-refinement: a rule smaller than the 80-point instantiation of k_iter_tile, padded with zero-weight points (one
workgroup per element on 80 points / 60 test functions) against the rule as it is on the separate launches, over the element
count -- where hpv_rule1d_pad_max (csrc/hpv_mfma.h) has to draw the line.   rule1d_sweep.py [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd.drivers import poisson1d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402
from hp_vpinns_amd.vpinn import VPINN1D  # noqa: E402

L = [1, 20, 20, 20, 1]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1000


def run(ne, q, nt, mode):
    for k in ("HPV_NO_RULE_PADDING", "HPV_FORCE_RULE_PADDING"):
        os.environ.pop(k, None)
    if mode:
        os.environ[mode] = "1"
    s = poisson1d.setup(N_Element=ne, N_testfcn=nt, N_Quad=q)
    m = VPINN1D(s["X_u_train"], s["u_train"], s["X_quad_train"], s["W_quad_train"], s["F_ext_total"], s["grid"], s["X_test"],
                s["u_test"], L, var_form=1, init_params=xavier_init(L, 1234))
    n = iters if ne <= 4096 else max(50, iters // 8)
    m._step(min(50, n), False)
    m.h.sync()
    t0 = time.perf_counter()
    m._step(n, False)
    m.h.sync()
    us = (time.perf_counter() - t0) / n * 1e6
    v = m.h.kernel_variant().split("<")[0]
    del m
    return us, v


print("| elements | rule (points / test fcns) | padded onto 80 / 60: us / iteration | rule as it is: us / iteration | library's advice (default) |\n|---|---|---|---|---|")
for q, nt in ((10, 5), (20, 10), (40, 20), (60, 30)):
    for ne in (16, 128, 256, 512, 1024, 2048, 4096, 16384):
        if ne == 3:
            continue
        a, va = run(ne, q, nt, "HPV_FORCE_RULE_PADDING")
        b, vb = run(ne, q, nt, "HPV_NO_RULE_PADDING")
        c, vc = run(ne, q, nt, None)
        print("| %d | %d / %d | %.1f (%s) | %.1f (%s) | %.1f (%s) |" % (ne, q, nt, a, va, b, vb, c, vc), flush=True)
