#!/usr/bin/env python3
"""What one rank owns at N = 1 / 2 / 4 / 8 of the SCALED batch (the config-4 element shape on a 64 x 64-element
grid, 4 096 / N elements per GPU), on ONE GPU: us per iteration with the single-GPU tail (Adam fused into finalize) and with the
multi-GPU tail on a 1-rank communicator (finalize, ncclAllReduce of the packed buffer, k_adam) -- the per-rank compute side of the
strong-scaling curve nobody has measured yet (verdict round 4, item 1b).   large_shards.py [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd.drivers import poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

L = [2, 20, 20, 20, 1]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
G = 2 * sum(L[i] * L[i + 1] for i in range(len(L) - 1))
print("| shard of N | elements | points | single-GPU tail: us / iteration | 1-rank RCCL tail: us / iteration | tail cost | algorithmic TFLOP/s (frac of 78.6) | kernels |\n|---|---|---|---|---|---|---|---|")
for n in (1, 2, 4, 8):
    s = poisson2d.setup(N_el_x=64, N_el_y=64 // n, N_test_x=10, N_test_y=10, N_quad=20, with_test_grid=False, assemble="device")
    row = []
    for mode in ("none", "rccl"):
        m = poisson2d.build_model(s, L, init_params=xavier_init(L, 1234))
        if mode == "rccl":
            m.h.rccl_connect(1, 0, m.h.rccl_unique_id())
        m.h.step(40, False)
        m.h.sync()
        t0 = time.perf_counter()
        m.h.step(iters, False)
        m.h.sync()
        row.append((time.perf_counter() - t0) / iters * 1e6)
        assert m.h.exchange_in_use() == mode
        variant = m.h.kernel_variant()
        del m
    ne = 64 * 64 // n
    fl = 3 * 3 * G * ne * 400 + 48000 * ne
    tf = fl / (row[0] * 1e-6) / 1e12
    print("| 1/%d | %d | %d | %.1f | %.1f | +%.1f | %.1f (%.3f) | `%s` |" % (n, ne, ne * 400, row[0], row[1], row[1] - row[0], tf, tf / 78.6, variant), flush=True)
