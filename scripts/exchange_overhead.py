#!/usr/bin/env python3
"""Cost of the multi-GPU iteration tail on ONE GPU: the shard one of N GPUs owns (config 4, 256 / N elements) with the
in-library exchange connected to a 1-rank "world" -- iteration = whole-iteration kernel, finalize, exchange, Adam -- next to the
single-GPU iteration whose finalize kernel applies Adam.  rccl: ncclAllReduce on a 1-rank communicator, the update deferred into the
next iteration's kernels (what the default multi-GPU path launches, minus the xGMI hops: two launches + one collective); rccl +
k_adam: the same with HPV_NO_DEFERRED_ADAM=1 (a k_adam launch behind every collective, rounds 3-4); p2p: the one-kernel mailbox
exchange + Adam.   exchange_overhead.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd.drivers import poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

L = [2, 20, 20, 20, 1]
print("| shard (elements) | Adam fused into finalize | 1-rank RCCL all-reduce, update deferred into the next iteration | 1-rank RCCL all-reduce + k_adam | 1-rank mailbox exchange + Adam |\n|---|---|---|---|---|")
for n in (1, 2, 4, 8):
    s = poisson2d.setup(N_el_x=16, N_el_y=16 // n, N_test_x=10, N_test_y=10, N_quad=20, with_test_grid=False)
    row = []
    for mode in ("none", "rccl", "rccl-k_adam", "p2p"):
        if mode == "rccl-k_adam":
            os.environ["HPV_NO_DEFERRED_ADAM"] = "1"
        m = poisson2d.build_model(s, L, init_params=xavier_init(L, 1234))
        os.environ.pop("HPV_NO_DEFERRED_ADAM", None)
        if mode == "p2p":
            m.h.p2p_connect(m.h.p2p_export(1, 0))
            out, timed_out = m.h.p2p_selftest(m.h.reduce_buffer()[1])
            assert not timed_out and abs(out[0] - 1.0) < 1e-15
        elif mode.startswith("rccl"):
            m.h.rccl_connect(1, 0, m.h.rccl_unique_id())
        m.h.step(200, False)
        m.h.sync()
        t0 = time.perf_counter()
        m.h.step(4000, False)
        m.h.sync()
        row.append((time.perf_counter() - t0) / 4000 * 1e6)
        assert m.h.exchange_in_use() == mode.split("-")[0]
    print("| 1/%d (%d) | %.1f us | %.1f us (%+.1f) | %.1f us (%+.1f) | %.1f us |" % (n, 256 // n, row[0], row[1], row[1] - row[0], row[2], row[2] - row[0], row[3]))
