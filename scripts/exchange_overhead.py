#!/usr/bin/env python3
"""Cost of the exchange kernel itself: config 4 on one GPU with the in-library exchange connected to a 1-rank
"world" (its own mailbox): iteration = forward, projection+reverse, finalize, exchange+Adam  vs  the single-GPU
iteration whose finalize kernel applies Adam."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd.drivers import poisson2d  # noqa: E402
from hp_vpinns_amd.init import xavier_init  # noqa: E402

L = [2, 20, 20, 20, 1]
s = poisson2d.setup(N_el_x=16, N_el_y=16, N_test_x=10, N_test_y=10, N_quad=20, with_test_grid=False)
for p2p in (False, True):
    m = poisson2d.build_model(s, L, init_params=xavier_init(L, 1234))
    if p2p:
        m.h.p2p_connect(m.h.p2p_export(1, 0))
        out, timed_out = m.h.p2p_selftest(m.h.reduce_buffer()[1])
        assert not timed_out and abs(out[0] - 1.0) < 1e-15
    m.h.step(200, False)
    t0 = time.perf_counter()
    m.h.step(2000, False)
    print(("exchange+Adam kernel (1-rank mailbox)" if p2p else "Adam fused into finalize         ") +
          ": %.1f us/iter" % ((time.perf_counter() - t0) / 2000 * 1e6))
