#!/usr/bin/env python3
"""Stand-alone run of the projection (residual + adjoint) kernel on the scaled synthetic batch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hp_vpinns_amd import _lib  # noqa: E402
from hp_vpinns_amd.quadrature import GaussLobattoJacobiWeights  # noqa: E402
from hp_vpinns_amd.testfcn import tables_1d  # noqa: E402

ne = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
adj = (sys.argv[3] != "0") if len(sys.argv) > 3 else True     # 0: residual only (SURVEY.md 8d byte count)
q, nt = 20, 10
h = _lib.Handle(_lib.PDE_POISSON2D, 1, _lib.ACT_TANH, [2, 20, 20, 20, 1], lossb_weight=10)
x, w = GaussLobattoJacobiWeights(q, 0, 0)
h.set_quadrature(x, w, x, w)
h.set_tables(tables_1d(nt, x), tables_1d(nt, x))
ms, by = h.bench_projection(ne, reps, do_adjoint=adj)
print(("residual+adjoint" if adj else "residual only") + " projection n_elem=%d: %.3f ms/launch, %.1f GB/s algorithmic (%.1f%% of 8 TB/s), %.0f bytes/launch"
      % (ne, ms, by / ms / 1e6, by / ms / 1e6 / 80, by))
