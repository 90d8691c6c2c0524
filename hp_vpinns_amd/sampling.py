"""Latin-hypercube sampling used by the drivers for boundary / collocation points.

The reference calls `pyDOE.lhs(n, samples)` (P1:303; P2:314-351; P3:358-391,467-475).
pyDOE is a third-party package that is neither vendored nor pinned by the reference
and is not installed in this image, so this is a restatement of the published
"classic" LHS algorithm (stratify [0,1) into `samples` equal bins per dimension, one
uniform draw per bin, independent random permutation per dimension) driven by the
global numpy RNG exactly as the drivers seed it (`np.random.seed(1234)`, P1:26).
Parity at this boundary is UNPINNED upstream; the sampled points are treated as
*inputs* of the hot path and are stored in the golden fixtures.
"""
import numpy as np


def lhs(n, samples=None):
    if samples is None:
        samples = n
    cut = np.linspace(0, 1, samples + 1)
    u = np.random.rand(samples, n)
    a = cut[:samples]
    b = cut[1:samples + 1]
    rdpoints = np.zeros_like(u)
    for j in range(n):
        rdpoints[:, j] = u[:, j] * (b - a) + a
    H = np.zeros_like(rdpoints)
    for j in range(n):
        order = np.random.permutation(range(samples))
        H[:, j] = rdpoints[order, j]
    return H
