"""Seeded parameter initialisation (semantics of `VPINN.initialize_NN` / `xavier_init`,
P1:110-126, P2:139-155, P3:200-216): W ~ truncated normal (|z| <= 2 sigma), sigma =
sqrt(2/(in+out)); zero biases `[1,out]`.  The reference's unused scalar `a = 0.01` never enters
the loss (no gradient, skipped by `minimize`) and is not represented.

TF1's seeded random stream cannot be reproduced outside TF, so the classes also accept explicit
`init_params` (packed as in include/hpvpinn.h); that is what every parity test uses.
"""
import numpy as np


def n_params(layers, extra=0):
    return sum(layers[l] * layers[l + 1] + layers[l + 1] for l in range(len(layers) - 1)) + extra


def xavier_init(layers, seed=1234, extra=()):
    rng = np.random.default_rng(seed)
    parts = []
    for l in range(len(layers) - 1):
        i, j = layers[l], layers[l + 1]
        std = np.sqrt(2.0 / (i + j))
        z = rng.standard_normal(i * j)
        bad = np.abs(z) > 2
        while bad.any():
            z[bad] = rng.standard_normal(bad.sum())
            bad = np.abs(z) > 2
        parts += [std * z, np.zeros(j)]
    parts.append(np.asarray(extra, dtype=np.float64))
    return np.concatenate(parts)


def unpack(theta, layers):
    """-> (weights [in,out], biases [1,out], trailing extras) views into theta."""
    ws, bs, o = [], [], 0
    for l in range(len(layers) - 1):
        i, j = layers[l], layers[l + 1]
        ws.append(theta[o:o + i * j].reshape(i, j)); o += i * j
        bs.append(theta[o:o + j].reshape(1, j)); o += j
    return ws, bs, theta[o:]
