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


MFMA_WIDTH = 20   # hidden width of the hand-tuned MFMA kernels (csrc/kernels_mfma.hip and the whole-iteration kernels)
WIDE_WIDTHS = (24, 32, 40, 48, 64)   # hidden widths the width-generic MFMA kernels are instantiated for (csrc/kernels_wide.hip)


def device_width(hidden):
    """The uniform hidden width the device runs a network with these hidden layer widths at: 20 when none is wider (the
    whole-iteration kernels), else the smallest instantiated width that holds the widest layer; None beyond 64 (generic kernels)."""
    w = max(hidden)
    for cand in (MFMA_WIDTH,) + WIDE_WIDTHS:
        if w <= cand:
            return cand
    return None


def pad_plan(layers, extra=0, width=None, max_hidden=None):
    """Zero-padding of a network onto kernels of ONE hidden width (`width`; default `device_width`: 20 for narrow networks --
    the reference defaults are 5 wide, P2:280, P3:46 --, else the next instantiated width, also for non-uniform hidden
    layers).  Returns (padded_layers, index) with theta_padded[index] = theta, or None when no padding is needed (every
    hidden layer already has that width) or possible (too wide, too deep).

    The padding is exact, not an approximation: a padded neuron has zero incoming weights and bias, so it outputs
    act(0) = 0 (tanh and sin) with zero tangents, its outgoing weights are zero, and every gradient entry that belongs
    to padding is exactly zero (h_pad = 0 kills dW rows, hbar_pad = 0 kills dW columns and db) -- TF1 Adam leaves them
    at zero.  Sums only gain exact-zero terms."""
    hidden = layers[1:-1]
    if not hidden or layers[-1] != 1 or layers[0] > 2:
        return None
    if width is None:
        width = device_width(hidden)
    if width is None:
        return None
    if max_hidden is None:          # depths the MFMA kernels are instantiated for: 6 at the widths 20, 24, 32 (kernels_mfma.hip, kernels_wide.hip), 4 beyond
        max_hidden = 6 if width <= 32 else 4
    if len(hidden) > max_hidden:
        return None
    if width is None or any(w > width for w in hidden) or all(w == width for w in hidden):
        return None
    padded = [layers[0]] + [width] * len(hidden) + [1]
    idx, o = [], 0
    for l in range(len(layers) - 1):
        i, j = layers[l], layers[l + 1]
        I, J = padded[l], padded[l + 1]
        idx += [o + r * J + c for r in range(i) for c in range(j)]
        o += I * J
        idx += [o + c for c in range(j)]
        o += J
    idx += [o + k for k in range(extra)]
    return padded, np.asarray(idx, dtype=np.int64)
