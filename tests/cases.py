"""Shared builders: (oracle, product) pairs on identical golden inputs and initial parameters."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def theta0(layers, seed, extra=()):
    from hp_vpinns_amd.init import xavier_init
    return xavier_init([int(v) for v in layers], seed, extra=extra)


def p1_args(g, layers=None):
    L = [int(v) for v in (g["Net_layer"] if layers is None else layers)]
    return (g["X_u_train"], g["u_train"], g["X_quad_train"], g["W_quad_train"], g["F_ext_total"], g["grid"],
            g["X_test"], g["u_test"], L, g["X_f_train"], g["f_train"])


def p2_args(g, layers=None):
    L = [int(v) for v in (g["Net_layer"] if layers is None else layers)]
    ntf = [[int(v) for v in g["N_test_x"]], [int(v) for v in g["N_test_y"]]]
    return (g["X_u_train"], g["u_train"], g["X_f_train"], g["f_train"], g["XY_quad_train"], g["WXY_quad_train"],
            None, g["F_ext_total"], g["grid_x"], g["grid_y"], ntf, g["X_test_head"], g["u_test_head"], L)


def quad2d(q):
    from hp_vpinns_amd import GaussLobattoJacobiWeights
    X, W = GaussLobattoJacobiWeights(q, 0, 0)
    xx, yy = np.meshgrid(X, X)
    wxx, wyy = np.meshgrid(W, W)
    XY = np.hstack((xx.flatten()[:, None], yy.flatten()[:, None]))
    WXY = np.hstack((wxx.flatten()[:, None], wyy.flatten()[:, None]))
    return XY, WXY, X, W


def p3_args(g, layers=None):
    L = [int(v) for v in (g["Net_layer"] if layers is None else layers)]
    ntf = [[int(v) for v in g["N_test_x"]], [int(v) for v in g["N_test_t"]]]
    q = int(g["N_quad"])
    XT, WXT, _, _ = quad2d(q)
    # the fixture stores (a prefix of) the reference's own arrays: check the rebuilt rule against it
    n = g["XT_quad_train"].shape[0]
    assert np.abs(XT[:n] - g["XT_quad_train"]).max() < 1e-14
    assert np.abs(WXT[:n] - g["WXT_quad_train"]).max() < 1e-14
    rng = np.random.default_rng(7)
    XT_test = np.stack([rng.uniform(-1, 1, 64), rng.uniform(0, 1, 64)], axis=1)
    return (g["XT_u_train"], g["u_train"], g["XT_f_train"], XT, WXT, g["T_quad"], g["WT_quad"], g["grid_x"],
            g["grid_t"], ntf, XT_test, None, L, None, None)


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
