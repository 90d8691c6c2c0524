"""Poisson 1-D driver: the `__main__` block of the reference script restated (P1:227-337).

u(x) = 0.1 sin(8 pi x) + tanh(80 x) on [-1,1], -u'' = f (P1:243-257); N_Element elements
(uniform grid, or the published 3-element grid [-1,-0.1,0.1,1], P1:264-273);
F_ext[e][k] = J sum_q w_q f(x_q) phi_k(xi_q) (P1:275-294); two boundary points; `VPINN(...)`
called with the reference's argument list (P1:333-334).  Plotting (P1:339-429) is not restated.
"""
import argparse

import numpy as np

from ..quadrature import GaussLobattoJacobiWeights
from ..sampling import lhs
from ..testfcn import Test_fcn

omega, amp, r1 = 8 * np.pi, 1, 80


def u_ext(x):                                                    # P1:248-250
    return amp * (0.1 * np.sin(omega * x) + np.tanh(r1 * x))


def f_ext(x):                                                    # P1:252-254
    gtemp = -0.1 * (omega ** 2) * np.sin(omega * x) - (2 * r1 ** 2) * (np.tanh(r1 * x)) / ((np.cosh(r1 * x)) ** 2)
    return -amp * gtemp


def setup(N_Element=1, N_testfcn=60, N_Quad=80, N_F=500, seed=1234, N_testfcn_total=None):
    """N_testfcn_total: per-element number of test functions (the list the reference builds at P1:268 / 273 and a user
    edits for p-refinement); F_ext_total / U_ext_total are then lists of columns of different lengths."""
    np.random.seed(seed)                                         # P1:26
    x_quad, w_quad = GaussLobattoJacobiWeights(N_Quad, 0, 0)     # P1:260
    NE = N_Element
    x_l, x_r = -1, 1
    delta_x = (x_r - x_l) / NE
    grid = np.asarray([x_l + i * delta_x for i in range(NE + 1)])   # P1:267
    if N_Element == 3:                                           # P1:270-273
        grid = np.array([-1, -0.1, 0.1, 1])
        NE = 3
    ntot = np.array(NE * [N_testfcn]) if N_testfcn_total is None else np.asarray(N_testfcn_total, dtype=int)   # P1:268
    if ntot.size != NE:
        raise ValueError("N_testfcn_total needs one entry per element")
    testfcn = Test_fcn(int(ntot.max()), x_quad)                  # (N_test, Q); element e uses its first ntot[e] rows (P1:280-281)
    U_ext_total, F_ext_total = [], []
    for e in range(NE):                                          # P1:277-291
        x_quad_element = grid[e] + (grid[e + 1] - grid[e]) / 2 * (x_quad + 1)
        jacobian = (grid[e + 1] - grid[e]) / 2
        te = testfcn[:ntot[e]]
        U_ext_total.append((jacobian * (te * (w_quad * u_ext(x_quad_element))).sum(axis=1))[:, None])
        F_ext_total.append((jacobian * (te * (w_quad * f_ext(x_quad_element))).sum(axis=1))[:, None])
    if len(set(ntot.tolist())) == 1:                             # P1:293-294 (dense when every element has the same count)
        U_ext_total, F_ext_total = np.asarray(U_ext_total), np.asarray(F_ext_total)
    X_u_train = np.asarray([-1.0, 1.0])[:, None]                 # P1:298-299
    u_train = u_ext(X_u_train)
    X_f_train = (2 * lhs(1, N_F) - 1)                            # P1:303
    f_train = f_ext(X_f_train)
    delta_test = 0.001                                           # P1:318-324
    xtest = np.arange(-1, 1 + delta_test, delta_test)
    X_test = xtest[:, None]
    u_test = u_ext(X_test)
    return dict(grid=grid, F_ext_total=F_ext_total, U_ext_total=U_ext_total, X_quad_train=x_quad[:, None],
                W_quad_train=w_quad[:, None], X_u_train=X_u_train, u_train=u_train, X_f_train=X_f_train,
                f_train=f_train, X_test=X_test, u_test=u_test)


def run(LR=0.001, Opt_Niter=1000 + 1, Opt_tresh=2e-32, var_form=1, N_Element=1, Net_layer=None, N_testfcn=60,
        N_Quad=80, N_F=500, lossb_weight=1, init_params=None, backend="auto", verbose=True):
    """P1:231-240 hyper-parameters (reference defaults) -> trained model, prediction and L2 error."""
    from ..vpinn import VPINN1D
    Net_layer = [1] + [20] * 4 + [1] if Net_layer is None else Net_layer      # P1:236
    s = setup(N_Element, N_testfcn, N_Quad, N_F)
    total_record = []
    model = VPINN1D(s["X_u_train"], s["u_train"], s["X_quad_train"], s["W_quad_train"], s["F_ext_total"], s["grid"],
                    s["X_test"], s["u_test"], Net_layer, s["X_f_train"], s["f_train"], var_form=var_form,
                    lossb_weight=lossb_weight, LR=LR, init_params=init_params, backend=backend,
                    total_record=total_record)                   # P1:333-334
    model.train(Opt_Niter, Opt_tresh)                            # P1:336
    u_pred = model.predict(s["X_test"])                          # P1:337
    err = np.linalg.norm(s["u_test"] - u_pred, 2) / np.linalg.norm(s["u_test"], 2)
    if verbose:
        print("relative L2 error of u: %.3e   final recorded loss: %.3e" % (err, total_record[-1][1]))
    return dict(model=model, u_pred=u_pred, rel_l2=err, total_record=total_record, setup=s)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=1001)
    ap.add_argument("--elements", type=int, default=1)
    ap.add_argument("--var-form", type=int, default=1)
    a = ap.parse_args()
    run(Opt_Niter=a.iters, N_Element=a.elements, var_form=a.var_form)
