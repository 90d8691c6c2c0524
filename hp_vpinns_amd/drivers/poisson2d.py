"""Poisson 2-D driver: the `__main__` block of the reference script restated (P2:265-435).

u = (0.1 sin(2 pi x) + tanh(10 x)) sin(2 pi y) on [-1,1]^2, Laplace(u) = f (P2:296-310);
N_el_x x N_el_y elements, tensor GLL rule with N_quad points per direction, N_test_x x N_test_y
Legendre-difference test functions per element; F_ext[ex,ey][k][r] (P2:381-414); 4 x N_bound
boundary points by LHS (P2:314-347).  Plotting (P2:443-534) is not restated.
"""
import argparse

import numpy as np

from ..quadrature import GaussLobattoJacobiWeights
from ..sampling import lhs
from ..testfcn import Test_fcn

omegax = omegay = 2 * np.pi
r1 = 10


def u_ext(x, y):                                                 # P2:300-302
    return (0.1 * np.sin(omegax * x) + np.tanh(r1 * x)) * np.sin(omegay * (y))


def f_ext(x, y):                                                 # P2:304-307
    return (-0.1 * (omegax ** 2) * np.sin(omegax * x) - (2 * r1 ** 2) * (np.tanh(r1 * x)) / ((np.cosh(r1 * x)) ** 2)) \
        * np.sin(omegay * (y)) + (0.1 * np.sin(omegax * x) + np.tanh(r1 * x)) * (-omegay ** 2 * np.sin(omegay * (y)))


def setup(N_el_x=4, N_el_y=4, N_test_x=5, N_test_y=5, N_quad=10, N_bound=80, N_residual=100, seed=1234,
          with_test_grid=True, assemble="host", device=0):
    np.random.seed(seed)                                         # P2:23
    ones = lambda a, v: np.full((len(a), 1), float(v))           # noqa: E731
    x_up = 2 * lhs(1, N_bound) - 1                               # P2:314-320
    x_up_train, u_up_train = np.hstack((x_up, ones(x_up, 1))), u_ext(x_up, ones(x_up, 1))
    x_lo = 2 * lhs(1, N_bound) - 1                               # P2:322-328
    x_lo_train, u_lo_train = np.hstack((x_lo, ones(x_lo, -1))), u_ext(x_lo, ones(x_lo, -1))
    y_ri = 2 * lhs(1, N_bound) - 1                               # P2:330-336
    x_ri_train, u_ri_train = np.hstack((ones(y_ri, 1), y_ri)), u_ext(ones(y_ri, 1), y_ri)
    y_le = 2 * lhs(1, N_bound) - 1                               # P2:338-344
    x_le_train, u_le_train = np.hstack((ones(y_le, -1), y_le)), u_ext(ones(y_le, -1), y_le)
    X_u_train = np.concatenate((x_up_train, x_lo_train, x_ri_train, x_le_train))
    u_train = np.concatenate((u_up_train, u_lo_train, u_ri_train, u_le_train))
    grid_pt = lhs(2, N_residual)                                 # P2:349-354 (PINN residual points)
    xf, yf = 2 * grid_pt[:, 0] - 1, 2 * grid_pt[:, 1] - 1
    X_f_train = np.hstack((xf[:, None], yf[:, None]))
    f_train = f_ext(xf, yf)[:, None]
    X_quad, WX_quad = GaussLobattoJacobiWeights(N_quad, 0, 0)    # P2:357-364
    xx, yy = np.meshgrid(X_quad, X_quad)
    wxx, wyy = np.meshgrid(WX_quad, WX_quad)
    XY_quad_train = np.hstack((xx.flatten()[:, None], yy.flatten()[:, None]))
    WXY_quad_train = np.hstack((wxx.flatten()[:, None], wyy.flatten()[:, None]))
    NE_x, NE_y = N_el_x, N_el_y                                  # P2:368-376
    delta_x, delta_y = 2 / NE_x, 2 / NE_y
    grid_x = np.asarray([-1 + i * delta_x for i in range(NE_x + 1)])
    grid_y = np.asarray([-1 + i * delta_y for i in range(NE_y + 1)])
    N_testfcn_total = [NE_x * [N_test_x], NE_y * [N_test_y]]
    tx, ty = Test_fcn(N_test_x, X_quad), Test_fcn(N_test_y, X_quad)      # (Nt, Q)
    ax, by = tx * WX_quad, ty * WX_quad
    F_ext_total = np.empty((NE_x, NE_y, N_test_y, N_test_x))
    if assemble == "device":                                     # same numbers from the projection kernel
        from ..rhs import assemble_F_ext_2d
        F_ext_total = assemble_F_ext_2d(f_ext, grid_x, grid_y, N_test_x, N_test_y, N_quad, device=device)
    for ex in range(NE_x if assemble != "device" else 0):        # P2:386-411
        xq = grid_x[ex] + (grid_x[ex + 1] - grid_x[ex]) / 2 * (X_quad + 1)
        for ey in range(NE_y):
            yq = grid_y[ey] + (grid_y[ey + 1] - grid_y[ey]) / 2 * (X_quad + 1)
            jacobian = ((grid_x[ex + 1] - grid_x[ex]) / 2) * ((grid_y[ey + 1] - grid_y[ey]) / 2)
            fq = f_ext(xq[None, :], yq[:, None])                 # [j (y)][i (x)]
            F_ext_total[ex, ey] = jacobian * (by @ fq @ ax.T)    # [k][r]
    out = dict(X_u_train=X_u_train, u_train=u_train, X_f_train=X_f_train, f_train=f_train,
               XY_quad_train=XY_quad_train, WXY_quad_train=WXY_quad_train, F_ext_total=F_ext_total,
               grid_x=grid_x, grid_y=grid_y, N_testfcn_total=N_testfcn_total)
    if with_test_grid:
        delta_test = 0.01                                        # P2:418-426
        xtest = np.arange(-1, 1 + delta_test, delta_test)
        ytest = np.arange(-1, 1 + delta_test, delta_test)
        Xg, Yg = np.meshgrid(xtest, ytest)                       # x fastest, like the reference's nested list
        out["X_test"] = np.hstack((Xg.flatten()[:, None], Yg.flatten()[:, None]))
        out["u_test"] = u_ext(out["X_test"][:, 0:1], out["X_test"][:, 1:2])
    return out


def build_model(s, Net_layer, var_form=1, init_params=None, backend="auto", loss_his=None, **kw):
    from ..vpinn import VPINN2D
    X_test = s.get("X_test", s["X_u_train"])
    u_test = s.get("u_test", s["u_train"])
    return VPINN2D(s["X_u_train"], s["u_train"], s["X_f_train"], s["f_train"], s["XY_quad_train"], s["WXY_quad_train"],
                   None, s["F_ext_total"], s["grid_x"], s["grid_y"], s["N_testfcn_total"], X_test, u_test, Net_layer,
                   var_form=var_form, init_params=init_params, backend=backend, loss_his=loss_his, **kw)   # P2:430-431


def run(scheme="VPINNs", Net_layer=None, var_form=1, N_el_x=4, N_el_y=4, N_test_x=5, N_test_y=5, N_quad=10,
        N_bound=80, N_residual=100, n_iter=10000 + 1, init_params=None, backend="auto", record_every=1, verbose=True):
    """P2:279-288 hyper-parameters (reference defaults) -> trained model, prediction and L2 error."""
    Net_layer = [2] + [5] * 3 + [1] if Net_layer is None else Net_layer        # P2:280
    s = setup(N_el_x, N_el_y, N_test_x, N_test_y, N_quad, N_bound, N_residual)
    loss_his = []
    model = build_model(s, Net_layer, var_form, init_params, backend, loss_his, scheme=scheme)
    model.train(n_iter, record_every=record_every)               # P2:434
    u_pred = model.predict()                                     # P2:435
    err = np.linalg.norm(s["u_test"] - u_pred, 2) / np.linalg.norm(s["u_test"], 2)
    if verbose:
        print("relative L2 error of u: %.3e   final loss: %.3e" % (err, loss_his[-1]))
    return dict(model=model, u_pred=u_pred, rel_l2=err, loss_his=loss_his, setup=s)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10001)
    ap.add_argument("--elements", type=int, default=4)
    ap.add_argument("--var-form", type=int, default=1)
    ap.add_argument("--width", type=int, default=5)
    ap.add_argument("--record-every", type=int, default=1)
    a = ap.parse_args()
    run(n_iter=a.iters, N_el_x=a.elements, N_el_y=a.elements, var_form=a.var_form,
        Net_layer=[2] + [a.width] * 3 + [1], record_every=a.record_every)
