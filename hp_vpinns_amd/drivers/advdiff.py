"""Advection-diffusion coefficient identification driver: the module constants and `__main__`
block of the reference script restated (P3:31-54, 347-494).

u_t + V u_x = eps u_xx on x in [-1,1], t in [0,T]; u(x,0) = -sin(pi x); u(+-1,t) = 0; exact eps =
gamma/pi with gamma = 0.1 (P3:41-42) is the quantity to IDENTIFY: the network and a trainable
epsilon (init 1.0, P3:63) are fitted to the variational residual plus boundary/initial data plus 15
interior measurements of the exact solution (P3:463-483).  The exact solution is the 801-term
Fourier series of P3:416-445.  `.mat` export and plots (P3:500-697) are not restated.
"""
import argparse

import numpy as np

from ..quadrature import GaussLobattoJacobiWeights
from ..sampling import lhs

gamma = 0.1
epsilon = gamma / (np.pi)        # P3:42 (exact value; the trainable one starts at 1.0)
V = 1.0
T = 1


def u_initial(x, t):                                             # P3:351-353
    return -np.sin(np.pi * x)


def u_ext(x, t, trunc=800):
    """Analytical solution as a Fourier series (P3:416-445); x, t scalars or equal-shape arrays."""
    x = np.asarray(x, dtype=np.float64)[..., None]
    t = np.asarray(t, dtype=np.float64)[..., None]
    p = np.arange(0, trunc + 1.0)
    D = epsilon
    c0 = 16 * np.pi ** 2 * D ** 3 * V * np.exp(V / D / 2 * (x - V * t / 2))
    c1_n = (-1) ** p * 2 * p * np.sin(p * np.pi * x) * np.exp(-D * p ** 2 * np.pi ** 2 * t)
    c1_d = V ** 4 + 8 * (V * np.pi * D) ** 2 * (p ** 2 + 1) + 16 * (np.pi * D) ** 4 * (p ** 2 - 1) ** 2
    c1 = np.sinh(V / D / 2) * np.sum(c1_n / c1_d, axis=-1, keepdims=True)
    c2_n = (-1) ** p * (2 * p + 1) * np.cos((p + 0.5) * np.pi * x) * np.exp(-D * (2 * p + 1) ** 2 * np.pi ** 2 * t / 4)
    c2_d = V ** 4 + (V * np.pi * D) ** 2 * (8 * p ** 2 + 8 * p + 10) + (np.pi * D) ** 4 * (4 * p ** 2 + 4 * p - 3) ** 2
    c2 = np.cosh(V / D / 2) * np.sum(c2_n / c2_d, axis=-1, keepdims=True)
    c = (c0 * (c1 + c2))[..., 0]
    return np.where(t[..., 0] == 0, u_initial(x[..., 0], t[..., 0]), c)   # P3:442-443


def setup(N_el_x=1, N_el_t=1, N_test_x=5, N_test_t=5, N_quad=10, N_bound=80, NPf=500, NPu_inter=5, seed=1234,
          with_test_grid=True):
    np.random.seed(seed)                                         # P3:27
    col = lambda a, v: np.full((len(a), 1), float(v))            # noqa: E731
    t_up = T * lhs(1, N_bound)                                   # P3:358-364: x = +1, u = 0
    x_up_train, u_up_train = np.hstack((col(t_up, 1), t_up)), col(t_up, 0.0)
    t_lo = T * lhs(1, N_bound)                                   # P3:366-372: x = -1, u = 0
    x_lo_train, u_lo_train = np.hstack((col(t_lo, -1), t_lo)), col(t_lo, 0.0)
    x_in = 2 * lhs(1, N_bound) - 1                               # P3:374-380: t = 0, u = -sin(pi x)
    x_in_train, u_in_train = np.hstack((x_in, col(x_in, 0))), u_initial(x_in, col(x_in, 0))
    grid_pt = lhs(2, NPf)                                        # P3:387-391
    XT_f_train = np.hstack(((2 * grid_pt[:, 0] - 1)[:, None], (T * grid_pt[:, 1])[:, None]))
    X_quad, WX_quad = GaussLobattoJacobiWeights(N_quad, 0, 0)    # P3:395-400
    xx, tt = np.meshgrid(X_quad, X_quad)
    wxx, wtt = np.meshgrid(WX_quad, WX_quad)
    XT_quad_train = np.hstack((xx.flatten()[:, None], tt.flatten()[:, None]))
    WXT_quad_train = np.hstack((wxx.flatten()[:, None], wtt.flatten()[:, None]))
    delta_x, delta_t = 2 / N_el_x, T / N_el_t                    # P3:404-410
    grid_x = np.asarray([-1 + i * delta_x for i in range(N_el_x + 1)])
    grid_t = np.asarray([0 + i * delta_t for i in range(N_el_t + 1)])
    N_testfcn_total = [N_el_x * [N_test_x], N_el_t * [N_test_t]]
    out = {}
    if with_test_grid:                                           # P3:448-458 (x fastest)
        xtest = np.linspace(-1, 1, 256)
        ttest = np.arange(0, T + 0.01, 0.01)
        Xg, Tg = np.meshgrid(xtest, ttest)
        out["XT_test"] = np.hstack((Xg.flatten()[:, None], Tg.flatten()[:, None]))
        out["u_test"] = u_ext(out["XT_test"][:, 0], out["XT_test"][:, 1])[:, None]
    # interior measurements for the inverse problem (P3:463-483)
    xs, ts = [], []
    for xv in (-0.5, 0.0, 0.5):
        xs.append(np.full((NPu_inter, 1), xv))
        ts.append(T * lhs(1, NPu_inter))
    XT_u_inter_train = np.hstack((np.concatenate(xs), np.concatenate(ts)))
    u_inter_train = u_ext(XT_u_inter_train[:, 0], XT_u_inter_train[:, 1])[:, None]
    XT_u_train = np.concatenate((x_up_train, x_lo_train, x_in_train, XT_u_inter_train))
    u_train = np.concatenate((u_up_train, u_lo_train, u_in_train, u_inter_train))
    out.update(XT_u_train=XT_u_train, u_train=u_train, XT_f_train=XT_f_train, XT_quad_train=XT_quad_train,
               WXT_quad_train=WXT_quad_train, T_quad=X_quad, WT_quad=WX_quad, grid_x=grid_x, grid_t=grid_t,
               N_testfcn_total=N_testfcn_total)
    return out


def build_model(s, Net_layer, var_form=0, LR=0.001, init_params=None, backend="auto", **kw):
    from ..vpinn import VPINNAdvDiff
    XT_test = s.get("XT_test", s["XT_u_train"])
    u_test = s.get("u_test", s["u_train"])
    lb, ub = XT_test.min(0), XT_test.max(0)                      # P3:460-461
    return VPINNAdvDiff(s["XT_u_train"], s["u_train"], s["XT_f_train"], s["XT_quad_train"], s["WXT_quad_train"],
                        s["T_quad"], s["WT_quad"], s["grid_x"], s["grid_t"], s["N_testfcn_total"], XT_test, u_test,
                        Net_layer, lb, ub, var_form=var_form, LR=LR, V=V, init_params=init_params, backend=backend,
                        **kw)                                    # P3:488-489


def export_mat(path, s, u_record, u_records_iterhis, total_record, total_time_train):
    """The reference's `<case>_record.mat` (P3:500-508): test grid, exact solution, element grids, the best prediction of the last
    tenth of the run, the records [iteration, loss, epsilon, 1] and the training time.  (The reference issues one `savemat` per
    variable into the same open file, which leaves only fragments readable; here ONE call writes all eight variables under the
    reference's names.)"""
    import scipy.io
    rec = np.array([[float(r[0]), float(r[1]), float(np.ravel(r[2])[0]), float(r[3])] for r in total_record]) if len(total_record) else np.zeros((0, 4))
    scipy.io.savemat(path, {"x_test": s["XT_test"], "u_test": s["u_test"], "grid_x": s["grid_x"], "grid_t": s["grid_t"],
                            "u_pred": np.zeros((0, 1)) if u_record is None else u_record,
                            "u_pred_his": np.asarray(u_records_iterhis, dtype=np.float64) if len(u_records_iterhis) else np.zeros((0, 1)),
                            "total": rec, "total_time_train": float(total_time_train)})


def run(LR=0.001, Opt_Niter=1500 + 1, Opt_tresh=2e-11, var_form=0, Net_layer=None, N_el_x=1, N_el_t=1, N_test_x=5,
        N_test_t=5, N_quad=10, N_bound=80, init_params=None, backend="auto", verbose=True, mat_path=None):
    """P3:31-54 hyper-parameters (reference defaults) -> identified epsilon, prediction, L2 error."""
    Net_layer = [2] + [5] * 3 + [1] if Net_layer is None else Net_layer        # P3:46
    s = setup(N_el_x, N_el_t, N_test_x, N_test_t, N_quad, N_bound)
    model = build_model(s, Net_layer, var_form, LR, init_params, backend)
    error_record, total_record, u_record, u_his, t_train = model.train(Opt_Niter, Opt_tresh)   # P3:493-494
    if mat_path is not None:                                                    # P3:500-508
        export_mat(mat_path, s, u_record, u_his, total_record, t_train)
    u_pred = model.predict()
    err = np.linalg.norm(s["u_test"] - u_pred, 2) / np.linalg.norm(s["u_test"], 2)
    eps_id = float(model.epsilon[0])
    if verbose:
        print("identified epsilon: %.6f (exact %.6f)   relative L2 error of u: %.3e   train time %.2fs"
              % (eps_id, epsilon, err, t_train))
    return dict(model=model, u_pred=u_pred, rel_l2=err, epsilon=eps_id, total_record=total_record, setup=s)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=1501)
    ap.add_argument("--elements-x", type=int, default=1)
    ap.add_argument("--quad", type=int, default=10)
    ap.add_argument("--width", type=int, default=5)
    ap.add_argument("--var-form", type=int, default=0)
    ap.add_argument("--mat", default=None, help="write the reference's <case>_record.mat here (P3:500-508)")
    a = ap.parse_args()
    run(Opt_Niter=a.iters, N_el_x=a.elements_x, N_quad=a.quad, var_form=a.var_form, Net_layer=[2] + [a.width] * 3 + [1],
        mat_path=a.mat)
