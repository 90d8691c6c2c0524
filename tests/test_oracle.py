"""The CPU oracle itself: known answers, finite differences, TF1 Adam rule, recording semantics.
(The TF1 graph cannot run here, so these are the anchors of the unpinned part -- see oracle header.)"""
import numpy as np
import pytest
import torch

from cases import gold, p1_args, p2_args, p3_args, theta0
from oracle import vpinn_oracle as O


def test_oracle_tables_match_reference_fixtures():
    g = gold("testfcn")
    x = O.GaussLobattoJacobiWeights(10, 0, 0)[0][:, None]
    assert np.abs(O.Test_fcn(5, x) - g["phi_5_10"]).max() < 1e-14
    d1, d2 = O.dTest_fcn(5, x)
    assert np.abs(d1 - g["dphi_5_10"]).max() < 1e-13 and np.abs(d2 - g["d2phi_5_10"]).max() < 1e-12
    q = gold("quadrature")
    xs, ws = O.GaussLobattoJacobiWeights(80, 0, 0)
    assert np.abs(xs - q["gll_x_80"]).max() == 0 and np.abs(ws - q["gll_w_80"]).max() == 0


@pytest.mark.parametrize("tag,lv", [("poisson1d_cfg1", 290.4593764435546), ("poisson1d_ne3", 407.0420338281929)])
def test_zero_network_loss_1d(tag, lv):
    """loss(0) = sum_e mean(F_e^2) + lossb; the 3-element value 407.04+1 is the ~4e2 initial plateau of
    the reference's published Results/loss.pdf."""
    g = gold(tag)
    a = p1_args(g)
    o = O.OracleVPINN1D(*a, init_params=np.zeros(O.n_params(a[8])))
    loss, lossb, lossv = (float(v.detach()) for v in o.loss_parts())
    F = g["F_ext_total"]
    assert abs(lossv - (F ** 2).mean(axis=(1, 2)).sum()) < 1e-9
    assert abs(lossv - lv) < 1e-8 and abs(lossb - 1.0) < 1e-12 and abs(loss - lv - 1.0) < 1e-8


def test_zero_network_loss_2d():
    g = gold("poisson2d_default")
    a = p2_args(g)
    o = O.OracleVPINN2D(*a, init_params=np.zeros(O.n_params(a[13])))
    loss, lossb, lossv = (float(v.detach()) for v in o.loss_parts())
    assert abs(lossv - 60.15859233615944) < 1e-9 and abs(lossb - 0.24711041894014868) < 1e-12
    assert abs(loss - (10 * lossb + lossv)) < 1e-12        # P2:127


@pytest.mark.parametrize("kind,vf", [("1d", 1), ("1d", 2), ("1d", 3), ("2d", 0), ("2d", 1), ("2d", 2), ("adv", 0), ("adv", 1)])
def test_gradient_vs_finite_differences(kind, vf):
    if kind == "1d":
        a = p1_args(gold("poisson1d_small")); o = O.OracleVPINN1D(*a, var_form=vf, init_params=theta0(a[8], 3))
    elif kind == "2d":
        a = p2_args(gold("poisson2d_small")); o = O.OracleVPINN2D(*a, var_form=vf, init_params=theta0(a[13], 3))
    else:
        a = p3_args(gold("advdiff_small")); o = O.OracleVPINNAdvDiff(*a, var_form=vf, init_params=theta0(a[12], 3, extra=[0.8]))
    _, g = o.loss_and_grad()
    th = o.get_params()
    rng = np.random.default_rng(0)
    idx = list(rng.choice(th.size, 6, replace=False)) + [th.size - 1]
    for i in idx:
        h = 1e-6 * max(1.0, abs(th[i]))
        vals = []
        for s in (+1, -1):
            t = th.copy(); t[i] += s * h
            o.theta = torch.tensor(t, requires_grad=True)
            vals.append(float(o.loss_parts()[0]))
        fd = (vals[0] - vals[1]) / (2 * h)
        assert abs(fd - g[i]) < 1e-5 * max(1.0, abs(g[i])), (i, fd, g[i])


def test_tf1_adam_rule():
    """theta -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps): first step is lr*g/(|g|+eps*sqrt(1-b2)),
    i.e. NOT torch.optim.Adam's eps placement."""
    a = p1_args(gold("poisson1d_small"))
    o = O.OracleVPINN1D(*a, init_params=theta0(a[8], 5))
    th0 = o.get_params()
    _, g = o.loss_and_grad()
    o.adam_step()
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    expect = th0 - lr_t * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-8)
    assert np.abs(o.get_params() - expect).max() < 1e-15
    g2 = o.loss_and_grad()[1]
    th1 = o.get_params()
    o.adam_step()
    m = 0.9 * 0.1 * g + 0.1 * g2
    v = 0.999 * 0.001 * g * g + 0.001 * g2 * g2
    lr_t = 1e-3 * np.sqrt(1 - 0.999 ** 2) / (1 - 0.9 ** 2)
    assert np.abs(o.get_params() - (th1 - lr_t * m / (np.sqrt(v) + 1e-8))).max() < 1e-15


def test_recording_semantics():
    a = p1_args(gold("poisson1d_small"))
    o = O.OracleVPINN1D(*a, init_params=theta0(a[8], 5))
    rec = o.train(25, 0.0)
    assert [int(r[0]) for r in rec] == [0, 10, 20]          # every 10 iterations (P1:210)
    o2 = O.OracleVPINN1D(*a, init_params=theta0(a[8], 5))
    o2.adam_step()
    assert abs(float(o2.loss_parts()[0]) - rec[0][1]) < 1e-14   # recorded AFTER the update (P1:208-211)
    a2 = p2_args(gold("poisson2d_small"))
    o3 = O.OracleVPINN2D(*a2, init_params=theta0(a2[13], 5))
    assert len(o3.train(4)) == 4                             # every iteration (P2:243-244)


def test_advdiff_epsilon_moves_and_5tuple():
    a = p3_args(gold("advdiff_small"))
    o = O.OracleVPINNAdvDiff(*a, init_params=theta0(a[12], 5, extra=[1.0]))
    out = o.train(12, 0.0)
    assert len(out) == 5 and len(out[1]) == 2
    assert float(o.get_params()[-1]) != 1.0


def _same(o1, o2):
    o2.vectorized = True
    (l1, g1), (l2, g2) = o1.loss_and_grad(), o2.loss_and_grad()
    assert np.abs(np.array(l1) - np.array(l2)).max() < 1e-12 * abs(l1[0]), (l1, l2)
    assert np.abs(g1 - g2).max() < 1e-11 * np.abs(g1).max()
    for _ in range(3):          # and the Adam trajectories stay together
        o1.adam_step(); o2.adam_step()
    assert np.abs(o1.get_params() - o2.get_params()).max() < 1e-12


@pytest.mark.parametrize("vf", [0, 1, 2])
def test_vectorized_oracle_equals_structured_2d(vf):
    """The batched/einsum variant (CPU baseline B; the checker of the full-size GPU parity tests) computes the same
    loss and gradient as the reference-structured element loop, for every var_form."""
    a = p2_args(gold("poisson2d_small"))
    th = theta0(a[13], 17)
    _same(O.OracleVPINN2D(*a, var_form=vf, init_params=th), O.OracleVPINN2D(*a, var_form=vf, init_params=th))


@pytest.mark.parametrize("tag,vf", [("poisson1d_small", 1), ("poisson1d_small", 2), ("poisson1d_small", 3), ("poisson1d_ne3", 3)])
def test_vectorized_oracle_equals_structured_1d(tag, vf):
    a = p1_args(gold(tag))
    th = theta0(a[8], 18)
    # non-zero first bias: with zero biases the sin network is odd, d loss / d b_out cancels to round-off and Adam
    # (which normalises by sqrt(v) + 1e-8) would amplify that noise into an O(lr) difference of one parameter
    th[a[8][1]: 2 * a[8][1]] = 0.1 * np.arange(a[8][1])
    _same(O.OracleVPINN1D(*a, var_form=vf, init_params=th), O.OracleVPINN1D(*a, var_form=vf, init_params=th))


@pytest.mark.parametrize("vf", [0, 1])
def test_vectorized_oracle_equals_structured_advdiff(vf):
    a = p3_args(gold("advdiff_small"))
    th = theta0(a[12], 19, extra=[0.6])
    _same(O.OracleVPINNAdvDiff(*a, var_form=vf, init_params=th), O.OracleVPINNAdvDiff(*a, var_form=vf, init_params=th))


# ---- consistency anchors of the restated variational forms against reference-produced data ---------------------
# F_ext_total in the fixtures was assembled by the reference's own driver code from its f_ext; feeding the reference's
# exact solution u_ext (instead of the network) into the restated loss graph must therefore make the variational
# residual vanish -- to round-off for the strong form, to the quadrature error of the integration by parts for the weak
# forms, decreasing under refinement.  A wrong sign, Jacobian factor or test-function derivative in the restatement
# would leave an O(1) residual (as var_form 2 of the 2-D reference itself does for Jx != 1: it carries no 1/Jx^2).
def _exact_1d(X):
    return 1.0 * (0.1 * torch.sin(8 * np.pi * X) + torch.tanh(80 * X))          # P1:248-250 (amp 1, omega 8 pi, r1 80)


def _exact_2d(X):
    x, y = X[:, 0:1], X[:, 1:2]
    return (0.1 * torch.sin(2 * np.pi * x) + torch.tanh(10 * x)) * torch.sin(2 * np.pi * y)   # P2:300-302


@pytest.mark.parametrize("tag,vf,tol", [("poisson2d_default", 0, 1e-25), ("poisson2d_default", 1, 1e-6),
                                        ("poisson2d_cfg3", 1, 1e-9), ("poisson2d_cfg4", 1, 1e-25)])
def test_exact_solution_annihilates_the_2d_variational_residual(tag, vf, tol):
    g = gold(tag)
    a = p2_args(g, layers=[2, 5, 1])
    o = O.OracleVPINN2D(*a, var_form=vf)
    o.neural_net = lambda X, theta=None: _exact_2d(X)
    o.vectorized = tag != "poisson2d_default"
    loss, lossb, lossv = (float(v.detach()) for v in o.loss_parts())
    zero_net = (g["F_ext_total"] ** 2).mean(axis=(2, 3)).sum()
    assert lossv < tol * zero_net, (lossv, zero_net)
    assert lossb < 1e-25                                        # the boundary fixtures are u_ext on the boundary


@pytest.mark.parametrize("tag,vf", [("poisson1d_cfg2", 1), ("poisson1d_cfg2", 2), ("poisson1d_cfg2", 3)])
def test_exact_solution_annihilates_the_1d_variational_residual(tag, vf):
    g = gold(tag)
    a = p1_args(g, layers=[1, 5, 1])
    o = O.OracleVPINN1D(*a, var_form=vf)
    o.neural_net = lambda X, theta=None: _exact_1d(X)
    loss, lossb, lossv = (float(v.detach()) for v in o.loss_parts())
    zero_net = (g["F_ext_total"] ** 2).mean(axis=(1, 2)).sum()
    assert lossv < 1e-6 * zero_net, (vf, lossv, zero_net)
    assert lossb < 1e-25


# ---- two independent restatements: autograd double-backward vs closed-form Taylor channels + hand-derived
#      reverse pass (oracle/closed_form.py, the mathematics of the HIP kernels) must agree to round-off ----
def _rel(a, b):
    a, b = np.asarray(a, float).ravel(), np.asarray(b, float).ravel()
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("vf", [0, 1, 2])
def test_closed_form_equals_autograd_poisson2d(vf):
    from oracle import closed_form as CF
    a = p2_args(gold("poisson2d_small"))
    L = a[13]
    th = theta0(L, 5)
    l3, g = O.OracleVPINN2D(*a, var_form=vf, init_params=th).loss_and_grad()
    q = int(round(np.sqrt(a[4].shape[0])))
    l3c, gc = CF.loss_and_grad_2d(th, L, "poisson2d", vf, a[4][:q, 0], a[5][:q, 0], a[8], a[9], a[10][0][0], a[10][1][0],
                                  a[7], a[0], a[1], 10.0)
    assert _rel(l3c, l3) < 1e-13 and _rel(gc, g) < 1e-12


@pytest.mark.parametrize("vf", [1, 2, 3])
def test_closed_form_equals_autograd_poisson1d(vf):
    from oracle import closed_form as CF
    a = p1_args(gold("poisson1d_small"))
    L = a[8]
    th = theta0(L, 11)
    l3, g = O.OracleVPINN1D(*a, var_form=vf, init_params=th).loss_and_grad()
    l3c, gc = CF.loss_and_grad_1d(th, L, vf, a[2][:, 0], a[3][:, 0], a[5], a[4].shape[1], a[4], a[0], a[1], 1.0)
    assert _rel(l3c, l3) < 1e-13 and _rel(gc, g) < 1e-11


@pytest.mark.parametrize("vf", [0, 1])
def test_closed_form_equals_autograd_advdiff(vf):
    from oracle import closed_form as CF
    a = p3_args(gold("advdiff_small"))
    L = a[12]
    th = theta0(L, 9, extra=[0.7])
    l3, g = O.OracleVPINNAdvDiff(*a, var_form=vf, init_params=th).loss_and_grad()
    q = int(a[5].size)
    l3c, gc = CF.loss_and_grad_2d(th, L, "advdiff", vf, a[3][:q, 0], a[4][:q, 0], a[7], a[8], a[9][0][0], a[9][1][0], None,
                                  a[0], a[1], 10.0, V=1.0)
    assert _rel(l3c, l3) < 1e-13 and _rel(gc, g) < 1e-12          # the last entry is d loss / d epsilon


# ---- the C / OpenMP restatement (CPU baseline B of BASELINE.md section 3) pinned to the autograd oracle ------------
@pytest.mark.parametrize("tag,threads", [("poisson2d_small", 1), ("poisson2d_small", 3), ("poisson2d_cfg3", 4)])
def test_c_closed_form_baseline_equals_the_autograd_oracle(tag, threads):
    from oracle.cpu_baseline import CPoisson2D
    a = p2_args(gold(tag), layers=[2, 20, 20, 20, 1])
    th = theta0(a[13], 21)
    o = O.OracleVPINN2D(*a, var_form=1, init_params=th)
    o.vectorized = True
    c = CPoisson2D(a[0], a[1], a[4], a[5], a[7], a[8], a[9], a[13], th, threads=threads)
    (lo, go), (lc, gc) = o.loss_and_grad(), c.loss_and_grad()
    assert np.abs(np.array(lo) - lc).max() < 1e-12 * abs(lo[0]), (lo, lc)
    assert np.linalg.norm(go - gc) < 1e-10 * np.linalg.norm(go)
    hist = c.train(5)                       # TF1 Adam in C == the oracle's adam_step
    for k in range(5):
        l3 = o.adam_step()
        assert abs(l3[0] - hist[k, 0]) < 1e-10 * abs(l3[0])
    assert np.abs(o.get_params() - c.theta).max() < 1e-10


def test_oracle_1d_per_element_test_function_counts():
    """F_ext_total[e] of different lengths (P1:66-67 reads Ntest_element per element): the element loop's variational loss equals
    sum_e mean(R[e, :n_e]^2) of the dense residuals -- the truncation of the same tables, nothing else."""
    from hp_vpinns_amd.drivers import poisson1d
    from hp_vpinns_amd.init import xavier_init
    from oracle.vpinn_oracle import OracleVPINN1D
    counts = [12, 60, 5]
    L = [1, 20, 20, 1]
    th = xavier_init(L, 3)
    th[20:40] = 0.1
    for vf in (1, 2, 3):
        sr = poisson1d.setup(N_Element=3, N_testfcn_total=counts)
        sd = poisson1d.setup(N_Element=3)
        mk = lambda s: OracleVPINN1D(s["X_u_train"], s["u_train"], s["X_quad_train"], s["W_quad_train"], s["F_ext_total"],
                                     s["grid"], s["X_test"], s["u_test"], L, s["X_f_train"], s["f_train"], var_form=vf,
                                     init_params=th)
        orag, oden = mk(sr), mk(sd)
        assert orag.ragged and not oden.ragged
        for e, n in enumerate(counts):
            assert np.array_equal(sr["F_ext_total"][e], sd["F_ext_total"][e][:n])
        oden.vectorized = True
        oden.loss_parts()
        R = oden.last["R"]
        want = sum(float(np.mean(R[e, :n] ** 2)) for e, n in enumerate(counts))
        got = float(orag.loss_parts()[2])
        assert abs(got - want) <= 1e-13 * max(1.0, abs(want))


# ---- var_form 2 of the 2-D class (P2:108-115) against reference-produced F_ext_total, by Green's identity ----------------
# P2:108-115 integrates u against the SECOND derivatives of the test functions in reference coordinates, multiplied by the
# element jacobian alone: neither the 1/Jx^2, 1/Jy^2 of the chain rule nor the element-edge terms of the second integration by
# parts are there (the 1-D var_form 3 has its edge term, P1:89-91).  With the exact solution in place of the network,
#   int_e u_xx v = (J/Jx^2) A_x - (Jy/Jx) E_x,   A_x = int int u phi_r'' phi_k,   E_x = int phi_k(eta) [u phi_r']_{xi=-1}^{xi=1} d eta
# (v = phi_r(xi) phi_k(eta), phi(+-1) = 0), so the residual the restated form leaves against the reference's own F_ext_total is
#   R = J (1 - 1/Jx^2) A_x + J (1 - 1/Jy^2) A_y + (Jy/Jx) E_x + (Jx/Jy) E_y
# -- the edge terms alone on ONE element with Jx = Jy = 1 (fixture e1), the (1 - 1/J^2) defect on top of them on a 2 x 2 grid
# (fixture e2).  A restatement with the chain-rule factors, a wrong sign or a wrong table would miss this by O(1).
def _u_exact_2d_np(x, y):
    return (0.1 * np.sin(2 * np.pi * x) + np.tanh(10 * x)) * np.sin(2 * np.pi * y)           # P2:300-302


@pytest.mark.parametrize("tag", ["poisson2d_e1", "poisson2d_e2"])
def test_2d_var_form_2_defect_against_reference_rhs(tag):
    g = gold(tag)
    a = p2_args(g, layers=[2, 5, 1])
    o = O.OracleVPINN2D(*a, var_form=2)
    o.neural_net = lambda X, theta=None: _exact_2d(X)
    o.vectorized = True
    o.loss_parts()
    R = o.last["R"]                                             # [ex][ey][k][r] = U - F_ext_total
    q = int(g["N_quad"])
    xi, w = O.GaussLobattoJacobiWeights(q, 0, 0)
    nt = 5
    phi = O.Test_fcn(nt, xi[:, None])[:, :, 0]                   # [r][i]
    d1, d2 = O.dTest_fcn(nt, xi[:, None])
    d2 = d2[:, :, 0]
    d1e = O.dTest_fcn(nt, np.array([[-1.0], [1.0]]))[0][:, :, 0]      # [r][edge]: phi_r'(-1), phi_r'(+1)
    gx, gy = g["grid_x"], g["grid_y"]
    worst, scale = 0.0, 0.0
    for ex in range(len(gx) - 1):
        for ey in range(len(gy) - 1):
            Jx, Jy = (gx[ex + 1] - gx[ex]) / 2, (gy[ey + 1] - gy[ey]) / 2
            J = Jx * Jy
            x = gx[ex] + Jx * (xi + 1)
            y = gy[ey] + Jy * (xi + 1)
            U = _u_exact_2d_np(x[None, :], y[:, None])           # [j][i]
            Ax = np.einsum("j,i,ji,ri,kj->kr", w, w, U, d2, phi)
            Ay = np.einsum("j,i,ji,ri,kj->kr", w, w, U, phi, d2)
            ux0, ux1 = _u_exact_2d_np(gx[ex], y), _u_exact_2d_np(gx[ex + 1], y)          # u on the two x-edges, [j]
            uy0, uy1 = _u_exact_2d_np(x, gy[ey]), _u_exact_2d_np(x, gy[ey + 1])          # u on the two y-edges, [i]
            Ex = np.einsum("j,kj,jr->kr", w, phi, ux1[:, None] * d1e[None, :, 1] - ux0[:, None] * d1e[None, :, 0])
            Ey = np.einsum("i,ri,ik->kr", w, phi, uy1[:, None] * d1e[None, :, 1] - uy0[:, None] * d1e[None, :, 0])
            want = J * (1 - 1 / Jx ** 2) * Ax + J * (1 - 1 / Jy ** 2) * Ay + (Jy / Jx) * Ex + (Jx / Jy) * Ey
            worst = max(worst, np.abs(R[ex, ey] - want).max())
            scale = max(scale, np.abs(want).max(), np.abs(g["F_ext_total"][ex, ey]).max())
            if tag.endswith("e1"):
                assert Jx == 1.0 and Jy == 1.0
                assert np.abs(want - (Ex + Ey)).max() == 0.0     # nothing but the edge terms on the unit-jacobian element
                assert np.abs(Ex).max() > 1.0                    # ... which are O(1): u = +-tanh(10) sin(2 pi y) on x = +-1
    # (tolerance = the quadrature error of the 60-point rule on tanh(10 x): 1.4e-6 on the element of width 2, 1e-12 on width 1)
    assert scale > 1.0 and worst < (1e-5 if tag.endswith("e1") else 1e-10) * scale, (worst, scale)
    print(tag, "defect identity: worst", worst, "scale", scale)


# ---- the AdvDiff forms (P3:161-174) against the reference's exact solution: the 801-term series of P3:416-445 with
#      epsilon = 0.1 / pi (P3:41-42) solves u_t + V u_x = epsilon u_xx, and the right-hand side is zero (P3:180): fed in place of
#      the network, the series must make `lossv` vanish under refinement of the rule / the grid -- with the exact epsilon only ----
def _advdiff_series(X, trunc=800):
    x, t = X[:, 0:1], X[:, 1:2]
    D, V = 0.1 / np.pi, 1.0
    p = torch.arange(0, trunc + 1.0, dtype=torch.float64).reshape(1, -1)
    c0 = 16 * np.pi ** 2 * D ** 3 * V * torch.exp(V / D / 2 * (x - V * t / 2))
    sgn = torch.where(p % 2 == 0, 1.0, -1.0)
    c1 = np.sinh(V / D / 2) * torch.sum(sgn * 2 * p * torch.sin(p * np.pi * x) * torch.exp(-D * p ** 2 * np.pi ** 2 * t)
                                        / (V ** 4 + 8 * (V * np.pi * D) ** 2 * (p ** 2 + 1) + 16 * (np.pi * D) ** 4 * (p ** 2 - 1) ** 2),
                                        dim=-1, keepdim=True)
    c2 = np.cosh(V / D / 2) * torch.sum(sgn * (2 * p + 1) * torch.cos((p + 0.5) * np.pi * x)
                                        * torch.exp(-D * (2 * p + 1) ** 2 * np.pi ** 2 * t / 4)
                                        / (V ** 4 + (V * np.pi * D) ** 2 * (8 * p ** 2 + 8 * p + 10)
                                           + (np.pi * D) ** 4 * (4 * p ** 2 + 4 * p - 3) ** 2), dim=-1, keepdim=True)
    return c0 * (c1 + c2)


def _advdiff_lossv(vf, nex, net, q, eps, ntest=5):
    from hp_vpinns_amd.drivers import advdiff
    s = advdiff.setup(N_el_x=nex, N_el_t=net, N_test_x=ntest, N_test_t=ntest, N_quad=q, with_test_grid=False)
    L = [2, 5, 1]
    th = theta0(L, 1, extra=[eps])
    o = O.OracleVPINNAdvDiff(s["XT_u_train"], s["u_train"], s["XT_f_train"], s["XT_quad_train"], s["WXT_quad_train"], s["T_quad"],
                             s["WT_quad"], s["grid_x"], s["grid_t"], s["N_testfcn_total"], s["XT_u_train"], None, L,
                             var_form=vf, init_params=th)
    o.neural_net = lambda X, theta=None: _advdiff_series(X)
    o.vectorized = True
    return float(o.loss_parts()[2].detach())


def test_advdiff_series_is_the_fixture_solution():
    """The torch series used below IS the reference's u_ext: compared with the grid the reference's own function produced."""
    g = gold("advdiff_default")
    xs, ts = g["uext_x"], g["uext_t"][1:]                        # (t = 0 is the reference's special case u_initial, P3:442-443)
    X = torch.as_tensor(np.stack(np.meshgrid(xs, ts), -1).reshape(-1, 2))
    assert np.abs(_advdiff_series(X).numpy().reshape(len(ts), len(xs)) - g["uext_grid"][1:]).max() < 2e-5        # (sinh(V / 2D) c1 and cosh(V / 2D) c2 cancel to 1e-7 of their size: the order of summation shows at 9e-6 in the boundary layer)


@pytest.mark.parametrize("vf", [0, 1])
def test_exact_solution_annihilates_the_advdiff_variational_residual(vf):
    """Measured here: var_form 0 (the strong-form integrand, satisfied point-wise) sits at the series' own round-off floor
    (3e-13: sinh(V/2D) c1 and cosh(V/2D) c2 cancel to 1e-7 of their size) on every rule; var_form 1 (integrated by parts: exact
    only up to the rule's error) falls 4.0e-4 -> 6.9e-10 -> 2.4e-12 from the reference's default 1 x 1 element / 10 x 10 points
    to 4 x 2 elements of 20 x 20 points.  The coefficient training starts from (1.0, P3:63) leaves 5.0, one 5 % off 1.35e-5."""
    eps = 0.1 / np.pi
    default = _advdiff_lossv(vf, 1, 1, 10, eps)                  # P3:47-52: one element, 10 x 10 points, 5 x 5 test functions
    coarse = _advdiff_lossv(vf, 4, 2, 10, eps)
    fine = _advdiff_lossv(vf, 4, 2, 20, eps)
    wrong = _advdiff_lossv(vf, 4, 2, 20, 1.0)
    off5 = _advdiff_lossv(vf, 4, 2, 20, 1.05 * eps)
    print("advdiff vf", vf, "lossv(exact series):", default, coarse, fine, "eps = 1:", wrong, "eps 5 % off:", off5)
    if vf == 0:
        assert max(default, coarse, fine) < 1e-11
    else:
        assert coarse < 1e-4 * default and fine < 1e-1 * coarse and fine < 1e-10
    assert fine < 1e-10 * wrong, (fine, wrong)
    assert off5 > 1e4 * fine                                     # the variational loss identifies the coefficient
