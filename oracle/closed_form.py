"""Second, independent CPU restatement of the hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rule as
vpinn_oracle.py: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package).

`vpinn_oracle.py` follows the reference's TF1 graphs with autograd (reverse mode applied twice).  This module
computes the same loss and its gradient in CLOSED FORM with numpy only -- Taylor-mode channels (value, first and
second input tangents) pushed forward through the MLP, sum-factorised projection onto the test functions, and a
hand-derived reverse pass with the third activation derivative -- i.e. the mathematics the HIP kernels implement,
written independently of them.  SURVEY.md section 8(c) asks for two restatements that agree to <= 1e-12; the CPU suite
(tests/test_oracle.py) checks exactly that, so the hand-derived reverse pass is validated without a GPU.

References: network P1:128-148 / P2:158-185 / P3:219-245; variational forms P1:82-96, P2:91-120, P3:157-182; loss
assembly P1:98-100, P2:122-129, P3:184-187.
"""
import numpy as np

from .vpinn_oracle import Test_fcn, dTest_fcn, unpack


def _act(kind, z):
    """sigma, sigma', sigma'', sigma''' at z."""
    if kind == "tanh":
        s = np.tanh(z)
        d1 = 1 - s * s
        return s, d1, -2 * s * d1, -2 * d1 * (1 - 3 * s * s)
    s, c = np.sin(z), np.cos(z)
    return s, c, -s, -c


def taylor_forward(theta, layers, X, kind, t1, t2):
    """Channels [u, du/dx_c (c in t1), d2u/dx_c^2 (c in t2)] at the rows of X, plus what the reverse pass needs."""
    ws, bs, _ = unpack(theta, layers)
    N, d = X.shape
    h = X
    hc = {c: np.tile(np.eye(d)[c], (N, 1)) for c in set(t1) | set(t2)}
    hcc = {c: np.zeros((N, d)) for c in t2}
    tape = []
    for l in range(len(ws) - 1):
        z = h @ ws[l] + bs[l]
        zc = {c: v @ ws[l] for c, v in hc.items()}
        zcc = {c: v @ ws[l] for c, v in hcc.items()}
        s, d1, d2, d3 = _act(kind, z)
        tape.append((h, hc, hcc, zc, zcc, d1, d2, d3))
        h = s
        hc = {c: d1 * zc[c] for c in zc}
        hcc = {c: d2 * zc[c] ** 2 + d1 * zcc[c] for c in zcc}
    tape.append((h, hc, hcc))
    W, b = ws[-1], bs[-1]
    out = [h @ W + b] + [hc[c] @ W for c in t1] + [hcc[c] @ W for c in t2]
    return out, tape


def taylor_backward(theta, layers, tape, t1, t2, gbar):
    """d(sum_k <gbar_k, channel_k>)/d theta for the channel list of taylor_forward (gbar_k has shape (N, 1))."""
    ws, bs, _ = unpack(theta, layers)
    grads_w, grads_b = [None] * len(ws), [None] * len(ws)
    h, hc, hcc = tape[-1]
    W = ws[-1]
    gu = gbar[0]
    gc = {c: gbar[1 + i] for i, c in enumerate(t1)}
    gcc = {c: gbar[1 + len(t1) + i] for i, c in enumerate(t2)}
    grads_w[-1] = h.T @ gu + sum(hc[c].T @ gc[c] for c in t1) + sum(hcc[c].T @ gcc[c] for c in t2)
    grads_b[-1] = gu.sum(0, keepdims=True)
    hb = gu @ W.T
    hcb = {c: (gc[c] @ W.T if c in gc else 0.0) for c in hc}
    hccb = {c: gcc[c] @ W.T for c in hcc}
    for l in range(len(ws) - 2, -1, -1):
        h, hc, hcc, zc, zcc, d1, d2, d3 = tape[l]
        zccb = {c: hccb[c] * d1 for c in zcc}
        zcb = {c: hcb[c] * d1 + (2 * hccb[c] * d2 * zc[c] if c in zcc else 0.0) for c in zc}
        zb = hb * d1
        for c in zc:
            zb = zb + hcb[c] * d2 * zc[c]
        for c in zcc:
            zb = zb + hccb[c] * (d3 * zc[c] ** 2 + d2 * zcc[c])
        grads_w[l] = h.T @ zb + sum(hc[c].T @ zcb[c] for c in zc) + sum(hcc[c].T @ zccb[c] for c in zcc)
        grads_b[l] = zb.sum(0, keepdims=True)
        hb = zb @ ws[l].T
        hcb = {c: zcb[c] @ ws[l].T for c in zc}
        hccb = {c: zccb[c] @ ws[l].T for c in zcc}
    return np.concatenate([np.concatenate([gw.reshape(-1), gb.reshape(-1)]) for gw, gb in zip(grads_w, grads_b)])


def _tables(n, x):
    t0 = Test_fcn(n, x[:, None])[:, :, 0]
    t1, t2 = dTest_fcn(n, x[:, None])
    return [t0, t1[:, :, 0], t2[:, :, 0]]


def loss_and_grad_2d(theta, layers, pde, var_form, xi, w, grid_x, grid_y, ntx, nty, F, Xd, ud, lossb_weight, V=1.0):
    """Poisson-2D (pde='poisson2d', tanh) or AdvDiff (pde='advdiff', tanh, theta[-1] = epsilon) on a tensor grid of
    uniform test-function counts.  Returns ((loss, lossb as reported, lossv), gradient)."""
    P = theta.size - (1 if pde == "advdiff" else 0)
    eps = theta[-1] if pde == "advdiff" else 0.0
    # integrand terms: (dx, dy, [(channel name, alpha)], coefficient(Jx, Jy), multiplied by eps)
    if pde == "poisson2d":
        terms = {0: [(0, 0, [("xx", 1.0), ("yy", 1.0)], lambda jx, jy: jx * jy, False)],
                 1: [(1, 0, [("x", 1.0)], lambda jx, jy: -jy, False), (0, 1, [("y", 1.0)], lambda jx, jy: -jx, False)],
                 2: [(2, 0, [("u", 1.0)], lambda jx, jy: jx * jy, False), (0, 2, [("u", 1.0)], lambda jx, jy: jx * jy, False)]}[var_form]
    else:   # x = channel 0 (space), y = channel 1 (time): u_t + V u_x - eps u_xx
        terms = {0: [(0, 0, [("y", 1.0), ("x", V), ("xx", -eps)], lambda jx, jy: jx * jy, False)],
                 1: [(0, 0, [("y", 1.0), ("x", V)], lambda jx, jy: jx * jy, False),
                     (1, 0, [("x", 1.0)], lambda jx, jy: jy, True)]}[var_form]
    names = sorted({n for t in terms for n, _ in t[2]})
    t1 = [c for c, n in enumerate("xy") if n in names]
    t2 = [c for c, n in enumerate(("xx", "yy")) if n in names]
    chan = {"u": 0}
    chan.update({"xy"[c]: 1 + i for i, c in enumerate(t1)})
    chan.update({("xx", "yy")[c]: 1 + len(t1) + i for i, c in enumerate(t2)})
    tx, ty = _tables(ntx, xi), _tables(nty, xi)
    Q = xi.size
    nex, ney = grid_x.size - 1, grid_y.size - 1
    NR = ntx * nty
    X = np.empty((nex * ney * Q * Q, 2))
    for ex in range(nex):
        for ey in range(ney):
            xq = grid_x[ex] + (grid_x[ex + 1] - grid_x[ex]) / 2 * (xi + 1)
            yq = grid_y[ey] + (grid_y[ey + 1] - grid_y[ey]) / 2 * (xi + 1)
            blk = X[(ex * ney + ey) * Q * Q:(ex * ney + ey + 1) * Q * Q]
            blk[:, 0], blk[:, 1] = np.tile(xq, Q), np.repeat(yq, Q)      # q = j * Q + i, x fastest (P2:362-365)
    th = theta[:P]
    out, tape = taylor_forward(th, layers, X, "tanh", t1, t2)
    ch = [o.reshape(nex, ney, Q, Q) for o in out]                          # [ex, ey, j, i]
    jx = (grid_x[1:] - grid_x[:-1]) / 2
    jy = (grid_y[1:] - grid_y[:-1]) / 2
    U = np.zeros((nex, ney, nty, ntx))
    parts = []
    for dx, dy, mix, cf, eps_mult in terms:
        G = sum(a * ch[chan[n]] for n, a in mix)
        A, B = tx[dx] * w, ty[dy] * w                                      # w_x phi_r^(dx)(xi_i), w_y phi_k^(dy)(xi_j)
        c = cf(jx[:, None], jy[None, :]) * (eps if eps_mult else 1.0)
        proj = np.einsum("kj,abji,ri->abkr", B, G, A)
        U += c[:, :, None, None] * proj
        parts.append((mix, A, B, c, proj, eps_mult, cf))
    R = U - (F if F is not None else 0.0)
    lossv = (R ** 2).mean(axis=(2, 3)).sum()
    ud_pred, tape_d = taylor_forward(th, layers, Xd, "tanh", [], [])
    msq = ((ud - ud_pred[0]) ** 2).mean() if len(Xd) else 0.0
    loss = lossb_weight * msq + lossv
    # ---- reverse ----
    Rb = 2.0 / NR * R
    gbar = [np.zeros_like(ch[0]) for _ in ch]
    deps = 0.0
    for mix, A, B, c, proj, eps_mult, cf in parts:
        Gb = np.einsum("kj,abkr,ri->abji", B, Rb * c[:, :, None, None], A)
        for n, a in mix:
            gbar[chan[n]] += a * Gb
        if pde == "advdiff":
            if eps_mult:
                deps += (Rb * cf(jx[:, None], jy[None, :])[:, :, None, None] * proj).sum()
            for n, a in mix:
                if n == "xx":                                              # alpha = -eps
                    deps += -(Gb * ch[chan[n]]).sum()
    g = taylor_backward(th, layers, tape, t1, t2, [gb.reshape(-1, 1) for gb in gbar])
    if len(Xd):
        g = g + taylor_backward(th, layers, tape_d, [], [], [-2.0 * lossb_weight / len(Xd) * (ud - ud_pred[0])])
    if pde == "advdiff":
        g = np.concatenate([g, [deps]])
    lossb_rep = lossb_weight * msq if pde == "advdiff" else msq
    return (loss, lossb_rep, lossv), g


def loss_and_grad_1d(theta, layers, var_form, xi, w, grid, ntest, F, Xd, ud, lossb_weight):
    """Poisson-1D (sin): var_form 1, 2, 3 incl. the element-edge term of var_form 3 (P1:88-91)."""
    t1 = [0] if var_form == 2 else []
    t2 = [0] if var_form == 1 else []
    tabs = _tables(ntest, xi)
    d1b = dTest_fcn(ntest, np.array([[-1.0], [1.0]]))[0][:, :, 0]          # phi'_k(-1), phi'_k(+1)
    ne, Q = grid.size - 1, xi.size
    J = (grid[1:] - grid[:-1]) / 2
    X = np.concatenate([grid[e] + J[e] * (xi + 1) for e in range(ne)])[:, None]
    out, tape = taylor_forward(theta, layers, X, "sin", t1, t2)
    ch = [o.reshape(ne, Q) for o in out]
    if var_form == 1:
        A, c, G = tabs[0] * w, -J, ch[1]
    elif var_form == 2:
        A, c, G = tabs[1] * w, np.ones(ne), ch[1]
    else:
        A, c, G = tabs[2] * w, -1.0 / J, ch[0]
    U = c[:, None] * (G @ A.T)
    if var_form == 3:
        Xe = np.stack([grid[:-1], grid[1:]], 1).reshape(-1, 1)
        oute, tape_e = taylor_forward(theta, layers, Xe, "sin", [], [])
        ue = oute[0].reshape(ne, 2)
        U = U + (1.0 / J)[:, None] * (ue[:, 1:2] * d1b[None, :, 1] - ue[:, 0:1] * d1b[None, :, 0])
    R = U - F.reshape(ne, ntest)
    lossv = (R ** 2).mean(axis=1).sum()
    ud_pred, tape_d = taylor_forward(theta, layers, Xd, "sin", [], [])
    msq = ((ud - ud_pred[0]) ** 2).mean()
    loss = lossb_weight * msq + lossv
    Rb = 2.0 / ntest * R
    Gb = (Rb * c[:, None]) @ A
    gbar = [np.zeros((ne * Q, 1)) for _ in ch]
    gbar[0 if var_form == 3 else 1] = Gb.reshape(-1, 1)
    g = taylor_backward(theta, layers, tape, t1, t2, gbar)
    if var_form == 3:
        ge = np.stack([-(Rb * d1b[None, :, 0]).sum(1) / J, (Rb * d1b[None, :, 1]).sum(1) / J], 1).reshape(-1, 1)
        g = g + taylor_backward(theta, layers, tape_e, [], [], [ge])
    g = g + taylor_backward(theta, layers, tape_d, [], [], [-2.0 * lossb_weight / len(Xd) * (ud - ud_pred[0])])
    return (loss, msq, lossv), g
