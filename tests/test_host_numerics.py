"""Host-side numerics (quadrature, test-function tables, drivers' set-up) against the golden
fixtures produced by RUNNING the reference's numpy code (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from cases import gold
from hp_vpinns_amd import DJacobi, GaussJacobiWeights, GaussLobattoJacobiWeights, Jacobi, Test_fcn, dTest_fcn
from hp_vpinns_amd.drivers import poisson1d, poisson2d
from hp_vpinns_amd.testfcn import tables_1d


def _close(a, b, tol):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-300), np.abs(a - b).max()


@pytest.mark.parametrize("q", [5, 10, 20, 80])
def test_gll_rule_matches_reference(q):
    g = gold("quadrature")
    x, w = GaussLobattoJacobiWeights(q, 0, 0)
    _close(x, g[f"gll_x_{q}"], 1e-14)
    _close(w, g[f"gll_w_{q}"], 1e-13)


@pytest.mark.parametrize("q", [3, 10, 20, 80])
def test_gll_analytic_properties(q):
    x, w = GaussLobattoJacobiWeights(q, 0, 0)
    assert abs(w.sum() - 2.0) < 1e-13                      # integrates constants
    assert np.abs(x + x[::-1]).max() < 1e-15 and np.abs(w - w[::-1]).max() < 1e-13   # symmetry
    assert x[0] == -1 and x[-1] == 1 and np.all(np.diff(x) > 0)
    for deg in range(0, 2 * q - 2):                        # exact to degree 2Q-3
        exact = 0.0 if deg % 2 else 2.0 / (deg + 1)
        assert abs((w * x ** deg).sum() - exact) < 1e-12, deg


def test_jacobi_matches_reference():
    g = gold("quadrature")
    xs = g["jac_x"]
    for k in g.files:
        if k.startswith("jac_") and k != "jac_x":
            n, a, b = (int(v) for v in k.split("_")[1:])
            _close(Jacobi(n, a, b, xs), g[k], 1e-13)
    _close(DJacobi(6, 0, 0, xs, 2), g["djac_6_0_0_2"], 1e-13)
    x, w = GaussJacobiWeights(7, 0, 0)
    _close(x, g["gj_x_7"], 1e-14)
    _close(w, g["gj_w_7"], 1e-13)


@pytest.mark.parametrize("nt,q", [(60, 80), (5, 10), (10, 20)])
def test_test_function_tables_match_reference(nt, q):
    g = gold("testfcn")
    x = GaussLobattoJacobiWeights(q, 0, 0)[0][:, None]
    t = Test_fcn(nt, x)
    d1, d2 = dTest_fcn(nt, x)
    _close(t, g[f"phi_{nt}_{q}"], 1e-13)
    _close(d1, g[f"dphi_{nt}_{q}"], 1e-12)
    _close(d2, g[f"d2phi_{nt}_{q}"], 1e-12)
    tab = tables_1d(nt, x[:, 0])
    assert tab.shape == (3, nt, q)
    _close(tab[0], g[f"phi_{nt}_{q}"][:, :, 0], 1e-13)


def test_tables_same_in_all_three_reference_classes():
    g = gold("testfcn")
    for k in ("p2_phix_5_10", "p2_phiy_5_10", "p3_phi_5_10"):
        _close(g[k], g["phi_5_10"], 1e-15)
    _close(g["p2_dphi_5_10"], g["dphi_5_10"], 1e-15)
    _close(g["p3_d2phi_5_10"], g["d2phi_5_10"], 1e-15)


def test_test_functions_vanish_at_the_edges_and_edge_slopes():
    g = gold("testfcn")
    xb = np.array([[-1.0], [1.0]])
    assert np.abs(Test_fcn(60, xb)).max() < 1e-12
    d1, d2 = dTest_fcn(60, xb)
    _close(d1, g["dphi_edge_60"], 1e-13)
    _close(d2, g["d2phi_edge_60"], 1e-12)


@pytest.mark.parametrize("tag,kw", [("cfg1", {}), ("ne3", dict(N_Element=3)), ("cfg2", dict(N_Element=16)),
                                    ("small", dict(N_Element=4, N_testfcn=6, N_Quad=12))])
def test_poisson1d_driver_setup_matches_reference(tag, kw):
    g, s = gold("poisson1d_" + tag), poisson1d.setup(**kw)
    for k in ("grid", "F_ext_total", "U_ext_total", "X_quad_train", "W_quad_train", "X_u_train", "u_train",
              "X_f_train", "f_train", "X_test", "u_test"):
        _close(s[k], g[k], 1e-12)


@pytest.mark.parametrize("tag,kw", [
    ("default", {}), ("cfg3", dict(N_el_x=8, N_el_y=8)),
    ("cfg4", dict(N_el_x=16, N_el_y=16, N_test_x=10, N_test_y=10, N_quad=20)),
    ("small", dict(N_el_x=3, N_el_y=2, N_test_x=4, N_test_y=3, N_quad=6, N_bound=10, N_residual=10))])
def test_poisson2d_driver_setup_matches_reference(tag, kw):
    g, s = gold("poisson2d_" + tag), poisson2d.setup(**kw)
    for k in ("grid_x", "grid_y", "F_ext_total", "XY_quad_train", "WXY_quad_train", "X_u_train", "u_train",
              "X_f_train", "f_train"):
        _close(s[k], g[k], 1e-13)
    assert tuple(g["X_test_shape"]) == s["X_test"].shape
    _close(s["X_test"][:500], g["X_test_head"], 1e-15)
    _close(s["u_test"][:500], g["u_test_head"], 1e-14)
    assert abs(s["u_test"].sum() - g["u_test_sum"]) < 1e-9


@pytest.mark.parametrize("tag,kw", [
    ("default", {}), ("cfg5", dict(N_el_x=8, N_quad=80)),
    ("small", dict(N_el_x=3, N_el_t=2, N_test_x=4, N_test_t=3, N_quad=6, N_bound=10))])
def test_advdiff_driver_setup_matches_reference(tag, kw):
    from hp_vpinns_amd.drivers import advdiff
    g, s = gold("advdiff_" + tag), advdiff.setup(**kw, with_test_grid=False)
    for k in ("grid_x", "grid_t", "XT_u_train", "u_train", "XT_f_train", "T_quad", "WT_quad"):
        _close(s[k], g[k], 1e-13)
    n = g["XT_quad_train"].shape[0]
    _close(s["XT_quad_train"][:n], g["XT_quad_train"], 1e-14)
    _close(s["WXT_quad_train"][:n], g["WXT_quad_train"], 1e-13)
    X, T = np.meshgrid(g["uext_x"], g["uext_t"])
    _close(advdiff.u_ext(X, T), g["uext_grid"], 1e-13)     # the 801-term Fourier series (P3:416-445)
    assert abs(advdiff.epsilon - float(g["epsilon_exact"])) < 1e-16


def test_zero_padding_of_a_narrow_network_is_exact():
    """init.pad_plan: a 5-wide network (reference default, P2:280) zero-padded onto 20-wide layers has the same loss, and
    its gradient restricted to the original entries equals the original gradient while every padding entry is exactly 0
    (checked with the CPU oracle; on the GPU the classes use this to run narrow networks on the MFMA kernels)."""
    from cases import gold, p2_args, theta0
    from hp_vpinns_amd.init import n_params, pad_plan
    from oracle import vpinn_oracle as O
    a = list(p2_args(gold("poisson2d_small"), layers=[2, 5, 7, 1]))
    th = theta0(a[13], 3)
    padded_layers, idx = pad_plan(a[13])
    assert padded_layers == [2, 20, 20, 1] and idx.size == th.size == n_params(a[13])
    thp = np.zeros(n_params(padded_layers))
    thp[idx] = th
    l3, g = O.OracleVPINN2D(*a, init_params=th).loss_and_grad()
    a[13] = padded_layers
    l3p, gp = O.OracleVPINN2D(*a, init_params=thp).loss_and_grad()
    assert np.abs(np.array(l3p) - np.array(l3)).max() < 1e-13 * abs(l3[0])
    assert np.abs(gp[idx] - g).max() < 1e-13 * np.abs(g).max()
    pad = np.ones(thp.size, bool)
    pad[idx] = False
    assert pad.sum() == thp.size - th.size and np.all(gp[pad] == 0.0)
    assert pad_plan([2, 20, 20, 1]) is None and pad_plan([2, 24, 1]) is None and pad_plan([2] + [5] * 7 + [1]) is None
    assert pad_plan([2] + [5] * 6 + [1])[0] == [2] + [20] * 6 + [1] and pad_plan([2] + [30] * 5 + [1])[0] == [2] + [32] * 5 + [1]
    assert pad_plan([2] + [36] * 5 + [1]) is None     # depth: 6 up to width 32, 4 beyond
    # wider networks go to the next hidden width the width-generic MFMA kernels are instantiated for (csrc/kernels_wide.hip)
    assert pad_plan([2, 30, 30, 1])[0] == [2, 32, 32, 1] and pad_plan([2, 20, 40, 20, 1])[0] == [2, 40, 40, 40, 1]
    assert pad_plan([1, 21, 1])[0] == [1, 24, 1] and pad_plan([2, 50, 64, 1])[0] == [2, 64, 64, 1]
    assert pad_plan([2, 32, 32, 32, 1]) is None and pad_plan([2, 70, 70, 1]) is None      # exact width / beyond 64: no padding
    pl, idx = pad_plan([2, 3, 30, 1], extra=1)
    assert pl == [2, 32, 32, 1] and idx.size == n_params([2, 3, 30, 1], 1) and idx[-1] == n_params(pl)


def test_recorded_run_indices_and_stop_without_a_device():
    """vpinn._VPINNBase._recorded_run: which iterations of a chunk are recording iterations (index % 10 == 0), where the
    run stops (first recorded loss below the threshold) and that the state is rolled back and re-run to exactly that
    iteration -- with the device replaced by a stub."""
    from hp_vpinns_amd.vpinn import _VPINNBase

    class Stub(_VPINNBase):
        def __init__(self, losses):
            self.losses, self.calls = np.asarray(losses, float), []

            class H:
                def get_state(s):
                    return "state"

                def set_state(s, st):
                    self.calls.append(("set_state", st))
            self.h = H()

        def _step_record(self, n):
            self.calls.append(("record", n))
            return np.stack([self.losses[:n], 0 * self.losses[:n], self.losses[:n]], 1), np.zeros(n)

        def _step(self, n, read_loss):
            self.calls.append(("step", n))

    m = Stub(np.linspace(10, 1, 30))
    recs, stop = m._recorded_run(7, 30, 0.0)                 # iterations 7..36: recording ones are 10, 20, 30
    assert [r[0] for r in recs] == [10, 20, 30] and stop is None and m.calls == [("record", 30)]
    assert [float(r[1][0]) for r in recs] == [float(m.losses[k]) for k in (3, 13, 23)]
    m = Stub(np.linspace(10, 1, 30))
    recs, stop = m._recorded_run(0, 30, 6.5)                 # losses at iterations 0, 10, 20: 10, 6.9, 3.8 -> stop at 20
    assert [r[0] for r in recs] == [0, 10, 20] and stop == 20
    assert m.calls == [("record", 30), ("set_state", "state"), ("step", 21)]


def test_advdiff_mat_export_holds_the_reference_variables(tmp_path):
    """drivers.advdiff.export_mat: the eight variables of the reference's `hpPINN_ADE_Iden_record.mat` (P3:500-508) in one file."""
    import scipy.io
    from hp_vpinns_amd.drivers import advdiff
    s = dict(XT_test=np.random.rand(7, 2), u_test=np.random.rand(7, 1), grid_x=np.array([-1.0, 1.0]), grid_t=np.array([0.0, 1.0]))
    total = [np.array([10 * i, 1.0 / (i + 1), np.array([1.0 - 0.01 * i]), 1], dtype=object) for i in range(5)]
    path = str(tmp_path / "rec.mat")
    advdiff.export_mat(path, s, np.ones((7, 1)), [], total, 1.25)
    m = scipy.io.loadmat(path)
    assert set(["x_test", "u_test", "grid_x", "grid_t", "u_pred", "u_pred_his", "total", "total_time_train"]) <= set(m)
    assert m["total"].shape == (5, 4) and abs(m["total"][3, 1] - 0.25) < 1e-15 and abs(m["total"][4, 2] - 0.96) < 1e-15
    assert m["u_pred"].shape == (7, 1) and float(m["total_time_train"]) == 1.25
    advdiff.export_mat(path, s, None, [], [], 0.0)           # (a run that never reached its last tenth)
    assert scipy.io.loadmat(path)["total"].shape == (0, 4)


def test_zero_weight_padding_of_quadrature_rules():
    """vpinn._pad_rule / _device_rule_2d: a rule between two instantiated ones goes to the device with zero-weight points appended --
    every quadrature sum is unchanged, the choice of the device rule follows the kernels' instantiations and grid limits."""
    from hp_vpinns_amd.quadrature import GaussLobattoJacobiWeights
    from hp_vpinns_amd.testfcn import tables_1d
    from hp_vpinns_amd.vpinn import _device_rule_2d, _pad_rule
    x, w = GaussLobattoJacobiWeights(14, 0, 0)
    xp, wp = _pad_rule(x, w, 16)
    assert xp.size == 16 and np.array_equal(xp[:14], x) and np.all(wp[14:] == 0.0) and np.all(xp[14:] == x[-1])
    f = lambda t: np.cos(1.3 * t) + t ** 5
    assert sum(a * b for a, b in zip(wp, f(xp))) == sum(a * b for a, b in zip(w, f(x)))      # the appended terms are exact zeros
    # the tables the kernels contract with are w * phi: the padded columns vanish whatever phi is there
    t0, tp = tables_1d(7, x), tables_1d(7, xp)
    assert tp.shape == (3, 7, 16) and np.array_equal(tp[..., :14], t0) and np.all(np.isfinite(tp))
    sel = lambda q, ntx, nty, ne, **k: _device_rule_2d(*(GaussLobattoJacobiWeights(q, 0, 0) * 2), ntx, nty, ne, **k)[0].size
    assert [sel(q, q // 2, q // 2, 256) for q in (6, 10, 11, 12, 14, 16, 18, 20, 22)] == [10, 10, 12, 12, 16, 16, 20, 20, 22]
    assert sel(14, 9, 3, 256) == 20 and sel(14, 11, 3, 256) == 14          # the test-function counts must fit the instantiation too
    assert sel(7, 4, 4, 2048) == 7 and sel(14, 7, 7, 2048) == 14 and sel(18, 9, 9, 4096) == 20     # grids the kernel leaves to the separate launches
    # the plan the advice evaluates is the dispatch's own (advisor, round 5): 1 024 elements of a 14-point rule under three hidden layers are
    # four full rounds -> the element loop takes the 16x16 kernel's grid -> no padding (until round 5 the advice said 16 and the launch
    # then ran the loop on the padded rule); 768 elements (three rounds: one workgroup per element) are padded; the loop starts at six
    # rounds under two hidden layers, so 1 024 elements are padded there
    assert sel(14, 7, 7, 1024, n_hidden=3) == 14 and sel(14, 7, 7, 768, n_hidden=3) == 16 and sel(14, 7, 7, 1024, n_hidden=2) == 16
    assert sel(14, 7, 7, 1536, n_hidden=2) == 14
    assert sel(8, 5, 5, 64, exact_counts=True, only=10) == 10 and sel(8, 4, 5, 64, exact_counts=True, only=10) == 8
    assert sel(11, 6, 6, 64, exact_counts=True, only=10) == 11                # (an instantiation the caller did not ask for)
    # 1-D: the 80 / 60 tile kernel takes a smaller rule only on shards where one workgroup per element pays (hpv_rule1d_pad_max)
    from hp_vpinns_amd._lib import rule_advice
    assert rule_advice(0, 1, 80, 60, 1, 16) == (80, 60) and rule_advice(0, 1, 80, 12, 1, 16) == (80, 60)
    assert rule_advice(0, 1, 40, 20, 1, 16) == (80, 60) and rule_advice(0, 1, 10, 5, 1, 256) == (80, 60)
    assert rule_advice(0, 1, 10, 5, 1, 10000) == (10, 5)                     # h-refinement: 10 k elements of 10 points stay as they are
    assert rule_advice(0, 1, 40, 20, 1, 257) == (40, 20) and rule_advice(0, 1, 60, 30, 1, 512) == (80, 60) and rule_advice(0, 1, 60, 30, 1, 513) == (60, 30)
    assert rule_advice(0, 1, 90, 20, 1, 4) == (90, 20) and rule_advice(0, 1, 60, 61, 1, 4) == (60, 61)
    xa, _ = GaussLobattoJacobiWeights(12, 0, 0)
    xb, wb = GaussLobattoJacobiWeights(14, 0, 0)
    assert _device_rule_2d(xa, _, xb, wb, 5, 5, 64)[0].size == 12           # different rules per direction: left alone


def test_grid_plan_of_the_whole_iteration_kernel():
    """hpv_grid_plan = the dispatch's own hpv_fused_grid_plan (256 CUs when no device can be queried): one workgroup per element up to
    the chip; the element loop on many full rounds where it is built (not for three hidden layers on 20x20 points); round 6: a ragged
    last round of 20x20-point elements from two full rounds on = full rounds + the tail in split mode (1 600 = 6 x 256 + 64), with ONE
    full round or smaller elements the separate launches (measured: profiles/r06_notes.md 3); tails beyond half a round keep the
    80 %-full rule."""
    from hp_vpinns_amd._lib import HpvError, grid_plan
    assert grid_plan(-1, 20, 3, 256) == 1 and grid_plan(-1, 20, 3, 100) == 1 and grid_plan(-1, 12, 2, 1) == 1
    assert grid_plan(-1, 20, 3, 1600) == 3 and grid_plan(-1, 20, 3, 1296) == 3 and grid_plan(-1, 20, 3, 552) == 3
    assert grid_plan(-1, 20, 3, 289) == 0                        # one full round + 33: the separate launches (90 against 85 us)
    assert grid_plan(-1, 20, 3, 4096) == 1 and grid_plan(-1, 20, 3, 512) == 1          # whole rounds
    assert grid_plan(-1, 20, 3, 400) == 0 and grid_plan(-1, 20, 3, 480) == 1            # 1.56 / 1.88 rounds: below / above 80 % full
    assert grid_plan(-1, 16, 3, 552) == 0 and grid_plan(-1, 16, 3, 289) == 0            # 16x16 points: ragged grids on the separate launches
    assert grid_plan(-1, 16, 3, 1024) == 2 and grid_plan(-1, 16, 3, 768) == 1           # the element loop from four full rounds on
    assert grid_plan(-1, 20, 2, 1536) == 2 and grid_plan(-1, 20, 2, 1600) == 3          # two hidden layers: the loop exists on 20x20 points
    assert grid_plan(-1, 12, 3, 1024) == 0                                              # many small elements: beyond hpv_elem_resident_max
    for bad in ((14, 3, 10), (20, 4, 10), (20, 3, 0)):
        with pytest.raises(HpvError):
            grid_plan(-1, *bad)


def test_device_tanh_algorithm_in_exact_arithmetic():
    """csrc/hpv_math.h, round 5 (25 fp64 operations: k and 2^k from the bits of one fma, degree-9 interpolant of (e^(-2w) - 1 + 2w) / w^2):
    the same operation sequence with every operation rounded once (rational arithmetic), the polynomial READ FROM THE HEADER, against
    mpmath at 60 digits -- relative error <= 4e-16 incl. |x| -> 0, the reduction boundaries (k + 1/2) ln2 / 2 and the clamp."""
    import importlib.util
    import math
    import os
    import random
    spec = importlib.util.spec_from_file_location(
        "tanh_proto", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "tanh_proto.py"))
    tp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tp)
    C = tp.device_coeffs()
    assert len(C) == 10
    xs = tp.samples(1200) + [0.5 * math.log(2.0) * (k + 0.5) for k in range(0, 90)] + [1e-300, 5e-324, 31.999, 32.0, 33.0, 1e300]
    err, where = tp.maxerr(lambda x: tp.tanh_new(x, C, True), xs)
    assert err < 4e-16, (err, where)
    random.seed(7)
    for x in [random.uniform(-3, 3) for _ in range(50)]:
        assert tp.tanh_new(-x, C, True) == -tp.tanh_new(x, C, True)
    assert tp.tanh_new(40.0, C, True) == 1.0 and tp.tanh_new(-1e300, C, True) == -1.0 and tp.tanh_new(0.0, C, True) == 0.0


def test_device_sincos_algorithm_in_exact_arithmetic():
    """csrc/hpv_math.h, round 5: the trimmed sincos (quadrant from the bits of one fma, sin = r (1 + z p), cos = 1 - z/2 + z^2 q in three
    operations) with every operation rounded once, against mpmath -- <= 2 ulp for |x| <= 1e6 incl. the doubles nearest to multiples
    of pi/2, and no worse than the round-4 form by more than half an ulp (scripts/sincos_proto.py; the device side: tests/test_gpu_parity.py)."""
    import importlib.util
    import math
    import os
    import random
    import mpmath as mp
    spec = importlib.util.spec_from_file_location(
        "sincos_proto", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "sincos_proto.py"))
    sp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sp)
    random.seed(11)
    xs = [random.uniform(-40, 40) for _ in range(600)] + [random.uniform(-1e6, 1e6) for _ in range(300)] + [k * (math.pi / 2) for k in range(0, 300)]
    worst = {False: [0.0, 0.0], True: [0.0, 0.0]}
    for x in xs:
        rs, rc = mp.sin(mp.mpf(x)), mp.cos(mp.mpf(x))
        for trimmed in (False, True):
            a, b = sp.sincos(x, trimmed)
            worst[trimmed][0] = max(worst[trimmed][0], sp.ulp(a, rs))
            worst[trimmed][1] = max(worst[trimmed][1], sp.ulp(b, rc))
    assert worst[True][0] <= 2.0 and worst[True][1] <= 2.0, worst
    assert worst[True][0] <= worst[False][0] + 0.5 and worst[True][1] <= worst[False][1] + 0.5, worst
