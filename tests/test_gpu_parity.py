"""Parity of the HIP path (through the C-ABI) against the CPU oracle on identical golden inputs and
identical initial parameters.  Tolerances: 1e-9 relative on loss / gradient / residuals (fp64 kernels
vs fp64 autograd, different summation orders), and the north_star bar of 1e-5 relative L2 on u(x) and
on the loss trajectory -- we assert 1e-7, two orders tighter."""
import numpy as np
import pytest

from cases import gold, p1_args, p2_args, p3_args, rel, theta0

pytestmark = pytest.mark.gpu

TOL = 1e-9
TRAJ_TOL = 1e-7


def _vec(o, vectorized):
    # the vectorised oracle variant (proven equal to the reference-structured element loops for every var_form in
    # tests/test_oracle.py) keeps the many-step comparisons fast on a loaded host; the three *_small tests below use the
    # element loops themselves
    o.vectorized = vectorized
    return o


def _pair_1d(tag, vf, layers=None, backend="auto", vectorized=True):
    from hp_vpinns_amd.vpinn import VPINN1D
    from oracle.vpinn_oracle import OracleVPINN1D
    g = gold(tag)
    a = p1_args(g, layers)
    th = theta0(a[8], 11)
    th[1 * a[8][1] + 0: 1 * a[8][1] + a[8][1]] = 0.1 * np.arange(a[8][1])  # non-zero first bias
    return (_vec(OracleVPINN1D(*a, var_form=vf, init_params=th), vectorized), VPINN1D(*a, var_form=vf, init_params=th, backend=backend))


def _pair_2d(tag, vf, layers=None, backend="auto", vectorized=True):
    from hp_vpinns_amd.vpinn import VPINN2D
    from oracle.vpinn_oracle import OracleVPINN2D
    g = gold(tag)
    a = p2_args(g, layers)
    th = theta0(a[13], 12)
    return (_vec(OracleVPINN2D(*a, var_form=vf, init_params=th), vectorized), VPINN2D(*a, var_form=vf, init_params=th, backend=backend))


def _pair_adv(tag, vf, layers=None, backend="auto", vectorized=True):
    from hp_vpinns_amd.vpinn import VPINNAdvDiff
    from oracle.vpinn_oracle import OracleVPINNAdvDiff
    g = gold(tag)
    a = p3_args(g, layers)
    th = theta0(a[12], 13, extra=[0.7])
    return (_vec(OracleVPINNAdvDiff(*a, var_form=vf, init_params=th), vectorized), VPINNAdvDiff(*a, var_form=vf, init_params=th, backend=backend))


def _check_loss_grad(o, m):
    l3o, go = o.loss_and_grad()
    l3m, gm = m.loss_and_grad()
    assert rel(l3m, l3o) < TOL, (l3m, l3o)
    assert rel(gm, go) < TOL, (rel(gm, go), np.abs(gm - go).max())
    # forward-only evaluation gives the same loss
    assert rel(m.loss(), l3o) < TOL


def _check_traj(o, m, n=12):
    lo, lm = [], []
    for _ in range(n):
        o.adam_step()
        lo.append(float(o.loss_parts()[0]))
        lm.append(float(m._step(1, True)[0]))
    assert rel(lm, lo) < TRAJ_TOL, (lm, lo)
    assert rel(m.get_params(), o.get_params()) < TRAJ_TOL


@pytest.mark.parametrize("vf", [1, 2, 3])
def test_poisson1d_small(vf):
    o, m = _pair_1d("poisson1d_small", vf, backend="generic", vectorized=False)
    _check_loss_grad(o, m)
    _check_traj(o, m)
    x = np.linspace(-1, 1, 101)[:, None]
    assert rel(m.predict(x), o.predict(x)) < TRAJ_TOL


@pytest.mark.parametrize("vf", [0, 1, 2])
def test_poisson2d_small(vf):
    o, m = _pair_2d("poisson2d_small", vf, backend="generic", vectorized=False)
    _check_loss_grad(o, m)
    _check_traj(o, m)
    X = np.random.default_rng(3).uniform(-1, 1, (77, 2))
    assert rel(m.predict(X), o.predict(X)) < TRAJ_TOL


@pytest.mark.parametrize("vf", [0, 1])
def test_advdiff_small(vf):
    o, m = _pair_adv("advdiff_small", vf, backend="generic", vectorized=False)
    _check_loss_grad(o, m)
    _check_traj(o, m)
    assert abs(float(m.epsilon[0]) - float(o.get_params()[-1])) < 1e-9


def test_zero_network_known_answers():
    """loss(theta=0) = sum_e mean(F_e^2) + w*lossb: 290.459376+1 (1 element), 407.042034+1 (the
    published 3-element run, the ~4e2 plateau of the reference's Results/loss.pdf)."""
    from hp_vpinns_amd.vpinn import VPINN1D
    for tag, lv in (("poisson1d_cfg1", 290.4593764435546), ("poisson1d_ne3", 407.0420338281929)):
        g = gold(tag)
        a = p1_args(g)
        m = VPINN1D(*a, init_params=np.zeros(901))
        l3 = m.loss()
        assert abs(l3[2] - lv) < 1e-9 * lv and abs(l3[1] - 1.0) < 1e-12


def test_cfg1_reference_default_shape():
    """BASELINE config 1: Poisson-1D, 1 element, N_quad 80, N_test 60, [1,20,20,20,1], sin."""
    o, m = _pair_1d("poisson1d_cfg1", 1)
    _check_loss_grad(o, m)
    _check_traj(o, m, n=6)


def test_cfg2_16_elements():
    o, m = _pair_1d("poisson1d_cfg2", 1)
    _check_loss_grad(o, m)


def test_cfg3_loss_grad():
    """BASELINE config 3: Poisson-2D 8x8 elements, 10x10 quad, 5x5 test, [2,20,20,20,1]."""
    o, m = _pair_2d("poisson2d_cfg3", 1)
    _check_loss_grad(o, m)
    _check_traj(o, m, n=3)


def test_shard_invariance():
    """Element shards evaluated separately (fake collective: sum of the packed partials) reproduce the
    unsharded loss and gradient -- the multi-GPU path's correctness argument on one device."""
    from hp_vpinns_amd import _lib
    from hp_vpinns_amd.testfcn import tables_1d
    from hp_vpinns_amd.vpinn import VPINN2D, _tensor_rule
    g = gold("poisson2d_small")
    a = p2_args(g)
    th = theta0(a[13], 12)
    full = VPINN2D(*a, init_params=th, backend="generic")
    l3, gr = full.loss_and_grad()
    xi, wx, yi, wy = _tensor_rule(a[4], a[5])
    ne = full.Nelementx * full.Nelementy
    acc = None
    for (b, e, with_data) in ((0, 2, True), (2, 5, False), (5, ne, False)):
        h = _lib.Handle(_lib.PDE_POISSON2D, 1, _lib.ACT_TANH, a[13], lossb_weight=10, backend=_lib.BACKEND_GENERIC)
        h.set_quadrature(xi, wx, yi, wy)
        h.set_tables(tables_1d(full.Ntestx, xi), tables_1d(full.Ntesty, yi))
        h.set_elements(a[8], a[9], b, e)
        h.set_rhs(np.asarray(a[7]).reshape(-1))
        if with_data:
            h.set_data(a[0], np.asarray(a[1]).reshape(-1))
        h.set_params(th)
        l3s, gs = h.loss_and_grad(True)
        part = np.concatenate([gs, [l3s[2], l3s[0] - l3s[2]]])
        acc = part if acc is None else acc + part
    assert rel(acc[:-2], gr) < 1e-12
    assert abs(acc[-2] - l3[2]) < 1e-12 * abs(l3[2])
    assert abs(acc[-2] + acc[-1] - l3[0]) < 1e-12 * abs(l3[0])


# ---------------------------------------------------------------------------------------------------
# MFMA fast path (20-wide hidden layers): every channel set the three problems use, 1..4 hidden layers,
# point counts that are not a multiple of the 16-point MFMA tile.
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,vf", [("1d", 1), ("1d", 2), ("1d", 3), ("2d", 0), ("2d", 1), ("2d", 2), ("adv", 0), ("adv", 1)])
def test_mfma_path_all_channel_sets(kind, vf):
    if kind == "1d":
        o, m = _pair_1d("poisson1d_small", vf, layers=[1, 20, 20, 20, 1], backend="mfma")
    elif kind == "2d":
        o, m = _pair_2d("poisson2d_small", vf, layers=[2, 20, 20, 20, 1], backend="mfma")
    else:
        o, m = _pair_adv("advdiff_small", vf, layers=[2, 20, 20, 20, 1], backend="mfma")
    assert m.backend() == "mfma"
    _check_loss_grad(o, m)
    _check_traj(o, m, n=8)


@pytest.mark.parametrize("nhid", [1, 2, 4])
def test_mfma_path_depths(nhid):
    o, m = _pair_2d("poisson2d_small", 1, layers=[2] + [20] * nhid + [1], backend="mfma")
    assert m.backend() == "mfma"
    _check_loss_grad(o, m)
    o, m = _pair_1d("poisson1d_small", 1, layers=[1] + [20] * nhid + [1], backend="mfma")
    _check_loss_grad(o, m)


def test_mfma_equals_generic_on_device():
    """Same inputs through both device paths (config-3 shape): loss/grad agree to rounding."""
    from hp_vpinns_amd.vpinn import VPINN2D
    g = gold("poisson2d_cfg3")
    a = p2_args(g)
    th = theta0(a[13], 12)
    m1 = VPINN2D(*a, init_params=th, backend="generic")
    m2 = VPINN2D(*a, init_params=th, backend="mfma")
    assert (m1.backend(), m2.backend()) == ("generic", "mfma")
    (l1, g1), (l2, g2) = m1.loss_and_grad(), m2.loss_and_grad()
    assert rel(l2, l1) < 1e-12 and rel(g2, g1) < 1e-11
    r1 = m1.h.residuals(64 * 25)
    r2 = m2.h.residuals(64 * 25)
    assert rel(r2, r1) < 1e-11


def test_mfma_unavailable_shape_raises():
    from hp_vpinns_amd import _lib
    with pytest.raises(_lib.HpvError):
        _pair_2d("poisson2d_small", 1, layers=[2, 80, 80, 1], backend="mfma")   # wider than any MFMA kernel (20; 24..64: kernels_wide.hip)


def test_device_tanh_accuracy():
    """The hand-written fp64 tanh: <= 4e-16 absolute and relative error against the math library / numpy."""
    from hp_vpinns_amd import _lib
    h = _lib.Handle(_lib.PDE_POISSON2D, 1, _lib.ACT_TANH, [2, 20, 1])
    x = np.concatenate([np.linspace(-8, 8, 40001), np.logspace(-14, 1.6, 3000), -np.logspace(-14, 1.6, 3000),
                        [0.0, 40.0, -40.0, 1e3, -1e3, 1e300]])
    a, a1, ref = h.debug_activation(x)
    t = np.tanh(x)
    assert np.abs(a - t).max() < 4e-16 and np.abs(ref - t).max() < 4e-16
    nz = np.abs(t) > 0
    assert (np.abs(a - t)[nz] / np.abs(t[nz])).max() < 6e-16
    assert np.abs(a1 - (1 - t * t)).max() < 1e-15
    # non-finite arguments: NaN propagates (a diverged hidden state must not turn into a finite loss), +-inf saturate
    a, a1, _ = h.debug_activation(np.array([np.nan, np.inf, -np.inf, -0.0]))
    assert np.isnan(a[0]) and np.isnan(a1[0])
    assert a[1] == 1.0 and a[2] == -1.0 and a1[1] == 0.0 and a1[2] == 0.0
    assert a[3] == 0.0 and np.signbit(a[3])


def test_device_sincos_accuracy():
    """The hand-written fp64 sincos of the 1-D drivers' activation (P1:134): <= 3 ulp against numpy and ocml for
    |x| <= 1e6 (incl. the doubles nearest to multiples of pi/2), ocml's own path beyond, NaN / inf -> NaN."""
    from hp_vpinns_amd import _lib
    h = _lib.Handle(_lib.PDE_POISSON1D, 1, _lib.ACT_SIN, [1, 20, 1])
    rng = np.random.default_rng(5)
    x = np.concatenate([np.linspace(-40, 40, 80001), rng.uniform(-1e6, 1e6, 20000), np.arange(0, 4000) * (np.pi / 2),
                        np.logspace(-300, 0, 2000), -np.logspace(-300, 0, 2000), [0.0, 1e6, -1e6]])
    a, a1, ref = h.debug_activation(x)
    s, c = np.sin(x), np.cos(x)
    tol = lambda t: 3.0 * np.maximum(np.spacing(np.abs(t)), 1e-30)   # (exact-zero neighbourhoods: absolute 3e-30)
    assert (np.abs(a - s) <= tol(s)).all() and (np.abs(a1 - c) <= tol(c)).all()
    assert (np.abs(ref - s) <= tol(s)).all()
    big = np.array([1.0000001e6, 3e9, -7.5e15, 1e300])
    a, a1, ref = h.debug_activation(big)
    assert np.array_equal(a, ref) and (np.abs(a - np.sin(big)) <= tol(np.sin(big))).all()
    assert (np.abs(a1 - np.cos(big)) <= tol(np.cos(big))).all()
    a, a1, _ = h.debug_activation(np.array([np.nan, np.inf, -np.inf, -0.0]))
    assert np.isnan(a[:3]).all() and np.isnan(a1[:3]).all()
    assert a[3] == 0.0 and np.signbit(a[3]) and a1[3] == 1.0


def test_checkpoint_path_without_suffix_and_point_array_validation(tmp_path):
    """np.savez appends '.npz': saving and loading with the same suffix-less path must round-trip; point arrays whose row
    length is not the problem dimension are refused before the C side reads past them."""
    o, m = _pair_2d("poisson2d_small", 1, layers=[2, 20, 20, 20, 1])
    m._step(3, False)
    p = str(tmp_path / "ckpt")
    m.save_checkpoint(p)
    th = m.get_params()
    m._step(2, False)
    m.load_checkpoint(p)
    assert np.array_equal(m.get_params(), th)
    with pytest.raises(ValueError):
        m.h.predict(np.zeros((5, 1)))
    with pytest.raises(ValueError):
        m.h.set_data(np.zeros(6), np.zeros(6))
    with pytest.raises(ValueError):
        m.h.set_data(np.zeros((4, 3)), np.zeros(4))


def test_rectangular_tensor_rule_is_accepted():
    """qx != qy: the C ABI takes the two 1-D rules separately, and the class recovers them from the flattened arrays."""
    from hp_vpinns_amd import GaussLobattoJacobiWeights
    from hp_vpinns_amd.vpinn import _tensor_rule
    X8, W8 = GaussLobattoJacobiWeights(8, 0, 0)
    X6, W6 = GaussLobattoJacobiWeights(6, 0, 0)
    xx, yy = np.meshgrid(X8, X6)
    wx, wy = np.meshgrid(W8, W6)
    XY = np.stack([xx.ravel(), yy.ravel()], 1)
    WXY = np.stack([wx.ravel(), wy.ravel()], 1)
    xi, wxi, yi, wyi = _tensor_rule(XY, WXY)
    assert np.array_equal(xi, X8) and np.array_equal(yi, X6) and np.array_equal(wxi, W8) and np.array_equal(wyi, W6)
    # through the library (generic kernels) against the closed-form numpy restatement is covered for square rules; here
    # the handle must at least accept the shapes and produce the zero-network known answer sum_e mean(F_e^2)
    from hp_vpinns_amd import _lib
    from hp_vpinns_amd.testfcn import tables_1d
    h = _lib.Handle(_lib.PDE_POISSON2D, 1, _lib.ACT_TANH, [2, 8, 8, 1], lossb_weight=10, backend=_lib.BACKEND_GENERIC)
    h.set_quadrature(xi, wxi, yi, wyi)
    h.set_tables(tables_1d(3, xi), tables_1d(2, yi))
    g = np.array([-1.0, 0.0, 1.0])
    h.set_elements(g, g)
    F = np.random.default_rng(0).normal(size=(2, 2, 2, 3))
    h.set_rhs(F.reshape(-1))
    h.set_params(np.zeros(h.num_params()))
    l3 = h.loss_and_grad(False)[0]
    assert abs(l3[2] - (F ** 2).mean(axis=(2, 3)).sum()) < 1e-12 * l3[2]


def test_reduce_buffer_is_a_zero_copy_torch_view():
    """The multi-GPU path all-reduces the library-owned packed buffer through a torch tensor that aliases
    it (__cuda_array_interface__); check the aliasing and the layout [grad | lossv | w*lossb | msq | pad]."""
    import torch
    from hp_vpinns_amd.dist import Reducer
    o, m = _pair_2d("poisson2d_small", 1, layers=[2, 20, 20, 20, 1])
    l3, g = m.loss_and_grad()
    ptr, n = m.h.reduce_buffer()
    assert n == g.size + 4
    t = Reducer(ptr, n, 0).tensor
    assert t.data_ptr() == ptr and t.dtype == torch.float64 and t.is_cuda
    v = t.cpu().numpy()
    assert np.array_equal(v[:g.size], g)
    assert abs(v[g.size] - l3[2]) == 0 and abs(v[g.size] + v[g.size + 1] - l3[0]) < 1e-15 * abs(l3[0])
    # single-process NCCL (RCCL) group: the collective path end to end on one GPU
    import os
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        dist.all_reduce(t)
        torch.cuda.synchronize()
        assert np.array_equal(t.cpu().numpy(), v)
        # the exact code path N GPUs run (forward_backward -> all_reduce -> apply_adam on torch's stream),
        # forced on with a 1-rank RCCL group: must reproduce the single-process trajectory
        os.environ["HPV_FORCE_DIST"] = "1"
        try:
            # default multi-GPU exchange: ncclAllReduce issued by the library inside its own iteration graphs
            o1, m1 = _pair_2d("poisson2d_small", 1, layers=[2, 20, 20, 20, 1])
            assert m1._dist and m1.exchange() == "rccl" and m1.h.exchange_in_use() == "rccl"
            _check_loss_grad(o1, m1)
            _check_traj(o1, m1, n=4)
            for _ in range(27):
                o1.adam_step()
            l3 = m1._step(27, True)           # 3 replays of the 8-iteration graph + one 3-iteration graph, all-reduce captured
            assert abs(l3[0] - float(o1.loss_parts()[0])) < TRAJ_TOL * abs(l3[0])
            assert rel(m1.get_params(), o1.get_params()) < TRAJ_TOL
            hist = m1._step_record(9)[0]
            for _ in range(9):
                o1.adam_step()
            assert abs(hist[-1, 0] - float(o1.loss_parts()[0])) < TRAJ_TOL * abs(hist[-1, 0])
            # the strong-form PINN branch through the same exchange (collocation points shard over the ranks)
            from hp_vpinns_amd.vpinn import VPINN2D
            from oracle.vpinn_oracle import OracleVPINN2D
            ap = p2_args(gold("poisson2d_small"), layers=[2, 20, 20, 20, 1])
            thp = theta0(ap[13], 31)
            op, mp_ = OracleVPINN2D(*ap, scheme="PINNs", init_params=thp), VPINN2D(*ap, scheme="PINNs", init_params=thp)
            assert mp_.exchange() == "rccl"
            _check_loss_grad(op, mp_)
            _check_traj(op, mp_, n=4)
            del m1, mp_
            # the fallback: torch.distributed all_reduce on torch's stream (HPV_EXCHANGE=torch)
            os.environ["HPV_EXCHANGE"] = "torch"
            o2, m2 = _pair_2d("poisson2d_small", 1, layers=[2, 20, 20, 20, 1])
            assert m2._dist and m2._reducer.active and m2.exchange() == "torch"
            _check_loss_grad(o2, m2)
            _check_traj(o2, m2, n=6)
            # 27 iterations in one call: 1 eager + 3 replays of the captured 8-iteration graph
            # (kernels + all-reduce + Adam inside one hipGraph) + 2 eager
            for _ in range(27):
                o2.adam_step()
            l3 = m2._step(27, True)
            assert m2._dist_graphs.get(8) is not None, "multi-GPU iteration graph was not captured"
            assert abs(l3[0] - float(o2.loss_parts()[0])) < TRAJ_TOL * abs(l3[0])
            assert rel(m2.get_params(), o2.get_params()) < TRAJ_TOL
            o3, m3 = _pair_2d("poisson2d_small", 1, layers=[2, 20, 20, 20, 1])
            hist = m3._step_record(21)[0]        # loss history through the multi-GPU path (k_adam records the reduced buffer)
            lo3 = []
            for _ in range(8):
                o3.adam_step()
                lo3.append(float(o3.loss_parts()[0]))
            assert rel(hist[:8, 0], lo3) < TRAJ_TOL and np.all(np.isfinite(hist))
            l3b = m2._step(11, True)             # a second size: 11-iteration graph
            for _ in range(11):
                o2.adam_step()
            assert m2._dist_graphs.get(11) is not None
            assert abs(l3b[0] - float(o2.loss_parts()[0])) < TRAJ_TOL * abs(l3b[0])
        finally:
            del os.environ["HPV_FORCE_DIST"]
            os.environ.pop("HPV_EXCHANGE", None)
    finally:
        dist.destroy_process_group()


def test_drivers_end_to_end_training_reduces_the_loss():
    """The restated drivers call the classes with the reference's argument lists; a short training run must
    drive the loss down and keep recording semantics (every 10 its / every it)."""
    from hp_vpinns_amd.drivers import advdiff, poisson1d, poisson2d
    r1 = poisson1d.run(Opt_Niter=301, N_Element=3, Net_layer=[1, 20, 20, 20, 1], verbose=False)
    rec = np.array(r1["total_record"])
    assert [int(v) for v in rec[:4, 0]] == [0, 10, 20, 30] and rec[-1, 1] < rec[0, 1]
    assert r1["u_pred"].shape == (2001, 1) and np.isfinite(r1["rel_l2"])
    r2 = poisson2d.run(n_iter=200, Net_layer=[2, 20, 20, 20, 1], verbose=False)
    assert len(r2["loss_his"]) == 200 and r2["loss_his"][-1] < r2["loss_his"][0]
    assert r2["u_pred"].shape == (40401, 1)
    r3 = advdiff.run(Opt_Niter=201, Net_layer=[2, 20, 20, 20, 1], verbose=False)
    assert len(r3["total_record"]) == 21 and r3["epsilon"] < 1.0       # moves towards 0.1/pi from 1.0


def test_device_rhs_assembly_matches_reference_fixtures():
    """Row N1: F_ext_total assembled by the projection kernel == the reference's own F_ext_total."""
    from hp_vpinns_amd.drivers import poisson1d, poisson2d
    from hp_vpinns_amd.rhs import assemble_F_ext_1d, assemble_F_ext_2d
    g = gold("poisson2d_cfg4")
    F = assemble_F_ext_2d(poisson2d.f_ext, g["grid_x"], g["grid_y"], 10, 10, 20)
    assert F.shape == g["F_ext_total"].shape and rel(F, g["F_ext_total"]) < 1e-13
    g = gold("poisson2d_small")
    F = assemble_F_ext_2d(poisson2d.f_ext, g["grid_x"], g["grid_y"], 4, 3, 6)      # generic-kernel shape
    assert rel(F, g["F_ext_total"]) < 1e-13
    g = gold("poisson1d_cfg2")
    F = assemble_F_ext_1d(poisson1d.f_ext, g["grid"], 60, 80)
    assert F.shape == g["F_ext_total"].shape and rel(F, g["F_ext_total"]) < 1e-12
    s = poisson2d.setup(N_el_x=8, N_el_y=8, with_test_grid=False, assemble="device")
    assert rel(s["F_ext_total"], gold("poisson2d_cfg3")["F_ext_total"]) < 1e-13


def test_device_gll_rule_and_test_function_tables_match_reference_fixtures():
    """Row N1: Gauss-Lobatto-Legendre nodes/weights and the phi / phi' / phi'' tables generated on the device against the
    fixtures produced by running the reference's quadrature module and VPINN.Test_fcn / dTest_fcn."""
    from hp_vpinns_amd import _lib
    h = _lib.Handle(_lib.PDE_POISSON2D, 1, _lib.ACT_TANH, [2, 20, 1])
    gq, gt = gold("quadrature"), gold("testfcn")
    for q in (5, 10, 20, 80):
        x, w = h.gll_rule(q)
        assert np.abs(x - gq[f"gll_x_{q}"]).max() < 2e-15 and rel(w, gq[f"gll_w_{q}"]) < 1e-13
        assert abs(w.sum() - 2.0) < 1e-14
    for nt, q in ((60, 80), (5, 10), (10, 20)):
        tab = h.test_tables(nt, gq[f"gll_x_{q}"])
        for d, name in enumerate(("phi", "dphi", "d2phi")):
            ref = gt[f"{name}_{nt}_{q}"][:, :, 0]
            assert rel(tab[d], ref) < 1e-12, (nt, q, name, rel(tab[d], ref))


def test_loss_history_equals_a_forward_pass_after_every_update():
    """P2:243-244 records the loss after every update with a second forward pass per iteration; the device-side history
    (hpv_step_record) must give the same values (the next iteration's forward pass computes them anyway) and leave the
    parameters bit-identical -- also across the history capacity and against the oracle's per-iteration losses."""
    from hp_vpinns_amd import _lib
    o, m1 = _pair_2d("poisson2d_small", 1, layers=[2, 20, 20, 20, 1])
    _, m2 = _pair_2d("poisson2d_small", 1, layers=[2, 20, 20, 20, 1])
    ref = np.array([m1._step(1, True) for _ in range(25)])
    hist = m2._step_record(25)[0]
    assert hist.shape == (25, 3) and rel(hist, ref) < 1e-13
    assert np.array_equal(m1.get_params(), m2.get_params())
    lo = []
    for _ in range(10):
        o.adam_step()
        lo.append(float(o.loss_parts()[0]))
    assert rel(hist[:10, 0], lo) < TRAJ_TOL
    n = _lib.HIST_CAP + 37                       # more iterations than the history holds: read back in chunks
    ref2 = m1._step(n, True)
    hist2 = m2._step_record(n)[0]
    assert hist2.shape == (n, 3) and rel(hist2[-1], ref2) < 1e-13 and np.array_equal(m1.get_params(), m2.get_params())
    m3 = _pair_2d("poisson2d_small", 1, layers=[2, 20, 20, 20, 1])[1]
    m3.loss_his = []
    m3.train(30)                                 # the class's own loop (record_every = 1) goes through the history
    assert rel(m3.loss_his[:25], ref[:, 0]) < 1e-13 and len(m3.loss_his) == 30


def test_chunked_recording_keeps_the_reference_semantics():
    """P1:208-218 / P3:309-330: the loss (and epsilon) recorded every 10th iteration AFTER that iteration's update, and the
    loop left at the first recorded loss below the threshold.  The chunked runs (device-side history, one read-back per
    chunk, roll-back to the stop iteration) must give the records and final parameters of iteration-by-iteration stepping."""
    from hp_vpinns_amd.vpinn import VPINN1D, VPINNAdvDiff
    g = gold("poisson1d_small")
    a = p1_args(g)
    th = theta0(a[8], 11)
    ref = VPINN1D(*a, var_form=1, init_params=th)
    ref_rec = []
    for it in range(73):                         # one iteration at a time, a forward pass after every update
        l3 = ref._step(1, True)
        if it % 10 == 0:
            ref_rec.append((it, float(l3[0])))
    for chunk in (1000, 7):
        m = VPINN1D(*a, var_form=1, init_params=th, total_record=[])
        m._RECORD_CHUNK = chunk
        m.train(73, 0.0)
        assert [int(r[0]) for r in m.total_record] == [r[0] for r in ref_rec]
        assert rel([r[1] for r in m.total_record], [r[1] for r in ref_rec]) < 1e-13
        assert np.array_equal(m.get_params(), ref.get_params())
    tresh = 0.5 * (ref_rec[3][1] + ref_rec[4][1])          # first recorded loss below it: iteration 40
    assert ref_rec[4][1] < tresh < ref_rec[3][1]
    m = VPINN1D(*a, var_form=1, init_params=th, total_record=[])
    m.train(73, tresh)
    assert [int(r[0]) for r in m.total_record] == [0, 10, 20, 30, 40]
    stop = VPINN1D(*a, var_form=1, init_params=th)
    stop._step(41, False)                        # the reference leaves the loop right after iteration 40's update
    assert np.array_equal(m.get_params(), stop.get_params())
    # AdvDiff: epsilon travels with the records
    g3 = gold("advdiff_small")
    a3 = p3_args(g3)
    th3 = theta0(a3[12], 9, extra=[1.0])
    r3 = VPINNAdvDiff(*a3, init_params=th3)
    ref3 = []
    for it in range(31):
        l3 = r3._step(1, True)
        if it % 10 == 0:
            ref3.append((it, float(l3[0]), float(r3.epsilon[0])))
    m3 = VPINNAdvDiff(*a3, init_params=th3)
    m3._RECORD_CHUNK = 16
    out = m3.train(200, 0.0)                     # the first 90 % go through the history, the rest step by step
    rec = out[1]
    assert [int(r[0]) for r in rec[:4]] == [0, 10, 20, 30] and len(rec) == 20
    assert rel([r[1] for r in rec[:4]], [r[1] for r in ref3]) < 1e-12
    assert rel([float(r[2][0]) for r in rec[:4]], [r[2] for r in ref3]) < 1e-13


def test_checkpoint_resume_is_bit_exact_and_l2_error(tmp_path):
    """Row N4: save after 7 iterations, restore into a fresh model, continue: identical to 15 straight."""
    from hp_vpinns_amd.vpinn import VPINN2D
    g = gold("poisson2d_small")
    a = p2_args(g, layers=[2, 20, 20, 20, 1])
    th = theta0(a[13], 33)
    m1 = VPINN2D(*a, init_params=th)
    m1._step(15, False)
    m2 = VPINN2D(*a, init_params=th)
    m2._step(7, False)
    ck = str(tmp_path / "ck.npz")
    m2.save_checkpoint(ck)
    m3 = VPINN2D(*a, init_params=np.zeros_like(th))
    m3.load_checkpoint(ck)
    m3._step(8, False)
    assert np.array_equal(m3.h.get_state(), m1.h.get_state())
    X = np.random.default_rng(1).uniform(-1, 1, (4000, 2))          # predict through the MFMA path
    u = m1.predict(X)
    o = __import__("oracle.vpinn_oracle", fromlist=["x"]).OracleVPINN2D(*a, init_params=m1.get_params())
    assert rel(u, o.predict(X)) < 1e-12
    assert abs(m1.rel_l2_error(X, u + 1e-3 * np.abs(u)) - 1e-3) < 1e-6


@pytest.mark.parametrize("backend,layers", [("generic", [2, 8, 8, 1]), ("mfma", [2, 20, 20, 20, 1])])
def test_pinn_strong_form_branch(backend, layers):
    """Row N3: scheme='PINNs' (P2:128-129): loss = 10 lossb + mean((u_xx+u_yy-f)^2) at the collocation points."""
    from hp_vpinns_amd.vpinn import VPINN2D
    from oracle.vpinn_oracle import OracleVPINN2D
    a = p2_args(gold("poisson2d_default"), layers)
    th = theta0(a[13], 44)
    o = OracleVPINN2D(*a, scheme="PINNs", init_params=th)
    m = VPINN2D(*a, scheme="PINNs", init_params=th, backend=backend)
    assert m.backend() == backend or backend == "mfma"
    _check_loss_grad(o, m)
    _check_traj(o, m, n=8)
    assert m.backend() == backend


# ---------------------------------------------------------------------------------------------------
# Edge cases of the boundary: empty sets, empty shards, non-uniform grids, recording semantics, full size.
# ---------------------------------------------------------------------------------------------------
def test_no_boundary_points_and_empty_element_shard():
    """n_data = 0 -> loss = lossv; a handle that owns NO elements (more ranks than elements) contributes only
    its data term; the two partial buffers still add up to the full loss / gradient."""
    from hp_vpinns_amd import _lib
    from hp_vpinns_amd.testfcn import tables_1d
    from hp_vpinns_amd.vpinn import VPINN2D, _tensor_rule
    g = gold("poisson2d_small")
    a = p2_args(g, layers=[2, 20, 20, 20, 1])
    th = theta0(a[13], 5)
    full = VPINN2D(*a, init_params=th)
    l3, gr = full.loss_and_grad()
    xi, wx, yi, wy = _tensor_rule(a[4], a[5])

    def handle(eb, ee, with_data):
        h = _lib.Handle(_lib.PDE_POISSON2D, 1, _lib.ACT_TANH, a[13], lossb_weight=10)
        h.set_quadrature(xi, wx, yi, wy)
        h.set_tables(tables_1d(full.Ntestx, xi), tables_1d(full.Ntesty, yi))
        h.set_elements(a[8], a[9], eb, ee)
        h.set_rhs(np.asarray(a[7]).reshape(-1))
        if with_data:
            h.set_data(a[0], np.asarray(a[1]).reshape(-1))
        h.set_params(th)
        return h
    hv = handle(0, 6, False)                      # all elements, no boundary points
    l3v, gv = hv.loss_and_grad(True)
    assert abs(l3v[0] - l3v[2]) == 0 and abs(l3v[2] - l3[2]) < 1e-12 * abs(l3[2]) and l3v[1] == 0
    hd = handle(3, 3, True)                       # empty shard, boundary points only
    l3d, gd = hd.loss_and_grad(True)
    assert l3d[2] == 0 and abs(l3d[0] - (l3[0] - l3[2])) < 1e-12 * abs(l3[0])
    assert rel(gv + gd, gr) < 1e-12
    hv.step(3, True)
    hd.step(3, True)                              # Adam on a data-only handle runs too


def test_nonuniform_grid_published_3_element_run():
    """The reference's published run: grid [-1,-0.1,0.1,1] (P1:270-273), 60 test functions, 80 GLL points."""
    o, m = _pair_1d("poisson1d_ne3", 1)
    _check_loss_grad(o, m)
    _check_traj(o, m, n=5)


def test_reference_default_networks():
    """Reference-default shapes: P1 [1,20,20,20,20,1] sin (P1:236), P2/P3 [2,5,5,5,1] tanh (P2:280, P3:46)."""
    from hp_vpinns_amd.vpinn import VPINN1D, VPINN2D, VPINNAdvDiff
    from oracle.vpinn_oracle import OracleVPINN1D, OracleVPINN2D, OracleVPINNAdvDiff
    a = p1_args(gold("poisson1d_default"))
    assert a[8] == [1, 20, 20, 20, 20, 1]
    th = theta0(a[8], 1)
    _check_loss_grad(OracleVPINN1D(*a, init_params=th), VPINN1D(*a, init_params=th))
    a = p2_args(gold("poisson2d_default"))
    assert a[13] == [2, 5, 5, 5, 1]
    th = theta0(a[13], 2)
    o, m = OracleVPINN2D(*a, init_params=th), VPINN2D(*a, init_params=th)
    assert m.backend() == "mfma" and m._dev_layers == [2, 20, 20, 20, 1]   # 5-wide net zero-padded onto the MFMA kernels
    _check_loss_grad(o, m)
    _check_traj(o, m, n=5)
    mg = VPINN2D(*a, init_params=th, backend="generic")      # the same network on the generic kernels, unpadded
    assert mg.backend() == "generic"
    l3p, gp = VPINN2D(*a, init_params=th).loss_and_grad()
    l3g, gg = mg.loss_and_grad()
    assert gp.shape == gg.shape == th.shape and rel(gp, gg) < 1e-12 and rel(l3p, l3g) < 1e-13
    m.h.step(40, False)                                      # padding stays exactly zero under training
    full = m.h.get_params()
    pad = np.ones(full.size, bool)
    pad[m._pad_idx] = False
    assert pad.sum() > 0 and np.all(full[pad] == 0.0)
    a = p3_args(gold("advdiff_default"))
    th = theta0(a[12], 3, extra=[1.0])
    o, m = OracleVPINNAdvDiff(*a, init_params=th), VPINNAdvDiff(*a, init_params=th)
    _check_loss_grad(o, m)
    _check_traj(o, m, n=5)


def test_train_recording_semantics_and_early_stop():
    """P1:201-224: recorded every 10 iterations after the update, early exit below `tresh`; P2:243-244: every
    iteration; P3:341: the 5-tuple with [it, loss, epsilon, 1] records."""
    from hp_vpinns_amd.vpinn import VPINN1D, VPINN2D, VPINNAdvDiff
    from oracle.vpinn_oracle import OracleVPINN1D
    a = p1_args(gold("poisson1d_small"), layers=[1, 20, 20, 20, 1])
    th = theta0(a[8], 9)
    rec = []
    m = VPINN1D(*a, init_params=th, total_record=rec)
    m.train(35, 0.0)
    assert [int(r[0]) for r in rec] == [0, 10, 20, 30]
    o = OracleVPINN1D(*a, init_params=th)
    ro = o.train(35, 0.0)
    assert rel([r[1] for r in rec], [r[1] for r in ro]) < 1e-7
    m2 = VPINN1D(*a, init_params=th)
    m2.train(1000, 1e30)                                   # first recorded loss is already below tresh -> stops at it 0
    assert len(m2.total_record) == 1
    o1 = OracleVPINN1D(*a, init_params=th)
    o1.adam_step()
    assert rel(m2.get_params(), o1.get_params()) < 1e-9   # exactly one update happened before the stop
    a2 = p2_args(gold("poisson2d_small"), layers=[2, 20, 20, 20, 1])
    m3 = VPINN2D(*a2, init_params=theta0(a2[13], 9))
    m3.train(7)
    assert len(m3.loss_his) == 7
    a3 = p3_args(gold("advdiff_small"), layers=[2, 20, 20, 20, 1])
    m4 = VPINNAdvDiff(*a3, init_params=theta0(a3[12], 9, extra=[1.0]))
    out = m4.train(21, 0.0)
    assert len(out) == 5 and [int(r[0]) for r in out[1]] == [0, 10, 20] and out[1][0][3] == 1


def test_full_size_config4_properties():
    """BASELINE config 4 at full size (102 400 points): size-independent properties instead of the oracle --
    (i) two half-domain shards sum to the whole; (ii) loss(theta = 0) = sum_e mean(F_e^2) + w*mean(u_d^2);
    (iii) the separate projection + reverse launches (HPV_FUSE=n) agree with the default fused reverse kernel."""
    import os
    from hp_vpinns_amd.drivers import poisson2d
    from hp_vpinns_amd.init import xavier_init
    s = poisson2d.setup(N_el_x=16, N_el_y=16, N_test_x=10, N_test_y=10, N_quad=20, with_test_grid=False)
    L = [2, 20, 20, 20, 1]
    th = xavier_init(L, 77)
    m = poisson2d.build_model(s, L, init_params=th)
    l3, g = m.loss_and_grad()
    m0 = poisson2d.build_model(s, L, init_params=np.zeros_like(th))
    z3 = m0.loss()
    F = s["F_ext_total"]
    assert abs(z3[2] - (F ** 2).mean(axis=(2, 3)).sum()) < 1e-10 * z3[2]
    assert abs(z3[1] - (s["u_train"] ** 2).mean()) < 1e-13
    os.environ["HPV_FUSE"] = "n"
    try:
        mw = poisson2d.build_model(s, L, init_params=th)
        l3w, gw = mw.loss_and_grad()
        assert "k_project_wg<20x20/10x10>" in mw.h.kernel_variant()      # 256 elements: one workgroup per element
        os.environ["HPV_PJ_WG_SMALL"] = "0"                               # ... against "a lane owns a line" on the same grid
        from hp_vpinns_amd import _lib as _l
        with _l.library(_l.TEST_HOOKS_LIB_PATH):                          # (an A/B switch of the -DHPV_EXPERIMENTS build)
            mt = poisson2d.build_model(s, L, init_params=th)
            l3t, gt = mt.loss_and_grad()
            assert "k_project_tp<20x20/10x10>" in mt.h.kernel_variant()
    finally:
        del os.environ["HPV_FUSE"]
        os.environ.pop("HPV_PJ_WG_SMALL", None)
    assert rel(gw, g) < 1e-11 and rel(l3w, l3) < 1e-13
    assert rel(gt, g) < 1e-11 and rel(l3t, l3) < 1e-13
    from hp_vpinns_amd import _lib
    from hp_vpinns_amd.testfcn import tables_1d
    from hp_vpinns_amd.vpinn import _tensor_rule
    xi, wx, yi, wy = _tensor_rule(s["XY_quad_train"], s["WXY_quad_train"])
    acc = 0
    for (b, e, d) in ((0, 128, True), (128, 256, False)):
        h = _lib.Handle(_lib.PDE_POISSON2D, 1, _lib.ACT_TANH, L, lossb_weight=10)
        h.set_quadrature(xi, wx, yi, wy)
        h.set_tables(tables_1d(10, xi), tables_1d(10, yi))
        h.set_elements(s["grid_x"], s["grid_y"], b, e)
        h.set_rhs(F.reshape(-1))
        if d:
            h.set_data(s["X_u_train"], s["u_train"].reshape(-1))
        h.set_params(th)
        acc = acc + h.loss_and_grad(True)[1]
    assert rel(acc, g) < 1e-11


def test_large_grid_known_answer_and_shards():
    """Maximum sizes: the config-4 element shape on a 64x64-element grid (1.6 M quadrature points, 1.8 GB activation
    store): zero-network known answer, and the gradient of two half-domain shards sums to the whole (index widths,
    grid-stride loops, the fused kernel with more workgroups than CUs)."""
    from hp_vpinns_amd.drivers import poisson2d
    from hp_vpinns_amd.init import xavier_init
    s = poisson2d.setup(N_el_x=64, N_el_y=64, N_test_x=10, N_test_y=10, N_quad=20, with_test_grid=False, assemble="device")
    L = [2, 20, 20, 20, 1]
    m0 = poisson2d.build_model(s, L, init_params=np.zeros(921))
    z3 = m0.loss()
    assert abs(z3[2] - (s["F_ext_total"] ** 2).mean(axis=(2, 3)).sum()) < 1e-10 * z3[2]
    del m0
    th = xavier_init(L, 3)
    m = poisson2d.build_model(s, L, init_params=th)
    l3, g = m.loss_and_grad()
    del m
    half = poisson2d.setup(N_el_x=64, N_el_y=64, N_test_x=10, N_test_y=10, N_quad=20, with_test_grid=False, assemble="device")
    acc_l, acc_g = 0.0, 0.0
    for sl in (slice(0, 32), slice(32, 64)):     # the two x-halves as separate problems on their own grids
        hs = dict(half)
        hs["grid_x"] = half["grid_x"][sl.start:sl.stop + 1]
        hs["F_ext_total"] = half["F_ext_total"][sl]
        hs["N_testfcn_total"] = [half["N_testfcn_total"][0][sl], half["N_testfcn_total"][1]]
        if sl.start:                             # boundary term only once
            hs["X_u_train"], hs["u_train"] = half["X_u_train"][:0], half["u_train"][:0]
        mh = poisson2d.build_model(hs, L, init_params=th)
        lh, gh = mh.loss_and_grad()
        acc_l, acc_g = acc_l + lh[0], acc_g + gh
        del mh
    assert abs(acc_l - l3[0]) < 1e-11 * abs(l3[0]) and rel(acc_g, g) < 1e-10


@pytest.mark.parametrize("vf", [1, 2, 3])
def test_tall_element_projection_1d(vf):
    """(80 quad, 60 test) 1-D elements go through the workgroup-per-element projection kernel, incl. the edge
    term of var_form 3 (P1:89-91)."""
    o, m = _pair_1d("poisson1d_ne3", vf)
    _check_loss_grad(o, m)
    _check_traj(o, m, n=4)


@pytest.mark.parametrize("vf", [0, 1])
def test_tall_element_projection_advdiff_config5(vf):
    """BASELINE config 5: AdvDiff, 8 elements, 80x80 GLL points and 5x5 test functions per element."""
    o, m = _pair_adv("advdiff_cfg5", vf)
    _check_loss_grad(o, m)


@pytest.mark.parametrize("nex,ney", [(2, 3), (8, 8), (8, 16)])
def test_fused_projection_reverse_config4_element_shape(nex, ney):
    """20x20-point / 10x10-test elements (BASELINE config 4 shape): the reverse kernel runs in element-block mode with
    the projection fused in, and small shards (what one GPU of an 8-GPU run owns) spread an element over 8 / 4 / 2
    workgroups that each project it; parity against the oracle (smallest case) and against the unfused launches."""
    import os
    from hp_vpinns_amd.drivers import poisson2d
    from hp_vpinns_amd.init import xavier_init
    from oracle.vpinn_oracle import OracleVPINN2D
    s = poisson2d.setup(N_el_x=nex, N_el_y=ney, N_test_x=10, N_test_y=10, N_quad=20, N_bound=13, with_test_grid=False)
    L = [2, 20, 20, 20, 1]
    th = xavier_init(L, 5)
    m = poisson2d.build_model(s, L, init_params=th)
    assert m.backend() == "mfma"
    if nex * ney <= 6:
        o = OracleVPINN2D(s["X_u_train"], s["u_train"], s["X_f_train"], s["f_train"], s["XY_quad_train"], s["WXY_quad_train"],
                          None, s["F_ext_total"], s["grid_x"], s["grid_y"], s["N_testfcn_total"], None, None, L, init_params=th)
        o.vectorized = True
        _check_loss_grad(o, m)
        _check_traj(o, m, n=6)
        m = poisson2d.build_model(s, L, init_params=th)
    l3f, gf = m.loss_and_grad()
    r_f = m.h.residuals(nex * ney * 100)
    os.environ["HPV_FUSE"] = "n"      # separate projection and reverse launches (read when the handle is built)
    try:
        m2 = poisson2d.build_model(s, L, init_params=th)
        l3, g = m2.loss_and_grad()
        r_u = m2.h.residuals(nex * ney * 100)
    finally:
        del os.environ["HPV_FUSE"]
    assert rel(gf, g) < 1e-12 and rel(l3f, l3) < 1e-13 and rel(r_f, r_u) < 1e-12
    for _ in range(2):                # duplicate projections by the workgroups of an element: results must be reproducible
        l3b, gb = m.loss_and_grad()
        assert np.array_equal(gb, gf) and np.array_equal(l3b, l3f)
    # the element-resident whole-iteration kernel (forced also for these small shards) against the separate launches
    os.environ["HPV_FUSE"] = "i"
    try:
        m3 = poisson2d.build_model(s, L, init_params=th)
        l3i, gi = m3.loss_and_grad()
        r_i = m3.h.residuals(nex * ney * 100)
        for _ in range(2):
            l3b, gb = m3.loss_and_grad()
            assert np.array_equal(gb, gi) and np.array_equal(l3b, l3i)
        m3._step(5, False)
        m2._step(5, False)
        assert rel(m3.get_params(), m2.get_params()) < 1e-11
    finally:
        del os.environ["HPV_FUSE"]
    assert rel(gi, g) < 1e-12 and rel(l3i, l3) < 1e-13 and rel(r_i, r_u) < 1e-12


@pytest.mark.parametrize("nhid", [2, 3])
def test_whole_iteration_kernel_depths_and_forms(nhid):
    """kernels_fused.hip on the config-4 element shape: 2 and 3 hidden layers, against the oracle (small grid, forced) --
    loss, gradient, residuals, a short Adam trajectory; var_form 2 is a two-term form that is NOT one-hot and must fall
    back to the separate kernels with identical results."""
    import os
    from hp_vpinns_amd.drivers import poisson2d
    from hp_vpinns_amd.init import xavier_init
    from oracle.vpinn_oracle import OracleVPINN2D
    s = poisson2d.setup(N_el_x=2, N_el_y=2, N_test_x=10, N_test_y=10, N_quad=20, N_bound=9, with_test_grid=False)
    L = [2] + [20] * nhid + [1]
    th = xavier_init(L, 8)
    os.environ["HPV_FUSE"] = "i"
    try:
        for vf in (1, 2):
            m = poisson2d.build_model(s, L, var_form=vf, init_params=th)
            o = OracleVPINN2D(s["X_u_train"], s["u_train"], s["X_f_train"], s["f_train"], s["XY_quad_train"], s["WXY_quad_train"],
                              None, s["F_ext_total"], s["grid_x"], s["grid_y"], s["N_testfcn_total"], None, None, L,
                              var_form=vf, init_params=th)
            o.vectorized = True
            _check_loss_grad(o, m)
            assert rel(m.h.residuals(400), o.last["R"].reshape(-1)) < TOL
            _check_traj(o, m, n=5)
    finally:
        del os.environ["HPV_FUSE"]


@pytest.mark.parametrize("nex,ney", [(16, 8), (16, 4), (16, 2), (5, 3)])
def test_split_whole_iteration_kernel_on_small_shards(nex, ney):
    """Shards of <= 128 config-4 elements (what one GPU of a 2 / 4 / 8-GPU run owns): the whole-iteration kernel runs in SPLIT
    mode -- 2 / 4 / 8 workgroups share an element, exchange u_x, u_y through write-through stores and meet at a barrier in device
    memory.  Against the forward + split reverse kernels (HPV_FUSE=s): loss / gradient / residuals, bitwise reproducible, an Adam
    trajectory; then thousands of back-to-back launches (every one crosses the barrier) must neither time out nor drift."""
    import os
    from hp_vpinns_amd.drivers import poisson2d
    from hp_vpinns_amd.init import xavier_init
    s = poisson2d.setup(N_el_x=nex, N_el_y=ney, N_test_x=10, N_test_y=10, N_quad=20, N_bound=13, with_test_grid=False)
    L = [2, 20, 20, 20, 1]
    th = xavier_init(L, 6)
    m = poisson2d.build_model(s, L, init_params=th)
    l3, g = m.loss_and_grad()
    r = m.h.residuals(nex * ney * 100)
    for _ in range(3):
        l3b, gb = m.loss_and_grad()
        assert np.array_equal(gb, g) and np.array_equal(l3b, l3)
    os.environ["HPV_FUSE"] = "s"
    try:
        m2 = poisson2d.build_model(s, L, init_params=th)
        l3s, gs = m2.loss_and_grad()
        rs = m2.h.residuals(nex * ney * 100)
        m2._step(40, False)
    finally:
        del os.environ["HPV_FUSE"]
    assert rel(g, gs) < 1e-12 and rel(l3, l3s) < 1e-13 and rel(r, rs) < 1e-12
    m._step(40, False)
    assert rel(m.get_params(), m2.get_params()) < 1e-10
    m._step(4000, False)                 # raises HpvError (-7) if a barrier ever times out
    assert np.all(np.isfinite(m.get_params())) and np.isfinite(m.loss()[0])


@pytest.mark.parametrize("nhid", [2, 3])
def test_small_element_iteration_kernel(nhid):
    """k_iter_small (10x10-point / 5x5-test elements: one workgroup of eight waves per element, one tile per wave, BASELINE
    config 3's shape) against the oracle and, bit for bit reproducible, against the separate launches (HPV_FUSE=b); a grid
    with more boundary tiles than elements must fall back to the separate launches with the same results."""
    import os
    from hp_vpinns_amd.drivers import poisson2d
    from hp_vpinns_amd.init import xavier_init
    from oracle.vpinn_oracle import OracleVPINN2D
    L = [2] + [20] * nhid + [1]
    th = xavier_init(L, 4)
    for nex, ney, nb in ((3, 2, 20), (2, 2, 40)):          # 6 elements / 5 boundary tiles; 4 elements / 10 boundary tiles (fallback)
        s = poisson2d.setup(N_el_x=nex, N_el_y=ney, N_test_x=5, N_test_y=5, N_quad=10, N_bound=nb, with_test_grid=False)
        m = poisson2d.build_model(s, L, init_params=th)
        o = OracleVPINN2D(s["X_u_train"], s["u_train"], s["X_f_train"], s["f_train"], s["XY_quad_train"], s["WXY_quad_train"],
                          None, s["F_ext_total"], s["grid_x"], s["grid_y"], s["N_testfcn_total"], None, None, L, init_params=th)
        o.vectorized = True
        _check_loss_grad(o, m)
        assert rel(m.h.residuals(nex * ney * 25), o.last["R"].reshape(-1)) < TOL
        l3a, ga = m.loss_and_grad()
        l3b, gb = m.loss_and_grad()
        assert np.array_equal(ga, gb) and np.array_equal(l3a, l3b)
        os.environ["HPV_FUSE"] = "b"
        try:
            m2 = poisson2d.build_model(s, L, init_params=th)
            l3s, gs = m2.loss_and_grad()
        finally:
            del os.environ["HPV_FUSE"]
        assert rel(ga, gs) < 1e-12 and rel(l3a, l3s) < 1e-13
        _check_traj(o, m, n=6)


def _p2p_worker(rank, world, port, out_path):
    """One rank of the in-library exchange test: its own process, both ranks on cuda:0 (IPC works within a device)."""
    import os
    import pickle
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["HPV_EXCHANGE"] = "p2p"            # (the default, in-library RCCL, needs one GPU per rank)
    os.environ["HPV_P2P_TIMEOUT_MS"] = "1500"
    os.environ["HPV_FUSE"] = "s"                  # two processes share the GPU: no cross-workgroup barriers (SPLIT mode) here
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hp_vpinns_amd.drivers import poisson2d
        from hp_vpinns_amd.init import xavier_init
        s = poisson2d.setup(N_el_x=4, N_el_y=6, N_test_x=10, N_test_y=10, N_quad=20, N_bound=13, with_test_grid=False)
        L = [2, 20, 20, 20, 1]
        m = poisson2d.build_model(s, L, init_params=xavier_init(L, 5), device=0)
        res = {"p2p": m._p2p, "world": m.world}
        if m._p2p:
            l3, g = m.loss_and_grad()
            hist = m._step_record(9)[0]
            l3b = m._step(13, True)
            res.update(l3=l3, g=g, hist=hist, l3b=l3b, theta=m.get_params())
            m._step(1500, False)                 # many back-to-back exchanges (graph replays, both mailbox parities)
            res.update(l3c=m.loss(), theta_long=m.get_params())
            # a peer that does not arrive: rank 1 stays away for longer than the wait budget.  Rank 0 must report the
            # failure with its replica (parameters, Adam state) UNTOUCHED, and the late rank must fail as well (poisoned
            # arrival counters) instead of applying an update its peer never made.
            import time
            from hp_vpinns_amd import _lib
            dist.barrier()
            state0 = m.h.get_state()
            if rank == 1:
                time.sleep(4.0)
            try:
                m._step(2, False)
                res["timeout_raised"] = False
            except _lib.HpvError as e:
                res["timeout_raised"] = "peer did not arrive" in str(e)
            res["state_untouched"] = bool(np.array_equal(m.h.get_state(), state0))
        with open(f"{out_path}.{rank}", "wb") as f:
            pickle.dump(res, f)
    finally:
        dist.destroy_process_group()


def test_in_library_exchange_two_ranks_on_one_gpu(tmp_path):
    """The multi-GPU iteration without a collective library call: two processes (two element shards) exchange the packed
    buffer through IPC-mapped mailboxes and apply Adam in the same kernel.  Both ranks must hold bit-identical
    parameters, and losses / gradient / trajectory must equal the single-process model (up to the summation order)."""
    import pickle
    import torch.multiprocessing as mp
    from hp_vpinns_amd.drivers import poisson2d
    from hp_vpinns_amd.init import xavier_init
    out = str(tmp_path / "p2p")
    mp.spawn(_p2p_worker, args=(2, 29541, out), nprocs=2, join=True)
    r0, r1 = (pickle.load(open(f"{out}.{r}", "rb")) for r in (0, 1))
    assert r0["p2p"] and r1["p2p"] and r0["world"] == 2, "the in-library exchange did not connect"
    assert np.array_equal(r0["theta"], r1["theta"]) and np.array_equal(r0["hist"], r1["hist"]) and np.array_equal(r0["g"], r1["g"])
    assert np.array_equal(r0["theta_long"], r1["theta_long"]) and np.array_equal(r0["l3c"], r1["l3c"]) and np.all(np.isfinite(r0["theta_long"]))
    s = poisson2d.setup(N_el_x=4, N_el_y=6, N_test_x=10, N_test_y=10, N_quad=20, N_bound=13, with_test_grid=False)
    L = [2, 20, 20, 20, 1]
    m = poisson2d.build_model(s, L, init_params=xavier_init(L, 5))
    l3, g = m.loss_and_grad()
    hist = m._step_record(9)[0]
    l3b = m._step(13, True)
    assert rel(r0["l3"], l3) < 1e-12 and rel(r0["g"], g) < 1e-11
    assert rel(r0["hist"], hist) < 1e-9 and rel(r0["l3b"], l3b) < 1e-9 and rel(r0["theta"], m.get_params()) < 1e-9
    for r in (r0, r1):
        assert r["timeout_raised"] is True and r["state_untouched"] is True, (r["timeout_raised"], r["state_untouched"])


def _tile_vs_separate(build, n_res, o=None, n_traj=6):
    """default build (must run the whole-iteration tile kernel) against HPV_FUSE=n (separate launches) and, if given, the oracle."""
    import os
    m = build()
    l3, g = m.loss_and_grad()
    assert m.h.pass_structure() == "whole-iteration-tile"
    r = m.h.residuals(n_res)
    l3b, gb = m.loss_and_grad()
    assert np.array_equal(g, gb) and np.array_equal(l3, l3b)          # fixed summation order: bitwise reproducible
    os.environ["HPV_FUSE"] = "n"
    try:
        m2 = build()
        l3s, gs = m2.loss_and_grad()
        assert m2.h.pass_structure() == "separate"
        rs = m2.h.residuals(n_res)
        m2._step(30, False)
    finally:
        del os.environ["HPV_FUSE"]
    assert rel(g, gs) < 1e-11 and rel(l3, l3s) < 1e-12 and rel(r, rs) < 1e-11
    m._step(30, False)
    assert rel(m.get_params(), m2.get_params()) < 1e-9
    if o is not None:
        m3 = build()
        _check_loss_grad(o, m3)
        _check_traj(o, m3, n=n_traj)


@pytest.mark.parametrize("nhid,vf,nel", [(3, 1, 1), (3, 1, 5), (4, 1, 3), (3, 2, 3), (2, 2, 2)])
def test_tile_iteration_kernel_poisson1d(nhid, vf, nel):
    """kernels_tile.hip on the 1-D rule (80 points, 60 test functions per element; sin; P1:82-87): one workgroup per element,
    the boundary tile on a free wave -- var_form 1 (u'') and 2 (u'), 2..4 hidden layers (4 = the reference default, P1:236),
    against the separate launches and the oracle."""
    from hp_vpinns_amd.drivers import poisson1d
    from hp_vpinns_amd.init import xavier_init
    from hp_vpinns_amd.vpinn import VPINN1D
    from oracle.vpinn_oracle import OracleVPINN1D
    s = poisson1d.setup(N_Element=nel)
    L = [1] + [20] * nhid + [1]
    th = xavier_init(L, 21)
    th[L[1]:2 * L[1]] = 0.1          # (a non-zero first bias: the odd sin network otherwise has a round-off-level output-bias gradient)
    args = (s["X_u_train"], s["u_train"], s["X_quad_train"], s["W_quad_train"], s["F_ext_total"], s["grid"], s["X_test"],
            s["u_test"], L, s["X_f_train"], s["f_train"])
    build = lambda: VPINN1D(*args, var_form=vf, init_params=th)
    o = OracleVPINN1D(*args, var_form=vf, init_params=th)
    o.vectorized = True
    _tile_vs_separate(build, nel * 60, o)


@pytest.mark.parametrize("case", ["advdiff-vf0", "advdiff-vf1", "poisson2d-vf0", "poisson2d-vf2"])
def test_tile_iteration_kernel_small_2d_elements(case):
    """kernels_tile.hip on 10x10-point / 5x5-test elements with the channel sets k_iter_small does not take: AdvDiff with its
    trainable epsilon (u_t, u_x, u_xx; P3:161-174; hundreds of data points -> extra workgroups of data tiles), Poisson-2D
    var_form 0 (u_xx, u_yy) and var_form 2 (u against the second derivatives of the test functions)."""
    from hp_vpinns_amd.drivers import advdiff, poisson2d
    from hp_vpinns_amd.init import xavier_init
    L = [2, 20, 20, 20, 1]
    if case.startswith("advdiff"):
        vf = int(case[-1])
        s = advdiff.setup(N_el_x=3, N_quad=10, with_test_grid=False)
        th = xavier_init(L, 17, extra=[1.0])
        build = lambda: advdiff.build_model(s, L, var_form=vf, init_params=th)
        n_res = 3 * 25
    else:
        vf = int(case[-1])
        s = poisson2d.setup(N_el_x=3, N_el_y=2, N_test_x=5, N_test_y=5, N_quad=10, N_bound=30, with_test_grid=False)
        th = xavier_init(L, 17)
        build = lambda: poisson2d.build_model(s, L, var_form=vf, init_params=th)
        n_res = 6 * 25
    _tile_vs_separate(build, n_res)


@pytest.mark.parametrize("vf,backend", [(1, "auto"), (2, "auto"), (3, "auto"), (1, "generic")])
def test_poisson1d_per_element_test_function_counts(vf, backend):
    """p-refinement of the 1-D driver: F_ext_total[e] of different lengths (the reference reads Ntest_element =
    len(F_ext_total[e]) per element, P1:66-67, and builds the list from N_testfcn_total, P1:268-281).  Loss, gradient,
    residuals (zero rows beyond an element's count) and an Adam trajectory against the oracle's element loop."""
    from hp_vpinns_amd.drivers import poisson1d
    from hp_vpinns_amd.init import xavier_init
    from hp_vpinns_amd.vpinn import VPINN1D
    from oracle.vpinn_oracle import OracleVPINN1D
    counts = [60, 45, 7, 33]
    s = poisson1d.setup(N_Element=4, N_testfcn_total=counts)
    assert [f.shape[0] for f in s["F_ext_total"]] == counts
    L = [1, 20, 20, 20, 1]
    th = xavier_init(L, 31)
    th[L[1]:2 * L[1]] = 0.1
    args = (s["X_u_train"], s["u_train"], s["X_quad_train"], s["W_quad_train"], s["F_ext_total"], s["grid"], s["X_test"],
            s["u_test"], L, s["X_f_train"], s["f_train"])
    m = VPINN1D(*args, var_form=vf, init_params=th, backend=backend)
    o = OracleVPINN1D(*args, var_form=vf, init_params=th)
    _check_loss_grad(o, m)
    if backend == "auto" and vf != 3:
        assert m.h.pass_structure() == "whole-iteration-tile"
    r = m.h.residuals(4 * 60).reshape(4, 60)
    for e, n in enumerate(counts):
        assert np.all(r[e, n:] == 0.0) and np.count_nonzero(r[e, :n]) >= n - 3     # (a high-order row can round to exactly 0)
    _check_traj(o, m, n=8)
    # the same counts written as a dense F with trailing zeros and NO counts is a different (and wrong) loss
    Fd = np.zeros((4, 60, 1))
    for e, f in enumerate(s["F_ext_total"]):
        Fd[e, :f.shape[0]] = f
    a2 = list(args)
    a2[4] = Fd
    m2 = VPINN1D(*a2, var_form=vf, init_params=th, backend=backend)
    assert abs(m2.loss_and_grad()[0][0] - m.loss_and_grad()[0][0]) > 1e-6


@pytest.mark.parametrize("nt", [40, 5])
@pytest.mark.parametrize("vf", [1, 2])
def test_poisson1d_fewer_test_functions_run_on_the_element_resident_kernel(nt, vf):
    """N_testfcn is a free hyper-parameter of the 1-D driver (P1:238).  With the reference's 80-point rule, any count below the
    instantiated 60 runs on the whole-iteration tile kernel as a p-refinement with equal counts (vpinn.py pads tables and F, the
    library divides by the count): loss, gradient, residuals and a trajectory against the oracle built with the count itself."""
    from hp_vpinns_amd.drivers import poisson1d
    from hp_vpinns_amd.init import xavier_init
    from hp_vpinns_amd.vpinn import VPINN1D
    from oracle.vpinn_oracle import OracleVPINN1D
    s = poisson1d.setup(N_Element=3, N_testfcn_total=[nt] * 3)
    L = [1, 20, 20, 20, 1]
    th = xavier_init(L, 32)
    th[L[1]:2 * L[1]] = 0.1
    args = (s["X_u_train"], s["u_train"], s["X_quad_train"], s["W_quad_train"], s["F_ext_total"], s["grid"], s["X_test"],
            s["u_test"], L, s["X_f_train"], s["f_train"])
    m = VPINN1D(*args, var_form=vf, init_params=th)
    o = OracleVPINN1D(*args, var_form=vf, init_params=th)
    assert m.N_test == nt and o.N_test == nt
    _check_loss_grad(o, m)
    assert m.h.pass_structure() == "whole-iteration-tile", m.h.kernel_variant()
    r = m.h.residuals(3 * 60).reshape(3, 60)
    assert np.all(r[:, nt:] == 0.0) and np.count_nonzero(r[:, :nt]) >= 3 * nt - 3
    assert abs(float(m.loss_and_grad()[0][2]) - float(np.mean(r[:, :nt] ** 2, axis=1).sum())) < 1e-10 * max(1.0, float(np.abs(r).max()) ** 2)   # lossv = sum_e mean over ITS nt residuals
    _check_traj(o, m, n=8)


@pytest.mark.parametrize("backend", ["auto", "generic"])
def test_poisson1d_shards_with_test_function_counts(backend):
    """What the ranks of a multi-GPU run own: the element range [e_begin, e_end) with F and the per-element test-function
    counts sliced inside the library, the boundary term on one shard only (the other runs the tile kernel without data
    tiles).  Variational losses and gradients of the shards add up to the whole problem's."""
    from hp_vpinns_amd.drivers import poisson1d
    from hp_vpinns_amd.init import xavier_init
    from hp_vpinns_amd.vpinn import VPINN1D
    counts = [60, 20, 45, 7, 33]
    s = poisson1d.setup(N_Element=5, N_testfcn_total=counts)
    L = [1, 20, 20, 20, 1]
    th = xavier_init(L, 9)
    th[20:40] = 0.1
    args = (s["X_u_train"], s["u_train"], s["X_quad_train"], s["W_quad_train"], s["F_ext_total"], s["grid"], s["X_test"],
            s["u_test"], L, s["X_f_train"], s["f_train"])
    full = VPINN1D(*args, init_params=th, backend=backend)
    l3, g = full.h.loss_and_grad()
    parts = []
    for eb, ee, with_data in ((0, 2, True), (2, 5, False)):
        m = VPINN1D(*args, init_params=th, backend=backend)
        m.h.set_elements(s["grid"], None, eb, ee)
        if not with_data:
            m.h.set_data(None, None)
        parts.append(m.h.loss_and_grad())
        if backend == "auto":
            assert m.h.pass_structure() == "whole-iteration-tile"
        r = m.h.residuals((ee - eb) * 60).reshape(ee - eb, 60)
        for e in range(eb, ee):
            assert np.all(r[e - eb, counts[e]:] == 0.0) and np.count_nonzero(r[e - eb, :counts[e]]) >= counts[e] - 3
    (la, ga), (lb, gb) = parts
    assert lb[1] == 0.0 and rel(la[1], l3[1]) < 1e-13                      # lossb lives on the shard with the data points
    assert rel(la[2] + lb[2], l3[2]) < 1e-12 and rel(ga + gb, g) < 1e-11


def test_active_test_counts_argument_checks_and_reset():
    """hpv_set_active_tests: counts outside 1..ntest, a wrong number of entries and 2-D problems are refused (the 2-D drivers
    reshape F_ext_total into a dense array, P2:414); NULL restores the uniform case exactly."""
    from hp_vpinns_amd import _lib
    from hp_vpinns_amd.drivers import poisson1d, poisson2d
    from hp_vpinns_amd.init import xavier_init
    from hp_vpinns_amd.vpinn import VPINN1D
    s = poisson1d.setup(N_Element=3)
    L = [1, 20, 20, 1]
    th = xavier_init(L, 2)
    th[20:40] = 0.1
    m = VPINN1D(s["X_u_train"], s["u_train"], s["X_quad_train"], s["W_quad_train"], s["F_ext_total"], s["grid"], s["X_test"],
                s["u_test"], L, s["X_f_train"], s["f_train"], init_params=th)
    l0, g0 = m.h.loss_and_grad()
    for bad in ([60, 61, 10], [0, 5, 5], [10, 10]):
        with pytest.raises(_lib.HpvError):
            m.h.set_active_tests(bad)
    m.h.set_active_tests([60, 30, 60])
    l1, _ = m.h.loss_and_grad()
    assert abs(l1[2] - l0[2]) > 1e-9
    m.h.set_active_tests(None)
    l2, g2 = m.h.loss_and_grad()
    assert np.array_equal(l2, l0) and np.array_equal(g2, g0)
    s2 = poisson2d.setup(N_el_x=2, N_el_y=2, with_test_grid=False)
    m2 = poisson2d.build_model(s2, [2, 20, 20, 1], init_params=xavier_init([2, 20, 20, 1], 2))
    with pytest.raises(_lib.HpvError):
        m2.h.set_active_tests([5, 5, 5, 5])


def test_single_workgroup_grid_finishes_the_iteration_in_the_kernel():
    """BASELINE config 1 (one 1-D element = ONE workgroup): the whole-iteration tile kernel writes the packed buffer, applies TF1
    Adam and records the loss itself; against the same kernel followed by k_finalize (HPV_NO_INKERNEL_FINALIZE=1): loss triple,
    gradient bit for bit (the same reduction); 25-step trajectory, per-iteration loss history and Adam state to round-off (two
    compilations of the same update expression may contract their multiply-adds differently); and against the oracle."""
    import os
    from hp_vpinns_amd.vpinn import VPINN1D
    from oracle.vpinn_oracle import OracleVPINN1D
    a = p1_args(gold("poisson1d_cfg1"), layers=[1, 20, 20, 20, 1])
    th = theta0(a[8], 41)
    th[20:40] = 0.03 * np.arange(20)
    m = VPINN1D(*a, init_params=th)
    l3, g = m.loss_and_grad()
    assert m.h.pass_structure() == "whole-iteration-tile"
    hist = m._step_record(25)[0]
    st = m.h.get_state()
    os.environ["HPV_NO_INKERNEL_FINALIZE"] = "1"
    try:
        from hp_vpinns_amd import _lib as _l
        with _l.library(_l.TEST_HOOKS_LIB_PATH):       # (an A/B switch of the -DHPV_EXPERIMENTS build)
            m2 = VPINN1D(*a, init_params=th)
            l3b, gb = m2.loss_and_grad()
            assert m2.h.pass_structure() == "whole-iteration-tile"
            hist2 = m2._step_record(25)[0]
            st2 = m2.h.get_state()
    finally:
        del os.environ["HPV_NO_INKERNEL_FINALIZE"]
    assert np.array_equal(l3, l3b) and np.array_equal(g, gb)
    assert rel(hist, hist2) < 1e-13 and rel(st, st2) < 1e-12, (rel(hist, hist2), rel(st, st2))
    o = OracleVPINN1D(*a, init_params=th)
    o.vectorized = True
    lo = []
    for _ in range(25):
        o.adam_step()
        lo.append(float(o.loss_parts()[0]))
    assert rel(hist[:, 0], lo) < TRAJ_TOL and rel(m.get_params(), o.get_params()) < TRAJ_TOL


def test_persistent_single_workgroup_launch_is_the_same_iteration(monkeypatch):
    """HPV_PERSIST=1 (measured no faster: kernels_tile.hip, tile_body; libhpvpinn_testhooks.so only): MANY iterations of the one-workgroup grid of config 1
    in ONE launch -- the same body function behind a call.  Loss history, parameters, Adam moments and beta powers must equal the
    one-launch-per-iteration run to round-off (two compilations of the same body), the recorded iterations must be all there, and
    early-stop chunking on top of it must keep the reference's semantics."""
    from hp_vpinns_amd.vpinn import VPINN1D
    a = p1_args(gold("poisson1d_cfg1"), layers=[1, 20, 20, 20, 1])
    th = theta0(a[8], 41)
    th[20:40] = 0.03 * np.arange(20)
    ref = VPINN1D(*a, init_params=th)
    h_ref = ref._step_record(37)[0]
    ref._step(100, False)
    from hp_vpinns_amd import _lib
    monkeypatch.setenv("HPV_PERSIST", "1")
    mp = VPINN1D(*a, init_params=th)                   # the product library neither carries the persistent launch nor reads the switch
    mp._step(3, False)
    assert "persistent" not in mp.h.kernel_variant(), mp.h.kernel_variant()
    with _lib.library(_lib.TEST_HOOKS_LIB_PATH):       # -DHPV_EXPERIMENTS
        m = VPINN1D(*a, init_params=th)
    h = m._step_record(37)[0]
    assert "persistent" in m.h.kernel_variant(), m.h.kernel_variant()
    m._step(100, False)
    assert m.h.updates_applied() == 137
    assert rel(h, h_ref) < 1e-12 and rel(m.h.get_state(), ref.h.get_state()) < 1e-11
    rec = []
    with _lib.library(_lib.TEST_HOOKS_LIB_PATH):
        m2 = VPINN1D(*a, init_params=th, total_record=rec)
    m2.train(41, 0.0)
    assert [int(r[0]) for r in rec] == [0, 10, 20, 30, 40] and rel([r[1] for r in rec], h_ref[[0, 10, 20, 30], 0].tolist() + [rec[-1][1]]) < 1e-12


def test_large_batch_projection_plans_agree(monkeypatch):
    """The stand-alone projection on a LARGE synthetic batch (the HBM-roofline measurement of SURVEY.md 8d) has three plans:
    the streaming residual kernel (LDS-staged batches of 6 elements, register double-buffered: residual only), the
    column-in-registers kernel k_project_tp, and the general k_project.  On the same seeded data they must give the same R and
    element losses (sums of the same terms in different orders) -- incl. a batch size that is not a multiple of 6 or 3."""
    from hp_vpinns_amd import _lib
    from hp_vpinns_amd.quadrature import GaussLobattoJacobiWeights
    from hp_vpinns_amd.testfcn import tables_1d
    x, w = GaussLobattoJacobiWeights(20, 0, 0)

    def sums(adj, backend=_lib.BACKEND_AUTO, n=10007, experiments=False):
        if experiments:                                # the measured-slower plans live in the -DHPV_EXPERIMENTS build only
            with _lib.library(_lib.TEST_HOOKS_LIB_PATH):
                return sums(adj, backend, n)
        h = _lib.Handle(_lib.PDE_POISSON2D, 1, _lib.ACT_TANH, [2, 20, 20, 20, 1], lossb_weight=10, backend=backend)
        h.set_quadrature(x, w, x, w)
        h.set_tables(tables_1d(10, x), tables_1d(10, x))
        return h.bench_checksums(n, adj)

    monkeypatch.setenv("HPV_PJ_STREAM", "1")        # (opt-in: equal speed to k_project_tp since that kernel's tables are SGPR operands)
    s_stream = sums(False, experiments=True)
    monkeypatch.delenv("HPV_PJ_STREAM")
    monkeypatch.setenv("HPV_PJ_DMA", "1")           # the LDS-DMA stream (three batch buffers filled by global_load_lds)
    s_dma, s_dma2 = sums(False, experiments=True), sums(False, n=6 * 4096 + 5, experiments=True)
    monkeypatch.setenv("HPV_PJ_DMA", "2")           # every wave its own LDS-DMA loader and consumer
    s_wd, s_wd2 = sums(False, experiments=True), sums(False, n=6 * 4096 + 5, experiments=True)
    monkeypatch.delenv("HPV_PJ_DMA")
    s_tp = sums(False)
    assert rel(s_wd[[0, 1, 2, 5]], s_tp[[0, 1, 2, 5]]) < 1e-12, (s_wd, s_tp)
    assert rel(s_wd2[[0, 1, 2, 5]], s_dma2[[0, 1, 2, 5]]) < 1e-12, (s_wd2, s_dma2)
    assert rel(s_dma[[0, 1, 2, 5]], s_tp[[0, 1, 2, 5]]) < 1e-12, (s_dma, s_tp)
    assert rel(s_dma2[[0, 1, 2, 5]], sums(False, n=6 * 4096 + 5)[[0, 1, 2, 5]]) < 1e-12
    s_gen = sums(False, _lib.BACKEND_GENERIC)
    assert np.all(np.isfinite(s_stream)) and s_stream[1] > 0
    assert rel(s_stream[[0, 1, 2, 5]], s_tp[[0, 1, 2, 5]]) < 1e-12, (s_stream, s_tp)
    assert rel(s_tp[[0, 1, 2, 5]], s_gen[[0, 1, 2, 5]]) < 1e-12, (s_tp, s_gen)
    a_tp, a_gen = sums(True), sums(True, _lib.BACKEND_GENERIC)
    assert rel(a_tp, a_gen) < 1e-12 and a_tp[4] > 0
