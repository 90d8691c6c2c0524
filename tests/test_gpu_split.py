"""SPLIT mode of the whole-iteration kernel (kernels_fused.hip): what one GPU of a 2 / 4 / 8-GPU run of BASELINE config 4 owns
(128 / 64 / 32 elements), where 2 / 4 / 8 workgroups share an element and meet at a barrier in device memory.

  * against the ORACLE (not against another device path): loss triple, gradient, residuals and a short TF1-Adam trajectory of
    shards cut out of the reference-generated config-4 fixture with hpv_set_elements(e_begin, e_end) -- the shard's oracle is the
    vectorised restatement of P2:68-129 on the same rows of the grid and of F_ext_total; the shard gradients summed = the
    full-grid oracle's;
  * a barrier TIMEOUT (forced with the test knob HPV_DEBUG_SPLIT_SKIP=k: one partner of element 0 stays away from the k-th
    launch on) must leave the replica bit-identical -- parameters, Adam moments, beta powers -- and, with
    HPV_EXCHANGE_FALLBACK=0, raise HpvError(-7) from every entry point that runs iterations and leave the handle usable; the
    same through the 1-rank in-library RCCL path, where the failure travels in the pad slot of the all-reduced buffer;
  * by default hpv_step / hpv_step_record FINISH such a run on the launch structures without an exchange: the parameters and
    the recorded loss history equal those of an undisturbed run.
"""
import os

import numpy as np
import pytest

from cases import gold, p2_args, rel, theta0

pytestmark = pytest.mark.gpu

TOL = 1e-9
TRAJ_TOL = 1e-7
L4 = [2, 20, 20, 20, 1]


def _shard_pair(a, th, nshard, r):
    """(oracle on the rows of shard r, product re-sharded to [eb, ee)) -- both keep the boundary term."""
    from hp_vpinns_amd.dist import shard_range
    from hp_vpinns_amd.vpinn import VPINN2D
    from oracle.vpinn_oracle import OracleVPINN2D
    nex, ney = len(a[8]) - 1, len(a[9]) - 1
    eb, ee = shard_range(nex * ney, r, nshard)
    assert eb % ney == 0 and ee % ney == 0          # contiguous blocks of whole ex-rows
    rb, re_ = eb // ney, ee // ney
    m = VPINN2D(*a, init_params=th)
    m.h.set_elements(a[8], a[9], eb, ee)
    ao = list(a)
    ao[7] = a[7][rb:re_]
    ao[8] = a[8][rb:re_ + 1]
    ao[10] = [a[10][0][rb:re_], a[10][1]]
    o = OracleVPINN2D(*ao, init_params=th)
    o.vectorized = True
    return o, m, (ee - eb)


@pytest.mark.parametrize("nshard", [2, 4, 8])
def test_split_mode_shards_of_config4_against_the_oracle(nshard):
    from oracle.vpinn_oracle import OracleVPINN2D
    a = p2_args(gold("poisson2d_cfg4"), layers=L4)
    th = theta0(L4, 77)
    full = OracleVPINN2D(*a, init_params=th)
    full.vectorized = True
    l3_full, g_full = full.loss_and_grad()
    g_sum, lv_sum = np.zeros_like(g_full), 0.0
    for r in range(nshard):
        o, m, ne = _shard_pair(a, th, nshard, r)
        l3m, gm = m.loss_and_grad()
        assert m.h.pass_structure() == "whole-iteration-split", m.h.pass_structure()
        rm = m.h.residuals(ne * 100)
        g_sum += gm
        lv_sum += l3m[2]
        if r in (0, nshard // 2, nshard - 1):
            l3o, go = o.loss_and_grad()
            assert rel(l3m, l3o) < TOL, (l3m, l3o)
            assert rel(gm, go) < TOL, rel(gm, go)
            assert rel(rm, o.last["R"].reshape(-1)) < TOL
            l3b, gb = m.loss_and_grad()                  # partners project the element redundantly: bitwise reproducible
            assert np.array_equal(gb, gm) and np.array_equal(l3b, l3m)
        if r == nshard - 1:                              # a short trajectory of the last shard: 10 updates, loss after each
            lo, lm = [], []
            for _ in range(10):
                o.adam_step()
                lo.append(float(o.loss_parts()[0]))
                lm.append(float(m._step(1, True)[0]))
            assert rel(lm, lo) < TRAJ_TOL and rel(m.get_params(), o.get_params()) < TRAJ_TOL
        del m
    # every shard carried the boundary term (rank 0's job in a real run): remove the surplus copies before comparing
    lb_grad = full  # noqa: F841  (documentation: g_full = grad(lossv) + grad(10 lossb))
    ob = OracleVPINN2D(*a, init_params=th)
    ob.vectorized = True
    import torch
    lossb = 10 * torch.mean(torch.square(ob.utrain - ob.net_u(ob.x, ob.y)))
    gb_ = torch.autograd.grad(lossb, ob.theta)[0].numpy()
    assert rel(g_sum - (nshard - 1) * gb_, g_full) < TOL
    assert abs(lv_sum - l3_full[2]) < TOL * abs(l3_full[2])


def _build_small_shard(seed=6, nex=16, ney=4):
    from hp_vpinns_amd.drivers import poisson2d
    from hp_vpinns_amd.init import xavier_init
    s = poisson2d.setup(N_el_x=nex, N_el_y=ney, N_test_x=10, N_test_y=10, N_quad=20, N_bound=13, with_test_grid=False)
    return poisson2d.build_model(s, L4, init_params=xavier_init(L4, seed))


def test_split_barrier_timeout_leaves_the_replica_intact(monkeypatch):
    from hp_vpinns_amd import _lib
    monkeypatch.setenv("HPV_EXCHANGE_FALLBACK", "0")      # report the failure instead of finishing on the barrier-free kernels
    ref = _build_small_shard()
    ref._step(24, False)
    m = _with_knob(1, _build_small_shard)
    state0 = m.h.get_state()
    for call in (lambda: m._step(11, False), lambda: m._step(3, True), lambda: m._step_record(5), lambda: m.loss_and_grad(),
                 lambda: m._step(1, False)):
        with pytest.raises(_lib.HpvError, match="did not meet at their barrier") as ei:
            call()
        assert ei.value.code == _lib.EXCHANGE_TIMEOUT
        assert m.h.pass_structure() == "whole-iteration-split" and m.h.shared_element_kernels()
        assert np.array_equal(m.h.get_state(), state0), "a timed-out iteration touched the replica"
    # forward-only evaluations never enter SPLIT mode and keep working on the untouched parameters
    assert np.isfinite(m.loss()[0])
    # without the knob the same shard trains, and identically to the untouched state's continuation
    m2 = _build_small_shard()
    m2.h.set_state(state0)
    m2._step(24, False)
    assert rel(m2.get_params(), ref.get_params()) < 1e-12


def _rccl_timeout_worker(rank, port, out_path):
    import pickle
    import torch
    import torch.distributed as dist
    from hp_vpinns_amd import _lib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HPV_FORCE_DIST="1", HPV_EXCHANGE_FALLBACK="0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    res = {}
    try:
        m = _with_knob(1, _build_small_shard)
        res["exchange"] = m.exchange()
        state0 = m.h.get_state()
        raised = []
        for call in (lambda: m._step(19, False), lambda: m._step_record(4), lambda: m.loss_and_grad()):
            try:
                call()
                raised.append(False)
            except _lib.HpvError as e:
                raised.append("did not meet at their barrier" in str(e))
            res.setdefault("intact", []).append(bool(np.array_equal(m.h.get_state(), state0)))
        res["raised"] = raised
        res["structure"] = m.h.pass_structure()
        # default behaviour: the partner leaves at the 9th launch, the run is finished without the in-kernel exchange -- the
        # all-reduce inside the iteration stays in place
        del os.environ["HPV_EXCHANGE_FALLBACK"]
        ref = _build_small_shard()
        ref._step(20, False)
        res["fallback"] = []
        # (launch 9 = the first iteration of an iteration graph; 11 and 16 = in the middle / at the end of one: the update of the iteration
        #  before -- deferred into the failing launch's kernels, hpv_api.hip flush_adam -- must have been applied, the failing one not)
        for launch in (9, 11, 16):
            m2 = _with_knob(launch, _build_small_shard)
            m2._step(20, False)
            res["fallback"].append((float(rel(m2.get_params(), ref.get_params())), float(rel(m2.h.get_state(), ref.h.get_state())),
                                    m2.h.updates_applied(), m2.h.shared_element_kernels(), m2.exchange(), ref.h.pass_structure()))
            del m2
    finally:
        with open(out_path, "wb") as f:
            pickle.dump(res, f)
        dist.destroy_process_group()


def test_split_barrier_timeout_through_the_in_library_rccl_path(tmp_path):
    """forward+backward -> finalize (pad slot = this rank's flag) -> ncclAllReduce -> k_adam: the update is skipped on every rank."""
    import pickle
    import torch.multiprocessing as mp
    out = str(tmp_path / "rccl_timeout.pkl")
    mp.spawn(_rccl_timeout_worker, args=(29547, out), nprocs=1, join=True)
    r = pickle.load(open(out, "rb"))
    assert r["exchange"] == "rccl" and r["structure"] == "whole-iteration-split", r
    assert r["raised"] == [True, True, True] and r["intact"] == [True, True, True], r
    assert len(r["fallback"]) == 3
    for fr, fs, applied, shared, exch, ref_structure in r["fallback"]:
        assert fr < 1e-10 and fs < 1e-10 and applied == 20 and not shared and exch == "rccl" and ref_structure == "whole-iteration-split", r


# ---- few tall elements (BASELINE config 5: AdvDiff, 8 elements x 80x80 points): kernels_tall.hip ----
@pytest.mark.parametrize("vf,nhid", [(0, 3), (1, 3), (0, 2), (1, 2)])
def test_tall_element_kernel_against_the_oracle_and_the_separate_launches(vf, nhid):
    """80x80-point elements split over 32 workgroups each (partial residual sums exchanged in device memory, one barrier per
    element): loss triple, gradient incl. d/d epsilon, residuals against the ORACLE; bit-reproducible; trajectory and epsilon
    against the separate launches (HPV_FUSE=n) and the oracle."""
    from cases import p3_args
    from hp_vpinns_amd.vpinn import VPINNAdvDiff
    from oracle.vpinn_oracle import OracleVPINNAdvDiff
    L = [2] + [20] * nhid + [1]
    a = p3_args(gold("advdiff_cfg5"), layers=L)
    th = theta0(L, 23, extra=[0.6])
    o = OracleVPINNAdvDiff(*a, var_form=vf, init_params=th)
    o.vectorized = True
    m = VPINNAdvDiff(*a, var_form=vf, init_params=th)
    l3o, go = o.loss_and_grad()
    l3m, gm = m.loss_and_grad()
    assert m.h.pass_structure() == "whole-iteration-tall", m.h.pass_structure()
    assert rel(l3m, l3o) < TOL and rel(gm, go) < TOL, (l3m, l3o, rel(gm, go))
    assert abs(gm[-1] - go[-1]) < TOL * max(abs(go[-1]), 1e-12)              # d loss / d epsilon
    assert rel(m.h.residuals(8 * 25), o.last["R"].reshape(-1)) < TOL
    l3b, gb = m.loss_and_grad()
    assert np.array_equal(gb, gm) and np.array_equal(l3b, l3m)
    os.environ["HPV_FUSE"] = "n"
    try:
        m2 = VPINNAdvDiff(*a, var_form=vf, init_params=th)
        l3s, gs = m2.loss_and_grad()
        assert m2.h.pass_structure() == "separate"
        m2._step(30, False)
    finally:
        del os.environ["HPV_FUSE"]
    assert rel(gm, gs) < 1e-11 and rel(l3m, l3s) < 1e-12
    m._step(30, False)
    assert rel(m.get_params(), m2.get_params()) < 1e-9
    lo, lm = [], []
    o2 = OracleVPINNAdvDiff(*a, var_form=vf, init_params=th)
    o2.vectorized = True
    m3 = VPINNAdvDiff(*a, var_form=vf, init_params=th)
    for _ in range(8):
        o2.adam_step()
        lo.append(float(o2.loss_parts()[0]))
        lm.append(float(m3._step(1, True)[0]))
    assert rel(lm, lo) < TRAJ_TOL and rel(m3.get_params(), o2.get_params()) < TRAJ_TOL


def test_tall_element_kernel_on_shards_and_its_barrier_timeout(monkeypatch):
    """Shards of config 5 (4 / 1 of its 8 elements: what a GPU of a 2 / 8-GPU run owns) run the same kernel with 64 workgroups
    per element; the shard gradients add up to the full-grid model's.  A partner that stays away (HPV_DEBUG_SPLIT_SKIP=1) must
    leave the replica bit-identical and (HPV_EXCHANGE_FALLBACK=0) raise."""
    monkeypatch.setenv("HPV_EXCHANGE_FALLBACK", "0")
    from cases import p3_args
    from hp_vpinns_amd import _lib
    from hp_vpinns_amd.dist import shard_range
    from hp_vpinns_amd.vpinn import VPINNAdvDiff
    L = [2, 20, 20, 20, 1]
    a = p3_args(gold("advdiff_cfg5"), layers=L)
    th = theta0(L, 29, extra=[0.9])
    full = VPINNAdvDiff(*a, init_params=th)
    l3f, gf = full.loss_and_grad()
    nodata = VPINNAdvDiff(*a, init_params=th)
    nodata.h.set_data(None, None)
    l3n, gn = nodata.loss_and_grad()                     # variational term only
    gb_ = gf - gn                                        # gradient of the boundary / data term
    for nshard in (2, 8):
        g_sum, lv = np.zeros_like(gf), 0.0
        for r in range(nshard):
            m = VPINNAdvDiff(*a, init_params=th)
            eb, ee = shard_range(8, r, nshard)
            m.h.set_elements(a[7], a[8], eb, ee)
            l3, g = m.loss_and_grad()
            assert m.h.pass_structure() == "whole-iteration-tall"
            g_sum += g
            lv += l3[2]
            del m
        assert rel(g_sum - (nshard - 1) * gb_, gf) < 1e-11 and abs(lv - l3f[2]) < 1e-12 * abs(l3f[2])
    m = _with_knob(1, lambda: VPINNAdvDiff(*a, init_params=th))
    state0 = m.h.get_state()
    for call in (lambda: m._step(9, False), lambda: m._step_record(3), lambda: m.loss_and_grad()):
        with pytest.raises(_lib.HpvError, match="did not meet at their barrier"):
            call()
        assert np.array_equal(m.h.get_state(), state0)


def _with_knob(k, build):
    """The model built on libhpvpinn_testhooks.so (the -DHPV_TEST_HOOKS build: the product library neither reads the knob nor
    carries the branch) with partner 1 of element 0 staying away from the in-kernel exchange from the k-th launch on."""
    from hp_vpinns_amd import _lib
    os.environ["HPV_DEBUG_SPLIT_SKIP"] = str(k)      # read when the handle's kernels are set up
    try:
        with _lib.library(_lib.TEST_HOOKS_LIB_PATH):
            m = build()
        assert m.h.build_info()["test_hooks"] == "1"
        return m
    finally:
        del os.environ["HPV_DEBUG_SPLIT_SKIP"]


def _fallback_note(capfd):
    """(iterations that took place before the timeout, iterations requested) from the library's one-line notice on stderr."""
    import re
    m = re.search(r"timed out after (\d+) of (\d+) iterations", capfd.readouterr().err)
    assert m, "no fallback notice"
    return int(m.group(1)), int(m.group(2))


def test_exchange_timeout_mid_run_is_finished_on_the_barrier_free_kernels(capfd):
    """hpv_step: the partner stays away from the 14th launch on -- the iterations before it were applied, the failing one and
    the rest of the run are redone without an in-kernel exchange; end state = an undisturbed run's."""
    ref = _build_small_shard()
    ref._step(30, False)
    m = _with_knob(14, _build_small_shard)
    capfd.readouterr()
    l3 = m._step(30, True)
    done, asked = _fallback_note(capfd)
    assert asked == 30 and 0 < done < 30
    assert m.h.updates_applied() == 30 and not m.h.shared_element_kernels()
    assert m.h.pass_structure() in ("whole-iteration", "fused-reverse")      # (small shard: the two-kernel path is the faster one)
    assert rel(m.get_params(), ref.get_params()) < 1e-10
    assert rel(l3, ref.loss()) < 1e-9
    # the handle stays on those kernels: further runs need no second fallback
    m._step(10, False)
    ref._step(10, False)
    assert "timed out" not in capfd.readouterr().err and m.h.updates_applied() == 40
    assert rel(m.get_params(), ref.get_params()) < 1e-10


def test_exchange_timeout_mid_run_keeps_the_recorded_history(capfd):
    """hpv_step_record: the losses recorded before the timeout are kept, the rest follows from the barrier-free kernels."""
    ref = _build_small_shard(seed=8)
    h_ref, _ = ref._step_record(21)
    m = _with_knob(6, lambda: _build_small_shard(seed=8))
    capfd.readouterr()
    h, _ = m._step_record(21)
    done, asked = _fallback_note(capfd)
    assert asked == 21 and 0 < done < 21
    assert rel(h, h_ref) < 1e-9 and rel(m.get_params(), ref.get_params()) < 1e-10


def test_tall_element_kernel_timeout_mid_run_is_finished_on_separate_launches(capfd):
    from cases import p3_args
    from hp_vpinns_amd.vpinn import VPINNAdvDiff
    L = [2, 20, 20, 20, 1]
    a = p3_args(gold("advdiff_cfg5"), layers=L)
    th = theta0(L, 31, extra=[0.8])
    ref = VPINNAdvDiff(*a, init_params=th)
    h_ref, e_ref = ref._step_record(12)
    assert ref.h.pass_structure() == "whole-iteration-tall"
    m = _with_knob(5, lambda: VPINNAdvDiff(*a, init_params=th))
    capfd.readouterr()
    h, e = m._step_record(12)
    done, asked = _fallback_note(capfd)
    assert asked == 12 and 0 < done < 12
    assert m.h.pass_structure() in ("separate", "fused-reverse") and m.h.updates_applied() == 12
    assert rel(h, h_ref) < 1e-9 and rel(e, e_ref) < 1e-10 and rel(m.get_params(), ref.get_params()) < 1e-10
