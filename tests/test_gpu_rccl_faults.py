"""Fault injection for the multi-GPU default exchange (the library's own ncclAllReduce inside its iteration graphs), on ONE
rank with a 1-rank nccl group -- the path the 8-GPU SCALE run takes, with the failures a first real multi-GPU run may meet:

  * a collective that REFUSES STREAM CAPTURE (HPV_TEST_RCCL_FAIL=capture): hpv_step must drop to eager launches
    (hpv_api.hip, enqueue_iterations) and really train -- same parameters as an undisturbed run -- and the communicator must
    still be usable afterwards;
  * a collective that FAILS EAGERLY in the middle of a run (HPV_TEST_RCCL_FAIL=eager:k): -6 from hpv_step, the failing
    iteration NOT applied (parameters, Adam moments, beta powers = the state after the last good iteration), the count of
    applied updates right.

The failures are injected by libhpvpinn_testhooks.so (-DHPV_TEST_HOOKS; the product library has no such switch).  Each scenario
runs in its own process (a process group per process)."""
import os
import pickle
import socket

import numpy as np
import pytest

from cases import rel

pytestmark = pytest.mark.gpu

L4 = [2, 20, 20, 20, 1]


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(hooks):
    from hp_vpinns_amd import _lib
    from hp_vpinns_amd.drivers import poisson2d
    from hp_vpinns_amd.init import xavier_init
    s = poisson2d.setup(N_el_x=4, N_el_y=4, N_test_x=10, N_test_y=10, N_quad=20, N_bound=13, with_test_grid=False)
    if hooks:
        with _lib.library(_lib.TEST_HOOKS_LIB_PATH):
            return poisson2d.build_model(s, L4, init_params=xavier_init(L4, 6))
    return poisson2d.build_model(s, L4, init_params=xavier_init(L4, 6))


def _worker(rank, port, out_path, mode):
    import torch
    import torch.distributed as dist
    from hp_vpinns_amd import _lib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HPV_FORCE_DIST="1", HPV_FUSE="s")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    res = {}
    try:
        ref = _build(False)
        res["ref_exchange"] = ref.exchange()
        if mode == "capture":
            os.environ["HPV_TEST_RCCL_FAIL"] = "capture"
            m = _build(True)
            res["exchange"], res["hooks"] = m.exchange(), m.h.build_info()["test_hooks"]
            l3 = m._step(21, True)                      # 2 graph replays of 8 + a remainder of 5 -- were capture possible
            res["graphs"] = m.h.graphs_in_use()
            res["applied"] = m.h.updates_applied()
            l3r = ref._step(21, True)
            res["graphs_ref"] = ref.h.graphs_in_use()
            res["par"] = float(rel(m.get_params(), ref.get_params()))
            res["loss"] = float(rel(l3, l3r))
            # the communicator survived the refused captures: more iterations, a known-answer all-reduce
            m._step(8, False)
            ref._step(8, False)
            res["par2"] = float(rel(m.get_params(), ref.get_params()))
            n = m.h.reduce_buffer()[1]
            res["selftest"] = float(np.abs(m.h.rccl_selftest(n) - (1.0 + 1e-3 * np.arange(n))).max())
        else:
            os.environ["HPV_NO_GRAPH"] = "1"
            os.environ["HPV_TEST_RCCL_FAIL"] = "eager:5"    # calls 1, 2: the connection self-test; 3, 4: iterations 1, 2; 5: iteration 3
            m = _build(True)
            res["exchange"] = m.exchange()
            try:
                m._step(10, False)
                res["raised"] = None
            except _lib.HpvError as e:
                res["raised"] = (e.code, "ncclAllReduce failed" in str(e))
            res["applied"] = m.h.updates_applied()
            ref._step(2, False)
            res["intact"] = bool(np.array_equal(m.h.get_state(), ref.h.get_state()))
            # the library keeps working on the same handle: the collective is healthy again (only the 5th call failed)
            m._step(3, False)
            ref._step(3, False)
            res["par_after"] = float(rel(m.get_params(), ref.get_params()))
            res["applied_after"] = m.h.updates_applied()
    finally:
        with open(out_path, "wb") as f:
            pickle.dump(res, f)
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["capture", "eager"])
def test_injected_collective_failures(tmp_path, mode):
    import torch.multiprocessing as mp
    out = str(tmp_path / f"rccl_{mode}.pkl")
    mp.spawn(_worker, args=(_port(), out, mode), nprocs=1, join=True)
    r = pickle.load(open(out, "rb"))
    assert r["ref_exchange"] == "rccl" and r["exchange"] == "rccl", r
    if mode == "capture":
        assert r["hooks"] == "1" and r["graphs"] is False and r["graphs_ref"] is True, r     # eager fallback taken; the undisturbed run replays graphs
        assert r["applied"] == 21 and r["par"] < 1e-12 and r["loss"] < 1e-12 and r["par2"] < 1e-12, r
        assert r["selftest"] < 1e-12, r
    else:
        assert r["raised"] == (-6, True), r
        assert r["applied"] == 2 and r["intact"] is True, r
        assert r["par_after"] < 1e-12 and r["applied_after"] == 5, r
