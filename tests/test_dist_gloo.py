"""The N>1 path on CPU: world_size-2 `gloo` processes run the element sharding + packed all-reduce
plumbing of hp_vpinns_amd.dist (the kernels' partial sums are stood in for by oracle partials, which
is allowed for tests) and must reproduce the unsharded loss / gradient."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cases import gold, p2_args, theta0
    from hp_vpinns_amd.dist import Reducer, dist_info, shard_range
    from oracle.vpinn_oracle import OracleVPINN2D
    assert dist_info()[:2] == (rank, world)
    a = p2_args(gold("poisson2d_small"))
    th = theta0(a[13], 21)
    o = OracleVPINN2D(*a, init_params=th)
    ne = o.Nelementx * o.Nelementy
    o.e_range = shard_range(ne, rank, world)          # this rank's contiguous element block
    o.use_data = rank == 0                            # boundary term lives on rank 0 only
    (loss, lossb, lossv), g = o.loss_and_grad()
    # packed buffer exactly as the library lays it out: [grad (P) | lossv | w*lossb | msq | pad]
    buf = torch.tensor(np.concatenate([g, [lossv, loss - lossv, lossb, 0.0]]))
    red = Reducer(tensor=buf)
    assert red.active
    out = red.allreduce().numpy().copy()
    q.put((rank, o.e_range, out))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_everything():
    from hp_vpinns_amd.dist import shard_range
    for ne in (1, 2, 5, 6, 16, 256, 257):
        for w in (1, 2, 3, 4, 8):
            r = [shard_range(ne, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == ne
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_world2_gloo_allreduce_matches_unsharded():
    sys.path.insert(0, HERE)
    from cases import gold, p2_args, rel, theta0
    from oracle.vpinn_oracle import OracleVPINN2D
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    a = p2_args(gold("poisson2d_small"))
    o = OracleVPINN2D(*a, init_params=theta0(a[13], 21))
    (loss, lossb, lossv), g = o.loss_and_grad()
    ranges = sorted(r[1] for r in res)
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == 6
    for _, _, out in res:                         # every rank holds the same, complete sums
        P = g.size
        assert rel(out[:P], g) < 1e-12
        assert abs(out[P] - lossv) < 1e-12 * abs(lossv)
        assert abs(out[P] + out[P + 1] - loss) < 1e-12 * abs(loss)
        assert abs(out[P + 2] - lossb) < 1e-14
    assert np.array_equal(res[0][2], res[1][2])   # bitwise identical on both ranks -> replicas stay in sync


def _p2p_logic_worker(rank, world, port, q):
    """The agreement / fallback logic of VPINN._connect_p2p with the device replaced by a stub (3 scenarios)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hp_vpinns_amd import _lib
    from hp_vpinns_amd.vpinn import _VPINNBase

    class Stub(_VPINNBase):
        def __init__(self, export_fails=False, wrong_answer=False, timeout=False):
            outer = self
            self.rank, self.world, self.log = rank, world, []

            class H:
                def p2p_export(s, w, r):
                    if export_fails:
                        raise _lib.HpvError("no ipc")
                    return bytes([r]) * 128

                def p2p_connect(s, handles):
                    outer.log.append(("connect", len(handles)))

                def reduce_buffer(s):
                    return 0, 10

                def p2p_selftest(s, n):
                    v = world * (world + 1) / 2 + world * 1e-3 * np.arange(n)
                    return (v + (1.0 if wrong_answer else 0.0)), int(timeout)

                def p2p_disconnect(s):
                    outer.log.append("disconnect")
            self.h = H()

    out = []
    m = Stub()
    out.append((m._connect_p2p(), m.log))                                   # everything works
    m = Stub(export_fails=(rank == 1))
    out.append((m._connect_p2p(), m.log))                                   # one rank cannot export
    m = Stub(wrong_answer=(rank == 0))
    out.append((m._connect_p2p(), m.log))                                   # one rank reads a wrong sum
    m = Stub(timeout=(rank == 1))
    out.append((m._connect_p2p(), m.log))                                   # one rank's peer never arrives
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_in_library_exchange_setup_agrees_on_fallback():
    """Every rank must reach the same decision -- use the in-library exchange or fall back to the collective path --
    whichever rank the failure happens on, and a rank that exported a mailbox must release it on fallback."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_p2p_logic_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank in (0, 1):
        ok, log = res[rank][0]
        assert ok is True and log == [("connect", 256)]
        ok, log = res[rank][1]
        assert ok is False and ("disconnect" in log) == (rank == 0) and ("connect", 256) not in log
        for scenario in (2, 3):
            ok, log = res[rank][scenario]
            assert ok is False and log == [("connect", 256), "disconnect"]


def _rccl_logic_worker(rank, world, port, q):
    """The agreement / fallback logic of VPINN._connect_rccl with the device replaced by a stub."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hp_vpinns_amd import _lib
    from hp_vpinns_amd.vpinn import _VPINNBase

    class Stub(_VPINNBase):
        def __init__(self, no_library=False, init_fails=False, wrong_answer=False, not_loadable=False, init_hangs=False,
                     selftest_hangs=False, slow_s=4.0):
            outer = self
            self.rank, self.world, self.log = rank, world, []

            class H:
                def rccl_available(s):
                    return not not_loadable

                def rccl_unique_id(s):
                    if no_library:
                        raise _lib.HpvError("librccl.so could not be loaded")
                    return b"\x07" * 128

                abandoned = leaked = False

                def rccl_connect(s, w, r, uid):
                    if init_fails:
                        raise _lib.HpvError("ncclCommInitRank failed")
                    if init_hangs:
                        import time
                        time.sleep(slow_s)   # longer than HPV_RCCL_TIMEOUT_S below: the caller must give up on it
                        outer.late.append(s.abandoned and s.leaked and outer.h is not s)   # the late return finds itself disowned
                        return
                    outer.log.append(("connect", w, r, len(uid)))

                def reduce_buffer(s):
                    return 0, 10

                def rccl_selftest(s, n):
                    if selftest_hangs:
                        import time
                        time.sleep(slow_s)
                        outer.late.append(s.abandoned and s.leaked and outer.h is not s)
                    return world * (world + 1) / 2 + world * 1e-3 * np.arange(n) + (1.0 if wrong_answer else 0.0)

                def rccl_disconnect(s):
                    outer.log.append("disconnect")

                def rccl_abandon(s):
                    s.abandoned = True

                def leak(s):
                    s.leaked = True
            self.late = []
            self._H = H
            self.h = H()

        def _new_handle(self):               # (the real one builds a fresh _lib.Handle and hands it the problem again)
            self.log.append("fresh-handle")
            return self._H()

    out = []
    os.environ["HPV_RCCL_TIMEOUT_S"] = "1.5"
    slow = world - 1                         # the last rank is the slow / failing one
    keep = []
    for kw in ({}, {"no_library": True}, {"init_fails": rank == slow}, {"wrong_answer": rank == 0}, {"not_loadable": rank == slow},
               {"init_hangs": rank == slow}, {"selftest_hangs": rank == slow}):
        m = Stub(**kw)
        out.append((m._connect_rccl(), list(m.log)))
        keep.append(m)
    import time
    time.sleep(4.5)                          # let the abandoned helper threads come back: they must find themselves disowned
    out.append([list(m.late) for m in keep])
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_in_library_rccl_setup_agrees_on_fallback():
    """The multi-GPU default: rank 0's ncclUniqueId reaches every rank, every rank joins and checks a known answer; a
    failure on ANY rank (library missing or not loadable on one rank, communicator refused or hanging, wrong sum) makes EVERY
    rank fall back, and ranks that had joined leave the communicator."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_logic_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank in (0, 1):
        ok, log = res[rank][0]
        assert ok is True and log == [("connect", 2, rank, 128)]
        ok, log = res[rank][1]
        assert ok is False and log == []                                   # no id: nobody even tries to join
        ok, log = res[rank][2]
        assert ok is False and log == ([("connect", 2, 0, 128), "disconnect"] if rank == 0 else [])
        ok, log = res[rank][3]
        assert ok is False and log == [("connect", 2, rank, 128), "disconnect"]
        ok, log = res[rank][4]
        assert ok is False and log == []        # one rank cannot load the library: agreed BEFORE anyone enters ncclCommInitRank
        ok, log = res[rank][5]                  # one rank's ncclCommInitRank does not return: wall-clock bound, everyone falls back
        assert ok is False and log == ([("connect", 2, 0, 128), "disconnect"] if rank == 0 else ["fresh-handle"])
        ok, log = res[rank][6]                  # one rank's self-test all-reduce never completes: that rank trains on a FRESH handle
        assert ok is False and log == ([("connect", 2, 0, 128), "disconnect"] if rank == 0 else [("connect", 2, 1, 128), "fresh-handle"])
        late = res[rank][7]
        assert late[5] == ([True] if rank == 1 else []) and late[6] == ([True] if rank == 1 else [])


@pytest.mark.timeout(600)
def test_world8_rccl_setup_agrees_with_one_slow_rank():
    """The world size the SCALE run uses: eight ranks agree on every step of the communicator set-up, the element shards
    partition the grid, and ONE slow rank (its ncclCommInitRank / self-test outlasts the wall-clock bound) sends all eight
    to the fallback -- the seven that had joined leave the communicator, the slow one moves to a fresh handle."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_logic_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=500) for _ in range(8))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from hp_vpinns_amd.dist import shard_range
    r = [shard_range(256, k, 8) for k in range(8)]
    assert r == [(32 * k, 32 * k + 32) for k in range(8)]
    for rank in range(8):
        ok, log = res[rank][0]
        assert ok is True and log == [("connect", 8, rank, 128)]
        assert res[rank][1] == (False, []) and res[rank][4] == (False, [])
        ok, log = res[rank][2]
        assert ok is False and log == ([("connect", 8, rank, 128), "disconnect"] if rank != 7 else [])
        ok, log = res[rank][5]
        assert ok is False and log == ([("connect", 8, rank, 128), "disconnect"] if rank != 7 else ["fresh-handle"])
        ok, log = res[rank][6]
        assert ok is False and log == ([("connect", 8, rank, 128), "disconnect"] if rank != 7 else [("connect", 8, 7, 128), "fresh-handle"])


class _ShardRecordingHandle:
    """Stand-in for hp_vpinns_amd._lib.Handle that keeps the ARGUMENTS of every set_* call (no GPU here); the in-library RCCL
    set-up answers as a healthy communicator would, so the class ends up on its multi-GPU default."""
    instances = []

    def __init__(self, pde, var_form, act, layers, **kw):
        self.kw, self.calls, self.theta = kw, [], None
        _ShardRecordingHandle.instances.append(self)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)

        def rec(*a, **k):
            self.calls.append((name, a, k))
        return rec

    def set_params(self, theta):
        self.theta = np.array(theta, dtype=np.float64)

    def backend_in_use(self):
        return 2

    def rccl_available(self):
        return True

    def rccl_unique_id(self):
        return b"\x05" * 128

    def reduce_buffer(self):
        return 0, 12

    def rccl_selftest(self, n):
        w = int(os.environ["WORLD_SIZE"])
        return w * (w + 1) / 2 + w * 1e-3 * np.arange(n)


def _torchrun_env_worker(rank, world, port, q, mode):
    """What `torchrun driver.py` gives a reference driver with ONLY the import swap: the launcher's environment, no
    init_process_group anywhere in the script (P2:430-434 unchanged)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    if mode != "no-master":
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    else:
        os.environ.pop("MASTER_ADDR", None)
        os.environ.pop("MASTER_PORT", None)
    if mode == "replicas":
        os.environ["HPV_NO_AUTO_DIST"] = "1"
    torch.set_num_threads(1)
    torch.cuda.set_device = lambda d: None            # (no GPU in this container; the class binds LOCAL_RANK's device)
    from cases import gold, p2_args, theta0
    from hp_vpinns_amd import _lib
    from hp_vpinns_amd import vpinn
    _lib.Handle = _ShardRecordingHandle
    assert not dist.is_initialized()
    a = p2_args(gold("poisson2d_small"))
    try:
        m = vpinn.VPINN2D(*a, init_params=theta0(a[13], 3), var_form=1)
    except RuntimeError as e:
        q.put((rank, "error", str(e)))
        return
    h = _ShardRecordingHandle.instances[-1]
    el = [c for c in h.calls if c[0] == "set_elements"][0][1]
    data = [c for c in h.calls if c[0] == "set_data"]
    q.put((rank, "ok", dict(e_begin=int(el[2]), e_end=int(el[3]), n_data_calls=len(data), world=m.world, mrank=m.rank,
                            device=h.kw["device"], exchange=m.exchange(), group=dist.is_initialized(),
                            group_world=dist.get_world_size() if dist.is_initialized() else 0)))
    if dist.is_initialized():
        dist.barrier()


def _run_torchrun_env(mode, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_torchrun_env_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, status, v = q.get(timeout=240)
        res[r] = (status, v)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


@pytest.mark.timeout(300)
def test_unchanged_driver_under_torchrun_shards_its_elements():
    """Verdict round 5, item 1a: WORLD_SIZE=2 in the environment, no process group -> dist_info() creates one; the two
    processes own DISJOINT contiguous element blocks that partition the grid, bind devices 0 and 1, the boundary term lives on
    rank 0 only, and both end on the in-library RCCL exchange.  (Until round 5: two full replicas on device 0, silently.)"""
    res = _run_torchrun_env("torchrun")
    assert all(s == "ok" for s, _ in res.values()), res
    r0, r1 = res[0][1], res[1][1]
    assert (r0["e_begin"], r0["e_end"], r1["e_begin"], r1["e_end"]) == (0, 3, 3, 6)      # 3 x 2 elements of the small fixture
    assert (r0["world"], r1["world"], r0["mrank"], r1["mrank"]) == (2, 2, 0, 1)
    assert (r0["device"], r1["device"]) == (0, 1)
    assert (r0["n_data_calls"], r1["n_data_calls"]) == (1, 0)
    assert r0["exchange"] == r1["exchange"] == "rccl" and r0["group"] and r0["group_world"] == 2


@pytest.mark.timeout(300)
def test_torchrun_env_without_rendezvous_raises_instead_of_replicating():
    res = _run_torchrun_env("no-master")
    for s, v in res.values():
        assert s == "error" and "MASTER_ADDR" in v and "init_process_group" in v and "HPV_NO_AUTO_DIST" in v, v


@pytest.mark.timeout(300)
def test_independent_replicas_only_when_asked_for():
    res = _run_torchrun_env("replicas")
    for s, v in res.values():
        assert s == "ok" and (v["e_begin"], v["e_end"], v["world"], v["group"], v["n_data_calls"]) == (0, 6, 1, False, 1), v
