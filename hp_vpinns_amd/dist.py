"""Element sharding and the one collective of the path.

`lossv = sum_e loss_e` (P1:96, P2:120, P3:182) is a sum of independent per-element terms that
share only the replicated parameter vector, so elements shard over GPUs in contiguous blocks
(one process per GPU).  Each rank's kernels fill a packed buffer
`[grad (P) | d eps | lossv | w*lossb | msq | pad]` with *partial sums*; one all-reduce(sum) per
iteration (RCCL over xGMI through torch.distributed's "nccl" backend) makes it global and every
rank applies the identical TF1-Adam update to its replica -- no parameter broadcast after step 0.
The boundary/data term lives on rank 0 only.
"""
import os


def shard_range(n_elem, rank, world):
    """Contiguous block [begin, end) of the flattened element index owned by `rank`."""
    base, rem = divmod(int(n_elem), int(world))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


_auto_group = [False]      # this module created the process group (it then also tears it down at exit)


def dist_info():
    """(rank, world, local_rank) of this process.

    * a torch.distributed process group exists: its rank / world size;
    * none exists but the process was started by `torchrun` / `python -m torch.distributed.run` (WORLD_SIZE > 1 in the
      environment): the group is created HERE from the launcher's environment (env:// rendezvous; "cpu:gloo,cuda:nccl" with
      the device set to LOCAL_RANK when a GPU is visible, "gloo" otherwise; HPV_DIST_BACKEND overrides), so that a reference driver with
      nothing but the one-line import swap (P2:430-434 unchanged) shards its elements over the N processes.  Until round 5 this
      case fell through to (0, 1, 0): N processes each trained the WHOLE problem on device 0, silently;
    * WORLD_SIZE > 1 but the launcher's variables are incomplete, or the group cannot be created: RuntimeError naming the line to
      add -- never a silent single-process run.  HPV_NO_AUTO_DIST=1 states that independent replicas are intended.
    * otherwise (a plain `python driver.py`): (0, 1, 0).
    """
    local = int(os.environ.get("LOCAL_RANK", "0") or 0)
    world_env = int(os.environ.get("WORLD_SIZE", "1") or 1)
    try:
        import torch.distributed as dist
        have = dist.is_available()
    except Exception:
        dist, have = None, False
    if have and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size(), local
    if world_env <= 1 or os.environ.get("HPV_NO_AUTO_DIST") == "1":
        return 0, 1, 0
    hint = ("add `import torch.distributed as dist; dist.init_process_group('nccl')` before the VPINN(...) constructor, or set "
            "HPV_NO_AUTO_DIST=1 if %d independent replicas are what you want" % world_env)
    if not have:
        raise RuntimeError("hp_vpinns_amd: WORLD_SIZE=%d but torch.distributed is not available; %s" % (world_env, hint))
    missing = [k for k in ("RANK", "MASTER_ADDR", "MASTER_PORT") if not os.environ.get(k)]
    if missing:
        raise RuntimeError("hp_vpinns_amd: WORLD_SIZE=%d but %s not set (start the driver with torchrun, or %s)"
                           % (world_env, ", ".join(missing), hint))
    import torch
    # (a GPU is visible: the control plane -- object collectives of the communicator set-up, barriers -- on gloo, "cuda:nccl" registered
    #  for the `torch` exchange fallback and created lazily, only if that fallback is ever used: the library's own communicator
    #  (hpv_rccl_*) is then the only RCCL communicator of the rank, as in bench.py)
    backend = os.environ.get("HPV_DIST_BACKEND") or ("cpu:gloo,cuda:nccl" if torch.cuda.is_available() else "gloo")
    try:
        if "nccl" in backend:
            torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    except Exception as e:
        raise RuntimeError("hp_vpinns_amd: WORLD_SIZE=%d and no process group: creating one (%s) failed: %s; %s"
                           % (world_env, backend, e, hint)) from e
    _auto_group[0] = True
    import atexit
    atexit.register(_teardown)
    if dist.get_rank() == 0:
        import sys
        print("hp_vpinns_amd: started under torchrun without a process group -- created one (%s, %d ranks): elements shard "
              "over the ranks, one process per GPU" % (backend, dist.get_world_size()), file=sys.stderr)
    return dist.get_rank(), dist.get_world_size(), local


def _teardown():
    try:
        import torch.distributed as dist
        if _auto_group[0] and dist.is_initialized():
            dist.destroy_process_group()
    except Exception:      # pragma: no cover - interpreter shutdown
        pass
    _auto_group[0] = False


class Reducer:
    """All-reduce of the packed buffer.  `tensor_of(ptr, n, device)` wraps library-owned device
    memory as a torch tensor without a copy (`__cuda_array_interface__`)."""

    class _DevBuf:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}

    def __init__(self, ptr=None, n=0, device=0, tensor=None, force=False):
        import torch
        import torch.distributed as dist
        self.dist = dist
        self.tensor = tensor if tensor is not None else torch.as_tensor(self._DevBuf(ptr, n), device=f"cuda:{device}")
        ready = dist.is_available() and dist.is_initialized()
        self.active = ready and (dist.get_world_size() > 1 or force)

    def allreduce(self):
        if self.active:
            self.dist.all_reduce(self.tensor, op=self.dist.ReduceOp.SUM)
        return self.tensor
