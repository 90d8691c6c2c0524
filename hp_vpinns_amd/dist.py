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


def dist_info():
    """(rank, world, local_rank) from torch.distributed if initialised, else the torchrun env."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size(), int(os.environ.get("LOCAL_RANK", 0))
    except Exception:
        pass
    return 0, 1, 0


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
