"""Multi-GPU support for the conjugate-update path: one process per GPU, torch.distributed over RCCL
(backend "nccl" on ROCm) on a node's xGMI links.

Two sharding patterns cover the whole path:

1. Independent units (batched NormalInverseWishart.ss_update, BASELINE config 2): the batch axis is cut
   into contiguous slices, one per rank; the posteriors are independent, so there is NO exchange.
   `shard_bounds` gives the slice.

2. Sample-axis sharding (GMM / LDS / DMBD): every rank holds N/g samples (series), runs the E-step and
   the local moment reductions, then ALL statistics of the VB iteration travel in ONE flat buffer through
   ONE all-reduce(sum) (`SuffStatReducer`); afterwards every rank runs the (cheap, replicated) ss_update.
   The messages are KB-sized (8.8 KB for a K=4, D=16 fp64 GMM), i.e. latency-bound on xGMI, which is why
   they are packed into a single collective rather than bucketed or overlapped.
   Sums arrive in a different order than a single-process .sum(0): compare at 1e-10, not bitwise.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """[lo, hi) of the contiguous slice of an axis of length n owned by `rank` (sizes differ by <= 1)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class SuffStatReducer():
    """Packs a list of statistic tensors into one flat buffer, sums it over the process group with a single
    all-reduce and hands the tensors back in their original shapes."""

    def __init__(self, group=None):
        self.group = group
        self.calls = 0  # number of collectives issued (tests assert: one per VB iteration)

    @property
    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def all_reduce(self, tensors):
        tensors = [t if isinstance(t, torch.Tensor) else torch.as_tensor(t) for t in tensors]
        if not dist.is_initialized() or self.world_size == 1:
            return list(tensors)
        dt = tensors[0].dtype
        for t in tensors:
            dt = torch.promote_types(dt, t.dtype)
        flat = torch.cat([t.reshape(-1).to(dt) for t in tensors])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        self.calls += 1
        out, off = [], 0
        for t in tensors:
            n = t.numel()
            out.append(flat[off:off + n].reshape(t.shape).to(t.dtype))
            off += n
        return out
