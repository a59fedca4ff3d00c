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


class _Packed():
    """one persistent flat buffer + the views of its slots (one slot per statistic of a given exchange)"""

    def __init__(self, shapes, dtype, device):
        sizes = [int(torch.Size(sh).numel()) for sh in shapes]
        self.flat = torch.zeros(sum(sizes), dtype=dtype, device=device)
        self.views, off = [], 0
        for sh, n in zip(shapes, sizes):
            self.views.append(self.flat[off:off + n].view(sh))
            off += n


class SuffStatReducer():
    """Sums a list of statistic tensors over the process group with a SINGLE all-reduce of one packed buffer.

    The buffer is persistent: it is laid out once per exchange signature (shapes, dtype, device of the statistics -- an
    exchange of a VB iteration has the same signature every iteration) and reused, so that an iteration neither
    concatenates nor splits the statistics.  They are written into their slots with one multi-tensor copy, the flat buffer
    is reduced in place, and the results are handed back as views of ONE clone of it (fresh memory: the caller may keep them).
    `slots(...)` hands the slot views out beforehand for producers that can write their result straight into the
    packed buffer; statistics passed back as those very views are returned in place (valid until the next exchange)."""

    def __init__(self, group=None):
        self.group = group
        self.calls = 0  # number of collectives issued (tests assert: one per exchange)
        self._packed = {}

    @property
    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def _buffer(self, shapes, dtype, device):
        key = (tuple(tuple(sh) for sh in shapes), dtype, str(device))
        pk = self._packed.get(key)
        if pk is None:
            pk = self._packed[key] = _Packed(key[0], dtype, device)
        return pk

    def slots(self, shapes, dtype, device):
        """views into the persistent packed buffer of this signature, in order; fill them, then `all_reduce(slots)`"""
        return list(self._buffer(shapes, dtype, device).views)

    def all_reduce(self, tensors):
        """sum of every tensor over the group, ONE collective.  Returns fresh tensors (views of one clone of the packed buffer),
        so a result kept across iterations -- or passed to a second model that shares this reducer -- is never overwritten by the
        next exchange of the same signature.  Zero-copy only for inputs that ARE this signature's slot views (`slots()`): those
        come back as the slots themselves, valid until the next exchange."""
        tensors = [t if isinstance(t, torch.Tensor) else torch.as_tensor(t) for t in tensors]
        if not dist.is_initialized() or self.world_size == 1:
            return list(tensors)
        dt = tensors[0].dtype
        for t in tensors:
            dt = torch.promote_types(dt, t.dtype)
        pk = self._buffer([t.shape for t in tensors], dt, tensors[0].device)
        own = [t.dtype == dt and t.data_ptr() == v.data_ptr() and t.shape == v.shape and t.stride() == v.stride()
               for v, t in zip(pk.views, tensors)]
        # an input that lives inside the packed buffer at ANOTHER position (a slot view passed back in a different place) would
        # be clobbered by the copies of the slots before it: take it out first
        store = pk.flat.untyped_storage().data_ptr()
        tensors = [t.clone() if (not o and t.numel() > 0 and t.untyped_storage().data_ptr() == store) else t
                   for o, t in zip(own, tensors)]
        todo = [(v, t) for o, v, t in zip(own, pk.views, tensors) if not o]
        if todo:  # statistics that were not produced in place: one multi-tensor copy into the slots
            torch._foreach_copy_([v for v, _ in todo], [t.expand(v.shape) for v, t in todo])
        dist.all_reduce(pk.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.calls += 1
        if all(own):
            return list(pk.views)
        fresh, out, off = pk.flat.clone(), [], 0
        for o, v, t in zip(own, pk.views, tensors):
            r = v if o else fresh[off:off + v.numel()].view(v.shape)
            off += v.numel()
            out.append(r if t.dtype == dt else r.to(t.dtype))
        return out
