"""Multi-GPU decomposition of the hot path.

Walks: the CSR graph is replicated per GPU and start vertices are split by
contiguous range; walker RNG keys depend only on (seed, start vertex, ordinal),
so the union of the shards' outputs equals the single-GPU output and NO collective
is on the walk path.  SGNS: each rank trains on the walks it generated against a
full replica of the model; sgns.DeltaAllReduce is the one exchange step.
"""
from typing import Tuple


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of rank's contiguous share of n items; shares differ by at most 1."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank {rank} / world {world}")
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def sentence_base(rank: int, world: int, rows_per_rank_max: int, epoch: int = 0) -> int:
    """Disjoint SGNS sentence-id ranges per (epoch, rank): RNG keys never collide."""
    return (epoch * world + rank) * rows_per_rank_max


def all_reduce(t, op=None, group=None, dist=None):
    """torch.distributed.all_reduce of `t` in place.  RCCL (backend "nccl") reduces device
    tensors directly over xGMI; under gloo -- the CPU tests, and the 2-ranks-on-one-GPU rehearsal
    of tests/test_multirank_gpu.py (RCCL refuses two ranks on one device) -- a device tensor is
    staged through host memory, so every multi-rank code path runs unchanged on both."""
    if dist is None:
        import torch.distributed as dist
    op = dist.ReduceOp.SUM if op is None else op
    backend = dist.get_backend(group) if hasattr(dist, "get_backend") else None  # tests stand in
    if backend is None:
        dist.all_reduce(t, op=op, group=group)
        return t
    if t.is_cuda and backend == "gloo":
        if t.dtype.is_floating_point and t.element_size() < 4:
            host = t.float().cpu()  # gloo has no bf16 / fp16 sum
        else:
            host = t.cpu()
        dist.all_reduce(host, op=op, group=group)
        t.copy_(host.to(t.dtype))
        return t
    if not t.is_cuda and t.dtype.is_floating_point and t.element_size() < 4:
        host = t.float()
        dist.all_reduce(host, op=op, group=group)
        t.copy_(host.to(t.dtype))
        return t
    dist.all_reduce(t, op=op, group=group)
    return t
