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
