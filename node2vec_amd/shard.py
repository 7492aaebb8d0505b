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


def _rank_ordered_reduce(parts, world: int, m: int, out):
    """out[i] = parts[0, i] + parts[1, i] + ... accumulated in fp32 in rank order, stored in
    out's dtype (bf16: one rounding of the fp32 sum).  HIP kernel on device tensors
    (n2v_delta_reduce), the same IEEE additions spelled out with torch on CPU tensors."""
    import torch

    if parts.is_cuda:
        from node2vec_amd import _lib

        L = _lib.load()
        with torch.cuda.device(parts.device):
            _lib.check(L.n2v_delta_reduce(parts.data_ptr(),
                                          _lib.WIRE_F32 if parts.dtype == torch.float32 else _lib.WIRE_BF16,
                                          world, m, out.data_ptr(), _lib.current_stream_ptr()),
                       "n2v_delta_reduce")
        return out
    p = parts.view(world, m)
    acc = p[0].float().clone()
    for r in range(1, world):
        acc += p[r].float()  # one rounded fp32 addition per rank, in rank order
    out.copy_(acc)  # fp32 -> bf16: round to nearest even, as the kernel does
    return out


def ordered_sum_shard(n: int, world: int) -> int:
    """elements of the shard every rank sums in ordered_sum of an n-element tensor (the tensor is padded to
    world shards): what crosses EVERY link, in each direction, once in the all-to-all and once in the all-gather"""
    return -(-int(n) // max(int(world), 1))


def ordered_sum(t, group=None, dist=None, scratch=None, force=False):
    """Sum of the 1-D tensor `t` (fp32 or bf16) over the ranks, in place, with numerics that do NOT
    depend on the collective library: the ranks exchange BYTES only -- an all-to-all hands rank r
    the r-th shard of every rank's tensor, rank r adds its `world` shards in fp32 in rank order
    (bf16: the fp32 sum is rounded once), an all-gather returns the summed shards to everybody.
    Bytes per rank on the links: 2 (world - 1) / world x the tensor, what a reduce-scatter +
    all-gather all-reduce moves; on the full-mesh xGMI both phases use all 7 links at once.
    RCCL's own bf16 all-reduce would round in bf16 at every hop and in an order that depends on
    its algorithm; gloo has no bf16 sum at all.  Here "nccl" (RCCL), gloo on CPU tensors and gloo
    with device tensors staged through the host give the same bits, and every rank ends with
    identical values (each element is summed by exactly one rank).  `scratch`: an optional dict
    the buffers are kept in between calls; `force`: run the collectives also on ONE rank (the
    rehearsal of `bench.py --gpus 1` under torch.distributed.run: the RCCL calls, a sum of one)."""
    import torch

    if dist is None:
        import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return t
    if t.dtype not in (torch.float32, torch.bfloat16) or t.dim() != 1 or not t.is_contiguous():
        raise TypeError("ordered_sum wants a contiguous 1-D float32 / bfloat16 tensor")
    n = t.numel()
    m = ordered_sum_shard(n, world)
    bits = torch.uint8  # moved as plain bytes (every backend carries them; gloo has no 16-bit type)
    key = (t.device, t.dtype, world * m)
    buf = None if scratch is None else scratch.get(key)
    if buf is None:
        buf = (torch.zeros(world * m, dtype=t.dtype, device=t.device),
               torch.empty(world * m, dtype=t.dtype, device=t.device),
               torch.empty(m, dtype=t.dtype, device=t.device))
        if scratch is not None:
            scratch.clear()  # one size at a time: the blocks of one exchange are equal but the last
            scratch[key] = buf
    send, recv, shard = buf
    send[:n].copy_(t)
    if n < world * m:
        send[n:].zero_()
    staged = t.is_cuda and dist.get_backend(group) == "gloo"  # gloo moves host memory
    if staged:
        h_send, h_recv = send.cpu().view(bits), torch.empty(world * m * t.element_size(), dtype=bits)
        dist.all_to_all_single(h_recv, h_send, group=group)
        recv.copy_(h_recv.view(t.dtype))
    else:
        dist.all_to_all_single(recv.view(bits), send.view(bits), group=group)
    _rank_ordered_reduce(recv, world, m, shard)
    if staged:
        h_full = torch.empty(world * m * t.element_size(), dtype=bits)
        dist.all_gather_into_tensor(h_full, shard.cpu().view(bits), group=group)
        send.copy_(h_full.view(t.dtype))
    else:
        dist.all_gather_into_tensor(send.view(bits), shard.view(bits), group=group)
    t.copy_(send[:n])
    return t
