"""The exchange step of the multi-GPU SGNS path on the device (sgns.DeltaSync over
n2v_delta_ref_init / n2v_delta_pack / n2v_delta_apply, csrc/n2v_sync.hip).  One GPU cannot hold two
RCCL ranks, so the collective itself is stood in for (the sum over two replicas is formed with
torch on the wire buffers); the HIP passes before and after it are compared with the same protocol
run on CPU tensors, which tests/test_dist_gloo.py runs through a real world-2 all-reduce."""
import pytest
import torch

pytestmark = pytest.mark.gpu


class _TwoRanks:
    """stands in for torch.distributed inside shard.ordered_sum as rank 0 of two: `other` holds
    what rank 1 would put on the wire at each exchange, in call order.  The two collectives only
    move bytes: all_to_all_single hands rank 0 the first shard of both ranks' buffers,
    all_gather_into_tensor returns rank 0's summed shard and -- rank 1's work, done here with the
    same rank-ordered fp32 additions -- the sum of the second shards."""

    def __init__(self, other):
        self.other = list(other)
        self.ReduceOp = torch.distributed.ReduceOp

    def get_world_size(self, group=None):
        return 2

    def get_backend(self, group=None):
        return "stand-in"

    def all_to_all_single(self, recv, send, group=None):
        o = self.other.pop(0)
        m = send.numel() // 2  # bytes per shard
        theirs = torch.zeros(send.numel(), dtype=torch.uint8, device=send.device)
        ob = o.contiguous().view(torch.uint8).to(send.device)
        theirs[: ob.numel()] = ob
        recv[:m].copy_(send[:m])
        recv[m:].copy_(theirs[:m])
        self._second = (send[m:].clone(), theirs[m:].clone(), o.dtype)

    def all_gather_into_tensor(self, full, shard, group=None):
        mine, theirs, dtype = self._second
        total = (mine.view(dtype).float() + theirs.view(dtype).float()).to(dtype)
        m = shard.numel()
        full[:m].copy_(shard)
        full[m:].copy_(total.view(torch.uint8))


def _replicas(device, shapes, seed):
    g = torch.Generator().manual_seed(seed)
    base = [torch.randn(s, generator=g) for s in shapes]
    d0 = [0.01 * torch.randn(s, generator=g) for s in shapes]
    d1 = [0.01 * torch.randn(s, generator=g) for s in shapes]
    r0 = [(b + d).to(device).contiguous() for b, d in zip(base, d0)]
    r1 = [(b + d).to(device).contiguous() for b, d in zip(base, d1)]
    return [b.to(device).contiguous() for b in base], r0, r1


def _run(device, wire, shapes, block_rows):
    """two replicas trained apart from a common state, one blocking exchange; returns rank 0's
    tensors (and references) afterwards"""
    from node2vec_amd.sgns import DeltaSync

    base, r0, r1 = _replicas(device, shapes, 7)
    orig0, orig1 = [t.cpu().clone() for t in r0], [t.cpu().clone() for t in r1]
    syncs = []
    for rep in (r0, r1):
        s = DeltaSync(rep, wire=wire, block_rows=block_rows, overlap=False)
        s.active, s.world = True, 2
        if wire == "bf16":
            s.refs = [s._ref_init(b) for b in base]  # the synchronised state both ranks share
        syncs.append(s)
    # what rank 1 puts on the wire, block by block, in the order _exchange walks them
    contrib = []
    s1 = syncs[1]
    for k, t in enumerate(s1.tensors):
        ref = None if s1.refs is None else s1.refs[k]
        before, w = s1._buffers(t)
        for lo in range(0, t.shape[0], block_rows):
            hi = min(t.shape[0], lo + block_rows)
            n = t[lo:hi].numel()
            s1._pack(t[lo:hi], None if ref is None else ref[lo:hi], None, w[:n])
            contrib.append(w[:n].clone())
    s0 = syncs[0]
    s0.dist = _TwoRanks(contrib)
    s0._exchange(exact=True)
    if device != "cpu":
        torch.cuda.synchronize()
    return ([t.cpu() for t in s0.tensors], None if s0.refs is None else [r.cpu() for r in s0.refs],
            orig0, orig1)


@pytest.mark.parametrize("wire", ["fp32", "bf16"])
@pytest.mark.parametrize("shapes", [[(1000, 128), (1000, 128)], [(37, 7), (5, 3)]])
def test_device_passes_equal_the_cpu_protocol(wire, shapes):
    got, got_ref, _, _ = _run("cuda", wire, shapes, block_rows=256)
    want, want_ref, r0, r1 = _run("cpu", wire, shapes, block_rows=256)
    for g, w in zip(got, want):
        assert torch.equal(g, w)  # same fp32 operations, same bf16 roundings (nearest even)
    if wire == "bf16":
        for g, w in zip(got_ref, want_ref):
            assert torch.equal(g.view(torch.int16), w.view(torch.int16))
    # and it IS the mean of the two replicas (bf16: up to the rounding of the deltas and the reference)
    tol = 1e-6 if wire == "fp32" else 1e-3
    for g, a, b in zip(got, r0, r1):
        assert torch.allclose(g, (a + b) / 2, atol=tol)


def test_overlapped_exchange_keeps_what_was_trained_meanwhile():
    """non-blocking form: rows get mean - snapshot ADDED, so an update made after the snapshot
    (here: by hand, before apply is reached on the side stream) survives"""
    from node2vec_amd.sgns import DeltaSync

    t = torch.randn(512, 64, device="cuda")
    before_all = t.clone()
    s = DeltaSync([t], wire="fp32", block_rows=128, overlap=False)
    s.active, s.world = True, 2

    class _Same(_TwoRanks):  # the other rank holds the same rows: the mean is the tensor itself
        def all_to_all_single(self, recv, send, group=None):
            self.other = [send.clone().view(torch.float32)]
            super().all_to_all_single(recv, send, group)

    s.dist = _Same([])
    s._exchange(exact=False)
    torch.cuda.synchronize()
    assert torch.allclose(t, before_all, atol=1e-6)


def test_delta_reduce_is_the_rank_ordered_fp32_sum():
    """n2v_delta_reduce (the local half of shard.ordered_sum) against the same additions spelled
    out with torch on the host: fp32 accumulation in rank order, one rounding to the wire type --
    so RCCL (device tensors), gloo (host tensors) and gloo with staged device tensors agree"""
    from node2vec_amd.shard import _rank_ordered_reduce

    gen = torch.Generator().manual_seed(9)
    for dtype in (torch.float32, torch.bfloat16):
        for world, m in ((2, 1000), (8, 4099), (5, 1)):
            scale = torch.logspace(-3, 3, world).repeat_interleave(m)
            parts = (torch.randn(world * m, generator=gen) * scale).to(dtype)
            want = _rank_ordered_reduce(parts, world, m, torch.empty(m, dtype=dtype))
            got = _rank_ordered_reduce(parts.cuda(), world, m, torch.empty(m, dtype=dtype, device="cuda"))
            torch.cuda.synchronize()
            assert torch.equal(got.cpu().view(torch.uint8), want.view(torch.uint8)), (dtype, world, m)
