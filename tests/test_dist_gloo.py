"""world_size-2 tests of the multi-GPU logic on CPU (gloo): the SGNS delta
all-reduce (the path's one exchange step) and the collective-free walk sharding."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from node2vec_amd.sgns import DeltaAllReduce
        from node2vec_amd.shard import shard_range

        torch.manual_seed(0)
        base0, base1 = torch.randn(37, 8), torch.randn(37, 8)  # same on every rank
        syn0, syn1 = base0.clone(), base1.clone()
        sync = DeltaAllReduce([syn0, syn1], mean=True, block_rows=16)
        g = torch.Generator().manual_seed(100 + rank)
        d0, d1 = torch.randn(37, 8, generator=g), torch.randn(37, 8, generator=g)
        syn0 += d0  # "local training" since the last sync
        syn1 += d1
        sync()
        # every rank must hold base + mean of all ranks' deltas
        all_d0 = [torch.randn(37, 8, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
        want0 = base0 + sum(all_d0) / world
        ok = torch.allclose(syn0, want0, atol=1e-6)
        gathered = [torch.zeros_like(syn1) for _ in range(world)]
        dist.all_gather(gathered, syn1)
        ok = ok and all(torch.equal(gathered[0], x) for x in gathered)
        # a second round starts from the synchronised copy
        syn0 += 1.0 if rank == 0 else 3.0
        sync()
        ok = ok and torch.allclose(syn0, want0 + 2.0, atol=1e-6)
        # walk sharding: ranges are disjoint and cover all start vertices
        lo, hi = shard_range(1001, rank, world)
        cover = torch.zeros(1001)
        cover[lo:hi] = 1
        dist.all_reduce(cover)
        ok = ok and bool((cover == 1).all())
        # vocabulary from globally summed counts: identical on every rank
        from node2vec_amd.sgns import build_vocab

        shards = [torch.tensor([[5, 5, 2, -1], [9, 2, 5, 5]], dtype=torch.int32),
                  torch.tensor([[9, 9, 9, 7], [11, 2, 2, 2]], dtype=torch.int32)]
        v = build_vocab(shards[rank], min_count=2)
        ok = ok and v.ids.tolist() == [2, 5, 9] and v.counts.tolist() == [5, 4, 4]  # ties: id asc
        ok = ok and v.index_of.tolist() == [-1, -1, 0, -1, -1, 1, -1, -1, -1, 2, -1, -1]
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_delta_all_reduce_and_sharding_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_delta_all_reduce_is_identity_on_one_rank():
    from node2vec_amd.sgns import DeltaAllReduce

    t = torch.randn(5, 4)
    want = t.clone() + 1
    sync = DeltaAllReduce([t])
    t += 1
    sync()
    assert torch.equal(t, want) and np.isfinite(t.numpy()).all()
