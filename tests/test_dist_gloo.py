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
        from node2vec_amd.sgns import DeltaSync
        from node2vec_amd.shard import shard_range

        torch.manual_seed(0)
        base0, base1 = torch.randn(37, 8), torch.randn(37, 8)  # same on every rank
        ok = True
        for wire, tol in (("fp32", 1e-6), ("bf16", 2e-2)):
            syn0, syn1 = base0.clone(), base1.clone()
            sync = DeltaSync([syn0, syn1], block_rows=16, sync_every=1, wire=wire)
            g = torch.Generator().manual_seed(100 + rank)
            d0, d1 = torch.randn(37, 8, generator=g), torch.randn(37, 8, generator=g)
            syn0 += d0  # "local training" since the last exchange
            syn1 += d1
            sync.step()
            # every rank must hold base + mean of all ranks' deltas
            all_d0 = [torch.randn(37, 8, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
            want0 = base0 + sum(all_d0) / world
            ok = ok and torch.allclose(syn0, want0, atol=tol)
            gathered = [torch.zeros_like(syn1) for _ in range(world)]
            dist.all_gather(gathered, syn1)
            ok = ok and all(torch.equal(gathered[0], x) for x in gathered)  # identical replicas
            # a second round starts from the synchronised state
            syn0 += 1.0 if rank == 0 else 3.0
            sync.step()
            ok = ok and torch.allclose(syn0, want0 + 2.0, atol=2 * tol)
            sync.finish()
            ok = ok and sync.syncs == 3
            # sgns.exchange_plan (what bench.py prints for a world of 8) against the buffers an exchange really holds
            # (rows a multiple of the block: ordered_sum keeps the buffers of the LAST block size only)
            from node2vec_amd.sgns import exchange_plan

            ok = ok and exchange_plan([(37, 8), (37, 8)], world, wire, block_rows=16)["blocks_per_sync"] * 3 == sync.exchanged_blocks
            t2 = torch.zeros(32, 8) + rank
            s2 = DeltaSync([t2], block_rows=16, sync_every=1, wire=wire)
            s2.step()
            plan = exchange_plan([(32, 8)], world, wire, block_rows=16)
            wb = 4 if wire == "fp32" else 2
            held = s2._before.numel() * 4 + s2._wire.numel() * wb
            sc = sum(b.numel() * b.element_size() for bufs in s2._scratch.values() for b in bufs)
            refs = 0 if s2.refs is None else sum(r.numel() * 2 for r in s2.refs)
            ok = ok and plan["block_buffers_bytes"] == held and plan["ordered_sum_buffers_bytes"] == sc
            ok = ok and plan["bf16_reference_bytes"] == refs and plan["wire_bytes_per_rank_per_sync"] == s2.wire_bytes
            ok = ok and plan["blocks_per_sync"] == s2.exchanged_blocks == 2
            ok = ok and plan["bytes_per_link_per_direction_per_sync"] == 2 * 2 * wb * (16 * 8 // world)
        # the sum itself (shard.ordered_sum): bytes through the backend, additions in fp32 in rank
        # order -- bit-equal to that sum spelled out locally, for both wire types, with a length
        # that does not divide by the world size; identical on every rank
        from node2vec_amd.shard import ordered_sum

        for dtype in (torch.float32, torch.bfloat16):
            parts = [(torch.randn(1001, generator=torch.Generator().manual_seed(7 + r)) * 3.0 ** r).to(dtype)
                     for r in range(world)]
            acc = parts[0].float().clone()
            for r in range(1, world):
                acc += parts[r].float()
            want = acc.to(dtype)
            mine, scratch = parts[rank].clone(), {}
            ordered_sum(mine, None, dist, scratch)
            ok = ok and torch.equal(mine.view(torch.uint8), want.view(torch.uint8))
            again = parts[rank].clone()
            ordered_sum(again, None, dist, scratch)  # the buffers are reused
            ok = ok and torch.equal(again, mine)
        # period logic: 7 launches at sync_every=3 -> exchanges after launches 3 and 6 + finish()
        t = torch.zeros(4, 2) + rank
        sync = DeltaSync([t], sync_every=3)
        seen = []
        for k in range(7):
            t += 1.0
            sync.step()
            seen.append(sync.syncs)
        sync.finish()
        ok = ok and seen == [0, 0, 1, 1, 1, 2, 2] and sync.syncs == 3
        ok = ok and torch.allclose(t, torch.full((4, 2), 7.5))  # mean of (0 + 7, 1 + 7)
        # SgnsModel.train with UNEVEN shards: rank 0 holds 5 rows, rank 1 holds 9; with blocks of
        # 4 rows rank 0 has 2 blocks of its own, rank 1 has 3 -- both must run the grid of the
        # largest shard (3 blocks x 2 epochs) or the collectives dead-lock
        from node2vec_amd import sgns

        class _Model(sgns.SgnsModel):
            def __init__(self):  # no vocabulary, no kernel: the loop structure only
                self.sentences_seen = 0
                self.launched = []

            def train_block(self, walks_idx, alpha, sentence_base, deterministic=False):
                self.launched.append((walks_idx.shape[0], sentence_base))

        class _Sync:
            def __init__(self):
                self.steps = 0

            def step(self):
                self.steps += 1
                x = torch.ones(1)
                dist.all_reduce(x)  # a real collective: uneven call counts would hang here

            def finish(self):
                self.steps += 100

        rows = 5 if rank == 0 else 9
        rmax = torch.tensor([rows])
        dist.all_reduce(rmax, op=dist.ReduceOp.MAX)
        m, sy = _Model(), _Sync()
        m.train(torch.zeros((rows, 3), dtype=torch.int32), epochs=2, block_rows=4, sync=sy,
                rows_global_max=int(rmax))
        ok = ok and sy.steps == 106
        want_rows = [4, 1, 4, 1] if rank == 0 else [4, 4, 1, 4, 4, 1]
        ok = ok and [r for r, _ in m.launched] == want_rows
        ok = ok and len({b for _, b in m.launched}) == len(m.launched)  # sentence ids never repeat
        # autotuned period with an EMPTY shard: rank 1 trains nothing, so its own
        # launch time is ~0; the period must come from the MAX of the measured times (rank 0's
        # 20 ms launches), be the same on both ranks and stay below the launches of one epoch
        import time as _time

        class _Slow(sgns.SgnsModel):
            def __init__(self):
                self.sentences_seen = 0
                self.launched = 0

            def train_block(self, walks_idx, alpha, sentence_base, deterministic=False):
                self.launched += 1
                _time.sleep(0.02)

        tt = torch.zeros(64, 4) + rank
        auto = DeltaSync([tt], sync_every=None, comm_share=0.5)
        rows = 40 if rank == 0 else 0
        ms = _Slow()
        ms.train(torch.zeros((rows, 3), dtype=torch.int32), epochs=1, block_rows=4, sync=auto,
                 rows_global_max=40)
        every = torch.tensor([auto.sync_every or 0])
        both = [torch.zeros_like(every) for _ in range(world)]
        dist.all_gather(both, every)
        ok = ok and int(both[0]) == int(both[1]) and 1 <= int(both[0]) <= 10  # 10 launches per epoch
        ok = ok and auto.max_every == 10 and ms.launched == (10 if rank == 0 else 0)
        ok = ok and auto.syncs >= 2  # the timed first exchange + finish(), at least
        # the period function itself: a zero launch time is clamped, not propagated
        ok = ok and auto.period_for(1.0, 0.0) == 10
        auto.max_every = None
        ok = ok and auto.period_for(0.1, 1.0) == 1 and auto.period_for(1.0, 0.1) == 10
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
        # the capacity-bounded mailboxes of the partitioned walk (partitioned.Outboxes): box d of every rank
        # reaches rank d whole with the capacities as FIXED split sizes, and a list start written relative to
        # the pool of its destination points behind the pools of the ranks before this one
        from node2vec_amd import partitioned as P

        caps_h, caps_w = [[2, 3], [4, 1]], [[5, 6], [2, 7]]  # [source][destination]
        bx = P.Outboxes(world, rank, caps_h, caps_w, torch.device("cpu"))
        ok = ok and bx.box_starts.tolist() == ([0, 2, 5, 0, 5, 11] if rank == 0 else [0, 4, 5, 0, 2, 9])
        ok = ok and bx.off_add.tolist() == ([0] * 5 if rank == 0 else [5] * 4 + [6])

        def sent(src, kind):  # what rank `src` puts into its send arrays
            nh, nw = sum(caps_h[src]), sum(caps_w[src])
            if kind == "head":
                return (torch.arange(nh * P.HEAD_COLS, dtype=torch.int64) + 1000 * src).view(nh, P.HEAD_COLS)
            return torch.arange(nh, dtype=torch.int64) + 100 * src if kind == "off" else \
                torch.arange(nw, dtype=torch.int32) + 10 * src

        bx.send_head.copy_(sent(rank, "head")), bx.send_off.copy_(sent(rank, "off")), bx.send_words.copy_(sent(rank, "words"))
        P._exchange_bounded(bx, None, dist, True, True)
        for kind, got, caps in (("head", bx.recv_head, caps_h), ("off", bx.recv_off, caps_h),
                                ("words", bx.recv_words, caps_w)):
            want = []
            for src in range(world):  # box `rank` of every source, in source order
                at = sum(caps[src][:rank])
                want.append(sent(src, kind)[at:at + caps[src][rank]])
            ok = ok and torch.equal(got, torch.cat(want))
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


def test_delta_sync_is_identity_on_one_rank():
    from node2vec_amd.sgns import DeltaSync

    t = torch.randn(5, 4)
    want = t.clone() + 1
    sync = DeltaSync([t])
    t += 1
    sync.step()
    sync.finish()
    assert torch.equal(t, want) and np.isfinite(t.numpy()).all()
