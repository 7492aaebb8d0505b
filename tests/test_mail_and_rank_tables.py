"""Host logic of round 4 that needs no GPU: the class / head tables of the degree-ranked form
(graph.rank_tables: plain torch) and the exchange of walkers that travel as Mail
(partitioned._exchange_mail) between two gloo ranks."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _rows(first, off, head, n, n_edges):
    """(row start, degree) of every rank the way walk_uniform_kernel<3> derives them"""
    first = first.numpy().astype(np.uint32).astype(np.int64)
    off = off.numpy().astype(np.uint32).astype(np.int64)
    P = first.size
    assert P & (P - 1) == 0 and P >= 2 and first[-1] == n and off[-1] == n_edges
    r = np.arange(n, dtype=np.int64)
    c = np.zeros(n, np.int64)
    half = P >> 1
    while half:
        c = np.where(first[np.minimum(c + half, P - 1)] <= r, c + half, c)
        half >>= 1
    deg = (off[c + 1] - off[c]) // np.maximum(first[c + 1] - first[c], 1)
    row = off[c] + (r - first[c]) * deg
    if head is not None:
        h = head.numpy().astype(np.uint64)
        row[:h.size] = (h & np.uint64((1 << 40) - 1)).astype(np.int64)
        deg[:h.size] = (h >> np.uint64(40)).astype(np.int64)
    return row, deg


@pytest.mark.parametrize("max_classes", [8191, 64, 3, 1])
@pytest.mark.parametrize("kind", ["power", "all_distinct", "flat", "with_zeros"])
def test_rank_tables_give_every_rank_its_row(kind, max_classes):
    from node2vec_amd.graph import rank_tables

    rng = np.random.default_rng(3)
    n = 5000
    if kind == "power":
        deg = np.minimum(rng.pareto(1.1, n) * 3, 4000).astype(np.int64)
    elif kind == "all_distinct":
        deg = rng.permutation(n).astype(np.int64)
    elif kind == "flat":
        deg = np.full(n, 7, np.int64)
    else:
        deg = rng.integers(0, 4, n).astype(np.int64)
    out = rank_tables(torch.from_numpy(deg), max_classes, 1 << 22)
    assert out is not None
    rank_vertex, rank_of, rank_rowptr, head, first, off = out
    rv = rank_vertex.numpy()
    assert np.array_equal(rv, np.argsort(-deg, kind="stable"))
    assert np.array_equal(rank_of.numpy()[rv], np.arange(n))
    row, d = _rows(first, off, head, n, int(deg.sum()))
    assert np.array_equal(d, deg[rv])
    assert np.array_equal(row, rank_rowptr.numpy()[:-1])
    n_cls = len(np.unique(deg))
    if n_cls > max_classes:
        assert head is not None and int(first[0]) == head.numel()
        assert int((first.numpy().astype(np.uint32) < n).sum()) == max_classes
    else:
        assert head is None
    # too many vertices to list one by one: declined
    if kind == "all_distinct":
        assert rank_tables(torch.from_numpy(deg), 8, 100) is None


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _mail_for(src, dst):
    """what rank `src` sends to rank `dst`: (5 + src + 2 dst) walkers with lists of 0 .. 3 words"""
    from node2vec_amd import partitioned as P

    k = 5 + src + 2 * dst
    head = torch.arange(k * P.HEAD_COLS, dtype=torch.int64).reshape(k, P.HEAD_COLS) + 1000 * src + 100000 * dst
    lens = torch.arange(k) % 4
    off = torch.cumsum(lens, 0) - lens  # (any order would do: here back to back)
    words = torch.arange(int(lens.sum()), dtype=torch.int32) + 7 * src + 70 * dst
    return P.Mail(head, off.to(torch.int64), words), lens


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from node2vec_amd import partitioned as P

        out = [_mail_for(rank, d)[0] for d in range(world)]
        got = P._exchange_mail(out, None, dist, torch.device("cpu"), status=0)
        assert len(got) == 1
        m = got[0]
        ok, at = True, 0
        for src in range(world):  # segments arrive in source order; every list is found at its start
            want, lens = _mail_for(src, rank)
            seg = slice(at, at + len(want))
            ok = ok and torch.equal(m.head[seg], want.head)
            for i in range(len(want)):
                o = int(m.off[at + i])
                ok = ok and torch.equal(m.words[o:o + int(lens[i])], want.words[int(want.off[i]):int(want.off[i]) + int(lens[i])])
            at += len(want)
        ok = ok and at == len(m)
        # a status word on one rank raises on every rank
        raised = False
        try:
            P._exchange_mail(out, None, dist, torch.device("cpu"), status=1 if rank == 1 else 0)
        except ZeroDivisionError:
            raised = True
        ret[rank] = bool(ok) and raised
    finally:
        dist.destroy_process_group()


def test_mail_exchange_between_two_gloo_ranks():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret[0] and ret[1]


def test_mail_cat_rebases_list_starts():
    from node2vec_amd import partitioned as P

    a, la = _mail_for(0, 0)
    b, lb = _mail_for(1, 0)
    m = P.Mail.cat([a, P.Mail.empty("cpu"), b], "cpu")
    assert len(m) == len(a) + len(b) and m.words.numel() == a.words.numel() + b.words.numel()
    for i in range(len(b)):
        o = int(m.off[len(a) + i])
        assert torch.equal(m.words[o:o + int(lb[i])], b.words[int(b.off[i]):int(b.off[i]) + int(lb[i])])
    assert P.Mail.cat([], "cpu").head.shape == (0, P.HEAD_COLS)
