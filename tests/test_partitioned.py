"""Graph-partitioned walking (node2vec_amd/partitioned.py) on CPU: the
partition, the walker migration and the path assembly, with the CPU oracle standing in for
the step function (the HIP one needs a GPU: tests/test_partitioned_gpu.py).  In one process
(list-transpose exchange, 1 to 5 parts) and as two gloo ranks exchanging through
torch.distributed.all_to_all_single: the walks must equal the oracle's walk over the WHOLE
graph, row for row -- which is what n2v_walk produces on one GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def oracle_step(dst_ptr, dst_ids, dst_w, src_id, src_ptr, src_ids, keys, steps, p, q, seed):
    """next_step_random_walk row by row with the oracle (checker code, test-side only)"""
    import n2v_oracle as o

    dp, di = dst_ptr.numpy(), dst_ids.numpy()
    sp, si = src_ptr.numpy(), src_ids.numpy()
    out = np.zeros(len(keys), np.int32)
    for r in range(len(keys)):
        ids = di[dp[r]:dp[r + 1]]
        w = np.ones(len(ids)) if dst_w is None else dst_w.numpy()[dp[r]:dp[r + 1]].astype(np.float64)
        s = int(src_id[r])
        if s < 0:
            alias, probs = o.alias_tables(w)
        else:
            alias, probs = o.edge_alias_tables(s, si[sp[r]:sp[r + 1]].tolist(), ids, w, p, q)
        u1, u2 = o.uniform_bits(seed, int(keys[r]), int(steps[r]))
        out[r] = ids[o.sampling_from_alias(alias, probs, u1 / 2.0 ** 32, u2 / 2.0 ** 32)]
    return torch.from_numpy(out)


def _graph(weighted, seed=3, nv=90, ne=700):
    from node2vec_amd.graph import DeviceGraph

    rng = np.random.default_rng(seed)
    src = rng.integers(0, nv - 8, ne)   # ids >= nv - 8 never have out-edges: sinks
    dst = rng.integers(0, nv, ne)
    w = (rng.random(ne) * 1.5 + 0.25) if weighted else None
    return DeviceGraph.from_edges(src, dst, w, n_vertices=nv)


def _whole_graph_walks(g, start, W, L, p, q, seed):
    import n2v_oracle as o

    return o.random_walk(g.rowptr.numpy(), g.col.numpy(), None if g.unit_weights else g.w.numpy(),
                         start.numpy(), W, L, p, q, seed)


@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("n_parts", [1, 2, 5])
def test_partitioned_walks_equal_whole_graph_walks(oracle, weighted, n_parts):
    from node2vec_amd import partitioned as P

    g = _graph(weighted)
    deg = g.degrees()
    start = torch.nonzero(deg >= 0).reshape(-1).to(torch.int32)  # sinks included: they start nothing
    parts = P.partition_graph(g, n_parts)
    assert sum(pt.col.numel() for pt in parts) == g.n_edges and parts[0].lo == 0 and parts[-1].hi == g.n_vertices
    for p, q in ((1.0, 1.0), (0.5, 2.0), (3.0, 0.7)):
        walks, valid = P.walk_partitioned_local(parts, start, 3, 12, p, q, 21, step_fn=oracle_step)
        want, wv = _whole_graph_walks(g, start, 3, 12, p, q, 21)
        assert np.array_equal(valid.numpy(), wv)
        assert not wv.all() and wv.any()  # some walkers vanish at sinks, some survive
        assert np.array_equal(walks.numpy()[wv], want[wv])
        # a dropped walker's row holds its path up to the sink, like n2v_walk's
        dropped = ~wv & (want[:, 0] >= 0)
        assert np.array_equal(walks.numpy()[dropped], want[dropped])


def test_partition_by_vertices_and_walk_length_zero(oracle):
    from node2vec_amd import partitioned as P

    g = _graph(False, seed=8)
    start = torch.arange(0, g.n_vertices, 3, dtype=torch.int32)
    parts = P.partition_graph(g, 4, balance="vertices")
    assert [pt.hi - pt.lo for pt in parts] == [23, 23, 22, 22]
    walks, valid = P.walk_partitioned_local(parts, start, 2, 0, 0.5, 2.0, 5, step_fn=oracle_step)
    want, wv = _whole_graph_walks(g, start, 2, 0, 0.5, 2.0, 5)
    assert np.array_equal(valid.numpy(), wv) and np.array_equal(walks.numpy()[wv], want[wv])
    with pytest.raises(ValueError):
        P.walk_partitioned_local(parts, start, 2, 3, 0.0, 1.0, 5, step_fn=oracle_step)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys

        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
        from node2vec_amd import partitioned as P

        g = _graph(True, seed=5)
        start = torch.arange(g.n_vertices, dtype=torch.int32)
        part = P.partition_graph(g, world)[rank]  # this rank keeps ITS rows only
        walks, valid, rows = P.walk_partitioned(part, start, 2, 10, 0.5, 2.0, 9, step_fn=oracle_step)
        want, wv = _whole_graph_walks(g, start, 2, 10, 0.5, 2.0, 9)
        mine = rows.numpy()
        ok = np.array_equal(valid.numpy(), wv[mine]) and np.array_equal(walks.numpy()[valid.numpy()],
                                                                        want[mine][wv[mine]])
        ok = ok and len(mine) == 2 * int(((start >= part.lo) & (start < part.hi)).sum())
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_partitioned_walk_two_gloo_ranks(oracle):
    world = 2
    port = _free_port()
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


@pytest.mark.parametrize("n_parts", [255, 256, 300, 40000])
def test_log_by_home_with_more_parts_than_a_byte_holds(n_parts):
    """path records grouped by the rank that emits their row: the sort key must hold every rank AND
    the sentinel of the empty slots (round-5 advisor finding: a one-byte key wrapped rank 300 to 44 and,
    at 256 parts, the sentinel to rank 0); then the rows every rank assembles"""
    from node2vec_amd.partitioned import GraphPart, RankState

    nv, W, L = 2 * n_parts, 2, 3
    bounds = torch.arange(n_parts, dtype=torch.int64) * 2           # part r = vertices [2 r, 2 r + 2)
    rowptr = torch.tensor([0, 1, 2], dtype=torch.int64)
    part = GraphPart(7, 14, 16, rowptr, torch.zeros(2, dtype=torch.int32), None, bounds)
    st = RankState(part, W, L, 1.0, 1.0, 0, step_fn=None)
    start = torch.arange(nv, dtype=torch.int64)                      # every vertex starts W walks
    gen = torch.Generator().manual_seed(n_parts)
    rows = torch.randint(0, nv * W, (5000,), generator=gen)
    rows[::7] = -1                                                   # empty slots of a bounded inbox
    rec = torch.stack([rows, torch.randint(0, L + 1, (5000,), generator=gen),
                       torch.randint(0, nv, (5000,), generator=gen)], 1)
    st.log = [rec[:1234], rec[1234:]]
    by_home = st.log_by_home(start, n_parts)
    assert len(by_home) == n_parts
    real = rec[rows >= 0]
    assert sum(r.shape[0] for r in by_home) == real.shape[0]
    for r in (0, 1, 43, 44, n_parts // 2, n_parts - 2, n_parts - 1):
        lo, hi = 2 * r * W, 2 * (r + 1) * W                         # the output rows of rank r
        want = real[(real[:, 0] >= lo) & (real[:, 0] < hi)]
        got = by_home[r]
        assert bool(((got[:, 0] >= lo) & (got[:, 0] < hi)).all()), r
        assert sorted(map(tuple, got.tolist())) == sorted(map(tuple, want.tolist())), r
    # rank 7 assembles its rows from what it was handed: every record lands inside its own range
    walks, ok, out_rows = st.assemble([by_home[7]], start)
    assert walks.shape == (2 * W, L + 1) and out_rows.tolist() == list(range(14 * W, 16 * W))
