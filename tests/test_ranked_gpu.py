"""The degree-ranked form of a unit-weight graph (n2v_graph.rank_*, n2v_rank_hops_build) and the
p = q = 1 walk on it: table contents against numpy, walks bit-identical to the hop-table kernel,
the CSR kernel and the CPU oracle -- with and without the head table, sinks, hubs, multi-edges, walk
lengths 0, 1 and across output sectors."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _graph(seed=8, nv=4000):
    from node2vec_amd.graph import DeviceGraph

    rng = np.random.default_rng(seed)
    src = np.concatenate([rng.integers(0, nv - 60, 30000), rng.integers(0, 10, 12000), rng.integers(0, nv - 60, 9000)])
    dst = np.concatenate([rng.integers(0, nv, 30000), rng.integers(0, nv, 12000), rng.integers(0, 10, 9000)])
    return DeviceGraph.from_edges(src, dst, None, n_vertices=nv, device="cuda")


def _rows_from_tables(g):
    """(row start, degree) of every rank as the walk kernel derives them: head table, else class search"""
    n = g.n_vertices
    head = np.zeros(0, np.uint64) if g.rank_head is None else g.rank_head.cpu().numpy().astype(np.uint64)
    first = g.rank_class_first.cpu().numpy().astype(np.uint32).astype(np.int64)
    off = g.rank_class_off.cpu().numpy().astype(np.uint32).astype(np.int64)
    P = first.size
    assert P & (P - 1) == 0 and 2 <= P <= 8192 and first[0] == head.size
    assert first[-1] == n and off[-1] == g.n_edges  # the entry that closes the last class
    r = np.arange(n, dtype=np.int64)
    c = np.zeros(n, np.int64)
    half = P >> 1
    while half:  # the kernel's fixed-depth search
        c = np.where(first[np.minimum(c + half, P - 1)] <= r, c + half, c)
        half >>= 1
    deg = (off[c + 1] - off[c]) // (first[c + 1] - first[c])
    row = off[c] + (r - first[c]) * deg
    H = head.size
    row[:H] = (head & np.uint64((1 << 40) - 1)).astype(np.int64)
    deg[:H] = (head >> np.uint64(40)).astype(np.int64)
    return row, deg


@pytest.mark.parametrize("max_classes", [8191, 16, 1])
def test_ranked_tables_hold_the_graph(max_classes):
    g = _graph()
    g.RANK_MAX_CLASSES = max_classes  # few classes: most ranks go through the head table
    g.build_ranked()
    assert g.rank_hops is not None
    deg = g.degrees().cpu().numpy()
    rowptr, col = g.rowptr.cpu().numpy(), g.col.cpu().numpy()
    rv, ro = g.rank_vertex.cpu().numpy(), g.rank_of.cpu().numpy()
    assert np.array_equal(rv, np.argsort(-deg, kind="stable"))  # descending degree, ties by ascending id
    assert np.array_equal(ro[rv], np.arange(g.n_vertices))
    row, d = _rows_from_tables(g)
    assert np.array_equal(d, deg[rv])
    assert np.array_equal(row, np.concatenate([[0], np.cumsum(deg[rv])[:-1]]))
    if max_classes < 4096:
        assert g.rank_head is not None and g.rank_class_first.numel() == 2 * max_classes
    hops = g.rank_hops.cpu().numpy().astype(np.uint32)
    for r in list(range(0, 40)) + list(range(40, g.n_vertices, 37)):
        v = rv[r]
        assert np.array_equal(hops[row[r]:row[r] + d[r]], ro[col[rowptr[v]:rowptr[v + 1]]]), r
    # every entry at once
    src_rank = np.repeat(np.arange(g.n_vertices), d)
    at_csr = rowptr[rv[src_rank]] + (np.arange(col.size) - row[src_rank])
    assert np.array_equal(hops.astype(np.int64), ro[col[at_csr]].astype(np.int64))


@pytest.mark.parametrize("max_classes", [8191, 16, 1])
def test_ranked_walks_change_no_bit(oracle, max_classes):
    from node2vec_amd import randomwalk as rw

    g = _graph()
    g.RANK_MAX_CLASSES = max_classes
    start = rw.start_vertices(g)
    for L in (0, 1, 14, 15, 16, 30, 80):
        a, av = rw.walk(g, start, 3, L, 1.0, 1.0, 4, use_hops8=False)
        b, bv = rw.walk(g, start, 3, L, 1.0, 1.0, 4, use_ranked=True)
        c, cv = rw.walk(g, start, 3, L, 1.0, 1.0, 4, rank_ids=True)
        assert g.rank_hops is not None
        assert torch.equal(a, b) and torch.equal(av, bv) and torch.equal(av, cv)
        back = torch.where(c >= 0, g.rank_vertex[c.clamp(min=0).long()], c)
        assert torch.equal(back, a)
    rowptr, col = g.rowptr.cpu().numpy(), g.col.cpu().numpy()
    want, wv = oracle.random_walk(rowptr, col, None, start.cpu().numpy(), 3, 30, 1.0, 1.0, 4, n_threads=8)
    b, bv = rw.walk(g, start, 3, 30, 1.0, 1.0, 4, use_ranked=True)
    assert np.array_equal(bv.cpu().numpy(), wv) and np.array_equal(b.cpu().numpy()[wv], want[wv])
    assert not wv.all()
    # start vertices that are sinks or out of range, an odd number of walkers
    odd = torch.tensor([5, 0, 3999, 17, 5], dtype=torch.int32)
    a, av = rw.walk(g, odd, 7, 33, 1.0, 1.0, 9, use_hops8=False)
    b, bv = rw.walk(g, odd, 7, 33, 1.0, 1.0, 9, use_ranked=True)
    assert torch.equal(a, b) and torch.equal(av, bv)
    with pytest.raises(ValueError):
        rw.walk(g, torch.tensor([4000], dtype=torch.int32), 1, 3, 1.0, 1.0, 1, use_ranked=True)


def test_rank_ids_are_refused_where_the_form_does_not_exist():
    from node2vec_amd import randomwalk as rw
    from node2vec_amd.graph import DeviceGraph

    g = _graph()
    with pytest.raises(ValueError):
        rw.walk(g, rw.start_vertices(g), 1, 5, 0.5, 2.0, 1, rank_ids=True)
    w = DeviceGraph.from_edges(np.array([0, 1, 2]), np.array([1, 2, 0]), np.array([1.0, 2.0, 3.0]), n_vertices=3,
                               device="cuda")
    with pytest.raises(ValueError):
        rw.walk(w, rw.start_vertices(w), 1, 5, 1.0, 1.0, 1, rank_ids=True)
    assert w.build_ranked().rank_hops is None
