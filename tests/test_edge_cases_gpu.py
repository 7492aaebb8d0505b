"""Edge cases of the walk / alias / SGNS entry points on the GPU (ragged and empty
inputs, extreme lengths), each checked against the oracle where one exists."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _g(edges, nv=None):
    from node2vec_amd.graph import DeviceGraph

    e = np.array(edges, dtype=np.float64).reshape(-1, 3)
    return DeviceGraph.from_edges(e[:, 0].astype(np.int64), e[:, 1].astype(np.int64),
                                  e[:, 2].astype(np.float32), n_vertices=nv, device="cuda")


def _both(oracle, g, start, nw, wl, p, q, seed, mode="exact"):
    from node2vec_amd import randomwalk as rw

    got, gv = rw.walk(g, torch.as_tensor(start, dtype=torch.int32), nw, wl, p, q, seed, mode)
    want, wv = oracle.random_walk(g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.w.cpu().numpy(),
                                  np.asarray(start, np.int32), nw, wl, p, q, seed)
    return got.cpu().numpy(), gv.cpu().numpy().astype(bool), want, wv


def test_no_walkers_and_zero_counts():
    from node2vec_amd import randomwalk as rw

    g = _g([(0, 1, 1.0), (1, 0, 1.0)])
    for mode in ("exact", "fast"):
        w, v = rw.walk(g, torch.zeros(0, dtype=torch.int32), 3, 5, 0.5, 2.0, 1, mode)
        assert w.shape == (0, 6) and v.numel() == 0
        w, v = rw.walk(g, torch.tensor([0, 1], dtype=torch.int32), 0, 5, 0.5, 2.0, 1, mode)
        assert w.shape == (0, 6)


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_walk_length_zero_is_just_the_start_vertex(oracle, mode):
    g = _g([(0, 1, 1.0), (1, 0, 1.0), (2, 0, 1.0)], nv=4)
    got, gv, want, wv = _both(oracle, g, [0, 1, 2, 3], 2, 0, 0.5, 2.0, 5, mode)
    assert gv.tolist() == wv.tolist() == [True] * 6 + [False] * 2  # vertex 3 has no out-edges
    assert got[gv].reshape(-1).tolist() == [0, 0, 1, 1, 2, 2]


@pytest.mark.parametrize("wl", [127, 128, 300])
def test_walks_around_and_beyond_the_register_path_buffer(oracle, wl):
    from node2vec_amd import synthetic

    g = synthetic.rmat(10, 5000, device="cuda")
    start = list(range(0, 1024, 9))
    got, gv, want, wv = _both(oracle, g, start, 2, wl, 0.5, 2.0, 77)
    assert np.array_equal(gv, wv) and np.array_equal(got, want) and got.shape[1] == wl + 1


def test_self_loops_multi_edges_and_duplicate_starts(oracle):
    # s in N(s), repeated edges, a vertex whose only edge is a self-loop
    edges = [(0, 0, 1.0), (0, 1, 1.0), (0, 1, 1.0), (1, 0, 1.0), (1, 2, 1.0), (2, 2, 1.0),
             (1, 1, 1.0), (0, 2, 1.0), (2, 0, 1.0), (2, 0, 1.0)]
    for w in (1.0, 0.7):
        g = _g([(a, b, w if (a + b) % 2 else 1.0) for a, b, _ in edges])
        for p, q in ((0.5, 2.0), (4.0, 0.25), (3.0, 0.7)):
            got, gv, want, wv = _both(oracle, g, [2, 0, 0, 1, 2], 5, 40, p, q, 9)
            assert np.array_equal(gv, wv) and np.array_equal(got, want)
            assert np.array_equal(got[0:5], got[20:25])  # the same start vertex twice: same walks


def test_out_of_range_start_id_raises():
    from node2vec_amd import randomwalk as rw

    g = _g([(0, 1, 1.0), (1, 0, 1.0)])
    for bad in (5, -1):
        with pytest.raises(ValueError):
            rw.walk(g, torch.tensor([0, bad], dtype=torch.int32), 1, 3, 1.0, 1.0, 1)


def test_star_graph_hub_return_probabilities(oracle):
    """a star: from a leaf the only move is the hub; from the hub, p decides the return"""
    n = 3000
    edges = [(0, i, 1.0) for i in range(1, n)] + [(i, 0, 1.0) for i in range(1, n)]
    g = _g(edges)
    got, gv, want, wv = _both(oracle, g, list(range(0, n, 37)), 4, 30, 0.25, 4.0, 3)
    assert np.array_equal(gv, wv) and np.array_equal(got, want)
    # with p = 0.25 (return weight 4 against 1/4 for each of 2998 others) returns are rare but present
    w = got[gv]
    ret = (w[:, 2:] == w[:, :-2])
    assert 0.0 < ret.mean() < 1.0


@pytest.mark.parametrize("pq", [(0.5, 2.0), (4.0, 0.25), (0.25, 0.25), (4.0, 4.0), (1.0, 0.5),
                                (3.0, 0.7), (0.7, 1.3), (1.5, 1.0 / 3.0)])
def test_rows_longer_than_the_ballot_cache(oracle, pq):
    """rows of 20 000 neighbours (the unit kernel caches class ballots for 8192): reverse
    classification keeps return/shared slots below the cache by position (hub A seen from a
    leaf: 1-3 slots; from B: 300 slots, more than the list holds, so it falls back to
    searching), with multi-edges, for every arrangement of under/overfull classes; the
    non-dyadic p, q also exercise the run-by-run row sum across many binades"""
    rng = np.random.default_rng(5)
    A, B, C = 0, 1, 2
    leaves = np.arange(100, 20100)
    src = [np.full(leaves.size, A), leaves]          # A <-> every leaf
    dst = [leaves, np.full(leaves.size, A)]
    lb = leaves[:300]                                  # B <-> A and the 300 lowest leaves
    src += [np.full(lb.size, B), lb, [A, B]]
    dst += [lb, np.full(lb.size, B), [B, A]]
    lc = rng.choice(leaves, 9000, replace=True)        # C <-> 9000 random leaves (with repeats) and A
    src += [np.full(lc.size, C), lc, [A, C]]
    dst += [lc, np.full(lc.size, C), [C, A]]
    dup = leaves[::997]                                # multi-edges A <-> leaf
    src += [np.full(dup.size, A), dup]
    dst += [dup, np.full(dup.size, A)]
    src, dst = np.concatenate(src), np.concatenate(dst)
    from node2vec_amd.graph import DeviceGraph

    g = DeviceGraph.from_edges(src, dst, np.ones(src.size, np.float32), n_vertices=20100, device="cuda")
    assert g.unit_weights and int(g.degrees().max()) > 8192
    start = [A, B, C] + list(leaves[::400]) + list(lb[::60])
    got, gv, want, wv = _both(oracle, g, start, 3, 12, pq[0], pq[1], 77)
    assert np.array_equal(gv, wv) and np.array_equal(got, want)


def test_default_p_q_uses_the_first_order_tables_and_the_same_bits(oracle, monkeypatch):
    """p = q = 1 (the reference's defaults): exact mode reads the K1 tables when they exist and
    rebuilds per step when they do not; both equal the oracle on a weighted graph with sinks"""
    from node2vec_amd import randomwalk as rw
    from node2vec_amd.graph import DeviceGraph

    rng = np.random.default_rng(21)
    src, dst = rng.integers(0, 800, 9000), rng.integers(0, 800, 9000)
    keep = src % 11 != 3
    src, dst = src[keep], dst[keep]
    w = rng.uniform(0.1, 3.0, src.size).astype(np.float32)
    g = DeviceGraph.from_edges(src, dst, w, n_vertices=800, device="cuda")
    start = np.arange(0, 800, dtype=np.int32)
    want, wv = oracle.random_walk(g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.w.cpu().numpy(),
                                  start, 3, 30, 1.0, 1.0, 9)
    assert g.slots is None
    got, gv = rw.walk(g, torch.as_tensor(start), 3, 30, 1.0, 1.0, 9)  # builds and uses the slots
    assert g.slots is not None
    assert np.array_equal(gv.cpu().numpy().astype(bool), wv) and np.array_equal(got.cpu().numpy(), want)
    g.slots = g.pivots = None

    def refuse(self):
        raise ZeroDivisionError("pretend a row sums to zero")

    monkeypatch.setattr(DeviceGraph, "build_alias", refuse)
    got2, gv2 = rw.walk(g, torch.as_tensor(start), 3, 30, 1.0, 1.0, 9)  # per-step rebuild
    assert g.slots is None
    assert torch.equal(got, got2) and torch.equal(gv, gv2)


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_row_of_more_than_two_million_neighbours(oracle, mode):
    """a star with 2.2 M leaves: above 2^21 neighbours int(r1 * n) is no longer an exact
    integer product in the reference (fp64 rounds before int() truncates); the kernels switch
    to the same two fp64 operations.  Exact mode equals the oracle; fast mode at p = q = 1
    is the same draw"""
    from node2vec_amd import randomwalk as rw
    from node2vec_amd.graph import DeviceGraph

    n = 2_200_000
    leaves = torch.arange(1, n + 1, device="cuda")
    hub = torch.zeros(n, dtype=torch.int64, device="cuda")
    g = DeviceGraph.from_edges(torch.cat([hub, leaves]), torch.cat([leaves, hub]),
                               torch.ones(2 * n, device="cuda"), n_vertices=n + 1, device="cuda")
    assert int(g.degrees().max()) > (1 << 21)
    start = np.array([0, 1, 5, 70001, 1048577, 2097153, n], np.int32)
    p, q = (1.0, 1.0) if mode == "fast" else (0.5, 2.0)
    got, gv = rw.walk(g, torch.as_tensor(start), 3, 4, p, q, 11, mode=mode)
    want, wv = oracle.random_walk(g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.w.cpu().numpy(),
                                  start, 3, 4, p, q, 11, n_threads=8)
    assert np.array_equal(gv.cpu().numpy().astype(bool), wv) and np.array_equal(got.cpu().numpy(), want)


def test_alias_build_on_empty_and_single_rows(oracle):
    from node2vec_amd.graph import DeviceGraph

    g = DeviceGraph.from_edges([2], [0], [0.3], n_vertices=5, device="cuda").build_alias()
    assert g.alias.tolist() == [0] and g.prob.tolist() == [1.0]  # alias index 0 -> neighbour id 0
    a, p = oracle.alias_tables([float(np.float32(0.3))])
    assert (a, p) == ([0], [1.0])
    empty = DeviceGraph.from_edges([], [], [], n_vertices=3, device="cuda").build_alias()
    assert empty.n_edges == 0 and empty.slots.shape == (0, 4)


def test_sgns_rows_without_any_pair(oracle):
    """rows that are all out-of-vocabulary, or a single token: nothing is trained"""
    from node2vec_amd import sgns

    walks = torch.tensor([[-1, -1, -1, -1], [3, -1, -1, -1], [-1, 2, -1, -1]], dtype=torch.int32).cuda()
    vocab = sgns.Vocab(torch.arange(5).cuda(), torch.tensor([5, 4, 3, 2, 1]).cuda(),
                       torch.arange(5, dtype=torch.int32).cuda())
    m = sgns.SgnsModel(vocab, 32, 5, 5, seed=1)
    before = m.syn0.clone()
    m.train_block(walks, 0.025, 0)
    torch.cuda.synchronize()
    assert int(m.pairs.item()) == 0 and torch.equal(m.syn0, before) and float(m.syn1neg.abs().max()) == 0.0
