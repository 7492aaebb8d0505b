"""GPU tests for K1 (n2v_alias_build), trimming (n2v_trim_mark) and the fast
(rejection) walk sampler, all through the C ABI."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _rand_graph(seed, nv, ne, weighted, hubs=4):
    from node2vec_amd.graph import DeviceGraph

    rng = np.random.default_rng(seed)
    src, dst = rng.integers(0, nv, ne), rng.integers(0, nv, ne)
    for h in rng.integers(0, nv, hubs):
        k = int(rng.integers(100, 2500))
        src = np.concatenate([src, np.full(k, h)])
        dst = np.concatenate([dst, rng.integers(0, nv, k)])
    w = (rng.choice([0.25, 0.5, 1.0, 2.0, 1.7, 0.3], len(src)) if weighted else np.ones(len(src))).astype(np.float32)
    return DeviceGraph.from_edges(src, dst, w, n_vertices=nv, device="cuda")


@pytest.mark.parametrize("weighted", [False, True])
def test_alias_build_equals_generate_alias_tables(oracle, weighted):
    """K1 row tables == generate_alias_tables(row weights): alias ints and fp64
    probs identical to the oracle (itself pinned to the reference, G1)."""
    g = _rand_graph(11, 2000, 30000, weighted).build_alias()
    torch.cuda.synchronize()
    rowptr, w, col = g.rowptr.cpu().numpy(), g.w.cpu().numpy(), g.col.cpu().numpy()
    alias, prob = g.alias.cpu().numpy(), g.prob.cpu().numpy()  # alias: vertex behind the index
    assert np.array_equal(g.slots[:, 0].cpu().numpy(), g.col.cpu().numpy())
    checked = 0
    for v in range(g.n_vertices):
        b, e = rowptr[v], rowptr[v + 1]
        if e == b:
            continue
        a, p = oracle.alias_tables(w[b:e].astype(np.float64))
        assert alias[b:e].tolist() == col[b:e][np.array(a)].tolist(), v
        assert prob[b:e].tolist() == p, v
        checked += 1
    assert checked > 1500


def test_alias_build_golden_rows(oracle):
    """the reference's own G1 vectors, one row each -- ALL of them: the graph holds fp64 weights
    (n2v_graph.w64) because rows like [0.5, 0.8, 1.0] (tests/test_randomwalk.py:135) and the
    *_fp64 cases are not fp32-representable"""
    from node2vec_amd.graph import DeviceGraph

    cases = load_golden("g1_alias_tables.json")
    assert any(c["weights"] == [0.5, 0.8, 1.0] for c in cases) and len(cases) >= 35
    src = np.concatenate([np.full(len(c["weights"]), i) for i, c in enumerate(cases)])
    dst = np.concatenate([np.arange(len(c["weights"])) for c in cases])
    w = np.concatenate([np.array(c["weights"], np.float64) for c in cases])
    g = DeviceGraph.from_edges(src, dst, w, n_vertices=max(len(cases), int(dst.max()) + 1),
                               device="cuda").build_alias()
    assert g.w.dtype == torch.float64
    rowptr, col = g.rowptr.cpu().numpy(), g.col.cpu().numpy()
    alias, prob = g.alias.cpu().numpy(), g.prob.cpu().numpy()
    for i, c in enumerate(cases):
        b, e = rowptr[i], rowptr[i + 1]
        # the row's ids are 0..n-1 here, so the vertex behind an alias index IS the index
        assert col[b:e].tolist() == list(range(e - b))
        assert alias[b:e].tolist() == c["alias"] and prob[b:e].tolist() == c["probs"], i


def test_alias_build_zero_row_raises():
    from node2vec_amd.graph import DeviceGraph

    g = DeviceGraph.from_edges([0, 0, 1], [1, 2, 0], [0.0, 0.0, 1.0], device="cuda")
    with pytest.raises(ZeroDivisionError):
        g.build_alias()


def test_trim_mark_equals_oracle_and_reference_properties(oracle):
    """tests/test_randomwalk.py:194-224: exactly cap edges of a hot row survive, all
    from that row; other rows untouched.  Plus bit-equality with the oracle."""
    from node2vec_amd.fugue import trim_hotspot_edges

    rng = np.random.default_rng(5)
    src = np.concatenate([rng.integers(0, 300, 4000), np.full(5000, 7), np.full(901, 299)])
    srct = torch.from_numpy(src).cuda()
    keep = trim_hotspot_edges(srct, 900, 20).cpu().numpy()
    counts_before = np.bincount(src, minlength=300)
    counts_after = np.bincount(src[keep], minlength=300)
    assert counts_after[7] == 900 and counts_after[299] == 900
    cold = counts_before <= 900
    assert (counts_after[cold] == counts_before[cold]).all()
    # same marks as the oracle on the CSR ordering
    order = np.argsort(src, kind="stable")
    rowptr = np.concatenate([[0], np.cumsum(counts_before)])
    want = oracle.trim_mark(rowptr, 900, 20)
    assert np.array_equal(keep[order], want)
    # a different seed picks a different subset
    keep2 = trim_hotspot_edges(srct, 900, 21).cpu().numpy()
    assert (keep != keep2).any()


@pytest.mark.statistical
def test_trim_is_uniform():
    """each edge of a hot row survives with probability cap / degree"""
    from node2vec_amd.fugue import trim_hotspot_edges

    src = torch.zeros(200, dtype=torch.int64, device="cuda")
    hits = np.zeros(200)
    for seed in range(400):
        hits += trim_hotspot_edges(src, 50, seed).cpu().numpy()
    exp = 400 * 50 / 200
    chi2 = ((hits - exp) ** 2 / (exp * (1 - 0.25))).sum()
    assert chi2 < 300  # 199 dof, mean 199, sd 20


def test_fast_mode_first_step_equals_exact_mode():
    from node2vec_amd import randomwalk as rw

    g = _rand_graph(3, 1500, 20000, True)
    start = rw.start_vertices(g)
    a, va = rw.walk(g, start, 4, 1, 0.5, 2.0, 77, mode="exact")
    b, vb = rw.walk(g, start, 4, 1, 0.5, 2.0, 77, mode="fast")
    assert torch.equal(a, b) and torch.equal(va, vb)


@pytest.mark.statistical
@pytest.mark.parametrize("pq", [(0.5, 2.0), (4.0, 0.25), (1.0, 1.0), (1.0, 3.0)])
def test_fast_mode_transition_distribution(oracle, pq):
    """chi-square of the second step given (s, v) against the exact transition
    probabilities of the reference's bias rule (G6), on weighted karate."""
    from node2vec_amd import randomwalk as rw
    from node2vec_amd.graph import DeviceGraph

    p, q = pq
    rng = np.random.default_rng(1)
    edges = load_golden("karate_edges.json")
    und = {}
    for a, b, _ in edges:
        und[(min(a, b), max(a, b))] = float(np.float32(rng.uniform(0.3, 2.0)))
    src = [e[0] for e in edges]
    dst = [e[1] for e in edges]
    w = [und[(min(a, b), max(a, b))] for a, b, _ in edges]
    g = DeviceGraph.from_edges(src, dst, w, device="cuda")
    rowptr, col, ww = g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.w.cpu().numpy()
    nw = 30000
    walks, valid = rw.walk(g, torch.arange(34, dtype=torch.int32), nw, 2, p, q, 4242, mode="fast")
    assert bool(valid.all())
    wk = walks.cpu().numpy()
    chi2, dof = 0.0, 0
    for s in range(34):
        sub = wk[wk[:, 0] == s]
        for v in np.unique(sub[:, 1]):
            nxt = sub[sub[:, 1] == v][:, 2]
            if len(nxt) < 400:
                continue
            pr = oracle.transition_probs(rowptr, col, ww, s, int(v), p, q)
            nb = col[rowptr[v]:rowptr[v + 1]]
            obs = np.array([(nxt == x).sum() for x in nb], float)
            exp = pr * len(nxt)
            ok = exp >= 5
            chi2 += (((obs - exp) ** 2) / np.maximum(exp, 1e-12))[ok].sum()
            dof += int(ok.sum()) - 1
    assert dof > 300
    z = (chi2 - dof) / np.sqrt(2 * dof)
    assert z < 4.5, (chi2, dof, z)


def test_fast_mode_sinks_and_independence_of_sharding():
    from node2vec_amd import randomwalk as rw

    g = _rand_graph(9, 1200, 6000, False, hubs=2)  # sparse: has sinks
    start = torch.arange(g.n_vertices, dtype=torch.int32)
    walks, valid = rw.walk(g, start, 2, 15, 0.5, 2.0, 5, mode="fast")
    deg = g.degrees()
    assert bool((~valid).any()) and bool(valid.any())
    ok = walks[valid].long()
    assert bool((deg[ok[:, :-1]] > 0).all())  # every step left a vertex with out-edges
    # every hop is an edge of the graph
    key = set((g.rowptr.cpu().numpy().searchsorted(np.arange(g.n_edges), side="right") - 1) * 10000
              + g.col.cpu().numpy())
    hop = (ok[:, :-1] * 10000 + ok[:, 1:]).cpu().numpy().reshape(-1)
    assert all(int(h) in key for h in hop[:5000])
    parts = [rw.walk(g, c, 2, 15, 0.5, 2.0, 5, mode="fast") for c in torch.chunk(start, 4)]
    assert torch.equal(torch.cat([x[0] for x in parts]), walks)


@pytest.mark.parametrize("pq", [(0.5, 2.0), (4.0, 0.25), (2.0, 0.5)])
def test_fast_mode_pivot_index_changes_no_decision(pq):
    """the block-end search index (n2v_pivots_build) answers "x in N(s)" exactly like the
    plain binary search: fast walks with and without it are identical, on rows that start
    and end anywhere inside the 32-entry blocks, with multi-edges and hubs"""
    import torch

    from node2vec_amd import randomwalk as rw
    from node2vec_amd.graph import DeviceGraph

    rng = np.random.default_rng(11)
    nv = 5000
    src = np.concatenate([rng.integers(0, nv, 60000), np.full(4000, 7), rng.integers(0, nv, 4000),
                          np.full(900, 123), rng.integers(0, 50, 900)])
    dst = np.concatenate([rng.integers(0, nv, 60000), rng.integers(0, nv, 4000), np.full(4000, 7),
                          rng.integers(0, 50, 900), np.full(900, 123)])
    g = DeviceGraph.from_edges(src, dst, np.ones(src.size, np.float32), n_vertices=nv, device="cuda")
    g.build_alias()
    piv = g.pivots.cpu().numpy()
    col = g.col.cpu().numpy()
    assert np.array_equal(piv, col[np.minimum(np.arange(piv.size) * 32 + 31, col.size - 1)])
    start = rw.start_vertices(g)
    with_index = rw.walk(g, start, 4, 40, pq[0], pq[1], 5, mode="fast")
    g.pivots = None
    without = rw.walk(g, start, 4, 40, pq[0], pq[1], 5, mode="fast")
    assert torch.equal(with_index[0], without[0]) and torch.equal(with_index[1], without[1])
