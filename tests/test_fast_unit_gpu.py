"""Distribution parity of the UNIT-WEIGHT rejection sampler (walk_fast_kernel<true, true> with the
hop table, <true, false> without) against the reference's bias rule, generate_edge_alias_tables
(randomwalk.py:219-231): x == src -> w/p, x in N(src) -> w, else w/q.

Round 2's kernel folded the return edge out of the rejection envelope with the share
(nR/p) / (nR/p + (n - nR) b') while drawing the other candidates over ALL n entries: the return
edge came out over-weighted by n / (n - nR) (degree 2: P(return) 0.89 instead of 0.80 at p = 0.5,
q = 2).  These tests are the ones that catch it: unit weights, low-degree vertices, multi-edges
(nR > 1), triangles, with and without every optional table.  Expected values: the oracle's
exact transition probabilities (G6).  Test statistic: chi-square over all (s, v) contexts with
>= 400 samples, cells with expectation >= 5, as a z-score of the summed statistic; and the
return probability alone, per degree, within 5 binomial sigma.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = [pytest.mark.gpu, pytest.mark.statistical]


def _sym(pairs):
    e = []
    for a, b in pairs:
        if a != b:
            e.append((a, b))
            e.append((b, a))
    return e


def _low_degree_graph():
    """a 40-cycle (degree 2), 12 of its vertices joined by chords (degree 3), two pendant
    vertices (degree 1: the only move is the return) and two triangles hung on the cycle"""
    e = [(i, (i + 1) % 40) for i in range(40)]
    e += [(i, i + 20) for i in range(0, 12, 2)]
    e += [(5, 40), (17, 41)]
    e += [(30, 42), (42, 43), (43, 30), (33, 44), (44, 34)]
    return sorted(set(_sym(e)))


def _multi_edge_graph():
    """random 30-vertex graph whose undirected edges are repeated 1..3 times: nR in 1..3"""
    rng = np.random.default_rng(5)
    e = []
    for a in range(30):
        for b in rng.choice(30, 4, replace=False):
            if a != int(b):
                e += [(a, int(b))] * int(rng.integers(1, 4))
    return sorted(_sym(e))  # duplicates kept


def _triangle_rich_graph():
    """six 5-cliques in a ring + karate club shifted behind them"""
    e = []
    for c in range(6):
        base = 5 * c
        e += [(base + i, base + j) for i in range(5) for j in range(i + 1, 5)]
        e.append((base + 4, (base + 5) % 30))
    e = _sym(e)
    e += [(30 + int(a), 30 + int(b)) for a, b, _ in load_golden("karate_edges.json")]
    return sorted(set(e))


GRAPHS = {"low_degree": _low_degree_graph, "multi_edge": _multi_edge_graph,
          "triangle_rich": _triangle_rich_graph}
# tables: (edge classes, hop table, wedge table)
TABLES = {"hops+classes": (True, True, False), "classes": (True, False, False),
          "hops+classes+wedges": (True, True, True), "classes+wedges": (True, False, True),
          "bare": (False, False, False)}
# (with the wedge table the kernel draws from the LAYERS of a step's table -- n2v_walk_fast.hip,
#  kClassFirst --, without it candidates are rejected: two samplers, one distribution)


def _graph(name):
    from node2vec_amd.graph import DeviceGraph

    e = np.array(GRAPHS[name](), dtype=np.int64)
    g = DeviceGraph.from_edges(e[:, 0], e[:, 1], None, device="cuda")
    assert g.unit_weights
    return g


def _second_steps(g, p, q, tables, nw, seed):
    from node2vec_amd import randomwalk as rw

    classes, hops, wedges = tables
    if wedges:
        g.build_edge_classes()
        g.build_wedges()
        assert g.wedge_off is not None
    walks, valid = rw.walk(g, rw.start_vertices(g), nw, 2, p, q, seed, mode="fast",
                           use_edge_classes=classes, use_hops=hops, use_wedges=wedges)
    assert bool(valid.all())
    return walks.cpu().numpy()


@pytest.mark.parametrize("tables", list(TABLES))
@pytest.mark.parametrize("pq", [(0.5, 2.0), (0.25, 4.0), (0.5, 0.25)])
@pytest.mark.parametrize("name", list(GRAPHS))
def test_unit_fast_transition_distribution(oracle, name, pq, tables):
    p, q = pq
    g = _graph(name)
    rowptr, col = g.rowptr.cpu().numpy(), g.col.cpu().numpy()
    ones = np.ones(col.size, np.float32)
    wk = _second_steps(g, p, q, TABLES[tables], 20000, 991)
    chi2, dof = 0.0, 0
    ret_obs, ret_exp, ret_var = {}, {}, {}
    key = wk[:, 0].astype(np.int64) * g.n_vertices + wk[:, 1]
    order = np.argsort(key, kind="stable")
    key, nxt_all = key[order], wk[order, 2]
    cuts = np.flatnonzero(np.diff(key)) + 1
    for lo, hi in zip(np.r_[0, cuts], np.r_[cuts, key.size]):
        if hi - lo < 400:
            continue
        s, v = divmod(int(key[lo]), g.n_vertices)
        nxt = nxt_all[lo:hi]
        pr = oracle.transition_probs(rowptr, col, ones, s, v, p, q)
        nb = col[rowptr[v]:rowptr[v + 1]]
        ids, inv = np.unique(nb, return_inverse=True)  # multi-edges: one cell per vertex
        pv = np.bincount(inv, weights=pr)
        obs = np.array([(nxt == x).sum() for x in ids], float)
        assert obs.sum() == nxt.size  # every second step is a neighbour of v
        exp = pv * nxt.size
        ok = exp >= 5
        if ok.sum() >= 2:
            chi2 += (((obs - exp) ** 2) / np.maximum(exp, 1e-12))[ok].sum()
            dof += int(ok.sum()) - 1
        d = int(nb.size)
        pret = float(pv[ids == s].sum())
        ret_obs[d] = ret_obs.get(d, 0) + int((nxt == s).sum())
        ret_exp[d] = ret_exp.get(d, 0.0) + pret * nxt.size
        ret_var[d] = ret_var.get(d, 0.0) + pret * (1 - pret) * nxt.size
    assert dof > 60, dof
    z = (chi2 - dof) / np.sqrt(2 * dof)
    assert z < 4.5, (name, pq, tables, chi2, dof, z)
    for d in ret_obs:  # the return edge alone, per degree of v (what round 2 got wrong)
        if ret_var[d] > 0:
            zr = (ret_obs[d] - ret_exp[d]) / np.sqrt(ret_var[d])
            assert abs(zr) < 5.0, (name, pq, tables, "degree", d, ret_obs[d], ret_exp[d], zr)


def test_unit_fast_return_share_at_degree_two():
    """the smallest case by hand: on a cycle every vertex has degree 2 and no triangles, so at
    p = 0.5, q = 2: P(return) = (1/p) / (1/p + 1/q) = 0.8 (round 2's kernel gave 0.89)"""
    from node2vec_amd import randomwalk as rw
    from node2vec_amd.graph import DeviceGraph

    e = np.array(_sym([(i, (i + 1) % 64) for i in range(64)]), dtype=np.int64)
    g = DeviceGraph.from_edges(e[:, 0], e[:, 1], None, device="cuda")
    for hops in (True, False):
        walks, _ = rw.walk(g, rw.start_vertices(g), 4000, 2, 0.5, 2.0, 17, mode="fast", use_hops=hops)
        wk = walks.cpu().numpy()
        share = float((wk[:, 2] == wk[:, 0]).mean())
        n = wk.shape[0]
        assert abs(share - 0.8) < 5 * np.sqrt(0.8 * 0.2 / n), (hops, share)


def test_unit_fast_long_walks_match_exact_mode_statistically():
    """80-step fast and exact walks on the triangle-rich graph: for every context (prev, cur) the
    distribution of the next vertex is the same in both samples.  Per-context 2 x k contingency
    chi-squares (given the context the draws of a second-order chain are independent), summed."""
    from node2vec_amd import randomwalk as rw

    g = _graph("triangle_rich")
    start = rw.start_vertices(g)
    nv = g.n_vertices
    cnt = []
    for mode in ("exact", "fast"):
        walks, valid = rw.walk(g, start, 400, 80, 0.5, 2.0, 3, mode=mode)
        assert bool(valid.all())
        w = walks.long()
        tri = (w[:, :-2] * nv + w[:, 1:-1]) * nv + w[:, 2:]
        cnt.append(torch.bincount(tri.reshape(-1), minlength=nv ** 3).cpu().numpy()
                   .astype(float).reshape(nv * nv, nv))
    a, b = cnt
    chi2, dof = 0.0, 0
    for ctx in np.flatnonzero((a.sum(1) >= 200) & (b.sum(1) >= 200)):
        ra, rb = a[ctx], b[ctx]
        ok = (ra + rb) >= 20
        if ok.sum() < 2:
            continue
        ra, rb = ra[ok], rb[ok]
        na, nb_ = ra.sum(), rb.sum()
        ea, eb = (ra + rb) * na / (na + nb_), (ra + rb) * nb_ / (na + nb_)
        chi2 += (((ra - ea) ** 2) / ea + ((rb - eb) ** 2) / eb).sum()
        dof += int(ok.sum()) - 1
    z = (chi2 - dof) / np.sqrt(2 * dof)
    assert dof > 500 and z < 4.5, (chi2, dof, z)
