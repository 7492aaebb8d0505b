"""Adversarial rows through the HIP kernels: rows whose INTERNAL comparisons -- X_i - k D of the closed forms with
margins (n2v_unit_near.h), E_k - D_j / probs[pick] - 1 of the weighted decision (n2v_walk_wlanes.hip) -- were placed
at 0.01 ... 10 x the margin by exact rational arithmetic (scripts/models/margin_adversary.py), embedded in a graph
and walked.  The Python models of the two procedures pass the same rows on CPU (tests/test_closed_form_models.py);
this is the check that the KERNELS decide as their models do where random graphs never go.

Each row becomes three kinds of vertices: v, whose out-edges are the row (slot i -> vertex base + i; the return run
-> the vertex s, parallel edges); s, with edges to v (parallel ones: half of its walkers go there first)
and to the shared slots' vertices; sinks.  A walk of two steps from s stands on v with prev = s: its second step is a
draw from the table generate_edge_alias_tables builds for (s, N(s), N(v)) (randomwalk.py:193-232).  That table comes
from the oracle once per row; every walker's uniforms from the RNG restated below (pinned by rng_kat.json through
oracle.uniform_bits): the expected vertex of every walker in O(1)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts", "models"))

M64 = (1 << 64) - 1


def _mix64(z):
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _uniform_bits(seed, keys, step):
    """(u1, u2) of every walker key at `step`, numpy uint64 wrap-around arithmetic"""
    with np.errstate(over="ignore"):
        keys = keys.astype(np.uint64)
        h0 = _mix64(np.uint64(seed) ^ _mix64(keys + np.uint64(0x9E3779B97F4A7C15)))
        bits = _mix64(h0 + np.uint64(step + 1) * np.uint64(0xD1B54A32D192ED03))
    return (bits >> np.uint64(32)).astype(np.int64), (bits & np.uint64(0xFFFFFFFF)).astype(np.int64)


def test_rng_restatement_equals_the_oracle(oracle):
    keys = np.array([0, 1, 7, 12345, 2 ** 40 + 3, 2 ** 63 + 11], dtype=np.uint64)
    for seed in (0, 42, 2 ** 64 - 5):
        for step in (0, 1, 79):
            u1, u2 = _uniform_bits(seed, keys, step)
            for k, a, b in zip(keys.tolist(), u1.tolist(), u2.tolist()):
                assert oracle.uniform_bits(seed, k, step) == (a, b)


class _Rows:
    """adversarial rows laid out in one graph"""

    def __init__(self):
        self.src, self.dst, self.w, self.rows, self.base = [], [], [], [], 0

    def add(self, ids_in_row, weights, shared, rpos, info):
        """ids_in_row: slot -> offset inside the row's id block (return slots all = rpos); weights: per slot or None"""
        n = len(ids_in_row)
        base = self.base
        v, s = base + n, base + rpos
        ids = base + np.asarray(ids_in_row, dtype=np.int64)
        self.src.append(np.full(n, v))
        self.dst.append(ids)
        self.w.append(np.ones(n) if weights is None else np.asarray(weights, dtype=np.float64))
        sh = base + np.asarray(shared, dtype=np.int64)
        # as many parallel edges s -> v as s has other edges: half of the walkers go to v first.  (Parallel edges, not a
        # heavy one: the largest weight of the GRAPH enters the kernel's test for an exact row sum.)
        # (at most 200: the class word of the edge v -> s counts them in 8 bits, and 255 means "no lists")
        m = min(max(1, len(shared)), 200)
        self.src.append(np.full(m + len(sh), s))
        self.dst.append(np.concatenate([np.full(m, v), sh]))
        self.w.append(np.ones(m + len(sh)))
        self.rows.append(dict(info, v=v, s=s, ids=ids.astype(np.int32), n=n, shared=sh, to_v=m / (m + len(sh))))
        self.base = v + 1

    def graph(self, weighted):
        from node2vec_amd.graph import DeviceGraph

        src, dst = np.concatenate(self.src), np.concatenate(self.dst)
        w = np.concatenate(self.w) if weighted else None
        return DeviceGraph.from_edges(src, dst, w, n_vertices=self.base, device="cuda")


def _check_rows(oracle, g, rows, W, p, q, seed, walk_kwargs):
    """two steps from every s of `rows` (all with this p, q); every walker that stands on v after the first: its
    second vertex against the oracle's table.  Returns (walkers checked, hits on the placed slots)"""
    from node2vec_amd import randomwalk as rw

    start = torch.tensor(sorted(r["s"] for r in rows), dtype=torch.int32, device="cuda")
    walks, valid = rw.walk(g, start, W, 2, p, q, seed, mode="exact", **walk_kwargs)
    walks = walks.cpu().numpy().reshape(len(rows), W, 3)
    valid = valid.cpu().numpy().reshape(len(rows), W)
    checked = hits = 0
    rowptr, col, wts = g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.w.cpu().numpy()
    for k, r in enumerate(sorted(rows, key=lambda r: r["s"])):
        s, v, n = r["s"], r["v"], r["n"]
        nbs = col[rowptr[s]:rowptr[s + 1]]
        lo, hi = rowptr[v], rowptr[v + 1]
        assert hi - lo == n and np.array_equal(col[lo:hi], r["ids"])
        alias, probs = oracle.edge_alias_tables(s, nbs.tolist(), col[lo:hi], wts[lo:hi].astype(np.float64), p, q)
        alias, probs = np.asarray(alias), np.asarray(probs)
        at_v = walks[k, :, 1] == v
        assert at_v.sum() > 0.5 * W * r["to_v"]  # (the share of the edges of s that lead to v)
        ords = np.nonzero(at_v)[0]
        u1, u2 = _uniform_bits(seed, np.uint64(s) * np.uint64(W) + ords.astype(np.uint64), 1)
        pick = (u1 * n) >> 32
        r2 = u2 / 4294967296.0
        want = r["ids"][np.where(r2 < probs[pick], pick, alias[pick])]
        got = walks[k, ords, 2]
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (r["info"], "walker", int(ords[bad[0]]), "pick", int(pick[bad[0]]), "r2", float(r2[bad[0]]),
                               "got", int(got[bad[0]]), "want", int(want[bad[0]]))
        assert valid[k, ords].all() or True  # (the second vertex may be a sink: the walk then ends, fugue.py:147)
        checked += ords.size
        hits += int(np.isin(pick, np.asarray(r["slots"])).sum())
    return checked, hits


def _near_rows(n_max, per, seed):
    import margin_adversary as A

    got = []
    tot = A.attack_near(n_max, per, seed, out=lambda *_: None, collect=got)
    assert tot[3] == 0  # (the MODEL passes them: tests/test_closed_form_models.py)
    return got


def _weighted_rows(n_max, per, seed):
    import margin_adversary as A

    got = []
    tot = A.attack_weighted(n_max, per, seed, out=lambda *_: None, collect=got)
    assert tot[3] == 0
    return got


def _walkers_for(rows):
    # ~6 walkers on v per slot of the longest row (the placed comparison decides the draws of two or three slots)
    return int(min(max(max(6 * r["n"] / r["to_v"] for r in rows), 4096), 1 << 20))


@pytest.mark.parametrize("slots_kernel", [True, False])
def test_closed_forms_with_margins_on_rows_placed_at_the_margin(oracle, slots_kernel):
    """unit weights, 1/p or 1/q not dyadic: walk_exact_wedge_slots_kernel<2> (and, without the slots, the same forms
    through wedge_off) on rows of 8 ... 10^4 slots in all five class arrangements, the comparison nearest to a tie at
    +-{0.01 ... 10} x the margin of the counts stage or of the exact-sum stage"""
    placed = _near_rows(10_000, 1, 5 if slots_kernel else 6)
    assert len(placed) > 250 and len({r["arr"] for r in placed}) == 5
    L = _Rows()
    for r in placed:
        cls = r["cls"]
        rp = cls.index('R')
        ids = [rp if c == 'R' else i for i, c in enumerate(cls)]
        L.add(ids, None, [i for i, c in enumerate(cls) if c == 'M'], rp,
              dict(info=(r["n"], r["arr"], r["p"], r["q"], r["margin"], round(r["multiple"], 3)), slots=r["slots"],
                   p=r["p"], q=r["q"]))
    g = L.graph(weighted=False)
    assert g.unit_weights
    checked = hits = 0
    for r in L.rows:  # q is the knob of the placement: every row has its own
        c, h = _check_rows(oracle, g, [r], _walkers_for([r]), r["p"], r["q"], 77,
                           {} if slots_kernel else {"use_wedge_slots": False})
        checked, hits = checked + c, hits + h
    assert g.wedge_off is not None and g.wedge_slots is not None  # the kernels with the closed forms did run
    print(f"near rows {len(L.rows)}: {checked} walkers checked on the adversarial rows, {hits} draws of placed slots")
    assert hits > 4 * len(L.rows)


def test_weighted_decision_with_margins_on_rows_placed_at_the_margin(oracle):
    """weighted graphs: n2v_walk_weighted_step (lane kernel below 768 slots, wave kernel and block summaries above,
    second chance, exact kernel) on fp32 / fp64 / 24-decade / integer rows of 8 ... 10^4 slots whose crossing
    E_k - D_j or probs[pick] - 1 sits at +-{0.01 ... 10} x the margin"""
    placed = _weighted_rows(10_000, 1, 9)
    assert len(placed) > 150
    by_size = {}
    for r in placed:
        by_size.setdefault((r["n"], r["kind"] in ("fp32", "integers")), []).append(r)
    checked = hits = undecided = 0
    for (n, grid32), rows in sorted(by_size.items()):
        # one graph per row size and weight grid: the power of two that divides every stored weight is a property of
        # the GRAPH (row_sums[n_vertices]) -- rows of one size share the binade of their small weights
        L = _Rows()
        for r in rows:
            cls = np.asarray(r["cls"])
            rp = int(np.nonzero(cls == 2)[0][0])
            L.add(list(range(n)), r["w"], np.nonzero(cls == 1)[0].tolist(), rp,
                  dict(info=(n, r["kind"], r["where"], r["p"], r["q"], r["exact"], round(r["multiple"], 3)),
                       slots=r["slots"], p=r["p"], q=r["q"]))
        g = L.graph(weighted=True)
        assert not g.unit_weights
        groups = {}
        for r in L.rows:
            groups.setdefault((r["p"], r["q"]), []).append(r)
        for (p, q), rs in sorted(groups.items()):
            st = {}
            c, h = _check_rows(oracle, g, rs, _walkers_for(rs), p, q, 78, {"use_weighted_lanes": True, "stats": st})
            checked, hits = checked + c, hits + h
            assert "undecided" in st  # the step kernels with margins ran
            undecided += int(st["undecided"].item()) if torch.is_tensor(st["undecided"]) else int(st["undecided"])
    print(f"weighted rows {len(placed)}: {checked} walkers checked, {hits} draws of placed slots, "
          f"{undecided} walker-steps left to the exact kernel")
    assert hits > 4 * len(placed)
