"""Exact biased walks over LONG shared-position lists against the oracle.  The lists are those of `x in src_nbs_id`
(randomwalk.py:226) stored once per edge; on graphs trimmed at the reference's own cap (constants.py:6: 100 000)
hub-to-hub lists hold thousands of entries (cfg 4: up to 9 774, 2.75 M lists above 2 048), which the random graphs of
the other tests never reach: the 8-ary cuts of a search, the pivots of a slot, a row sum over thousands of runs."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PQ = [(0.5, 2.0), (4.0, 0.25), (4.0, 2.0), (0.25, 0.5), (3.0, 0.7), (0.7, 3.0)]


def _hub_graph(n_hubs, n_common, nv, extra, seed):
    """`n_hubs` hubs that share `n_common` neighbours (hub-to-hub lists of ~n_common entries) inside a sparse
    random graph; symmetrised, de-duplicated"""
    from node2vec_amd.graph import DeviceGraph

    rng = np.random.default_rng(seed)
    common = rng.choice(np.arange(n_hubs, nv), n_common, replace=False)
    src = [np.repeat(np.arange(n_hubs), n_common)]
    dst = [np.tile(common, n_hubs)]
    # every hub also has neighbours of its own, and the hubs are neighbours of one another
    for h in range(n_hubs):
        own = rng.choice(np.arange(n_hubs, nv), n_common // 3, replace=False)
        src.append(np.full(own.size, h))
        dst.append(own)
    hh = np.array([(a, b) for a in range(n_hubs) for b in range(n_hubs) if a != b])
    src += [hh[:, 0], rng.integers(0, nv, extra)]
    dst += [hh[:, 1], rng.integers(0, nv, extra)]
    src, dst = np.concatenate(src), np.concatenate(dst)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    key = np.unique(np.concatenate([src * nv + dst, dst * nv + src]))
    return DeviceGraph.from_edges(key // nv, key % nv, None, n_vertices=nv, device="cuda")


@pytest.mark.parametrize("wide_from,n_common", [(None, 2600), (None, 21_000), (3000, 21_000), (3000, 2600)])
def test_walks_over_hub_to_hub_lists_equal_the_oracle(wide_from, n_common):
    """hub-to-hub lists of ~2 600 and of ~21 000 entries, as 16-bit lists behind wedge slots and as the lists of
    wide rows: every kernel that reads the tables, every arrangement of the closed forms, values that are not
    dyadic -- the oracle's walks on a sample that starts on the hubs and their neighbours, and the kernels against
    one another over every start vertex and on walkers that stay among the hubs"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import n2v_oracle
    from node2vec_amd import randomwalk as rw

    n_hubs = 6
    g = _hub_graph(n_hubs, n_common, 3 * n_common + 2000, 8 * n_common, 11 + n_common)
    g.build_wedges(wide_from=wide_from)
    g.wedge_tried = True
    assert g.wedge_slots is not None
    assert int((g.edge_classes & 0xffffff).max()) >= n_common
    start = rw.start_vertices(g)
    rowptr, col = g.rowptr.cpu().numpy(), g.col.cpu().numpy()
    hubs = np.arange(n_hubs)
    sample = np.unique(np.concatenate([hubs, col[rowptr[0]:rowptr[0] + 150], col[rowptr[1]:rowptr[1] + 150]]))
    sample_t = torch.as_tensor(sample, dtype=torch.int32, device="cuda")
    for p, q in PQ:
        want, wv = n2v_oracle.random_walk(rowptr, col, None, sample, 3, 10, p, q, 5, n_threads=8)
        for kw in ({}, {"use_wedge_slots": False}):
            got, gv = rw.walk(g, sample_t, 3, 10, p, q, 5, **kw)
            assert np.array_equal(gv.cpu().numpy(), wv), (p, q, kw)
            assert np.array_equal(got.cpu().numpy(), want), (p, q, kw)
        a, av = rw.walk(g, start, 2, 16, p, q, 9)
        b, bv = rw.walk(g, start, 2, 16, p, q, 9, use_wedge_slots=False)
        assert torch.equal(av, bv) and torch.equal(a, b), (p, q)
        if (p, q) in ((3.0, 0.7), (0.7, 3.0)):
            # values that are not dyadic: the row sums of the steps into the hubs' rows were computed once
            # (n2v_edge_row_sums_build) -- the same walks with every such row added up by the lane that needs it,
            # and the stored sums equal Python's left-to-right float sum of the step's table (randomwalk.py:172)
            assert g.row_sums is not None and g.row_sums[1:3] == (p, q) and g.c_struct().row_sums
            c, cv = rw.walk(g, start, 2, 16, p, q, 9, use_row_sums=False)
            assert torch.equal(av, cv) and torch.equal(a, c), (p, q)
            sums = g.row_sums[0].cpu().numpy()
            checked = 0
            sources = [0, 1] + [int(x) for x in col[rowptr[0]:rowptr[0] + 400:100]]  # two hubs, some of their neighbours
            for s_ in sources:
                mine = col[rowptr[s_]:rowptr[s_ + 1]]
                nbr = set(mine.tolist())
                for k, v in enumerate(mine.tolist()):
                    if v >= n_hubs:  # (the rows of the hubs are the long ones)
                        continue
                    e = int(rowptr[s_]) + k
                    assert rowptr[v + 1] - rowptr[v] >= g.ROW_SUMS_FROM
                    row = col[rowptr[v]:rowptr[v + 1]].tolist()
                    assert sums[e] == sum((1.0 / p if x == s_ else (1.0 if x in nbr else 1.0 / q)) for x in row), (s_, v)
                    checked += 1
            assert checked >= 5
        # walkers that start on a hub and stay among the hubs' rows: most steps search a long list
        a, _ = rw.walk(g, sample_t[:n_hubs], 40, 30, p, q, 13)
        b, _ = rw.walk(g, sample_t[:n_hubs], 40, 30, p, q, 13, use_wedge_slots=False)
        c, _ = rw.walk(g, sample_t[:n_hubs], 40, 30, p, q, 13, use_wedge_kernel=False)
        assert torch.equal(a, b) and torch.equal(a, c), (p, q)
