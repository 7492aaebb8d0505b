"""n2v_wedge_build: the per-edge shared-position lists equal the set intersection the reference
computes at every step (randomwalk.py:318, :226) -- checked edge by edge against plain Python on
graphs with multi-edges, self-loops, sinks and hubs (lane path and wave path, both search
directions), in the 16-bit and the 32-bit position form."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _expected(rowptr, col):
    rows = [col[rowptr[v]:rowptr[v + 1]].tolist() for v in range(len(rowptr) - 1)]
    sets = [set(r) for r in rows]
    pos, rpos, cnt = [], [], []
    for s, row in enumerate(rows):
        for v in row:
            nv = rows[v]
            lst = [j for j, x in enumerate(nv) if x != s and x in sets[s]]
            r = [j for j, x in enumerate(nv) if x == s]
            pos.append(lst)
            rpos.append(r[0] if r else 0)
            cnt.append(len(r))
    return pos, rpos, cnt


@pytest.mark.parametrize("wide", [False, True])
def test_wedge_lists_equal_the_per_step_set_intersection(wide):
    from node2vec_amd.graph import DeviceGraph

    rng = np.random.default_rng(17)
    nv = 700
    src = np.concatenate([rng.integers(0, nv - 20, 9000), rng.integers(0, 6, 2500), rng.integers(0, nv - 20, 2500),
                          rng.integers(0, 40, 300)])
    dst = np.concatenate([rng.integers(0, nv, 9000), rng.integers(0, nv, 2500), rng.integers(0, 6, 2500),
                          rng.integers(0, 40, 300)])  # 6 hubs (wave path), multi-edges, self-loops
    g = DeviceGraph.from_edges(src, dst, None, n_vertices=nv, device="cuda")
    g.build_wedges(wide=wide)
    assert g.wedge_pos.dtype == (torch.int32 if wide else torch.int16)
    rowptr, col = g.rowptr.cpu().numpy(), g.col.cpu().numpy()
    want_pos, want_rpos, want_nr = _expected(rowptr, col)
    ec = g.edge_classes.cpu().numpy().astype(np.uint32)
    off = g.wedge_off.cpu().numpy().astype(np.uint64)
    pos = g.wedge_pos.cpu().numpy()
    pos = pos.astype(np.uint16 if pos.dtype == np.int16 else np.uint32).astype(np.int64)
    assert len(want_pos) == g.n_edges
    big = 0
    for e in range(g.n_edges):
        n_shared, n_ret = int(ec[e] & 0xffffff), int(ec[e] >> 24)
        o, rp = int(off[e] & np.uint64(0xffffffffff)), int(off[e] >> np.uint64(40))
        assert n_shared == len(want_pos[e]) and n_ret == min(want_nr[e], 255)
        assert pos[o:o + n_shared].tolist() == want_pos[e], e
        if n_ret:
            assert rp == want_rpos[e], e
        big += n_shared > 24
    assert big > 50  # the wave path was exercised
    assert int(off[-1] & np.uint64(0xffffffffff)) + int(ec[-1] & 0xffffff) == sum(len(p) for p in want_pos)


def test_wedge_table_respects_its_memory_bound():
    from node2vec_amd import synthetic

    g = synthetic.rmat(14, 200_000, device="cuda")
    g.build_wedges(max_bytes=1000)
    assert g.wedge_off is None and g.wedge_pos is None and g.edge_classes is not None
    g.build_wedges()
    assert g.wedge_off is not None
