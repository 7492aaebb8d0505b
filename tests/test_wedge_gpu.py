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
    if wide:
        assert g.wedge_slots is None  # the slots hold 16-bit positions
        return
    # the wedge slots (n2v_wedge_slots_build) restate the same lists: return position, entries below
    # it, and the list itself (<= 14 entries) or its offset and eight pivots
    slots = g.wedge_slots.cpu().numpy().astype(np.uint16).astype(np.int64)
    assert slots.shape == (g.n_edges, 16)
    n_long = 0
    for e in range(g.n_edges):
        lst, n_ret = want_pos[e], int(ec[e] >> 24)
        rp = want_rpos[e] if n_ret else int(off[e] >> np.uint64(40))
        assert slots[e, 0] == rp and slots[e, 1] == sum(1 for x in lst if x < rp), e
        if len(lst) <= 14:
            assert slots[e, 2:2 + len(lst)].tolist() == lst, e
        else:
            o = int(slots[e, 4] | (slots[e, 5] << 16) | (slots[e, 6] << 32) | (slots[e, 7] << 48))
            assert o == int(off[e] & np.uint64(0xffffffffff))
            assert slots[e, 8:16].tolist() == [lst[((k + 1) * len(lst)) // 9] for k in range(8)], e
            n_long += 1
    assert n_long > 100


def test_mixed_wedge_table_widens_only_the_lists_into_wide_rows():
    """n2v_wedge_build with wide = T >= 2 (production 65536: the reference's own trim cap is 100 000,
    constants.py:6): the list of an edge into a row of fewer than T entries is uint16 as ever, the
    list of an edge into a row of T entries or more is uint32 and lies behind all the 16-bit ones;
    the wedge slots exist for the first kind only; the hop table inlines return positions for the
    first kind only.  Same lists, edge by edge, as the reference's per-step set intersection."""
    from node2vec_amd.graph import DeviceGraph

    rng = np.random.default_rng(19)
    nv = 700
    src = np.concatenate([rng.integers(0, nv - 20, 9000), rng.integers(0, 6, 2500), rng.integers(0, nv - 20, 2500),
                          rng.integers(0, 40, 300)])
    dst = np.concatenate([rng.integers(0, nv, 9000), rng.integers(0, nv, 2500), rng.integers(0, 6, 2500),
                          rng.integers(0, 40, 300)])
    g = DeviceGraph.from_edges(src, dst, None, n_vertices=nv, device="cuda")
    T = 40
    g.build_wedges(wide_from=T)
    assert g.wedge_mode == T and g.c_struct().wedge_wide == T and g.wedge_pos.dtype == torch.int16
    rowptr, col = g.rowptr.cpu().numpy(), g.col.cpu().numpy()
    deg = np.diff(rowptr)
    want_pos, want_rpos, want_nr = _expected(rowptr, col)
    ec = g.edge_classes.cpu().numpy().astype(np.uint32)
    off = g.wedge_off.cpu().numpy().astype(np.uint64)
    pos16 = g.wedge_pos.cpu().numpy().astype(np.uint16)
    pos32 = pos16.view(np.uint32).astype(np.int64)
    pos16 = pos16.astype(np.int64)
    wide_e = deg[col] >= T
    assert 500 < wide_e.sum() < g.n_edges - 500
    n16 = sum(len(want_pos[e]) for e in range(g.n_edges) if not wide_e[e])
    lo32 = (n16 + 1) // 2
    for e in range(g.n_edges):
        n_shared, n_ret = int(ec[e] & 0xffffff), int(ec[e] >> 24)
        o, rp = int(off[e] & np.uint64(0xffffffffff)), int(off[e] >> np.uint64(40))
        assert n_shared == len(want_pos[e])
        if wide_e[e]:
            assert o >= lo32 and pos32[o:o + n_shared].tolist() == want_pos[e], e
        else:
            assert o + n_shared <= n16 and pos16[o:o + n_shared].tolist() == want_pos[e], e
        if n_ret:
            assert rp == want_rpos[e], e
    slots = g.wedge_slots.cpu().numpy().astype(np.uint16).astype(np.int64)
    # the edges into wide rows have FOLDED slots (n2v_wedge_slots_fold, round 6): a position below T as it is, one
    # from T on minus T; `nlow` entries lie below T; lists of more than 14 entries have a folded 16-bit copy behind
    # the 32-bit lists, and its offset and pivots in the slot
    assert g.slots_folded and (g.c_struct().reserved2 & 2)
    fold = lambda v: v if v < T else v - T
    n_long_folded = 0
    for e in np.nonzero(wide_e)[0]:
        lst, n_ret = want_pos[e], int(ec[e] >> 24)
        rp = want_rpos[e] if n_ret else int(off[e] >> np.uint64(40))
        below, nlow, upper = sum(1 for x in lst if x < rp), sum(1 for x in lst if x < T), int(rp >= T)
        assert slots[e, 0] == fold(rp), e
        if len(lst) <= 14:
            assert slots[e, 1] == below | nlow << 4 | upper << 8, e
            assert slots[e, 2:2 + len(lst)].tolist() == [fold(x) for x in lst], e
        else:
            assert slots[e, 1] == below & 0xffff and slots[e, 2] == nlow & 0xffff, e
            assert slots[e, 3] == upper | (below >> 16) << 4 | (nlow >> 16) << 8, e
            o = int(slots[e, 4] | (slots[e, 5] << 16) | (slots[e, 6] << 32) | (slots[e, 7] << 48))
            assert o >= 2 * lo32 and pos16[o:o + len(lst)].tolist() == [fold(x) for x in lst], e
            assert slots[e, 8:16].tolist() == [fold(lst[((k + 1) * len(lst)) // 9]) for k in range(8)], e
            n_long_folded += 1
    assert n_long_folded > 100
    # fold=False: the mixed table as rounds 4 - 5 built it -- no slot for an edge into a wide row, and the walks go
    # through the kernel that reads wedge_off
    g2 = DeviceGraph.from_edges(src, dst, None, n_vertices=nv, device="cuda")
    g2.build_wedges(wide_from=T, fold=False)
    assert not g2.slots_folded and not (g2.c_struct().reserved2 & 2)
    assert not g2.wedge_slots.cpu().numpy()[wide_e].any()
    for e in np.nonzero(~wide_e)[0][::7]:
        lst, n_ret = want_pos[e], int(ec[e] >> 24)
        rp = want_rpos[e] if n_ret else int(off[e] >> np.uint64(40))
        assert slots[e, 0] == rp and slots[e, 1] == sum(1 for x in lst if x < rp), e
        if len(lst) <= 14:
            assert slots[e, 2:2 + len(lst)].tolist() == lst, e
    assert g.can_inline_rpos()
    g.build_hops(inline_rpos=True)
    cls = g.hops.cpu().numpy()[:, 1].astype(np.uint32)
    inl = (cls & np.uint32(0x80000000)) != 0
    assert inl.any() and not inl[wide_e].any()
    assert np.array_equal(cls[wide_e], ec[wide_e])
    assert np.array_equal(inl[~wide_e], (ec[~wide_e] & 0xffffff) == 0)


PQ_MIXED = [(0.5, 2.0), (4.0, 0.25), (4.0, 2.0), (0.25, 0.5), (3.0, 0.7), (0.7, 3.0), (2.0, 1.0), (0.5, 0.5)]


@pytest.mark.parametrize("wide_from", [2, 24, 200])
def test_walks_over_a_mixed_wedge_table_equal_the_oracle(wide_from):
    """Exact biased walks on a graph whose wedge table is mixed (some rows "wide"): the slots kernel
    (steps on wide rows go through wedge_off with 32-bit lists, per lane), the kernel without slots,
    the class-count kernel and the passes over a workspace all give the oracle's walks, for every
    arrangement of the closed forms and for values that are not dyadic."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import n2v_oracle
    from node2vec_amd import randomwalk as rw
    from node2vec_amd.graph import DeviceGraph

    rng = np.random.default_rng(100 + wide_from)
    nv = 1500
    src = np.concatenate([rng.integers(0, nv, 9000), rng.integers(0, 8, 4000), rng.integers(0, 60, 2000)])
    dst = np.concatenate([rng.integers(0, nv, 9000), rng.integers(0, nv, 4000), rng.integers(0, 60, 2000)])
    keep = src != dst
    src, dst = src[keep], dst[keep]
    g = DeviceGraph.from_edges(np.concatenate([src, dst]), np.concatenate([dst, src]), None, n_vertices=nv,
                               device="cuda")
    g.build_wedges(wide_from=wide_from)
    g.wedge_tried = True
    assert g.wedge_mode == wide_from and g.wedge_slots is not None
    # the same table without folded slots (what a graph gets whose folded copies do not fit): the kernel that reads
    # wedge_off walks it
    g_plain = DeviceGraph.from_edges(np.concatenate([src, dst]), np.concatenate([dst, src]), None, n_vertices=nv,
                                     device="cuda")
    g_plain.build_wedges(wide_from=wide_from, fold=False)
    g_plain.wedge_tried = True
    assert not g_plain.slots_folded and g_plain.wedge_slots is not None
    start = rw.start_vertices(g)
    rowptr, col = g.rowptr.cpu().numpy(), g.col.cpu().numpy()
    for p, q in PQ_MIXED:
        want, wv = n2v_oracle.random_walk(rowptr, col, None, start.cpu().numpy(), 2, 25, p, q, 77, n_threads=8)
        assert g.slots_folded  # the slots kernel steps the wide rows through folded lists
        for kw in ({}, {"use_wedge_slots": False}, {"use_wedge_kernel": False}, {"use_workspace": True}):
            got, gv = rw.walk(g, start, 2, 25, p, q, 77, **kw)
            assert g.wedge_mode == wide_from  # (the table was not rebuilt)
            assert np.array_equal(gv.cpu().numpy(), wv), (p, q, kw)
            assert np.array_equal(got.cpu().numpy(), want), (p, q, kw)
        got, gv = rw.walk(g_plain, start, 2, 25, p, q, 77)
        assert np.array_equal(gv.cpu().numpy(), wv) and np.array_equal(got.cpu().numpy(), want), (p, q, "no fold")


def test_a_row_of_more_than_65535_entries_keeps_the_slots_kernel():
    """The reference's default trim cap is 100 000 (constants.py:6, randomwalk.py:252-253): ONE row of
    65 536 entries or more used to switch the wedge slots off for the whole graph.
    Now only the steps standing on such a row read 32-bit lists.  A hub of 70 000 neighbours inside a
    graph with triangles: the table is mixed, the slots exist, and the walks equal the table-free
    kernel's over all start vertices and the oracle's on a sample that starts on and next to the hub."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import n2v_oracle
    from node2vec_amd import randomwalk as rw
    from node2vec_amd.graph import DeviceGraph

    rng = np.random.default_rng(5)
    nv = 90_000
    leaves = rng.choice(np.arange(2, nv), 70_000, replace=False)
    a, b = rng.integers(0, nv, 300_000), rng.integers(0, nv, 300_000)
    second = rng.choice(np.arange(2, nv), 66_000, replace=False)  # a second wide row: hub-hub wedges are long
    src = np.concatenate([np.zeros_like(leaves), a, np.ones_like(second), [0]])
    dst = np.concatenate([leaves, b, second, [1]])
    keep = src != dst
    src, dst = src[keep], dst[keep]
    key = np.unique(np.concatenate([src * nv + dst, dst * nv + src]))
    g = DeviceGraph.from_edges(key // nv, key % nv, None, n_vertices=nv, device="cuda")
    assert int(g.degrees().max()) >= 70_000
    g.build_wedges()
    g.wedge_tried = True
    assert g.wedge_mode == 65536 and g.wedge_slots is not None and g.wedge_pos.dtype == torch.int16
    start_all = rw.start_vertices(g)
    rowptr, col = g.rowptr.cpu().numpy(), g.col.cpu().numpy()
    hub_nbrs = col[rowptr[0]:rowptr[0] + 40]
    sample = torch.tensor(np.unique(np.concatenate([[0, 1], hub_nbrs, col[rowptr[1]:rowptr[1] + 20]])),
                          dtype=torch.int32, device="cuda")
    for p, q in ((0.5, 2.0), (4.0, 0.25), (3.0, 0.7), (4.0, 2.0)):
        a1, v1 = rw.walk(g, start_all, 1, 12, p, q, 9)
        a2, v2 = rw.walk(g, start_all, 1, 12, p, q, 9, use_edge_classes=False)  # table-free, wave per walker
        assert torch.equal(v1, v2) and torch.equal(a1, a2), (p, q)
        a3, v3 = rw.walk(g, start_all, 1, 12, p, q, 9, use_wedge_slots=False)
        assert torch.equal(v1, v3) and torch.equal(a1, a3), (p, q)
        got, gv = rw.walk(g, sample, 2, 6, p, q, 9)
        want, wv = n2v_oracle.random_walk(rowptr, col, None, sample.cpu().numpy(), 2, 6, p, q, 9, n_threads=8)
        assert np.array_equal(gv.cpu().numpy(), wv) and np.array_equal(got.cpu().numpy(), want), (p, q)
        hub_steps = int(((got[:, :-1] == 0) | (got[:, :-1] == 1)).sum())
        assert hub_steps > 50  # the wide path was taken
    # fast mode reads the same table: walks start, stay in range and follow edges
    f, fv = rw.walk(g, sample, 2, 10, 4.0, 0.25, 9, mode="fast")
    assert bool(fv.all()) and int(f.min()) >= 0 and int(f.max()) < nv


def test_hop_table_with_inline_return_positions():
    """n2v_hops_build with N2V_HOPS_INLINE_RPOS: the class word of an edge WITHOUT shared neighbours
    is N2V_EC_INLINE | return count << 24 | return position, every other entry is edge_classes[e];
    exact biased walks through that table (slots kernel) equal those through the plain one"""
    from node2vec_amd import randomwalk as rw
    from node2vec_amd.graph import DeviceGraph

    rng = np.random.default_rng(23)
    nv = 900
    src = np.concatenate([rng.integers(0, nv, 7000), rng.integers(0, 5, 1500), rng.integers(0, nv, 1500)])
    dst = np.concatenate([rng.integers(0, nv, 7000), rng.integers(0, nv, 1500), rng.integers(0, 5, 1500)])
    g = DeviceGraph.from_edges(np.concatenate([src, dst]), np.concatenate([dst, src]), None, n_vertices=nv,
                               device="cuda")
    g.build_wedges()
    assert g.can_inline_rpos()
    g.build_hops(inline_rpos=True)
    assert g.hops_inline_rpos and g.c_struct().reserved2 == 1
    ec = g.edge_classes.cpu().numpy().astype(np.uint32)
    off = g.wedge_off.cpu().numpy().astype(np.uint64)
    cls = g.hops.cpu().numpy()[:, 1].astype(np.uint32)
    plain = (ec & 0xffffff) != 0
    assert 100 < plain.sum() < len(ec) - 100  # both kinds of edges
    assert np.array_equal(cls[plain], ec[plain])
    want = np.uint32(0x80000000) | (ec[~plain] & np.uint32(0x7f000000)) | (off[~plain] >> np.uint64(40)).astype(np.uint32)
    assert np.array_equal(cls[~plain], want)
    start = rw.start_vertices(g)
    for p, q in ((0.5, 2.0), (4.0, 0.25), (4.0, 2.0), (3.0, 0.7)):
        a, av = rw.walk(g, start, 3, 30, p, q, 5)
        assert g.hops_inline_rpos  # the default path keeps the inline form
        b, bv = rw.walk(g, start, 3, 30, p, q, 5, use_wedge_slots=False)
        assert not g.hops_inline_rpos  # another kernel: the table was rebuilt in the plain form
        assert torch.equal(a, b) and torch.equal(av, bv)
    # a caller that hands the inline form to another kernel is told so
    g.build_hops(inline_rpos=True)
    from node2vec_amd import _lib
    cs = g.c_struct()
    cs.wedge_slots = 0
    walks = torch.empty((start.numel(), 5), dtype=torch.int32, device="cuda")
    valid = torch.empty(start.numel(), dtype=torch.uint8, device="cuda")
    status = torch.zeros(4, dtype=torch.int32, device="cuda")
    rc = _lib.load().n2v_walk(cs, start.data_ptr(), start.numel(), 1, 4, 0.5, 2.0, 1, 0, walks.data_ptr(),
                              valid.data_ptr(), status.data_ptr(), None)
    assert rc == _lib.EINVAL


def test_wedge_table_respects_its_memory_bound():
    from node2vec_amd import synthetic

    g = synthetic.rmat(14, 200_000, device="cuda")
    g.build_wedges(max_bytes=1000)
    assert g.wedge_off is None and g.wedge_pos is None and g.edge_classes is not None
    assert g.wedge_slots is None
    g.build_wedges()
    assert g.wedge_off is not None and g.wedge_slots is not None
    need = 8 * g.n_edges + 2 * g.wedge_pos.numel()
    g.build_wedges(max_bytes=need + 8)  # room for the lists but not for 32 more bytes per edge
    assert g.wedge_off is not None and g.wedge_slots is None
