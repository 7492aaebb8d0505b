"""Graph-partitioned walking on the GPU: every rank of a vertex-range partition
stepped by the HIP step -- the fused one (n2v_partition_step: one launch per rank and step, N(v)
from the part's CSR, N(s) travelling only when q != 1) and the launch-per-stage one on
materialised tables (n2v_walk_uniforms + n2v_edge_bias + n2v_alias_build + n2v_alias_draw) --
walkers migrating between the parts, must reproduce the oracle and n2v_walk on the whole graph
bit for bit: unit, fp32 and fp64 weights, with sinks, for (p, q) with q == 1 and q != 1."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("step", ["fused", "fused_rows", "tables"])
@pytest.mark.parametrize("weighted", [False, True])
def test_partitioned_equals_the_oracle_bit_for_bit(oracle, weighted, step):
    """the HIP step function on every part, walkers migrating, against the ORACLE's walk over the
    whole graph (not only against n2v_walk): 3 parts, sinks, four (p, q)"""
    from node2vec_amd import partitioned as P
    from node2vec_amd import randomwalk as rw
    from node2vec_amd.graph import DeviceGraph

    rng = np.random.default_rng(5)
    nv, ne = 3000, 30_000
    src = np.concatenate([rng.integers(0, nv - 100, ne), rng.integers(0, 6, 3000)])
    dst = np.concatenate([rng.integers(0, nv, ne), rng.integers(0, nv, 3000)])
    w = (rng.random(len(src)) * 1.7 + 0.3) if weighted else None
    g = DeviceGraph.from_edges(src, dst, w, n_vertices=nv, device="cuda")
    # "fused": walkers carry wedge lists (unit weights; weighted parts have no such tables and
    # carry rows); "fused_rows": the same launch with whole rows travelling; "tables": launch per stage
    parts = P.partition_graph(g, 3, wedges=step == "fused")
    assert (parts[0].wedge_off is not None) == (step == "fused" and not weighted)
    start = rw.start_vertices(g)[::3].contiguous()
    step_fn = P.tables_step if step == "tables" else P.hip_step
    for p, q in ((1.0, 1.0), (0.5, 2.0), (4.0, 0.25), (0.3, 1.0)):
        want, wv = oracle.random_walk(g.rowptr.cpu().numpy(), g.col.cpu().numpy(),
                                      None if g.unit_weights else g.w.cpu().numpy(),
                                      start.cpu().numpy(), 2, 12, p, q, 9)
        walks, valid = P.walk_partitioned_local(parts, start, 2, 12, p, q, 9, step_fn=step_fn)
        got, gv = walks.cpu().numpy(), valid.cpu().numpy().astype(bool)
        assert np.array_equal(gv, wv) and not wv.all()
        assert np.array_equal(got[gv], want[wv])


@pytest.mark.parametrize("wedges", [True, False])
def test_a_part_without_edges_only_sees_walkers_vanish(oracle, wedges):
    """the last third of the vertices has no out-edges: that part stores no edge (NULL col, NULL
    per-edge tables), every walker that reaches it vanishes there (fugue.py:147) -- found by the
    step itself, one step after the arrival -- and the walks still equal the oracle's"""
    from node2vec_amd import partitioned as P
    from node2vec_amd import randomwalk as rw
    from node2vec_amd.graph import DeviceGraph

    rng = np.random.default_rng(12)
    g = DeviceGraph.from_edges(rng.integers(0, 200, 3000), rng.integers(0, 300, 3000), None,
                               n_vertices=300, device="cuda")
    parts = P.partition_graph(g, 3, balance="vertices", wedges=wedges)
    assert parts[2].col.numel() == 0 and parts[2].lo == 200
    start = rw.start_vertices(g)
    for p, q in ((1.0, 1.0), (0.5, 2.0), (2.0, 1.0)):
        want, wv = oracle.random_walk(g.rowptr.cpu().numpy(), g.col.cpu().numpy(), None,
                                      start.cpu().numpy(), 3, 9, p, q, 4)
        walks, valid = P.walk_partitioned_local(parts, start, 3, 9, p, q, 4)
        got, gv = walks.cpu().numpy(), valid.cpu().numpy().astype(bool)
        assert np.array_equal(gv, wv) and 0 < wv.sum() < wv.size
        assert np.array_equal(got[gv], want[wv])
        # dropped rows: the path up to the sink, then -1 (as n2v_walk emits them)
        ref, _ = rw.walk(g, start, 3, 9, p, q, 4)
        assert torch.equal(walks, ref)


@pytest.mark.parametrize("weighted", [False, True, "fp32"])
def test_partitioned_equals_n2v_walk_bit_for_bit(weighted):
    from node2vec_amd import partitioned as P
    from node2vec_amd import randomwalk as rw
    from node2vec_amd.graph import DeviceGraph

    rng = np.random.default_rng(31)
    nv, ne = 20_000, 260_000
    src = np.concatenate([rng.integers(0, nv - 300, ne), rng.integers(0, 12, 30_000)])  # 12 hubs
    dst = np.concatenate([rng.integers(0, nv, ne), rng.integers(0, nv, 30_000)])
    w = (rng.random(len(src)) * 1.7 + 0.3) if weighted else None  # fp64, not fp32-representable
    if weighted == "fp32":
        w = w.astype(np.float32)
    g = DeviceGraph.from_edges(src, dst, w, n_vertices=nv, device="cuda")
    parts = P.partition_graph(g, 4)
    assert max(pt.col.numel() for pt in parts) < 0.3 * g.n_edges  # a rank stores ~E / 4
    if not weighted:  # every part holds the slice of the wedge table of its own edges
        assert all(pt.wedge_off is not None and pt.wedge_off.numel() == pt.col.numel() for pt in parts)
        assert sum(pt.wedge_pos.numel() for pt in parts) <= g.wedge_pos.numel() + len(parts)
    start = rw.start_vertices(g)[::7].contiguous()
    for p, q in ((1.0, 1.0), (0.5, 2.0), (4.0, 0.25), (3.0, 1.0), (0.7, 1.3), (4.0, 2.0)):
        want, wv = rw.walk(g, start, 3, 15, p, q, 77)
        walks, valid = P.walk_partitioned_local(parts, start, 3, 15, p, q, 77)
        assert torch.equal(valid, wv)
        assert not bool(wv.all())  # walkers did vanish at sinks
        assert torch.equal(walks, want)  # dropped rows included: path up to the sink, then -1
        if not weighted:
            # unit weights: the call above forwarded the walkers with n2v_partition_forward (two
            # launches per part and step); the launch-per-stage routing that walk_partitioned's ranks
            # run must give the same rows
            assert P._forward_mode(parts, p, q, P.hip_step) == (1 if p == q == 1.0 else 3 if q == 1.0 else 2)
            w2, v2 = P.walk_partitioned_local(parts, start, 3, 15, p, q, 77, forwarding=False)
            assert torch.equal(v2, wv) and torch.equal(w2, want)
            # and every part stepped as a rank of walk_partitioned steps it: one forward launch into
            # per-destination outboxes, the walkers travelling as Mail
            w3, v3 = P.walk_partitioned_local(parts, start, 3, 15, p, q, 77, forwarding="ranks")
            assert torch.equal(v3, wv) and torch.equal(w3, want)
        else:
            assert P._forward_mode(parts, p, q, P.hip_step) == 0


def test_forwarding_repeats_a_step_whose_word_pool_was_too_small(monkeypatch):
    """the pool the wedge lists are appended to starts at 0 words: every step with lists overflows
    (N2V_ST_OVERFLOW), reports what it needs, and is repeated -- same walks"""
    from node2vec_amd import partitioned as P
    from node2vec_amd import randomwalk as rw
    from node2vec_amd.graph import DeviceGraph

    rng = np.random.default_rng(3)
    nv = 4000
    src = np.concatenate([rng.integers(0, nv, 60_000), rng.integers(0, 8, 8000)])
    dst = np.concatenate([rng.integers(0, nv, 60_000), rng.integers(0, nv, 8000)])
    g = DeviceGraph.from_edges(np.concatenate([src, dst]), np.concatenate([dst, src]), None, n_vertices=nv,
                               device="cuda")
    parts = P.partition_graph(g, 5)
    start = rw.start_vertices(g)
    want, wv = rw.walk(g, start, 2, 20, 0.5, 2.0, 8)
    monkeypatch.setattr(P, "FORWARD_WORDS_PER_WALKER", 0)
    monkeypatch.setattr(P, "FORWARD_MIN_WORDS", 0)
    t = {}
    walks, valid = P._walk_local_forwarding(parts, start, 2, 20, 0.5, 2.0, 8, 2, timings=t)
    assert t.get("pool_enlarged", 0) >= 1 and len(t["walkers_per_step"]) == 20
    assert torch.equal(valid, wv) and torch.equal(walks, want)
    walks, valid = P.walk_partitioned_local(parts, start, 2, 20, 0.5, 2.0, 8, forwarding="ranks")  # the ranks' form
    assert torch.equal(valid, wv) and torch.equal(walks, want)
    with pytest.raises(ValueError):
        P.walk_partitioned_local(P.partition_graph(g, 5, wedges=False), start, 2, 20, 0.5, 2.0, 8, forwarding=True)


@pytest.mark.parametrize("pq", [(0.5, 2.0), (1.0, 1.0), (3.0, 1.0)])
def test_ranks_walk_with_capacity_bounded_mailboxes(monkeypatch, pq):
    """walk_partitioned's ranks after the calibration steps: mailboxes of a fixed capacity per destination,
    exchanged whole (here: a transpose), empty slots skipped by the kernels, nothing read on the host until
    the last step -- same walks; boxes that are too small are reported once, at the end, and the walk is
    repeated with larger ones"""
    from node2vec_amd import partitioned as P
    from node2vec_amd import randomwalk as rw
    from node2vec_amd.graph import DeviceGraph

    rng = np.random.default_rng(5)
    nv = 6000
    src = np.concatenate([rng.integers(0, nv, 50_000), rng.integers(0, 12, 9000)])
    dst = np.concatenate([rng.integers(0, nv, 50_000), rng.integers(0, nv, 9000)])
    keep = src != dst
    src, dst = src[keep], dst[keep]
    a, b = np.concatenate([src, dst]), np.concatenate([dst, src])
    out = a < nv - 40  # the last vertices have in-edges only: walkers vanish there
    g = DeviceGraph.from_edges(a[out], b[out], None, n_vertices=nv, device="cuda")
    parts = P.partition_graph(g, 5)
    start = rw.start_vertices(g)
    p, q = pq
    want, wv = rw.walk(g, start, 3, 25, p, q, 19)
    assert not bool(wv.all())
    t = {}
    walks, valid = P.walk_partitioned_local(parts, start, 3, 25, p, q, 19, forwarding="ranks", timings=t)
    assert t["exact_steps"] == P.BOUNDED_CALIBRATION_STEPS and len(t["bounded_caps"]) == 1
    assert torch.equal(valid, wv) and torch.equal(walks, want)
    # boxes far too small: the attempt overflows, says so once, and the walk is repeated
    monkeypatch.setattr(P, "BOUNDED_SLACK", 0.2)
    monkeypatch.setattr(P, "BOUNDED_MIN_SLOTS", 1)
    t = {}
    walks, valid = P.walk_partitioned_local(parts, start, 3, 25, p, q, 19, forwarding="ranks", timings=t)
    slots = [sum(map(sum, caps_h)) for caps_h, _ in t["bounded_caps"]]  # one (slots, words) pair of matrices per attempt
    assert len(slots) >= 2 and slots[-1] > slots[0]
    assert torch.equal(valid, wv) and torch.equal(walks, want)
    # and with every step at exact sizes (round 4's form)
    monkeypatch.setattr(P, "BOUNDED", False)
    t = {}
    walks, valid = P.walk_partitioned_local(parts, start, 3, 25, p, q, 19, forwarding="ranks", timings=t)
    assert "bounded_caps" not in t
    assert torch.equal(valid, wv) and torch.equal(walks, want)


def test_partitioned_cfg2_sample_equals_n2v_walk():
    """BASELINE cfg 2 graph (R-MAT scale 20) cut into 8 parts, L = 80"""
    from node2vec_amd import partitioned as P
    from node2vec_amd import randomwalk as rw
    from node2vec_amd import synthetic

    g = synthetic.rmat(20, 5_000_000, device="cuda")
    parts = P.partition_graph(g, 8)
    start = rw.start_vertices(g)[::400].contiguous()
    want, wv = rw.walk(g, start, 2, 80, 0.5, 2.0, 42)
    import time

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    walks, valid = P.walk_partitioned_local(parts, start, 2, 80, 0.5, 2.0, 42)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"partitioned walking, 8 parts in one process: {int(valid.sum()) * 80 / dt / 1e6:.2f} M steps/s")
    assert bool(valid.all()) and torch.equal(valid, wv) and torch.equal(walks, want)


def test_partition_route_and_gathers_equal_plain_torch():
    """the C-ABI pieces of the routing on their own (n2v_partition_route, n2v_gather_rows,
    n2v_gather_wedges) against the index arithmetic they replace: finished walks and walkers that
    vanished are not forwarded, destinations by the parts' lower bounds (an EMPTY part in the
    middle owns nothing), lengths and sources by what travels"""
    from node2vec_amd import _lib
    from node2vec_amd.graph import DeviceGraph

    L = _lib.load()
    rng = np.random.default_rng(3)
    nv = 500
    g = DeviceGraph.from_edges(rng.integers(0, nv, 6000), rng.integers(0, nv, 6000), None,
                               n_vertices=nv, device="cuda")
    g.build_wedges()
    assert g.wedge_off is not None
    k, cols, walk_len, lo = 4000, 5, 7, 100
    bounds = torch.tensor([0, 100, 100, 320], dtype=torch.int64, device="cuda")  # part 1 is empty
    gen = torch.Generator().manual_seed(1)
    v = torch.randint(lo, 320, (k,), generator=gen)
    step = torch.randint(0, walk_len, (k,), generator=gen)
    nxt = torch.randint(0, nv, (k,), generator=gen).to(torch.int32)
    nxt[::17] = -1  # stood on a sink
    head = torch.stack([torch.arange(k), torch.randint(0, 1 << 40, (k,), generator=gen),
                        (torch.randint(0, nv, (k,), generator=gen) << 32) | v, step,
                        torch.randint(0, 1 << 30, (k,), generator=gen)], 1).cuda()
    rowptr = (g.rowptr[lo:321] - g.rowptr[lo]).contiguous()
    e0 = int(g.rowptr[lo])
    edge = (rowptr[(v - lo).cuda()] + torch.randint(0, 3, (k,), generator=gen).cuda()).clamp(max=int(rowptr[-1]) - 1)
    ec = g.edge_classes[e0:e0 + int(rowptr[-1])].contiguous()
    nxt_d = nxt.cuda()
    for carry in (0, 1, 2, 3):
        log = torch.empty((k, 3), dtype=torch.int64, device="cuda")
        ho = torch.empty((k, cols), dtype=torch.int64, device="cuda")
        dest = torch.empty(k, dtype=torch.int32, device="cuda")
        ln = torch.empty(k, dtype=torch.int64, device="cuda")
        src = torch.empty(k, dtype=torch.int64, device="cuda")
        _lib.check(L.n2v_partition_route(head.data_ptr(), cols, nxt_d.data_ptr(), edge.data_ptr(), k, walk_len,
                                         bounds.data_ptr(), 4, carry, rowptr.data_ptr(), lo, ec.data_ptr(),
                                         log.data_ptr(), ho.data_ptr(), dest.data_ptr(), ln.data_ptr(),
                                         src.data_ptr(), _lib.current_stream_ptr()), "route")
        gone = nxt_d < 0
        fwd = ~gone & (head[:, 3] + 1 < walk_len)
        assert torch.equal(log[:, 0], head[:, 0])
        assert torch.equal(log[:, 1], torch.where(gone, -1, head[:, 3] + 1))
        assert torch.equal(log[:, 2], torch.where(gone, -1, nxt_d.long()))
        want_dest = torch.searchsorted(bounds, nxt_d.long().clamp(min=0), right=True) - 1
        assert torch.equal(dest.long(), torch.where(fwd, want_dest, 4)) and not bool((dest == 1).any())
        assert torch.equal(ho[:, :2], head[:, :2]) and torch.equal(ho[:, 3], head[:, 3] + 1)
        assert torch.equal(ho[:, 2], ((head[:, 2] & 0xffffffff) << 32) | (nxt_d.long() & 0xffffffff))
        assert int(ho[:, 4].abs().sum()) == 0
        local = (head[:, 2] & 0xffffffff) - lo
        want_len = {0: torch.zeros_like(ln), 1: rowptr[local + 1] - rowptr[local],
                    2: (ec[edge] & 0xffffff).long(), 3: torch.zeros_like(ln)}[carry]
        assert torch.equal(ln, torch.where(fwd, want_len, 0))
        if carry:
            assert torch.equal(src[fwd], (local if carry == 1 else edge)[fwd])
        # grouping by destination: the stable counting sort against torch's stable sort
        work = torch.empty(((k + 255) // 256 + 1) * 5, dtype=torch.int64, device="cuda")
        hg, lg, sg = torch.empty_like(ho), torch.empty_like(ln), torch.empty_like(src)
        cuts = torch.empty(5, dtype=torch.int64, device="cuda")
        _lib.check(L.n2v_partition_group(dest.data_ptr(), ho.data_ptr(), cols, ln.data_ptr(), src.data_ptr(), k, 4,
                                         work.data_ptr(), hg.data_ptr(), lg.data_ptr(), sg.data_ptr(),
                                         cuts.data_ptr(), _lib.current_stream_ptr()), "group")
        ds, order = torch.sort(dest, stable=True)
        assert torch.equal(hg, ho[order]) and torch.equal(lg, ln[order]) and torch.equal(sg, src[order])
        assert torch.equal(cuts, torch.searchsorted(ds, torch.arange(5, dtype=torch.int32, device="cuda")))
        # the gathers over the forwarded walkers, in the given order
        idx = torch.nonzero(fwd).reshape(-1)
        ptr = torch.zeros(idx.numel() + 1, dtype=torch.int64, device="cuda")
        torch.cumsum(ln[idx], 0, out=ptr[1:])
        out = torch.full((max(int(ptr[-1]), 1),), -7, dtype=torch.int32, device="cuda")
        s_idx = src[idx].contiguous()
        if carry == 1:
            col = g.col[e0:e0 + int(rowptr[-1])].contiguous()
            _lib.check(L.n2v_gather_rows(rowptr.data_ptr(), col.data_ptr(), s_idx.data_ptr(), ptr.data_ptr(),
                                         idx.numel(), out.data_ptr(), _lib.current_stream_ptr()), "rows")
            want = torch.cat([col[rowptr[r]:rowptr[r + 1]] for r in s_idx.tolist()])
            assert torch.equal(out[:want.numel()], want)
        elif carry >= 2:
            off = g.wedge_off[e0:e0 + int(rowptr[-1])].contiguous()
            hs = ho[idx].contiguous()
            _lib.check(L.n2v_gather_wedges(ec.data_ptr(), off.data_ptr(), g.wedge_pos.data_ptr(),
                                           int(g.wedge_pos.dtype == torch.int32), s_idx.data_ptr(), ptr.data_ptr(),
                                           idx.numel(), out.data_ptr(), hs.data_ptr(), cols,
                                           _lib.current_stream_ptr()), "wedges")
            o = off[s_idx]
            assert torch.equal(hs[:, 4], (ec[s_idx].long() & 0xffffffff) | ((o >> 40) << 32))
            if carry == 2:
                want = torch.cat([g.wedge_pos[int(a):int(a) + int(c)].to(torch.int32)
                                  for a, c in zip((o & ((1 << 40) - 1)).tolist(), ln[idx].tolist())])
                assert torch.equal(out[:want.numel()], want)

