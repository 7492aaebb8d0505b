"""GPU parity tests for K2 (n2v_walk) through the C ABI: bit-exact against the
golden walks produced by the reference itself and against the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _graph_from_edges(edges, nv=None):
    from node2vec_amd.graph import DeviceGraph

    e = np.array(edges, dtype=np.float64).reshape(-1, 3)
    return DeviceGraph.from_edges(e[:, 0].astype(np.int64), e[:, 1].astype(np.int64),
                                  e[:, 2], n_vertices=nv, device="cuda")  # fp64 kept when not fp32-exact


def _hip_walks(g, start, nw, wl, p, q, seed, mode="exact"):
    from node2vec_amd import randomwalk as rw

    walks, valid = rw.walk(g, torch.as_tensor(start, dtype=torch.int32), nw, wl, p, q, seed, mode)
    torch.cuda.synchronize()
    return walks.cpu().numpy(), valid.cpu().numpy().astype(bool)


def test_library_loaded_is_in_tree():
    from node2vec_amd import _lib

    L = _lib.load()
    assert L.n2v_abi_version() == _lib.ABI_VERSION
    assert L.n2v_device_count() >= 1


def test_exact_walks_equal_reference_golden():
    """G4/G7: same walks, same dropped walkers as the reference driven with the
    same uniform stream (tests/golden/gen_golden.py)."""
    names = set()
    for c in load_golden("g4_walks.json"):
        g = _graph_from_edges(c["edges"])
        names.add(c["name"])
        if c["name"].endswith("_fp64"):
            assert g.w.dtype == torch.float64  # full-precision weights reach the kernel as fp64
        start = list(range(g.n_vertices)) if c["walk_seed"] is None else sorted(set(c["walk_seed"]))
        walks, valid = _hip_walks(g, start, c["num_walks"], c["walk_length"], c["p"], c["q"], c["seed"])
        got = {}
        for i, s in enumerate(start):
            for o in range(c["num_walks"]):
                r = i * c["num_walks"] + o
                if valid[r]:
                    got[(s, o + 1)] = walks[r].tolist()
        want = {(w["start"], w["ordinal"]): w["walk"] for w in c["walks"]}
        assert got.keys() == want.keys(), c["name"]
        for k in want:
            assert got[k] == want[k], (c["name"], k)
    assert {"karate_weighted_fp64", "multigraph_fp64", "decimal_weights_fp64",
            "decimal_weights_pq_fp64"} <= names


def _random_graph(rng, nv, ne, weighted, hubs=0, sinks=True):
    src = rng.integers(0, nv, size=ne)
    dst = rng.integers(0, nv, size=ne)
    if hubs:
        hs = rng.integers(0, nv, size=hubs)
        for h in hs:
            k = int(rng.integers(200, 3000))
            src = np.concatenate([src, np.full(k, h)])
            dst = np.concatenate([dst, rng.integers(0, nv, size=k)])
    if sinks:
        keep = src % 13 != 5
        src, dst = src[keep], dst[keep]
    w = rng.uniform(0.05, 4.0, size=len(src)).astype(np.float32) if weighted else np.ones(len(src), np.float32)
    return src, dst, w


@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("pq", [(1.0, 1.0), (0.5, 2.0), (4.0, 0.25), (3.0, 0.7), (1.0, 2.0), (0.5, 1.0)])
def test_exact_walks_equal_oracle_random_graphs(oracle, weighted, pq):
    from node2vec_amd.graph import DeviceGraph

    rng = np.random.default_rng(7 + int(weighted))
    src, dst, w = _random_graph(rng, 3000, 40000, weighted, hubs=6)
    g = DeviceGraph.from_edges(src, dst, w, n_vertices=3000, device="cuda")
    rowptr, col, ww = g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.w.cpu().numpy()
    start = np.arange(0, 3000, 3, dtype=np.int32)
    p, q = pq
    want, wvalid = oracle.random_walk(rowptr, col, ww, start, 2, 25, p, q, 1234, n_threads=8)
    got, gvalid = _hip_walks(g, start, 2, 25, p, q, 1234)
    assert (gvalid == wvalid).all()
    assert (got[gvalid] == want[wvalid]).all()
    assert gvalid.sum() > 100 and (~gvalid).sum() > 0  # both kept and dropped walkers seen


def test_exact_huge_row_beyond_lds_cache(oracle):
    """rows above 16384 neighbours take the uncached classification path"""
    from node2vec_amd.graph import DeviceGraph

    rng = np.random.default_rng(3)
    nv = 40000
    hub = np.arange(1, 30001)
    src = np.concatenate([np.zeros(len(hub), np.int64), hub, rng.integers(1, nv, 60000)])
    dst = np.concatenate([hub, np.zeros(len(hub), np.int64), rng.integers(1, nv, 60000)])
    w = np.ones(len(src), np.float32)
    g = DeviceGraph.from_edges(src, dst, w, n_vertices=nv, device="cuda")
    rowptr, col, ww = g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.w.cpu().numpy()
    start = np.arange(0, 400, dtype=np.int32)
    want, wvalid = oracle.random_walk(rowptr, col, ww, start, 2, 12, 0.5, 2.0, 99, n_threads=8)
    got, gvalid = _hip_walks(g, start, 2, 12, 0.5, 2.0, 99)
    assert (gvalid == wvalid).all() and (got[gvalid] == want[wvalid]).all()


def test_zero_weight_row_raises_zero_division():
    g = _graph_from_edges([(0, 1, 0.0), (1, 0, 1.0)])
    from node2vec_amd import randomwalk as rw

    with pytest.raises(ZeroDivisionError):
        rw.walk(g, torch.tensor([0], dtype=torch.int32), 1, 2, 1.0, 1.0, 1)


def test_zero_p_or_q_raises_value_error():
    g = _graph_from_edges([(0, 1, 1.0), (1, 0, 1.0)])
    from node2vec_amd import randomwalk as rw

    with pytest.raises(ValueError):
        rw.walk(g, torch.tensor([0], dtype=torch.int32), 1, 2, 0.0, 1.0, 1)
    with pytest.raises(ValueError):
        rw.walk(g, torch.tensor([0], dtype=torch.int32), 1, 2, 1.0, 0.0, 1)


def test_results_independent_of_sharding():
    """walks keyed by (seed, start vertex, ordinal): any split of start_ids gives
    the same rows (1-GPU walks == concatenated N-GPU walks)."""
    from node2vec_amd import synthetic

    g = synthetic.rmat(12, 40000, device="cuda")
    from node2vec_amd import randomwalk as rw

    start = rw.start_vertices(g)
    full, v = rw.walk(g, start, 3, 20, 0.5, 2.0, 42)
    halves = [rw.walk(g, part, 3, 20, 0.5, 2.0, 42) for part in torch.chunk(start, 3)]
    cat = torch.cat([h[0] for h in halves])
    assert torch.equal(full, cat) and bool(v.all())


@pytest.mark.parametrize("pq", [(1.0, 1.0), (0.5, 2.0), (1.0, 2.0), (0.25, 4.0), (4.0, 0.25), (3.0, 0.7),
                                (2.0, 0.5), (0.25, 0.25), (0.5, 0.25), (4.0, 4.0), (1.0, 0.5), (0.25, 0.5),
                                (2.0, 1.0), (4.0, 2.0), (2.0, 2.0), (0.5, 0.5),
                                (8.0, 2.0), (0.125, 0.25)])
def test_hop_table_and_class_counts_change_no_bit(oracle, pq):
    """unit-weight graph with sinks, multi-edges and hubs: the walks are the same bits with the
    hop table (one gather per step), with the CSR arrays + per-edge class counts, and with the
    wave-per-walker kernel that uses neither -- and equal to the oracle; each fast sampler draws the same
    walks with and without the hop table"""
    from node2vec_amd import randomwalk as rw
    from node2vec_amd.graph import DeviceGraph

    p, q = pq
    rng = np.random.default_rng(5)
    nv = 3000
    src = np.concatenate([rng.integers(0, nv - 50, 24000), rng.integers(0, 20, 9000), rng.integers(0, nv - 50, 9000)])
    dst = np.concatenate([rng.integers(0, nv, 24000), rng.integers(0, nv, 9000), rng.integers(0, 20, 9000)])
    g = DeviceGraph.from_edges(src, dst, None, n_vertices=nv, device="cuda")  # ids >= nv - 50 are sinks
    assert g.unit_weights
    start = rw.start_vertices(g)
    a, av = rw.walk(g, start, 3, 25, p, q, 9)
    if p == q == 1.0:
        assert g.hops8 is not None and g.hops is None  # the 8-byte hop table serves p = q = 1
        h16, h16v = rw.walk(g, start, 3, 25, p, q, 9, use_hops8=False)  # the 16-byte table
        assert g.hops is not None and torch.equal(a, h16) and torch.equal(av, h16v)
    assert g.hops is not None and (p == q == 1.0 or g.hops_have_classes or not rw.tables_regime(p, q))
    assert (g.wedge_off is not None) == rw.tables_regime(p, q)  # (3, 0.7) is not dyadic: tables all the same
    b, bv = rw.walk(g, start, 3, 25, p, q, 9, use_hops=False)
    c, cv = rw.walk(g, start, 3, 25, p, q, 9, use_hops=False, use_edge_classes=False)
    d, dv = rw.walk(g, start, 3, 25, p, q, 9, use_wedges=False)
    e, ev = rw.walk(g, start, 3, 25, p, q, 9, use_hops=False, use_wedges=False)
    f, fv = rw.walk(g, start, 3, 25, p, q, 9, use_wedge_kernel=False)  # lanes kernel, all tables
    assert torch.equal(a, f) and torch.equal(av, fv)
    assert torch.equal(a, b) and torch.equal(av, bv) and torch.equal(a, c) and torch.equal(av, cv)
    assert torch.equal(a, d) and torch.equal(av, dv) and torch.equal(a, e) and torch.equal(av, ev)
    want, wv = oracle.random_walk(g.rowptr.cpu().numpy(), g.col.cpu().numpy(), None,
                                  start.cpu().numpy(), 3, 25, p, q, 9, n_threads=8)
    assert np.array_equal(av.cpu().numpy(), wv)
    assert np.array_equal(a.cpu().numpy()[wv], want[wv])
    assert not bool(av.all())  # some walkers did vanish at sinks
    # fast mode: with the wedge table the class of a step is drawn first (one sampler), without it
    # candidates are rejected (another): each draws the same with and without the hop table; the
    # two agree in distribution (tests/test_fast_unit_gpu.py), in the first step (the unbiased
    # table, = exact mode) and -- where there is no bias -- in every draw
    fa, fav = rw.walk(g, start, 3, 25, p, q, 9, mode="fast")
    fb, fbv = rw.walk(g, start, 3, 25, p, q, 9, mode="fast", use_hops=False)
    fc, fcv = rw.walk(g, start, 3, 25, p, q, 9, mode="fast", use_wedges=False)
    fd, fdv = rw.walk(g, start, 3, 25, p, q, 9, mode="fast", use_wedges=False, use_hops=False)
    assert torch.equal(fa, fb) and torch.equal(fav, fbv)
    assert torch.equal(fc, fd) and torch.equal(fcv, fdv)
    # the layered sampler reads an edge's list from its wedge slot or through wedge_off: same draws
    fe, fev = rw.walk(g, start, 3, 25, p, q, 9, mode="fast", use_wedge_slots=False)
    assert (p == q == 1.0) or g.wedge_slots is not None
    assert torch.equal(fa, fe) and torch.equal(fav, fev)
    assert torch.equal(fa[:, :2], a[:, :2]) and torch.equal(fc[:, :2], a[:, :2])
    if p == q == 1.0:
        assert torch.equal(fa, a) and torch.equal(fc, a)


@pytest.mark.parametrize("bits", [(31, 30, 0), (29, 31, 0), (None, None, 0), (None, None, 3), (31, 30, 3)])
def test_hop8_table_changes_no_bit_with_and_without_escapes(oracle, bits):
    """the 8-byte hop table of the p = q = 1 kernel, rows as in the CSR (shift 0) or padded to
    multiples of 8 entries (shift 3): with wide id / row fields the degree field is 3 to 5 bits and
    many rows take the escape (degree read from rowptr); same walks as the 16-byte table, the CSR
    arrays and the oracle -- sinks, hubs, multi-edges, walk length 0 and 1"""
    from node2vec_amd import randomwalk as rw
    from node2vec_amd.graph import DeviceGraph

    rng = np.random.default_rng(8)
    nv = 4000
    src = np.concatenate([rng.integers(0, nv - 60, 30000), rng.integers(0, 10, 12000), rng.integers(0, nv - 60, 9000)])
    dst = np.concatenate([rng.integers(0, nv, 30000), rng.integers(0, nv, 12000), rng.integers(0, 10, 9000)])
    g = DeviceGraph.from_edges(src, dst, None, n_vertices=nv, device="cuda")
    g.build_hops8(force=True, col_bits=bits[0], row_bits=bits[1], align_shift=bits[2])
    assert g.hops8 is not None and g.hops8_shift == bits[2]
    cb, rb = g.hops8_bits
    esc = (1 << (64 - cb - rb)) - 1
    deg = g.degrees().cpu().numpy()
    if bits[0] is not None:
        assert int((deg >= esc).sum()) > 100  # the escape is exercised
    h = g.hops8.cpu().numpy().astype(np.uint64)
    col, rowptr = g.col.cpu().numpy(), g.rowptr.cpu().numpy()
    trow = rowptr if g.hops8_rowptr is None else g.hops8_rowptr.cpu().numpy()
    if bits[2]:
        assert np.all(trow % 8 == 0) and np.all(np.diff(trow) >= deg) and np.all(np.diff(trow) < deg + 8)
    src_of = np.repeat(np.arange(nv), deg)
    at = trow[src_of] + (np.arange(col.size) - rowptr[src_of])
    he = h[at]
    assert np.array_equal((he & np.uint64((1 << cb) - 1)).astype(np.int64), col.astype(np.int64))
    assert np.array_equal(((he >> np.uint64(cb)) & np.uint64((1 << rb) - 1)).astype(np.int64),
                          trow[col] >> bits[2])
    assert np.array_equal((he >> np.uint64(cb + rb)).astype(np.int64), np.minimum(deg[col], esc))
    start = rw.start_vertices(g)
    for L in (0, 1, 30):
        a, av = rw.walk(g, start, 3, L, 1.0, 1.0, 4)
        b, bv = rw.walk(g, start, 3, L, 1.0, 1.0, 4, use_hops8=False)
        c, cv = rw.walk(g, start, 3, L, 1.0, 1.0, 4, use_hops=False)
        assert torch.equal(a, b) and torch.equal(av, bv) and torch.equal(a, c) and torch.equal(av, cv)
    want, wv = oracle.random_walk(rowptr, col, None, start.cpu().numpy(), 3, 30, 1.0, 1.0, 4, n_threads=8)
    assert np.array_equal(av.cpu().numpy(), wv) and np.array_equal(a.cpu().numpy()[wv], want[wv])
    assert not wv.all()


def test_hop_table_is_refused_for_rows_it_cannot_pack():
    """a row of 2^24 neighbours or more does not fit n2v_hop: the graph stays without the table"""
    from node2vec_amd.graph import DeviceGraph

    n = (1 << 24) + 3
    rowptr = torch.tensor([0, n, n], dtype=torch.int64, device="cuda")
    col = torch.ones(n, dtype=torch.int32, device="cuda")
    g = DeviceGraph(rowptr, col, None)
    g.build_hops()
    assert g.hops is None


@pytest.mark.parametrize("pq", [(3.0, 0.7), (0.7, 3.0), (1.3, 1.3), (5.0, 3.0), (0.2, 0.6), (0.6, 0.2), (7.0, 1.0),
                                (0.3, 1.7), (10.0, 0.1), (0.001, 37.5)])
def test_values_that_are_not_dyadic_closed_forms_with_margins(oracle, pq):
    """1/p or 1/q not a power of two: the slots kernel decides most pairings by the closed forms on
    the values the counts give, with a margin that covers the reference loop's rounding
    (n2v_unit_near.h), and replays the rest.  Every arrangement of the three classes on the two
    stacks, hubs of a few thousand neighbours, multi-edges, sinks: the oracle's walks, bit for bit;
    and the same walks from the kernel that replays every pairing (no wedge slots)."""
    from node2vec_amd import randomwalk as rw
    from node2vec_amd.graph import DeviceGraph

    rng = np.random.default_rng(int(pq[0] * 1000 + pq[1] * 10))
    nv = 6000
    src = np.concatenate([rng.integers(0, nv - 50, 60000), rng.integers(0, 8, 16000), rng.integers(0, nv - 50, 9000)])
    dst = np.concatenate([rng.integers(0, nv, 60000), rng.integers(0, nv, 16000), rng.integers(0, 8, 9000)])
    g = DeviceGraph.from_edges(np.concatenate([src, dst[:70000]]), np.concatenate([dst, src[:70000]]), None,
                               n_vertices=nv, device="cuda")
    start = torch.unique(torch.cat([torch.arange(0, 40), torch.arange(0, nv, 7)])).to(torch.int32)
    p, q = pq
    got, gv = rw.walk(g, start, 3, 40, p, q, 99)
    assert g.wedge_slots is not None
    other, ov = rw.walk(g, start, 3, 40, p, q, 99, use_wedge_slots=False)
    assert torch.equal(got, other) and torch.equal(gv, ov)
    want, wv = oracle.random_walk(g.rowptr.cpu().numpy(), g.col.cpu().numpy(), None, start.cpu().numpy(), 3, 40,
                                  p, q, 99, n_threads=8)
    assert np.array_equal(gv.cpu().numpy(), wv) and np.array_equal(got.cpu().numpy()[wv], want[wv])
    assert wv.sum() > 1000


@pytest.mark.parametrize("pq", [(3.0, 3.0), (1.5, 3.0), (3.0, 1.5), (0.75, 0.375), (6.0, 3.0), (0.3, 0.9), (1.2, 0.6)])
def test_values_that_are_not_dyadic_on_rows_full_of_near_ties(oracle, pq):
    """Class values in small rational ratios (1/3, 2/3, 1 ...) on overlapping cliques: the cumulative
    sums of the pairing loop meet EXACTLY in real arithmetic on many rows (n_shared = n_other, an
    excess that is twice a deficit ...) and within a few ulp in fp64 -- the decisions the closed forms
    with margins must NOT make.  The oracle's walks, bit for bit."""
    from node2vec_amd import randomwalk as rw
    from node2vec_amd.graph import DeviceGraph

    rng = np.random.default_rng(5)
    src, dst = [], []
    nv = 0
    for size in [4, 5, 6, 7, 8, 9, 10, 12, 16, 24, 33, 48, 64, 65, 100] * 3:
        ids = np.arange(nv, nv + size)
        a, b = np.meshgrid(ids, ids)
        keep = a != b
        src.append(a[keep]); dst.append(b[keep])
        nv += size
    src, dst = np.concatenate(src), np.concatenate(dst)
    # bridges between cliques (symmetric), pendant vertices, a few duplicated edges
    extra = rng.integers(0, nv, (1500, 2))
    extra = extra[extra[:, 0] != extra[:, 1]]
    src = np.concatenate([src, extra[:, 0], extra[:, 1], extra[:40, 0]])
    dst = np.concatenate([dst, extra[:, 1], extra[:, 0], extra[:40, 1]])
    g = DeviceGraph.from_edges(src, dst, None, n_vertices=nv, device="cuda")
    start = rw.start_vertices(g)
    p, q = pq
    got, gv = rw.walk(g, start, 6, 60, p, q, 7)
    assert g.wedge_slots is not None
    want, wv = oracle.random_walk(g.rowptr.cpu().numpy(), g.col.cpu().numpy(), None, start.cpu().numpy(), 6, 60,
                                  p, q, 7, n_threads=8)
    assert np.array_equal(gv.cpu().numpy(), wv) and np.array_equal(got.cpu().numpy()[wv], want[wv])
    other, ov = rw.walk(g, start, 6, 60, p, q, 7, use_wedge_slots=False)
    assert torch.equal(got, other) and torch.equal(gv, ov)


@pytest.mark.parametrize("pq", [(0.25, 0.5), (0.125, 0.25), (0.0625, 0.125), (0.5, 2.0), (4.0, 2.0)])
def test_a_class_exactly_on_the_average(oracle, pq):
    """Dyadic p, q on rows where one class sits exactly on the row average (p = 1/4, q = 1/2 with two
    shared neighbours per return edge: "other" = 2 = avg): its slots are overfull with excess 0, the
    table's values are powers of two and the reference's loop is exact arithmetic -- the closed form of
    the shared-stack arrangement decides those rows with the ties taken as the loop takes them (a slot
    at exactly 1.0 stays overfull).  Cliques of every size with bridges and duplicated edges: the oracle's
    walks and the table-free kernel's, bit for bit."""
    from node2vec_amd import randomwalk as rw
    from node2vec_amd.graph import DeviceGraph

    rng = np.random.default_rng(11)
    src, dst, nv = [], [], 0
    for size in list(range(3, 40)) * 2 + [64, 65, 130]:
        ids = np.arange(nv, nv + size)
        a, b = np.meshgrid(ids, ids)
        keep = a != b
        src.append(a[keep]); dst.append(b[keep])
        nv += size
    extra = rng.integers(0, nv, (4000, 2))
    extra = extra[extra[:, 0] != extra[:, 1]]
    src = np.concatenate(src + [extra[:, 0], extra[:, 1], extra[:60, 0], extra[:60, 1]])
    dst = np.concatenate(dst + [extra[:, 1], extra[:, 0], extra[:60, 1], extra[:60, 0]])
    g = DeviceGraph.from_edges(src, dst, None, n_vertices=nv, device="cuda")
    start = rw.start_vertices(g)
    p, q = pq
    got, gv = rw.walk(g, start, 8, 50, p, q, 3)
    want, wv = oracle.random_walk(g.rowptr.cpu().numpy(), g.col.cpu().numpy(), None, start.cpu().numpy(), 8, 50,
                                  p, q, 3, n_threads=8)
    assert np.array_equal(gv.cpu().numpy(), wv) and np.array_equal(got.cpu().numpy()[wv], want[wv])
    other, ov = rw.walk(g, start, 8, 50, p, q, 3, use_wedges=False, use_edge_classes=False)
    assert torch.equal(got, other) and torch.equal(gv, ov)
