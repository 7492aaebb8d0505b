"""Size-independent properties at larger scale (where the CPU oracle would take too
long to check every walk): every hop is an edge, rows are complete, runs are
reproducible, exact and fast modes agree in distribution, and a sampled subset of
walks still equals the oracle bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _edge_keys(g):
    src = torch.repeat_interleave(torch.arange(g.n_vertices, device=g.device), g.degrees())
    return torch.sort(src * g.n_vertices + g.col.long()).values


def _all_hops_are_edges(g, walks, valid):
    w = walks[valid].long()
    hop = (w[:, :-1] * g.n_vertices + w[:, 1:]).reshape(-1)
    keys = _edge_keys(g)
    pos = torch.searchsorted(keys, hop).clamp_(max=keys.numel() - 1)
    return bool((keys[pos] == hop).all())


@pytest.mark.parametrize("pq", [(0.5, 2.0), (4.0, 0.25)])
def test_rmat18_exact_properties_and_oracle_sample(oracle, pq):
    from node2vec_amd import randomwalk as rw
    from node2vec_amd import synthetic

    p, q = pq
    g = synthetic.rmat(18, 1_200_000, device="cuda")
    start = rw.start_vertices(g)
    walks, valid = rw.walk(g, start, 4, 40, p, q, 2024)
    assert bool(valid.all())  # symmetrised graph: no sinks
    assert walks.shape == (start.numel() * 4, 41)
    assert torch.equal(walks[:, 0], start.repeat_interleave(4))  # to_path: src = path[0]
    assert _all_hops_are_edges(g, walks, valid)
    again, _ = rw.walk(g, start, 4, 40, p, q, 2024)
    assert torch.equal(walks, again)  # reproducible, independent of scheduling
    other, _ = rw.walk(g, start, 4, 40, p, q, 2025)
    assert not torch.equal(walks, other)
    # a sample of start vertices (incl. the biggest hubs) against the oracle
    deg = g.degrees()
    hubs = torch.topk(deg, 40).indices.to(torch.int32)
    sample = torch.unique(torch.cat([hubs, start[:: max(1, start.numel() // 400)]]))
    got, gv = rw.walk(g, sample, 4, 40, p, q, 2024)
    want, wv = oracle.random_walk(g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.w.cpu().numpy(),
                                  sample.cpu().numpy(), 4, 40, p, q, 2024, n_threads=8)
    assert np.array_equal(gv.cpu().numpy(), wv) and np.array_equal(got.cpu().numpy(), want)


def test_cfg2_every_walker_three_implementations():
    """BASELINE cfg 2, EVERY start vertex x 10 x 80: the closed-form kernel, the replay kernels and
    the table-free kernel give the same walks for one (p, q) per kernel instance"""
    from node2vec_amd import randomwalk as rw
    from node2vec_amd import synthetic
    from test_scale_cfg345_gpu import full_batch_differential

    g = synthetic.rmat(20, 5_000_000, device="cuda")
    start = rw.start_vertices(g)
    assert start.numel() > 400_000
    assert full_batch_differential(g, start) == 4 * start.numel() * 10 * 80


@pytest.mark.parametrize("pq", [(0.5, 2.0), (4.0, 0.25), (3.0, 0.7), (1.0, 1.0)])
def test_cfg2_full_size_properties_and_oracle_sample(oracle, pq):
    """BASELINE cfg 2 at its full size (R-MAT scale 20, ~9.7 M directed edges, W = 10, L = 80):
    the bench batch is walked once whole and once in 7 uneven shards -- same walks (a checksum of
    row checksums and the rows themselves), every hop is an edge, and a sample of start vertices
    (the biggest hubs included) equals the oracle bit for bit"""
    from node2vec_amd import randomwalk as rw
    from node2vec_amd import synthetic

    p, q = pq
    g = synthetic.rmat(20, 5_000_000, device="cuda")
    start = rw.start_vertices(g)[:47104].contiguous()
    walks, valid = rw.walk(g, start, 10, 80, p, q, 42)
    assert bool(valid.all()) and walks.shape == (471040, 81)
    assert _all_hops_are_edges(g, walks, valid)
    cuts = [0, 1, 777, 5000, 5001, 20000, 40001, 47104]
    parts = [rw.walk(g, start[a:b].contiguous(), 10, 80, p, q, 42)[0] for a, b in zip(cuts, cuts[1:])]
    sharded = torch.cat(parts)
    weights = torch.arange(1, 82, device="cuda", dtype=torch.int64)
    rows = (walks.long() * weights).sum(1)
    assert int(rows.sum()) == int((sharded.long() * weights).sum())
    assert torch.equal(walks, sharded)
    deg = g.degrees()
    sample = torch.unique(torch.cat([torch.topk(deg, 25).indices.to(torch.int32), start[::1500]]))
    got, gv = rw.walk(g, sample, 2, 30, p, q, 42)
    want, wv = oracle.random_walk(g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.w.cpu().numpy(),
                                  sample.cpu().numpy(), 2, 30, p, q, 42, n_threads=8)
    assert np.array_equal(gv.cpu().numpy(), wv) and np.array_equal(got.cpu().numpy(), want)


def test_cfg5_shape_bipartite_hubs(oracle):
    """BASELINE cfg 5 in miniature: hubs of ~10 k leaves + one hub per leaf, p=4 q=0.25.
    Bipartite => the 'shared neighbour' branch never fires; hub rows exceed the LDS
    class cache; the return slot is the only underfull one."""
    from node2vec_amd import randomwalk as rw
    from node2vec_amd import synthetic

    g = synthetic.hub_bipartite(300_000, 30, 10_000, device="cuda")
    deg = g.degrees()
    assert int(deg.max()) > 8192 * 2
    start = rw.start_vertices(g)
    sample = torch.cat([torch.arange(30, dtype=torch.int32, device="cuda"), start[30::997]])
    got, gv = rw.walk(g, sample, 3, 30, 4.0, 0.25, 7)
    want, wv = oracle.random_walk(g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.w.cpu().numpy(),
                                  sample.cpu().numpy(), 3, 30, 4.0, 0.25, 7, n_threads=8)
    assert np.array_equal(gv.cpu().numpy(), wv) and np.array_equal(got.cpu().numpy(), want)
    assert _all_hops_are_edges(g, got, gv)
    # walks alternate between the hub side (ids < 30) and the leaf side
    w = got[gv].long()
    side = w < 30
    assert bool((side[:, :-1] != side[:, 1:]).all())


def test_exact_and_fast_visit_the_same_distribution():
    """degree-bucketed visit frequencies of the two samplers agree (same Markov chain)"""
    from node2vec_amd import randomwalk as rw
    from node2vec_amd import synthetic

    g = synthetic.rmat(16, 400_000, device="cuda", weights="uniform")
    start = rw.start_vertices(g)
    a, _ = rw.walk(g, start, 6, 30, 0.5, 2.0, 1, mode="exact")
    b, _ = rw.walk(g, start, 6, 30, 0.5, 2.0, 2, mode="fast")
    deg = g.degrees()
    bucket = torch.log2(deg.clamp(min=1).double()).long()
    ha = torch.bincount(bucket[a[:, 1:].long()].reshape(-1), minlength=20).double()
    hb = torch.bincount(bucket[b[:, 1:].long()].reshape(-1), minlength=20).double()
    ha, hb = ha / ha.sum(), hb / hb.sum()
    assert float((ha - hb).abs().max()) < 5e-3


@pytest.mark.parametrize("pq", [(0.5, 2.0), (4.0, 0.25), (0.7, 1.3)])
def test_layered_fast_sampler_matches_exact_walks_at_scale(pq):
    """unit-weight R-MAT graph (2.6 x 10^5 vertices), all per-edge tables: the layered sampler of
    fast mode and the exact sampler are the same Markov chain -- over 6 x 10^6 steps each (fixed
    seeds: nothing here is random between runs) the share of RETURN moves (x == s: what round 2's
    fast sampler got wrong) agrees within 5 binomial sd, the degree-bucketed visit frequencies
    within 5e-3, and the share of moves that close a triangle (x in N(s), x != s) within 5 sd"""
    from node2vec_amd import randomwalk as rw
    from node2vec_amd import synthetic

    p, q = pq
    g = synthetic.rmat(18, 1_500_000, device="cuda")
    assert g.unit_weights
    start = rw.start_vertices(g)
    a, av = rw.walk(g, start, 1, 30, p, q, 1, mode="exact")
    b, bv = rw.walk(g, start, 1, 30, p, q, 2, mode="fast")
    assert g.wedge_off is not None  # fast mode walked the layers of the tables
    assert bool(av.all()) and bool(bv.all())
    keys = (torch.repeat_interleave(torch.arange(g.n_vertices, device="cuda"), g.degrees()) * g.n_vertices
            + g.col.long())  # ascending: rows are sorted

    def shares(w):
        w = w.long()
        ret = (w[:, 2:] == w[:, :-2])
        probe = (w[:, :-2] * g.n_vertices + w[:, 2:]).reshape(-1)  # is (s -> x) an edge?
        pos = torch.searchsorted(keys, probe).clamp_(max=keys.numel() - 1)
        tri = (keys[pos] == probe).reshape(ret.shape) & ~ret
        return float(ret.double().mean()), float(tri.double().mean()), ret.numel()

    (ra, ta, na), (rb, tb, nb) = shares(a), shares(b)
    for xa, xb in ((ra, rb), (ta, tb)):
        pbar = (xa * na + xb * nb) / (na + nb)
        z = (xa - xb) / (pbar * (1 - pbar) * (1 / na + 1 / nb)) ** 0.5
        assert abs(z) < 5.0, (pq, xa, xb, z)
    assert ra > 0.01 and ta > 0.001  # both events do occur
    bucket = torch.log2(g.degrees().clamp(min=1).double()).long()
    ha = torch.bincount(bucket[a[:, 1:].long()].reshape(-1), minlength=20).double()
    hb = torch.bincount(bucket[b[:, 1:].long()].reshape(-1), minlength=20).double()
    assert float((ha / ha.sum() - hb / hb.sum()).abs().max()) < 5e-3


def test_weighted_generic_kernel_large_rows(oracle):
    """weighted graph (generic kernel): hubs beyond the 1024-entry weight cache"""
    from node2vec_amd import randomwalk as rw
    from node2vec_amd import synthetic

    g = synthetic.rmat(15, 300_000, device="cuda", weights="uniform")
    assert not g.unit_weights and int(g.degrees().max()) > 2048
    hubs = torch.topk(g.degrees(), 30).indices.to(torch.int32)
    sample = torch.unique(torch.cat([hubs, rw.start_vertices(g)[::97]]))
    got, gv = rw.walk(g, sample, 3, 25, 0.5, 2.0, 11)
    want, wv = oracle.random_walk(g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.w.cpu().numpy(),
                                  sample.cpu().numpy(), 3, 25, 0.5, 2.0, 11, n_threads=8)
    assert np.array_equal(gv.cpu().numpy(), wv) and np.array_equal(got.cpu().numpy(), want)


def test_cfg2_full_size_sgns_properties(oracle):
    """K3 on the bench corpus at full size (471 040 rows of 81 tokens, 1 M-row model, dim 128):
    the number of trained pairs depends only on (seed, sentence ids) -- equal across launches,
    equal to the sum over row blocks launched separately, and equal to the oracle on a block
    of rows; with alpha = 0 the model comes back bit-identical; with alpha > 0 every row that
    occurs in the corpus moves and everything stays finite"""
    from node2vec_amd import randomwalk as rw
    from node2vec_amd import sgns, synthetic

    g = synthetic.rmat(20, 5_000_000, device="cuda")
    start = rw.start_vertices(g)[:47104].contiguous()
    walks, valid = rw.walk(g, start, 10, 80, 0.5, 2.0, 42)
    deg = g.degrees().clamp(min=1)
    order = torch.sort(deg, descending=True, stable=True).indices
    index_of = torch.empty(g.n_vertices, dtype=torch.int32, device="cuda")
    index_of[order] = torch.arange(g.n_vertices, dtype=torch.int32, device="cuda")
    m = sgns.SgnsModel(sgns.Vocab(order, deg[order], index_of), 128, 5, 5, seed=3, sample=0.0)
    idx = index_of[walks[valid].long()].contiguous()
    s0, s1 = m.syn0.clone(), m.syn1neg.clone()
    m.train_block(idx, 0.0, 0)  # alpha = 0: g = 0 for every target
    torch.cuda.synchronize()
    whole = int(m.pairs.item())
    assert torch.equal(m.syn0, s0) and torch.equal(m.syn1neg, s1)
    m.pairs.zero_()
    cuts = [0, 1000, 250000, 471040]
    for a, b in zip(cuts, cuts[1:]):
        m.train_block(idx[a:b].contiguous(), 0.0, a)  # sentence ids continue at a
    torch.cuda.synchronize()
    assert int(m.pairs.item()) == whole
    blk = idx[250000:251500].contiguous()
    m.pairs.zero_()
    m.train_block(blk, 0.0, 250000)
    torch.cuda.synchronize()
    o0, o1 = s0.cpu().numpy().copy(), s1.cpu().numpy().copy()
    n = oracle.sgns_train(blk.cpu().numpy(), o0, o1, m.cum_table.cpu().numpy(), None,
                          sgns.exp_table(), len(m.vocab), 250000, m.seed, 128, 5, 5, 0.0)
    assert int(m.pairs.item()) == n
    m.pairs.zero_()
    m.train_block(idx, 0.025, 0)
    torch.cuda.synchronize()
    assert int(m.pairs.item()) == whole
    assert bool(torch.isfinite(m.syn0).all()) and bool(torch.isfinite(m.syn1neg).all())
    seen = torch.zeros(g.n_vertices, dtype=torch.bool, device="cuda")
    seen[idx.long().reshape(-1)] = True
    moved = (m.syn0 != s0).any(1)
    assert bool(moved[seen].all()) and not bool(moved[~seen].any())


def test_trim_at_scale_properties():
    """trim_hotspot_vertices on a power-law graph of 2 M vertices (cfg 3's generator): every row
    above the cap keeps exactly `cap` edges, rows at or below it keep all, the mask depends only
    on the seed, and two seeds keep different samples of the same size"""
    from node2vec_amd import synthetic
    from node2vec_amd.fugue import trim_hotspot_edges

    g = synthetic.chung_lu(2_000_000, 20_000_000, device="cuda")
    deg = g.degrees()
    cap = 1000
    assert int(deg.max()) > 20 * cap
    src = torch.repeat_interleave(torch.arange(g.n_vertices, device="cuda"), deg)
    keep = trim_hotspot_edges(src, cap, 42)
    kept = torch.bincount(src[keep], minlength=g.n_vertices)
    assert torch.equal(kept, deg.clamp(max=cap))
    assert torch.equal(keep, trim_hotspot_edges(src, cap, 42))
    other = trim_hotspot_edges(src, cap, 43)
    assert int(other.sum()) == int(keep.sum()) and not torch.equal(other, keep)
    # the kept sample of the biggest row is spread over the whole row, not a prefix
    hub = int(torch.argmax(deg))
    lo, hi = int(g.rowptr[hub]), int(g.rowptr[hub + 1])
    pos = torch.nonzero(keep[lo:hi]).flatten().double() / (hi - lo)
    assert 0.4 < float(pos.mean()) < 0.6 and float(pos.max()) > 0.95 and float(pos.min()) < 0.05


def test_alias_tables_at_scale_encode_the_row_distribution():
    """K1 on a weighted R-MAT graph (262 k vertices, ~2.3 M edges): the table of every row,
    read back as a distribution -- slot i gives its own neighbour with prob_i / n and the
    neighbour behind its alias with (1 - prob_i) / n -- is w / sum(w) of that row to 1e-12"""
    from node2vec_amd import synthetic

    g = synthetic.rmat(18, 1_200_000, device="cuda", weights="uniform").build_alias()
    deg = g.degrees()
    row = torch.repeat_interleave(torch.arange(g.n_vertices, device="cuda"), deg)
    n = deg[row].double()
    prob = g.prob.double()
    keys = row * g.n_vertices + g.col.long()  # unique: the generator de-duplicates edges
    assert bool((keys[1:] > keys[:-1]).all())
    mass = prob / n
    alias_key = row * g.n_vertices + g.alias.long()
    pos = torch.searchsorted(keys, alias_key)
    assert bool((keys[pos.clamp(max=keys.numel() - 1)] == alias_key).all())  # alias is a neighbour
    mass = mass.index_add(0, pos, (1.0 - prob) / n)
    w = g.w.double()
    total = torch.zeros(g.n_vertices, dtype=torch.float64, device="cuda").index_add(0, row, w)
    assert float((mass - w / total[row]).abs().max()) < 1e-12
