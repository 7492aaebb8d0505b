"""Statistical parity of the full-speed (hogwild) trainer (last row; call site
embedding.py:126).  Bit parity is impossible under unsynchronised updates (as it is between two
gensim runs with workers > 1), and no gensim-produced vector exists to compare with -- the SGNS
half stays "parity unpinned" (DESIGN.md).  What is asserted, with the tolerances STATED here:

karate club (BASELINE cfg 1), reference-style defaults (iter = 10, sample = 1e-3, min_count = 10,
window 5, k = 5, dim 16), 5 seeds -- hogwild against the deterministic single-wave run that IS
bit-identical to the oracle (measured: identical vectors -- the trainer scales its concurrency
to the vocabulary, one wave per 32 rows, so a 34-row model is trained by one wave in order;
the planted-partition test of tests/test_sgns_gpu.py is the case with real concurrency):
  * mean per-vertex cosine after Procrustes alignment      >= 0.80   (measured 1.00)
  * overlap of the 5 nearest neighbours of every vertex    >= 0.55   (measured 1.00; chance 0.15)
  * edge-vs-non-edge AUC of the cosine similarity          |hogwild - deterministic| <= 0.03
R-MAT scale 20 (BASELINE cfg 2, all 471 k start vertices x 10 walks x 80 steps, 2.2 G pairs per
epoch, dim 128), 5 seeds, hogwild only (one wave would need hours).  Tolerances re-derived in
round 3 from 20 runs (profiles/r3a_hogwild_auc_runs.log, scripts/r3/hogwild_auc_runs.py): link AUC
mean 0.8970, sd 0.0024, min 0.8915, max 0.9020 -- round 2 had asserted a 5-seed spread <= 0.01
from ONE measurement (0.0026); the expected range of 5 draws at sd 0.0024 is 0.0056 and 0.01 is
exceeded about one time in 20, which is what the driver's box saw (0.0112).  The variance is
between seeds, not between launch geometries: capped at 2048 / 512 concurrent waves the same
statistic has sd 0.0020 / 0.0024 around 0.8974 / 0.8991.  (Capped at 64 waves -- gensim's
regime -- it is 0.908 .. 0.914: lost updates on hub rows cost ~0.014 AUC at the full 8192 waves;
DESIGN.md "hogwild".)
  * edge-vs-random-pair AUC of every seed within mean +- 6 sd: 0.8826 .. 0.9114
  * spread (max - min) of the 5 seeds <= 0.025 (10 sd; informative, not tight)
  * 10-nearest-neighbour overlap between seeds, over 2 000 probe vertices of degree >= 20: >= 0.45
    (20 runs: 0.559 .. 0.567; chance: 10 / 471 k)
"""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = [pytest.mark.gpu, pytest.mark.statistical]


def _procrustes_cos(x, y):
    x, y = x - x.mean(0), y - y.mean(0)
    u, _, vt = np.linalg.svd(x.T @ y)
    xr = x @ (u @ vt)
    return float(np.mean((xr * y).sum(1) / (np.linalg.norm(xr, axis=1) * np.linalg.norm(y, axis=1) + 1e-30)))


def _unit(v):
    v = v - v.mean(0)
    return v / (np.linalg.norm(v, axis=1, keepdims=True) + 1e-30)


def _knn(v, k, probe=None):
    u = _unit(v)
    s = u @ u.T if probe is None else u[probe] @ u.T
    rows = np.arange(s.shape[0])
    s[rows, rows if probe is None else probe] = -np.inf
    return np.argsort(-s, axis=1)[:, :k]


def _overlap(a, b):
    return float(np.mean([len(set(x) & set(y)) / len(x) for x, y in zip(a, b)]))


def test_karate_hogwild_vs_deterministic_five_seeds():
    from node2vec_amd import randomwalk as rw
    from node2vec_amd import sgns
    from node2vec_amd.graph import DeviceGraph

    edges = np.array(load_golden("karate_edges.json"), dtype=np.float64).reshape(-1, 3)
    g = DeviceGraph.from_edges(edges[:, 0].astype(np.int64), edges[:, 1].astype(np.int64), None,
                               n_vertices=34, device="cuda")
    adj = np.zeros((34, 34), bool)
    adj[edges[:, 0].astype(int), edges[:, 1].astype(int)] = True
    cos, knn, dauc = [], [], []
    for seed in range(5):
        walks, _ = rw.walk(g, rw.start_vertices(g), 10, 10, 1.0, 1.0, 100 + seed)  # cfg 1: W = 10, L = 10
        vocab = sgns.build_vocab(walks, 10)
        assert len(vocab) == 34
        idx = vocab.index_of[walks.long()]
        order = np.argsort(vocab.ids.cpu().numpy())

        def train(det):
            m = sgns.SgnsModel(vocab, 16, 5, 5, seed=seed, sample=1e-3)
            m.train(idx, epochs=10, alpha=0.025, min_alpha=1e-4, deterministic=det)
            torch.cuda.synchronize()
            return m.syn0.cpu().numpy()[order]  # row = vertex id

        def auc(v):
            s = _unit(v) @ _unit(v).T
            iu = np.triu_indices(34, 1)
            pos, neg = s[iu][adj[iu]], s[iu][~adj[iu]]
            return float((pos[:, None] > neg[None, :]).mean())

        det, hog = train(True), train(False)
        cos.append(_procrustes_cos(det, hog))
        knn.append(_overlap(_knn(det, 5), _knn(hog, 5)))
        dauc.append(abs(auc(det) - auc(hog)))
        assert auc(det) > 0.8 and auc(hog) > 0.8
    print("karate: procrustes cosine", cos, "knn@5 overlap", knn, "|dAUC|", dauc)
    assert min(cos) >= 0.80
    assert min(knn) >= 0.55
    assert max(dauc) <= 0.03


def test_rmat_1m_hogwild_quality_is_stable_over_five_seeds():
    from node2vec_amd import randomwalk as rw
    from node2vec_amd import sgns
    from node2vec_amd import synthetic

    g = synthetic.rmat(20, 5_000_000, device="cuda")
    start = rw.start_vertices(g)
    walks, valid = rw.walk(g, start, 10, 80, 0.5, 2.0, 42)
    assert bool(valid.all())
    vocab = sgns.build_vocab(walks, 10)
    idx = vocab.index_of[walks.long()]
    del walks
    index_of = vocab.index_of
    deg = g.degrees()
    gen = torch.Generator(device="cuda").manual_seed(1)
    # positive pairs: 200 k random edges; negatives: 200 k random vertex pairs (both in vocabulary)
    e = torch.randint(0, g.n_edges, (200_000,), generator=gen, device="cuda")
    src = torch.repeat_interleave(torch.arange(g.n_vertices, device="cuda"), deg)
    pa, pb = index_of[src[e]].long(), index_of[g.col[e].long()].long()
    del src
    ok = (pa >= 0) & (pb >= 0)
    pa, pb = pa[ok], pb[ok]
    na = torch.randint(0, len(vocab), (200_000,), generator=gen, device="cuda")
    nb = torch.randint(0, len(vocab), (200_000,), generator=gen, device="cuda")
    cand = torch.nonzero(deg[vocab.ids] >= 20).reshape(-1)
    probe = cand[torch.randperm(cand.numel(), generator=gen, device="cuda")[:2000]]
    aucs, nbrs = [], []
    for seed in range(5):
        m = sgns.SgnsModel(vocab, 128, 5, 5, seed=seed, sample=1e-3)
        m.hub_rows = 0  # plain stores everywhere: the statistic the 20 logged runs measured
        m.train(idx, epochs=1, alpha=0.025, min_alpha=1e-4)
        torch.cuda.synchronize()
        u = m.syn0 - m.syn0.mean(0)
        u = u / (u.norm(dim=1, keepdim=True) + 1e-30)
        sp, sn = (u[pa] * u[pb]).sum(1), (u[na] * u[nb]).sum(1)
        k = min(sp.numel(), 20000)
        aucs.append(float((sp[:k, None] > sn[None, :2000]).float().mean()))
        s = u[probe] @ u.T
        s[torch.arange(probe.numel(), device="cuda"), probe] = -1e9
        nbrs.append(torch.topk(s, 10, dim=1).indices.cpu().numpy())
        del m, u, s
    # hub_rows = 4096 (atomic adds on the most frequent rows): no wave overwrites what another
    # learned on a hub; measured 0.9094 +- 0.0015 over 5 seeds (profiles/r3k_hogwild_auc_hub_rows.log),
    # the level of the same trainer capped at 64 waves (0.908 .. 0.914) -- asserted at +- 6 sd
    # the DEFAULT (hub_rows None -> SgnsModel.auto_hub_rows: the rows held by >= 1.5 of the 8 192
    # waves at a time, 2 000 - 4 100 on this corpus): measured 0.9093 +- 0.0025 at 2 048 rows and
    # 0.9094 +- 0.0015 at 4 096 (profiles/r4n_hogwild_auc_hub_rows_knee.log, r3k_*) -- the level of
    # gensim's <= 16-thread regime -- for +17 .. +30 % of the epoch time; asserted in the same band
    # as the explicit 4096 below
    auto = []
    for seed in range(2):
        m = sgns.SgnsModel(vocab, 128, 5, 5, seed=seed, sample=1e-3)
        assert m.hub_rows is None
        m.train(idx, epochs=1, alpha=0.025, min_alpha=1e-4)
        torch.cuda.synchronize()
        assert 1300 <= m.hub_rows <= 4200, m.hub_rows
        u = m.syn0 - m.syn0.mean(0)
        u = u / (u.norm(dim=1, keepdim=True) + 1e-30)
        sp, sn = (u[pa] * u[pb]).sum(1), (u[na] * u[nb]).sum(1)
        k = min(sp.numel(), 20000)
        auto.append(float((sp[:k, None] > sn[None, :2000]).float().mean()))
        del m, u
    print("rmat-1m: link AUC with the default hub rows", auto)
    assert 0.9004 <= min(auto) and max(auto) <= 0.9240, auto
    hub = []
    for seed in range(2):
        m = sgns.SgnsModel(vocab, 128, 5, 5, seed=seed, sample=1e-3)
        m.hub_rows = 4096
        m.train(idx, epochs=1, alpha=0.025, min_alpha=1e-4)
        torch.cuda.synchronize()
        u = m.syn0 - m.syn0.mean(0)
        u = u / (u.norm(dim=1, keepdim=True) + 1e-30)
        sp, sn = (u[pa] * u[pb]).sum(1), (u[na] * u[nb]).sum(1)
        k = min(sp.numel(), 20000)
        hub.append(float((sp[:k, None] > sn[None, :2000]).float().mean()))
        del m, u
    print("rmat-1m: link AUC with hub_rows = 4096", hub)
    assert 0.9004 <= min(hub) and max(hub) <= 0.9240, hub  # above: +6 sd at the default trainer's sd
    ov = [_overlap(nbrs[0], nbrs[i]) for i in range(1, 5)]
    print("rmat-1m: link AUC per seed", aucs, "knn@10 overlap vs seed 0", ov)
    assert 0.8826 <= min(aucs) and max(aucs) <= 0.9114, aucs  # mean +- 6 sd of 20 runs
    assert max(aucs) - min(aucs) <= 0.025, aucs
    assert min(ov) >= 0.45
