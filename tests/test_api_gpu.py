"""End-to-end through the reference-shaped Python API on the GPU: BASELINE cfg 1
(karate club, p=q=1, walk_len=10, dim=16) and the assertions of the reference's
tests/test_fugue.py and tests/test_embedding.py."""
import os

import numpy as np
import pandas as pd
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _karate_df():
    e = load_golden("karate_edges.json")
    return pd.DataFrame(e, columns=["src", "dst", "weight"])


def test_cfg1_karate_walks_equal_reference_and_embed():
    from node2vec_amd.embedding import HipW2V, Node2VecHIP
    from node2vec_amd.fugue import random_walk

    params = {"num_walks": 10, "walk_length": 10, "return_param": 1.0, "inout_param": 1.0}
    df_walks = random_walk("hip", _karate_df(), params, random_seed=42)
    assert list(df_walks.columns) == ["src", "walk"] and len(df_walks) == 340
    assert all(len(w) == 11 and w[0] == s for s, w in zip(df_walks["src"], df_walks["walk"]))
    # identical to the walks the reference itself produced with this uniform stream (G4)
    g4 = [c for c in load_golden("g4_walks.json") if c["name"] == "karate_p1.0_q1.0"][0]
    want = sorted(w["walk"] for w in g4["walks"])
    assert sorted(df_walks["walk"].tolist()) == want

    w2v = {"min_count": 0, "iter": 40, "size": 16, "negative": 5}
    n2v = Node2VecHIP(df_walks, w2v, random_seed=1000)
    with pytest.raises(ValueError):
        n2v.embedding()
    model = n2v.fit()
    assert isinstance(model, HipW2V) and model.pairs_trained > 0
    emb = n2v.embedding()
    assert list(emb.columns) == ["id", "vector"] and len(emb) == 34
    assert all(len(v) == 16 for v in emb["vector"])
    assert len(n2v.get_vector(vertex_id="0")) == 16 and len(n2v.get_vector(vertex_id=1)) == 16
    # structure is learnt: adjacent vertices are closer than non-adjacent ones
    ids = emb["id"].to_numpy()
    V = np.array(emb["vector"].tolist())
    V = V - V.mean(axis=0, keepdims=True)  # remove the common direction SGNS gives tiny graphs
    V = V / np.linalg.norm(V, axis=1, keepdims=True)
    S = V @ V.T
    adj = np.zeros((34, 34), bool)
    for a_, b_, _ in load_golden("karate_edges.json"):
        adj[a_, b_] = True
    A = adj[np.ix_(ids, ids)]
    off = ~np.eye(34, dtype=bool)
    assert S[A].mean() > S[~A & off].mean() + 0.1


def test_save_load_round_trip(tmp_path):
    """tests/test_embedding.py:64-72"""
    from node2vec_amd.embedding import HipW2V, KeyedVectors, Node2VecHIP

    df = pd.DataFrame.from_dict({"walk": [[0, 1, 1, 0, 3, 4], [1, 2, 3, 2, 0, 4], [2, 3, 1, 0, 4, 4]]})
    params = {"min_count": 0, "iter": 1, "seed": 1000, "batch_words": 1, "size": 4, "workers": 4}
    n2v = Node2VecHIP(df, w2v_params=params)
    assert isinstance(n2v.fit(), HipW2V)
    df_res = n2v.embedding()
    assert isinstance(df_res, pd.DataFrame) and len(df_res) > 0 and list(df_res.columns) == ["id", "vector"]
    n2v.save_model(str(tmp_path), "tmp")
    assert os.path.exists(tmp_path / "tmp.model")
    assert isinstance(n2v.load_model(str(tmp_path), "tmp"), HipW2V)
    n2v.save_vectors(str(tmp_path), "tmp_vec")
    assert isinstance(n2v.load_vectors(str(tmp_path), "tmp_vec"), KeyedVectors)
    name_id = pd.DataFrame.from_dict({"name": ["a", "b", "c", "d", "e"], "id": [0, 1, 2, 3, 4]})
    n2v = Node2VecHIP(df, params, name_id=name_id)
    with pytest.raises(ValueError):
        n2v.embedding()
    n2v.fit()
    res = n2v.embedding()
    assert list(res.columns) == ["name", "vector"] and set(res["name"]) <= set("abcde")


def test_trim_index_counts():
    """tests/test_fugue.py:23-28, 38-41: cap 1 keeps one edge per source"""
    from node2vec_amd.fugue import trim_index

    df = pd.DataFrame({"src": ["a1", "a1", "a1", "a2", "a5", "b2"],
                       "dst": ["a5", "b2", "b6", "b2", "b2", "b6"]})
    e, name_id = trim_index(None, df, indexed=False, directed=True, max_out_deg=1, random_seed=1)
    assert len(e) == 4 and len(name_id) <= 5 and list(e.columns) == ["src", "dst", "weight"]
    e, name_id = trim_index(None, df, indexed=False, directed=False, max_out_deg=0)
    assert len(e) == 12 and len(name_id) == 5
    dfi = pd.DataFrame({"src": [0, 0, 0, 1], "dst": [1, 2, 3, 0], "weight": [1.0, 2.0, 3.0, 1.0]})
    e, none = trim_index(None, dfi, indexed=True, max_out_deg=2, random_seed=3)
    assert none is None and len(e) == 3 and set(map(tuple, e.values.tolist())) <= set(map(tuple, dfi.values.tolist()))


def test_random_walk_walk_seed_and_sinks():
    """fugue.py:132-134, 147: walk_seed filter and the sink drop (G7)"""
    from node2vec_amd.fugue import random_walk

    df = pd.DataFrame({"src": [0, 1, 3], "dst": [1, 2, 0], "weight": [1.0, 1.0, 1.0]})
    out = random_walk(None, df, {"num_walks": 1, "walk_length": 2}, random_seed=42)
    assert sorted(out["walk"].tolist()) == [[0, 1, 2], [3, 0, 1]]
    out = random_walk(None, df, {"num_walks": 2, "walk_length": 1}, random_seed=1,
                      walk_seed=pd.DataFrame({"id": [3, 2]}))
    assert out["src"].tolist() == [3, 3]
    with pytest.raises(ValueError):
        random_walk(None, df, {}, walk_seed=pd.DataFrame({"x": [1]}))
    with pytest.raises(ValueError):
        random_walk(None, df, {"return_param": 0.0, "walk_length": 3})


def test_on_device_corpus_feeds_sgns_without_dataframe():
    """walks stay in HBM from K2 to K3"""
    from node2vec_amd import synthetic
    from node2vec_amd.embedding import Node2VecHIP
    from node2vec_amd.fugue import random_walk_tensors

    g = synthetic.rmat(11, 20000, device="cuda")
    walks, valid = random_walk_tensors(g, {"num_walks": 4, "walk_length": 20, "return_param": 0.5,
                                           "inout_param": 2.0}, random_seed=3, mode="fast")
    assert walks.is_cuda and bool(valid.all())
    n2v = Node2VecHIP(walks, {"min_count": 1, "iter": 1, "size": 64, "sample": 1e-3}, random_seed=9)
    m = n2v.fit()
    assert m.wv.vectors.shape[1] == 64 and np.isfinite(m.wv.vectors).all()
    assert len(m.wv.vocab) == int(torch.unique(walks).numel())


def test_three_stage_example_pipeline(tmp_path):
    """examples/hip_pipeline.py: index | walk | embed through parquet stage files"""
    import subprocess
    import sys

    from conftest import ROOT

    names = [f"v{i}" for i in range(40)]
    rng = np.random.default_rng(0)
    df = pd.DataFrame({"src": rng.choice(names, 300), "dst": rng.choice(names, 300)})
    df = df[df["src"] != df["dst"]]
    csv = tmp_path / "edges.csv"
    df.to_csv(csv, index=False)
    work = str(tmp_path / "work")
    for stage, extra in (("index", [str(csv)]), ("walk", []), ("embed", [])):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "examples", "hip_pipeline.py"), stage, work] + extra)
    vec = pd.read_parquet(os.path.join(work, "graph_vectors.parquet"))
    assert list(vec.columns) == ["name", "vector"] and set(vec["name"]) <= set(names)
    assert len(vec) > 30 and all(len(v) == 128 for v in vec["vector"])
    assert open(os.path.join(work, "vectors.w2v")).readline().split()[1] == "128"


def test_streaming_pipeline_matches_materialised_corpus():
    """fit_streaming never holds the corpus; its vocabulary, counts and pair count equal
    those of training on the materialised walks, and with one deterministic wave per block
    the vectors are identical too."""
    from node2vec_amd import sgns, synthetic
    from node2vec_amd.fugue import random_walk_tensors
    from node2vec_amd.pipeline import fit_streaming

    g = synthetic.rmat(9, 3000, device="cuda")
    n2v = {"num_walks": 3, "walk_length": 12, "return_param": 0.5, "inout_param": 2.0}
    w2v = {"min_count": 2, "iter": 2, "size": 32, "negative": 5, "sample": 1e-2, "seed": 5}
    out, model = fit_streaming(g, dict(n2v), dict(w2v), random_seed=17, batch_vertices=100,
                               return_model=True)
    walks, valid = random_walk_tensors(g, dict(n2v), random_seed=17)
    vocab = sgns.build_vocab(walks[valid], 2)
    assert torch.equal(vocab.ids, model.vocab.ids) and torch.equal(vocab.counts, model.vocab.counts)
    assert len(out.wv.vocab) == len(vocab) and out.wv.vectors.shape == (len(vocab), 32)
    # same schedule on the materialised corpus: blocks of 100 start vertices x 3 walks
    ref = sgns.SgnsModel(vocab, 32, 5, 5, seed=5, sample=1e-2)
    idx = torch.where(valid.unsqueeze(1), vocab.index_of[walks.long()], torch.full_like(walks, -1))
    rows, done = idx.shape[0], 0
    for ep in range(2):
        for lo in range(0, rows, 300):
            a = max(1e-4, 0.025 - (0.025 - 1e-4) * (done / (2 * rows)))
            ref.train_block(idx[lo:lo + 300], a, ep * rows + lo)
            done += idx[lo:lo + 300].shape[0]
    torch.cuda.synchronize()
    assert int(ref.pairs.item()) == out.pairs_trained
    assert np.isfinite(out.wv.vectors).all()


@pytest.mark.parametrize("pq", [(0.5, 2.0), (1.0, 1.0)])
def test_fit_streaming_deterministic_equals_the_oracle_end_to_end(oracle, pq):
    """(at p = q = 1 the walks run on the degree-ranked form and come out in ranks: counts and the
    per-token lookup are translated once, nothing else changes)
    walk -> vocabulary -> SGNS through fit_streaming (HIP, "deterministic": True) against the
    same pipeline made of the ORACLE's walk, a numpy restatement of gensim's vocabulary / negative
    table / subsampling rule, and the oracle's trainer on the same batches: identical vocabulary,
    pair count and matrices, bit for bit.  (The initial syn0 is the seeded device draw of
    sgns.init_syn0 -- gensim's per-word hash seed is not reproducible across processes.)"""
    from node2vec_amd import sgns
    from node2vec_amd.graph import DeviceGraph
    from node2vec_amd.pipeline import fit_streaming

    rng = np.random.default_rng(8)  # directed, 5 hubs, the last 20 vertices have no out-edges
    src = np.concatenate([rng.integers(0, 580, 3500), rng.integers(0, 5, 500)])
    g = DeviceGraph.from_edges(src, rng.integers(0, 600, 4000), None, n_vertices=600, device="cuda")
    W, L, (p, q), seed, bv = 3, 12, pq, 17, 100
    n2v = {"num_walks": W, "walk_length": L, "return_param": p, "inout_param": q}
    w2v = {"min_count": 2, "iter": 2, "size": 32, "negative": 5, "sample": 1e-2, "seed": 5, "window": 4,
           "alpha": 0.025, "min_alpha": 1e-4, "deterministic": True}
    out, model = fit_streaming(g, dict(n2v), dict(w2v), random_seed=seed, batch_vertices=bv,
                               return_model=True)
    assert (g.rank_hops is not None) == (pq == (1.0, 1.0))
    # ---- the same with the oracle ------------------------------------------------------------
    rowptr, col = g.rowptr.cpu().numpy(), g.col.cpu().numpy()
    start = np.nonzero(np.diff(rowptr) > 0)[0].astype(np.int32)  # fugue.py:132
    walks, valid = oracle.random_walk(rowptr, col, None, start, W, L, p, q, seed)
    assert not valid.all()  # walkers vanished at sinks: their rows are no sentences
    counts = np.bincount(walks[valid].reshape(-1), minlength=g.n_vertices)
    ids = np.nonzero(counts >= 2)[0]
    order = np.argsort(-counts[ids], kind="stable")  # gensim: most frequent first (ties: by id)
    ids, cnt = ids[order], counts[ids][order]
    assert np.array_equal(model.vocab.ids.cpu().numpy(), ids)
    assert np.array_equal(model.vocab.counts.cpu().numpy(), cnt)
    pw = cnt.astype(np.float64) ** 0.75  # make_cum_table (gensim word2vec.py): domain 2^31 - 1
    cum = np.round(np.cumsum(pw) / pw.sum() * (2 ** 31 - 1)).astype(np.int64)
    cum[-1] = 2 ** 31 - 1
    assert np.array_equal(model.cum_table.cpu().numpy().astype(np.int64), cum)
    thr = 1e-2 * cnt.sum()  # sample < 1: a fraction of the retained words
    keep = np.minimum((np.sqrt(cnt / thr) + 1.0) * (thr / cnt), 1.0)
    sample_int = np.minimum(np.round(keep * 2.0 ** 32), 2.0 ** 32 - 1).astype(np.uint32)
    assert np.array_equal(model.sample_int.cpu().numpy().view(np.uint32), sample_int)
    index_of = np.full(g.n_vertices, -1, np.int32)
    index_of[ids] = np.arange(len(ids), dtype=np.int32)
    idx = np.where(valid[:, None], index_of[np.clip(walks, 0, None)], -1).astype(np.int32)
    s0 = sgns.init_syn0(len(ids), 32, 5, "cuda").cpu().numpy()
    s1 = np.zeros_like(s0)
    n_batches = -(-len(start) // bv)
    pairs = 0
    # the learning rate falls per JOB as in gensim (word2vec.py _job_producer / _get_next_alpha with
    # the reference's batch_words = 1000, constants.py:58): restated here in plain Python floats
    total, job_rows = len(start) * W, 1000 // (L + 1)

    def job_alpha(row, ep):
        pushed = (row // job_rows) * job_rows
        progress = (ep + 1.0 * pushed / total) / 2
        return float(np.float32(max(1e-4, 0.025 - (0.025 - 1e-4) * progress)))

    for ep in range(2):
        for k in range(n_batches):
            lo, hi = k * bv * W, min((k + 1) * bv * W, total)
            for j0 in range(lo - lo % job_rows, hi, job_rows):  # the jobs that overlap this batch
                a, b = max(j0, lo), min(j0 + job_rows, hi)
                pairs += oracle.sgns_train(idx[a:b], s0, s1, cum.astype(np.uint32), sample_int,
                                           sgns.exp_table(), len(ids), ep * total + a, 5, 32, 4, 5,
                                           job_alpha(a, ep))
    assert pairs == out.pairs_trained > 0
    assert np.array_equal(model.syn0.cpu().numpy(), s0)
    assert np.array_equal(model.syn1neg.cpu().numpy(), s1)
    assert np.array_equal(out.wv.vectors, s0) and np.abs(s1).max() > 0


def test_fit_streaming_raises_the_references_zero_division():
    """a visited row whose weights are all 0: generate_alias_tables divides by sum / n == 0
    (randomwalk.py:172-173).  fugue.random_walk raises it; the streaming pipeline walks with
    check=False (no per-batch host sync) and must still raise it, not train on truncated walks"""
    from node2vec_amd import randomwalk as rw
    from node2vec_amd.graph import DeviceGraph
    from node2vec_amd.pipeline import fit_streaming

    rng = np.random.default_rng(2)
    src, dst = rng.integers(0, 200, 2400), rng.integers(0, 200, 2400)
    w = rng.uniform(0.5, 2.0, 2400)
    w[src == 17] = 0.0  # every out-edge of vertex 17 weighs nothing
    g = DeviceGraph.from_edges(src, dst, w.astype(np.float32), n_vertices=200, device="cuda")
    n2v = {"num_walks": 2, "walk_length": 8, "return_param": 0.5, "inout_param": 2.0}
    w2v = {"min_count": 1, "iter": 1, "size": 32, "negative": 5, "window": 5}
    with pytest.raises(ZeroDivisionError):
        rw.walk(g, rw.start_vertices(g), 2, 8, 0.5, 2.0, 3)
    with pytest.raises(ZeroDivisionError):
        fit_streaming(g, dict(n2v), dict(w2v), random_seed=3, batch_vertices=64)


def test_corpus_count_and_index_equal_torch():
    """n2v_corpus_count / n2v_corpus_index (the passes between K2 and K3 of fit_streaming) against
    the framework ops they replace: dropped rows, negative tokens, out-of-range ids"""
    from node2vec_amd import sgns

    gen = torch.Generator().manual_seed(4)
    nv = 5000
    walks = torch.randint(-1, nv + 3, (3000, 41), generator=gen, dtype=torch.int32).cuda()
    valid = (torch.rand(3000, generator=gen) < 0.8).cuda()
    counts = torch.zeros(nv, dtype=torch.int64, device="cuda")
    counts[7] = 5  # accumulates
    sgns.corpus_count(walks, valid, counts)
    ok = valid.unsqueeze(1) & (walks >= 0) & (walks < nv)
    want = torch.bincount(walks[ok].long(), minlength=nv)
    want[7] += 5
    assert torch.equal(counts, want)
    c3 = torch.zeros(nv, dtype=torch.int64, device="cuda")
    c3[7] = 5
    sgns.corpus_count(walks, valid, c3, sort_above=1)  # the sort-based path of large batches
    assert torch.equal(c3, want)
    c2 = torch.zeros(nv, dtype=torch.int64, device="cuda")
    sgns.corpus_count(walks, None, c2)
    assert torch.equal(c2, torch.bincount(walks[(walks >= 0) & (walks < nv)].long(), minlength=nv))
    index_of = torch.randperm(nv, generator=gen).to(torch.int32).cuda()
    index_of[::7] = -1
    idx = sgns.corpus_index(walks, valid, index_of)
    ref = torch.where(ok, index_of[walks.clamp(0, nv - 1).long()], torch.full_like(walks, -1))
    assert idx.dtype == torch.int32 and torch.equal(idx, ref)
    assert sgns.corpus_index(walks[:0], valid[:0], index_of).shape == (0, 41)
