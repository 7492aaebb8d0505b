"""CPU tests of the host-side mirror of the reference interface (no GPU): the same
things the reference's tests pin for these layers (tests/test_constants.py,
tests/test_indexer.py, tests/test_fugue.py error paths, tests/test_embedding.py
constructor / base-class behaviour)."""
import os

import numpy as np
import pandas as pd
import pytest
import torch


def test_constants_match_reference_defaults():
    """tests/test_constants.py + constants.py:6-68"""
    from node2vec_amd import constants as c

    assert c.MAX_OUT_DEGREES == 100000 and c.NUM_PARTITIONS == 3000
    assert c.NODE2VEC_PARAMS == {"num_walks": 10, "walk_length": 20, "return_param": 1.0,
                                 "inout_param": 1.0}
    assert c.GENSIM_PARAMS == {"min_count": 10, "alpha": 0.025, "iter": 10, "seed": None,
                               "batch_words": 1000, "window": 5, "size": 128, "negative": 0,
                               "workers": 16}
    assert c.WORD2VEC_PARAMS["windowSize"] == 5 and c.WORD2VEC_PARAMS["vectorSize"] == 128
    assert set(c.WORD2VEC_PARAMS) == {"minCount", "numPartitions", "stepSize", "maxIter", "seed",
                                      "maxSentenceLength", "windowSize", "vectorSize"}


def _graph_df():
    # the 6-edge graph of tests/test_indexer.py / tests/test_fugue.py
    return pd.DataFrame({"src": ["a1", "a1", "a1", "a2", "a5", "b2"],
                         "dst": ["a5", "b2", "b6", "b2", "b2", "b6"]})


def test_index_graph_pandas_counts_and_errors():
    """tests/test_indexer.py:17-20, 39-42: 5 names -> 5 ids, edges x2 when undirected"""
    from node2vec_amd.indexer import index_graph_pandas

    df = _graph_df()
    e, name_id = index_graph_pandas(df.copy(), True)
    assert len(name_id) == 5 and len(e) == 6
    assert list(name_id.columns) == ["name", "id"] and list(e.columns) == ["src", "dst", "weight"]
    assert sorted(name_id["id"]) == list(range(5)) and (e["weight"] == 1.0).all()
    back = dict(zip(name_id["id"], name_id["name"]))
    assert [(back[s], back[d]) for s, d in zip(e["src"], e["dst"])] == list(zip(df["src"], df["dst"]))
    e2, _ = index_graph_pandas(df.copy(), False)
    assert len(e2) == 12
    with pytest.raises(ValueError):
        index_graph_pandas(df.rename(columns={"src": "source"}), True)


def test_trim_index_and_random_walk_validate_before_touching_the_gpu():
    """fugue.py:53-54 and :123-124"""
    from node2vec_amd.fugue import random_walk, trim_index

    with pytest.raises(ValueError):
        trim_index(None, pd.DataFrame({"src1": [0], "dst": [1]}))
    params = {"num_walks": 2}
    with pytest.raises(ValueError):
        random_walk(None, pd.DataFrame({"src": [0], "dst": [1], "weight": [1.0]}), params,
                    walk_seed=pd.DataFrame({"vid": [0]}))
    # defaults are filled into the CALLER's dict (fugue.py:120-122)
    assert params == {"num_walks": 2, "walk_length": 20, "return_param": 1.0, "inout_param": 1.0}


def test_kernels_fail_loudly_without_a_gpu():
    from node2vec_amd import randomwalk as rw
    from node2vec_amd.graph import DeviceGraph

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    g = DeviceGraph.from_edges([0, 1], [1, 0], [1.0, 1.0])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rw.walk(g, torch.tensor([0], dtype=torch.int32), 1, 2, 1.0, 1.0, 1)
    with pytest.raises(RuntimeError):
        g.build_alias()


def test_missing_library_raises_import_error(monkeypatch):
    from node2vec_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libn2v_hip.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_device_graph_csr_matches_presorted_adjacency():
    """fugue.py:130 partition(by=src, presort=dst): rows sorted by dst, multi-edges in
    input order, vertices without out-edges have empty rows"""
    from node2vec_amd.graph import DeviceGraph

    g = DeviceGraph.from_edges([3, 3, 0, 3, 0], [2, 0, 1, 0, 1], [1.0, 0.2, 5.0, 1.4, 6.0])
    assert g.rowptr.tolist() == [0, 2, 2, 2, 5]
    assert g.col.tolist() == [1, 1, 0, 0, 2]
    np.testing.assert_allclose(g.w.numpy(), [5.0, 6.0, 0.2, 1.4, 1.0], rtol=1e-7)
    assert g.degrees().tolist() == [2, 0, 0, 3]
    with pytest.raises(ValueError):
        DeviceGraph.from_edges([-1], [0], [1.0])
    with pytest.raises(KeyError):
        DeviceGraph.from_pandas(pd.DataFrame({"src": [0], "dst": [1]}))


def test_start_vertices_is_adjacency_ids_joined_with_walk_seed():
    """fugue.py:132-134"""
    from node2vec_amd import randomwalk as rw
    from node2vec_amd.graph import DeviceGraph

    g = DeviceGraph.from_edges([0, 1, 3], [1, 2, 0], [1.0, 1.0, 1.0])
    assert rw.start_vertices(g).tolist() == [0, 1, 3]
    assert rw.start_vertices(g, [3, 2, 3, 9, -1]).tolist() == [3]


def test_node2vec_base_is_abstract():
    """tests/test_embedding.py:17-31"""
    from node2vec_amd.embedding import Node2VecBase

    n2v = Node2VecBase()
    for call in (n2v.fit, n2v.embedding, lambda: n2v.get_vector(0),
                 lambda: n2v.save_model("file:///a", "b"), lambda: n2v.load_model("file:///a", "b")):
        with pytest.raises(NotImplementedError):
            call()


def test_node2vec_hip_constructor_contract():
    """tests/test_embedding.py:38-48, embedding.py:105-116, 133-134"""
    from node2vec_amd.embedding import Node2VecGensim, Node2VecHIP

    assert Node2VecGensim is Node2VecHIP
    df = pd.DataFrame.from_dict({"walk": [[0, 1, 1, 0, 3, 4], [1, 2, 3, 2, 0, 4], [2, 3, 1, 0, 4, 4]]})
    params = {}
    n2v = Node2VecHIP(df, params)
    assert set(params) >= {"min_count", "alpha", "iter", "seed", "batch_words", "window", "size",
                           "negative", "workers"}
    assert params["seed"] > 0 and n2v.w2v_params is params
    p2 = {"iter": 3}
    Node2VecHIP(df, w2v_params=p2, window_size=6, vector_size=64, random_seed=1000)
    assert (p2["window"], p2["size"], p2["seed"], p2["iter"]) == (6, 64, 1000, 3)
    for kw in ({"window_size": 3}, {"window_size": 31}, {"vector_size": 16}, {"vector_size": 2048}):
        with pytest.raises(ValueError):
            Node2VecHIP(df, {}, **kw)
    with pytest.raises(ValueError):
        Node2VecHIP(df, {"hs": 1})
    with pytest.raises(ValueError, match="Model is not available"):
        n2v.embedding()


def test_keyed_vectors_word2vec_text_round_trip(tmp_path):
    from node2vec_amd.embedding import HipW2V, KeyedVectors

    vec = np.random.default_rng(0).normal(size=(5, 8)).astype(np.float32)
    kv = KeyedVectors(["3", "0", "7", "1", "4"], vec)
    assert "7" in kv and np.array_equal(kv["7"], vec[2]) and list(kv.vocab) == ["3", "0", "7", "1", "4"]
    kv.save_word2vec_format(str(tmp_path / "v.txt"))
    assert open(tmp_path / "v.txt").readline() == "5 8\n"
    kv2 = KeyedVectors.load_word2vec_format(str(tmp_path / "v.txt"))
    assert kv2.index2word == kv.index2word and np.array_equal(kv2.vectors, vec)
    m = HipW2V(kv, vec * 2, {"size": 8}, 11)
    m.save(str(tmp_path / "m.model"))
    m2 = HipW2V.load(str(tmp_path / "m.model"))
    assert np.array_equal(m2.wv.vectors, vec) and m2.pairs_trained == 11 and os.path.exists(tmp_path / "m.model")


def test_sgns_host_tables():
    from node2vec_amd import sgns

    t = sgns.exp_table()
    assert t.dtype == np.float32 and len(t) == 1000
    assert abs(t[0] - 1 / (1 + np.exp(6.0))) < 1e-6 and abs(t[500] - 0.5) < 1e-6 and t[999] > 0.997
    walks = torch.tensor([[5, 5, 5, 2], [2, 9, 5, -1], [9, 2, 7, 7]], dtype=torch.int32)
    v = sgns.build_vocab(walks, min_count=2)
    assert v.ids.tolist() == [5, 2, 7, 9] and v.counts.tolist() == [4, 3, 2, 2]  # ties: id asc
    assert v.index_of.tolist() == [-1, -1, 1, -1, -1, 0, -1, 2, -1, 3]
    cum = sgns.make_cum_table(v.counts).numpy().astype(np.int64)
    p = np.array([4, 3, 2, 2], float) ** 0.75
    want = np.round(np.cumsum(p) / p.sum() * (2 ** 31 - 1)).astype(np.int64)
    assert cum.tolist() == want.tolist() and cum[-1] == 2 ** 31 - 1
    assert sgns.make_sample_int(v.counts, 0) is None
    si = sgns.make_sample_int(torch.tensor([1000, 10, 1]), 0.01).numpy().view(np.uint32).astype(np.int64)
    thr = 0.01 * 1011
    want0 = round((np.sqrt(1000 / thr) + 1) * (thr / 1000) * 2 ** 32)
    assert si[0] == want0 and si[1] == 2 ** 32 - 1 and si[2] == 2 ** 32 - 1  # prob clipped to 1
    assert sgns.split_rows(torch.arange(12, dtype=torch.int32).reshape(2, 6), 4).tolist() == [
        [0, 1, 2, 3], [4, 5, -1, -1], [6, 7, 8, 9], [10, 11, -1, -1]]


def test_shard_range_is_a_disjoint_cover():
    from node2vec_amd.shard import sentence_base, shard_range

    for n in (0, 1, 7, 8, 471785):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    with pytest.raises(ValueError):
        shard_range(5, 2, 2)
    bases = {sentence_base(r, 4, 100, e) for r in range(4) for e in range(3)}
    assert len(bases) == 12


def test_walk_stage_files_round_trip(tmp_path):
    """parquet stage files with the reference's [src, walk] schema (examples/fugue_spark.py:60)"""
    from node2vec_amd import io as n2v_io

    walks = torch.tensor([[3, 1, 2], [0, 2, 2], [5, 5, 1]], dtype=torch.int32)
    valid = torch.tensor([1, 0, 1], dtype=torch.uint8)
    path = str(tmp_path / "stage" / "walks.parquet")
    assert n2v_io.write_walks(path, walks, valid) == 2
    df = n2v_io.read_table(path)
    assert list(df.columns) == ["src", "walk"] and df["src"].tolist() == [3, 5]
    assert [list(w) for w in df["walk"]] == [[3, 1, 2], [5, 5, 1]]
    back = n2v_io.read_walks(path)
    assert back.dtype == torch.int32 and back.tolist() == [[3, 1, 2], [5, 5, 1]]
    assert n2v_io.write_walks(str(tmp_path / "empty.parquet"), walks[:0]) == 0
    vec = pd.DataFrame({"id": [1, 2], "vector": [[0.5, 1.0], [2.0, 3.0]]})
    n2v_io.write_vectors(str(tmp_path / "v.parquet"), vec)
    assert n2v_io.read_table(str(tmp_path / "v.parquet"))["id"].tolist() == [1, 2]


def test_index_graph_tensors_matches_the_pandas_indexer():
    """integer-named edge list: the torch (device) indexer gives the ids and the edge set
    of index_graph_pandas, directed and undirected, with duplicate and reversed edges"""
    import torch

    from node2vec_amd.indexer import index_graph_pandas, index_graph_tensors

    rng = np.random.default_rng(3)
    src = rng.choice([5, 17, 900, 42, 7, 123456789], 60)
    dst = rng.choice([5, 17, 900, 42, 8, 11], 60)
    w = rng.choice([0.5, 1.0, 2.0], 60)
    df = pd.DataFrame({"src": src, "dst": dst, "weight": w})
    for directed in (True, False):
        e, name_id = index_graph_pandas(df.copy(), directed)
        s_id, d_id, ww, names = index_graph_tensors(torch.from_numpy(src), torch.from_numpy(dst),
                                                     torch.from_numpy(w), directed)
        assert names.tolist() == name_id["name"].tolist()
        want = sorted(zip(e["src"], e["dst"], e["weight"].astype(np.float32)))
        got = sorted(zip(s_id.tolist(), d_id.tolist(), ww.numpy()))
        assert got == want
    s_id, d_id, ww, names = index_graph_tensors(torch.tensor([3, 3]), torch.tensor([9, 3]))
    assert ww.tolist() == [1.0, 1.0] and names.tolist() == [3, 9]


def test_pandas_twin_id_rule_numbers_by_first_appearance():
    """indexer.py:26-35: src.append(dst, ignore_index=True).drop_duplicates().reset_index() makes
    the id of a name the position of its first appearance in [src column, dst column]; on the
    reference's own test input (tests/test_indexer.py:14-16) that is a1 0, a2 1, a3 2, a4 3, b1 5,
    b2 6.  The counts the reference pins (6 vertices, 2 x 4 edges) hold for both rules."""
    import torch

    from node2vec_amd.indexer import index_graph_pandas, index_graph_tensors

    df = pd.DataFrame.from_dict({"src": ["a1", "a2", "a3", "a4"], "dst": ["a2", "b1", "b2", "a1"]})
    e, vid = index_graph_pandas(df.copy(), False, id_rule="first_appearance")
    assert list(vid.columns) == ["vertex_id", "vertex_name"]
    assert dict(zip(vid["vertex_name"], vid["vertex_id"])) == {"a1": 0, "a2": 1, "a3": 2, "a4": 3, "b1": 5, "b2": 6}
    assert len(vid) == 6 and len(e) == 8
    assert sorted(zip(e["src"], e["dst"]))[:3] == [(0, 1), (0, 3), (1, 0)]
    e2, vid2 = index_graph_pandas(df.copy(), False)
    assert len(vid2) == 6 and len(e2) == 8 and list(vid2.columns) == ["name", "id"]
    with pytest.raises(ValueError):
        index_graph_pandas(df.copy(), False, id_rule="alphabetical")
    # the device indexer follows the same rule
    rng = np.random.default_rng(4)
    src, dst = rng.integers(0, 500, 3000) * 7 + 3, rng.integers(0, 500, 3000) * 7 + 3
    w = rng.choice([0.5, 1.0], 3000)
    pdf = pd.DataFrame({"src": src, "dst": dst, "weight": w})
    for directed in (True, False):
        e, vid = index_graph_pandas(pdf.copy(), directed, id_rule="first_appearance")
        s_id, d_id, ww, names = index_graph_tensors(torch.from_numpy(src), torch.from_numpy(dst),
                                                     torch.from_numpy(w), directed, id_rule="first_appearance")
        assert sorted(zip(s_id.tolist(), d_id.tolist(), ww.numpy())) == \
            sorted(zip(e["src"], e["dst"], e["weight"].astype(np.float32)))
        assert names[torch.from_numpy(vid["vertex_id"].to_numpy())].tolist() == vid["vertex_name"].tolist()
        assert int((names >= 0).sum()) == len(vid)


def test_kernel_choice_predicates():
    """which (p, q) walk from the per-edge tables (randomwalk.tables_regime; the same rule as
    n2v_walk_exact_unit_try, csrc/n2v_walk_unit.hip): every pair but (1, 1) whose 1/p, 1/q are
    dyadic or of ordinary magnitude"""
    from node2vec_amd.randomwalk import lanes_regime, tables_regime

    grid = (0.25, 0.5, 1.0, 2.0, 4.0)
    assert sum(tables_regime(p, q) for p in grid for q in grid) == 24
    assert not tables_regime(1.0, 1.0)
    assert tables_regime(3.0, 0.7) and tables_regime(1.3, 1.3) and tables_regime(1000.0, 0.001)
    assert not tables_regime(1e-9, 3.0) and not tables_regime(3.0, 1e9)  # outside 2^-20 .. 2^20
    assert tables_regime(2.0 ** -10, 2.0) and not tables_regime(2.0 ** -30, 2.0)  # 1/p * 2^20 < 2^31
    # the class-count kernel (graphs without the wedge table): "other" alone and underfull
    assert lanes_regime(0.5, 2.0) and lanes_regime(1.0, 2.0) and lanes_regime(2.0, 2.0)
    assert not lanes_regime(4.0, 0.25) and not lanes_regime(4.0, 2.0) and not lanes_regime(3.0, 4.0)


def test_device_corpus_token_survives_pandas(tmp_path):
    """the frame random_walk() returns must survive to_parquet / concat / head /
    filters; pandas copies, compares and JSON-serialises DataFrame.attrs, so the frame carries an
    integer token and the device tensor lives in node2vec_amd.corpus"""
    import gc

    import pandas as pd
    import torch

    from node2vec_amd import corpus

    def frame(seed):
        w = torch.randint(0, 50, (200, 6), generator=torch.Generator().manual_seed(seed), dtype=torch.int32)
        df = pd.DataFrame({"src": w[:, 0].numpy().astype("int64"), "walk": w.numpy().tolist()})
        corpus.attach(df, w)
        return df, w

    a, wa = frame(1)
    b, wb = frame(2)
    assert corpus.lookup(a) is wa and corpus.lookup(b) is wb
    assert isinstance(a.attrs[corpus.ATTR], int)
    a.to_parquet(tmp_path / "walks.parquet")  # raised TypeError with a tensor in attrs
    both = pd.concat([a, b])  # raised RuntimeError (attrs compared with ==)
    assert len(both) == 400 and corpus.lookup(both) is None
    merged = a.merge(b, on="src", how="inner")
    assert corpus.lookup(merged) is None
    assert corpus.lookup(a.head(1)) is None  # derived frames: converted from their own rows
    assert corpus.lookup(a[a["src"] > 10]) is None
    back = pd.read_parquet(tmp_path / "walks.parquet")
    assert corpus.lookup(back) is None
    # an edited frame of the same length no longer matches its tensor
    c, wc = frame(3)
    c["walk"] = [[0] * 6] * len(c)
    assert corpus.lookup(c) is None
    # ONE replaced row (not among the sampled ones) is seen too -- every row must still
    # be the list object random_walk() put there; STRICT compares every element
    d, wd = frame(4)
    assert corpus.lookup(d) is wd
    d.at[97, "walk"] = list(d.at[97, "walk"])  # same contents, another object: conservative None
    assert corpus.lookup(d) is None
    e, we = frame(5)
    e.at[123, "walk"][2] += 1  # an edit inside an original list object
    corpus.STRICT = True
    try:
        assert corpus.lookup(e) is None
        f, wf = frame(6)
        assert corpus.lookup(f) is wf
    finally:
        corpus.STRICT = False
    n_before = len(corpus._registry)
    del a, both, merged
    gc.collect()
    assert len(corpus._registry) == n_before - 1  # the entry dies with its frame


def test_keyed_vectors_from_ids_is_lazy_and_reference_shaped(tmp_path):
    """a fitted model keeps integer ids and a (device) tensor: tokens, the token -> row map and
    the numpy matrix are made on access (no 10^8-element Python lists)"""
    import pandas as pd
    import torch

    from node2vec_amd.embedding import HipW2V, KeyedVectors, Node2VecHIP

    ids = np.array([30, 7, 1000003, 0, 12], dtype=np.int64)
    vec = torch.arange(5 * 4, dtype=torch.float32).reshape(5, 4)
    kv = KeyedVectors(torch.from_numpy(ids), vec)
    assert len(kv) == 5 and kv.vector_size == 4
    assert list(kv.vocab) == ["30", "7", "1000003", "0", "12"] == list(kv.index2word)
    assert kv.index2word[2] == "1000003" and kv.index2word[1:3] == ["7", "1000003"]
    assert "7" in kv and "8" not in kv and "07" not in kv and "x" not in kv
    assert kv.vocab["1000003"] == 2 and kv.vocab.get("5") is None and len(kv.vocab) == 5
    with pytest.raises(KeyError):
        kv.vocab["5"]
    assert np.array_equal(kv["0"], vec[3].numpy()) and isinstance(kv._vectors, torch.Tensor)
    assert np.array_equal(kv.rows(1, 3), vec[1:3].numpy())
    kv.save_word2vec_format(str(tmp_path / "v.txt"), chunk_rows=2)
    back = KeyedVectors.load_word2vec_format(str(tmp_path / "v.txt"))
    assert back.index2word == ["30", "7", "1000003", "0", "12"] and np.array_equal(back.vectors, vec.numpy())
    m = HipW2V(kv, vec * 2, {"size": 4}, 3)
    m.save(str(tmp_path / "m.model"))
    m2 = HipW2V.load(str(tmp_path / "m.model"))
    assert list(m2.wv.vocab) == list(kv.vocab) and np.array_equal(m2.syn1neg, (vec * 2).numpy())
    # embedding(): the reference's frame, from chunks; names through name_id
    n2v = Node2VecHIP(pd.DataFrame({"src": [0], "walk": [[0, 7]]}), {}, random_seed=1)
    n2v.model = m
    frame = n2v.embedding()
    assert list(frame.columns) == ["id", "vector"] and frame["id"].tolist() == ids.tolist()
    assert frame["vector"].iloc[2] == vec[2].tolist()
    chunks = list(n2v.iter_embedding(chunk_rows=2))
    assert [len(c) for c in chunks] == [2, 2, 1]
    n2v.name_id = pd.DataFrame({"name": ["a", "b", "c", "d", "e"], "id": [0, 7, 12, 30, 1000003]})
    named = n2v.embedding()
    assert list(named.columns) == ["name", "vector"] and named["name"].tolist() == ["d", "b", "e", "a", "c"]
    assert n2v.get_vector(7) == vec[1].tolist()


def test_index_graph_names_on_the_references_own_input():
    """reference tests/test_indexer.py:14-20: string names 'a1' .. 'b2', undirected -> 6 vertices
    and 2 x the edges; the device indexer's string front end (pyarrow dictionary codes, then
    index_graph_tensors on the codes; here on CPU tensors) numbers them like index_graph_pandas
    under both id rules, in chunks smaller than the input"""
    import pandas as pd
    import torch

    from node2vec_amd.indexer import index_graph_names, index_graph_pandas

    df = pd.DataFrame.from_dict({"src": ["a1", "a2", "a3", "a4"], "dst": ["a2", "b1", "b2", "a1"]})
    for rule in ("sorted", "first_appearance"):
        e, vid = index_graph_pandas(df.copy(), False, id_rule=rule)
        s, d, w, names = index_graph_names(df["src"], df["dst"], None, False, "cpu", rule, chunk_rows=3)
        assert len(s) == 2 * len(df) and sum(x is not None for x in names) == 6
        assert sorted(zip(s.tolist(), d.tolist())) == sorted(zip(e["src"], e["dst"]))
        assert bool((w == 1.0).all())
        if rule == "sorted":
            assert names.tolist() == vid["name"].tolist()
        else:
            assert [names[i] for i in vid["vertex_id"]] == vid["vertex_name"].tolist()
    with pytest.raises(ValueError):
        index_graph_names(["a"], ["b", "c"])
    # the two weight rules of the reference's twins: fp32 (Spark twin) merges weights that differ
    # below fp32 precision when de-duplicating, fp64 (pandas twin) keeps them apart
    src, dst = ["x", "y"], ["y", "x"]
    wt = [0.1, 0.1 + 1e-12]
    _, _, w32, _ = index_graph_names(src, dst, wt, False, "cpu", weight_dtype=torch.float32)
    _, _, w64, _ = index_graph_names(src, dst, wt, False, "cpu", weight_dtype=torch.float64)
    assert w32.dtype == torch.float32 and len(w32) == 2 and w64.dtype == torch.float64 and len(w64) == 4
    e64, _ = index_graph_pandas(pd.DataFrame({"src": src, "dst": dst, "weight": wt}), False)
    assert len(e64) == 4


def test_bench_gpus_n_spawns_n_ranks(monkeypatch):
    """bench.py --gpus N, started as one process, launches N ranks under torch.distributed.run as a
    child process: the command line it would run at N = 2, and the refusal of a
    WORLD_SIZE that contradicts --gpus.  (No GPU here: nothing is started.)"""
    import importlib.util
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("n2v_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    argv = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "cfg2"]
    cmd = bench.spawn_command(argv, 2, 29517)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[3:10] == ["--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", "29517"]
    assert cmd[10] == os.path.join(root, "bench.py") and cmd[11:] == argv
    assert "--spawn" not in bench.spawn_command(argv + ["--spawn"], 2, 1)
    # main(): one process + --gpus 2 -> the spawn path, with the parsed arguments; no torch import needed
    seen = {}

    def fake_spawn(args, av):
        seen["gpus"], seen["argv"] = args.gpus, list(av)
        return 0

    monkeypatch.setattr(bench, "spawn_ranks", fake_spawn)
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and seen == {"gpus": 2, "argv": argv}
    # a rank of a world of 1 claiming --gpus 2 is refused
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=1" in str(e.value.code)


def test_weighted_row_sums_and_the_grid_of_the_weights():
    """randomwalk.weighted_row_sums: what the margin kernels of weighted exact walks take the row sum of a step
    from -- per-row fp64 sums, then a power of two that divides every stored weight (fp32 storage only) and the
    largest weight; None when a weight is negative or not finite (the margins assume neither)"""
    import numpy as np
    import torch

    from node2vec_amd import randomwalk as rw
    from node2vec_amd.graph import DeviceGraph

    src = np.array([0, 0, 0, 1, 3, 3])
    dst = np.array([1, 2, 3, 0, 0, 2])
    w = np.array([0.5, 0.75, 3.0, 1.25, 0.1875, 2.0], dtype=np.float32)
    g = DeviceGraph.from_edges(src, dst, w, n_vertices=5)
    got = rw.weighted_row_sums(g)
    assert got.dtype == torch.float64 and got.numel() == 5 + 2
    assert got[:5].tolist() == [4.25, 1.25, 0.0, 2.1875, 0.0]
    # the smallest weight is 0.1875 = 0.75 * 2^-2: frexp exponent -2, 24 mantissa bits -> 2^-26 divides them all
    assert float(got[5]) == 2.0 ** -26 and float(got[6]) == 3.0
    assert all(float(x) % float(got[5]) == 0.0 for x in w)
    assert rw.weighted_row_sums(g) is got  # kept on the graph
    g64 = DeviceGraph.from_edges(src, dst, np.array([0.1, 0.3, 0.7, 1.1, 2.9, 0.2]), n_vertices=5)  # not fp32 values
    assert g64.w.dtype == torch.float64 and float(rw.weighted_row_sums(g64)[5]) == 0.0  # no grid claimed
    bad = DeviceGraph.from_edges(src, dst, np.array([0.5, -0.75, 3.0, 1.25, 0.1875, 2.0], dtype=np.float32), n_vertices=5)
    assert rw.weighted_row_sums(bad) is None


def test_weighted_hub_summaries_blocks_are_sorted_with_prefix_sums(monkeypatch):
    """randomwalk.weighted_hub_summaries (struct n2v_weighted_hubs): for every row of WEIGHTED_HUB_SLOTS slots or
    more, per block of 256 slots in row order, the weights sorted ascending (+inf behind the last weight of the
    row) and their fp64 prefix sums; -1 for the other rows"""
    import numpy as np
    import torch

    from node2vec_amd import randomwalk as rw
    from node2vec_amd.graph import DeviceGraph

    monkeypatch.setattr(rw, "WEIGHTED_HUB_SLOTS", 300)
    rng = np.random.default_rng(1)
    src = np.concatenate([rng.integers(0, 50, 400), np.repeat([3, 7], [700, 300])])
    dst = np.concatenate([rng.integers(0, 50, 400), rng.integers(0, 1000, 1000)])
    w = (rng.random(src.size) + 0.1).astype(np.float32)
    g = DeviceGraph.from_edges(src, dst, w, n_vertices=1000)
    st = rw.weighted_hub_summaries(g)
    assert st is not None and st.min_slots == 300 and rw.weighted_hub_summaries(g) is st  # kept on the graph
    _, block0, srt, prefix = g._weighted_hubs
    deg = g.degrees()
    hubs = torch.nonzero(deg >= 300).reshape(-1).tolist()
    assert hubs == [3, 7] and int((block0 >= 0).sum()) == 2
    assert srt.shape[1] == 256 and prefix.shape == (srt.shape[0], 257) and prefix.dtype == torch.float64
    nb = 0
    for v in hubs:
        assert int(block0[v]) == nb
        row = g.w[int(g.rowptr[v]):int(g.rowptr[v + 1])]
        for b in range((row.numel() + 255) // 256):
            seg = row[256 * b:256 * (b + 1)]
            got = srt[nb + b]
            assert torch.equal(got[:seg.numel()], torch.sort(seg).values) and bool(torch.isinf(got[seg.numel():]).all())
            want = torch.cumsum(torch.sort(seg).values.double(), 0)
            assert float(prefix[nb + b][0]) == 0.0 and torch.allclose(prefix[nb + b][1:seg.numel() + 1], want, rtol=1e-15)
            assert bool((prefix[nb + b][seg.numel():] == prefix[nb + b][seg.numel()]).all())  # (+inf adds nothing)
        nb += (row.numel() + 255) // 256
    assert nb == srt.shape[0]
    unit = DeviceGraph.from_edges(src, dst, None, n_vertices=1000)
    monkeypatch.setattr(rw, "WEIGHTED_HUB_SLOTS", 10 ** 9)
    g2 = DeviceGraph.from_edges(src, dst, w, n_vertices=1000)
    assert rw.weighted_hub_summaries(g2) is None and unit.unit_weights


def test_arrow_backed_walk_column_is_the_reference_frame_to_its_consumers(tmp_path):
    """random_walk() beyond corpus.LIST_COLUMN_MAX_VALUES vertices: the "walk" column is list<int32> over one flat
    buffer instead of a Python list per row.  What the reference's consumers do with the frame still works --
    np.array(df["walk"].tolist()) (embedding.py:125), iteration, .iloc, to_parquet with the schema of
    randomwalk.py:342 -- the device corpus is found through the (immutable) buffers, any derived or edited frame is
    not, and Node2VecHIP reads an Arrow column without building Python objects."""
    import numpy as np
    import pandas as pd
    import pyarrow.parquet as pq
    import torch

    from node2vec_amd import corpus
    from node2vec_amd.embedding import Node2VecHIP

    w = torch.randint(0, 50, (300, 7), generator=torch.Generator().manual_seed(8), dtype=torch.int32)
    assert isinstance(corpus.list_column(w.numpy(), "auto"), list)  # small: Python lists, as before
    with pytest.raises(ValueError):
        corpus.list_column(w.numpy(), "tuples")
    # "rows" (the default beyond LIST_COLUMN_MAX_VALUES values): one read-only ndarray view per row
    big = np.arange(3 * (corpus.LIST_COLUMN_MAX_VALUES // 2), dtype=np.int32).reshape(-1, 3)
    auto = corpus.list_column(big, "auto")
    assert isinstance(auto, np.ndarray) and auto.dtype == object and auto[7].base is not None and not auto[7].flags.writeable
    del big, auto
    rdf = pd.DataFrame({"src": w[:, 0].numpy().astype("int64"), "walk": corpus.list_column(w.numpy().copy(), "rows")})
    corpus.attach(rdf, w)
    assert corpus.lookup(rdf) is w
    assert np.array_equal(np.array(rdf["walk"].tolist()), w.numpy()) and len(rdf["walk"].iloc[0]) == 7
    assert all(r[0] == s for s, r in zip(rdf["src"], rdf["walk"]))
    with pytest.raises(ValueError):
        rdf.at[5, "walk"][2] += 1  # an edit INSIDE a row is refused: the views are read-only
    rdf.to_parquet(tmp_path / "rows.parquet")
    rback = pd.read_parquet(tmp_path / "rows.parquet")  # (round trip through pandas: ndarray rows again)
    assert np.array_equal(np.stack(rback["walk"].to_numpy()), w.numpy()) and corpus.lookup(rback) is None
    col = rdf["walk"].to_numpy().copy()
    col[9] = np.array(col[9])  # same contents, another object
    rdf["walk"] = col
    assert corpus.lookup(rdf) is None
    col = corpus.list_column(w.numpy(), "arrow")
    df = pd.DataFrame({"src": w[:, 0].numpy().astype("int64"), "walk": col})
    corpus.attach(df, w)
    assert corpus.lookup(df) is w
    assert df["walk"].iloc[3] == w[3].tolist() and isinstance(df["walk"].iloc[3], list)
    assert np.array_equal(np.array(df["walk"].tolist()), w.numpy())  # embedding.py:125
    assert [list(r) for r in df["walk"]][:2] == w[:2].tolist()
    assert np.array_equal(corpus.arrow_rows(df["walk"]), w.numpy())
    df.to_parquet(tmp_path / "walks.parquet")
    assert str(pq.read_schema(tmp_path / "walks.parquet").field("walk").type) in ("list<element: int32>", "list<item: int32>")
    # (pandas < 3 cannot read this file back itself -- it wrote its own dtype string into the metadata; pyarrow can:
    # why "arrow" is opt-in and "rows" the default for large results)
    back = pq.read_table(tmp_path / "walks.parquet").to_pandas(ignore_metadata=True)
    assert corpus.lookup(back) is None and np.array_equal(np.stack(back["walk"].to_numpy()), w.numpy())
    assert corpus.lookup(df.head(10)) is None and corpus.lookup(df[df["src"] > 10]) is None
    assert corpus.lookup(df.copy()) is None or corpus.lookup(df.copy()) is w  # (a copy is another frame: token checked)
    edited = df.copy()
    edited.attrs = dict(df.attrs)
    edited["walk"] = corpus.list_column(np.zeros((300, 7), np.int32), "arrow")
    assert corpus.lookup(edited) is None
    e2, w2 = pd.DataFrame({"src": w[:, 0].numpy().astype("int64"), "walk": corpus.list_column(w.numpy(), "arrow")}), w.clone()
    corpus.attach(e2, w2)
    e2["walk"] = e2["walk"].array.take(list(range(299, -1, -1)))  # any derived array: no longer the registered buffers
    assert corpus.lookup(e2) is None
    # a frame that was never attached: the trainer converts the Arrow column without Python objects
    plain = pd.DataFrame({"src": w[:, 0].numpy().astype("int64"), "walk": corpus.list_column(w.numpy(), "arrow")})
    t = Node2VecHIP(plain, {"size": 32, "iter": 1}, random_seed=3)._walk_tensor("cpu")
    assert t.dtype == torch.int32 and torch.equal(t, w)
    # ragged Arrow rows are refused by the fast path (and by the reference's np.array too)
    import pyarrow as pa

    ragged = pd.Series(pd.arrays.ArrowExtensionArray(pa.array([[1, 2], [3]], type=pa.list_(pa.int32()))))
    assert corpus.arrow_rows(ragged) is None


def test_exchange_plan_cfg4_on_eight_gpus_fits_and_names_its_link_bytes():
    """Multi-GPU readiness without a multi-GPU box: BASELINE cfg 4 at world 8 -- the model of the
    workload's own vocabulary (8.67 x 10^7 words x 128, two matrices), bf16 deltas -- from the size rules DeltaSync and
    shard.ordered_sum allocate by (tests/test_dist_gloo.py compares the plan with the live buffers of a two-rank
    exchange).  Per rank: replicas + bf16 reference + block buffers + the graph, its tables and a corpus batch
    <= 288 GB; per sync every link carries 2 x wire / world in each direction."""
    from node2vec_amd.sgns import exchange_plan

    n_vocab, dim = 86_700_000, 128
    # resident beside the model at cfg 4: CSR 3.8 GB + hop table 12.1 + ranked form 3.8 + a batch of
    # 2^20 x 10 walks of 81 tokens and its index (2 x 3.4 GB)
    resident = int((3.8 + 12.1 + 3.8 + 6.8) * 1e9)
    plan = exchange_plan([(n_vocab, dim), (n_vocab, dim)], 8, "bf16", resident_bytes=resident)
    assert plan["model_bytes"] == 2 * n_vocab * dim * 4 and plan["bf16_reference_bytes"] == 2 * n_vocab * dim * 2
    assert plan["fits"] and plan["hbm_bytes_per_rank"] < 170e9
    wire = 2 * n_vocab * dim * 2
    assert plan["wire_bytes_per_rank_per_sync"] == wire and plan["links_used"] == 7
    assert abs(plan["bytes_per_link_per_direction_per_sync"] - 2 * wire / 8) < 1e-3 * wire
    assert 0.05 < plan["link_seconds_per_sync_at_peak"] < 0.2  # ~0.15 s per exchange at the links' peak
    fp32 = exchange_plan([(n_vocab, dim), (n_vocab, dim)], 8, "fp32", resident_bytes=resident)
    assert fp32["bf16_reference_bytes"] == 0 and fp32["bytes_per_link_per_direction_per_sync"] == 2 * plan["bytes_per_link_per_direction_per_sync"]
    one = exchange_plan([(n_vocab, dim)], 1)
    assert one["bytes_per_link_per_direction_per_sync"] == 0 and one["links_used"] == 0
    # cfg 5: 5 x 10^7 x 256
    assert exchange_plan([(50_000_000, 256)] * 2, 8, "bf16", resident_bytes=int(20e9))["fits"]
