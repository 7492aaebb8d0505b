"""f1: the device indexer on the GPU against index_graph_pandas (the reference's
contract, indexer.py:9-49 / :52-84) for both id rules, and straight into the CSR + a walk."""
import numpy as np
import pandas as pd
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("id_rule", ["sorted", "first_appearance"])
@pytest.mark.parametrize("directed", [True, False])
def test_index_graph_tensors_on_the_gpu_equals_the_pandas_indexer(id_rule, directed):
    from node2vec_amd.indexer import index_graph_pandas, index_graph_tensors

    rng = np.random.default_rng(12)
    n_e = 400_000
    names = rng.choice(2 ** 40, 50_000, replace=False)  # sparse 64-bit names
    src, dst = names[rng.integers(0, len(names), n_e)], names[rng.integers(0, len(names), n_e)]
    w = rng.choice([0.25, 1.0, 3.0], n_e)
    e, vid = index_graph_pandas(pd.DataFrame({"src": src, "dst": dst, "weight": w}), directed, id_rule=id_rule)
    s_id, d_id, ww, nm = index_graph_tensors(torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda(),
                                             torch.from_numpy(w).cuda(), directed, id_rule=id_rule)
    assert s_id.is_cuda and nm.is_cuda
    got = torch.stack([s_id, d_id, (ww * 4).long()], 1).cpu().numpy()
    want = np.stack([e["src"].to_numpy(), e["dst"].to_numpy(), (e["weight"].to_numpy() * 4).astype(np.int64)], 1)
    assert np.array_equal(got[np.lexsort(got.T[::-1])], want[np.lexsort(want.T[::-1])])
    if id_rule == "sorted":
        assert np.array_equal(nm.cpu().numpy(), vid["name"].to_numpy())
    else:
        assert np.array_equal(nm.cpu().numpy()[vid["vertex_id"].to_numpy()], vid["vertex_name"].to_numpy())


def test_indexed_tensors_feed_the_csr_and_the_walk(oracle):
    """raw integer-named undirected edge list -> ids -> CSR -> walks == the oracle on the same CSR"""
    from node2vec_amd import randomwalk as rw
    from node2vec_amd.graph import DeviceGraph
    from node2vec_amd.indexer import index_graph_tensors

    rng = np.random.default_rng(2)
    src = torch.from_numpy(rng.integers(0, 3000, 20000) * 1000003).cuda()
    dst = torch.from_numpy(rng.integers(0, 3000, 20000) * 1000003).cuda()
    s_id, d_id, w, names = index_graph_tensors(src, dst, None, directed=False)
    g = DeviceGraph.from_edges(s_id, d_id, w, n_vertices=int(names.numel()), device="cuda")
    assert g.unit_weights
    start = rw.start_vertices(g)
    walks, valid = rw.walk(g, start, 2, 20, 0.5, 2.0, 1)
    want, wv = oracle.random_walk(g.rowptr.cpu().numpy(), g.col.cpu().numpy(), None, start.cpu().numpy(),
                                  2, 20, 0.5, 2.0, 1, n_threads=8)
    assert np.array_equal(valid.cpu().numpy(), wv) and np.array_equal(walks.cpu().numpy(), want)
    assert bool((names[walks[valid].long()] % 1000003 == 0).all())  # ids map back to the names


@pytest.mark.parametrize("id_rule", ["sorted", "first_appearance"])
def test_index_graph_names_strings_on_the_gpu_equal_the_pandas_indexer(id_rule):
    """string vertex names (the reference's own input type, tests/test_indexer.py:14-16) through
    the device indexer: 300 k edges over 40 k names, chunked dictionary encoding, undirected"""
    from node2vec_amd.indexer import index_graph_names, index_graph_pandas

    rng = np.random.default_rng(5)
    pool = np.array([f"v{int(x):09d}" if x % 3 else f"user/{int(x)}" for x in rng.choice(10 ** 9, 40_000, replace=False)],
                    dtype=object)
    src, dst = pool[rng.integers(0, len(pool), 300_000)], pool[rng.integers(0, len(pool), 300_000)]
    w = rng.choice([0.5, 1.0, 2.0], 300_000)
    e, vid = index_graph_pandas(pd.DataFrame({"src": src, "dst": dst, "weight": w}), False, id_rule=id_rule)
    s_id, d_id, ww, names = index_graph_names(src, dst, w, False, "cuda", id_rule, chunk_rows=70_000)
    assert s_id.is_cuda
    got = torch.stack([s_id, d_id, (ww * 4).long()], 1).cpu().numpy()
    want = np.stack([e["src"].to_numpy(), e["dst"].to_numpy(), (e["weight"].to_numpy() * 4).astype(np.int64)], 1)
    assert np.array_equal(got[np.lexsort(got.T[::-1])], want[np.lexsort(want.T[::-1])])
    if id_rule == "sorted":
        assert names.tolist() == vid["name"].tolist()
    else:
        assert [names[i] for i in vid["vertex_id"]] == vid["vertex_name"].tolist()
