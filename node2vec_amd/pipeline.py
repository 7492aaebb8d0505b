"""Walk -> SGNS streaming pipeline: the whole node2vec fit without ever holding the walk
corpus.

The reference materialises every walk as a DataFrame row, then as a numpy string array
(embedding.py:125), before training.  At BASELINE cfg 4 (100 M vertices x 10 walks x 81
tokens) that corpus is 324 GB, more than one GPU's HBM.  Here the corpus never exists:
walks are a pure function of (seed, start vertex, ordinal) (DESIGN.md "RNG"), so they
are regenerated batch by batch -- a counting pass for the vocabulary (walks cost ~2 % of
training), then per epoch each batch is walked (K2) and trained on (K3) while it sits in
HBM.  Row r of the virtual corpus is (start index r // W, ordinal r % W + 1), exactly the
row order of fugue.random_walk_tensors, and its SGNS sentence id is epoch * rows + r, so
training on the batches equals training on the materialised corpus with the same blocks.
"""
import logging
from typing import Any, Dict, Optional

import torch

from node2vec_amd import randomwalk as rw
from node2vec_amd import sgns
from node2vec_amd.constants import GENSIM_PARAMS, HIP_SGNS_PARAMS, NODE2VEC_PARAMS
from node2vec_amd.embedding import HipW2V, KeyedVectors
from node2vec_amd.graph import DeviceGraph


def fit_streaming(graph: DeviceGraph, n2v_params: Dict[str, Any], w2v_params: Dict[str, Any],
                  random_seed: int, batch_vertices: int = 65536, walk_seed_ids=None,
                  mode: str = "exact", return_model: bool = False):
    """node2vec end to end on the GPU: returns a HipW2V (and the SgnsModel when asked).

    n2v_params / w2v_params take the reference's keys (NODE2VEC_PARAMS, GENSIM_PARAMS plus
    the pass-through names of HIP_SGNS_PARAMS); missing keys are filled in the caller's
    dicts as the reference does (fugue.py:120-122, embedding.py:105-107)."""
    for k, v in NODE2VEC_PARAMS.items():
        n2v_params.setdefault(k, v)
    for k, v in GENSIM_PARAMS.items():
        w2v_params.setdefault(k, v)
    p = dict(HIP_SGNS_PARAMS)
    p.update(w2v_params)
    if p.get("hs", 0) or not p.get("sg", 1):
        raise ValueError("the HIP trainer implements sg=1, hs=0 (skip-gram, negative sampling)")
    negative = int(p["negative"]) if p["negative"] else int(HIP_SGNS_PARAMS["negative"])
    W, L = int(n2v_params["num_walks"]), int(n2v_params["walk_length"])
    pp, qq = float(n2v_params["return_param"]), float(n2v_params["inout_param"])
    seed = int(random_seed)
    start = rw.start_vertices(graph, walk_seed_ids)
    n_start = start.numel()
    batch_vertices = max(1, int(batch_vertices))
    dev = graph.device

    def batches():
        for lo in range(0, n_start, batch_vertices):
            yield lo, start[lo:lo + batch_vertices].contiguous()

    def walk(batch):
        return rw.walk(graph, batch, W, L, pp, qq, seed, mode)

    # ---- pass 1: token counts of the virtual corpus (vocabulary, cum_table, subsampling)
    counts = torch.zeros(graph.n_vertices, dtype=torch.int64, device=dev)
    rows_total = 0
    for _, b in batches():
        walks, valid = walk(b)
        w = walks[valid]
        rows_total += int(valid.numel())  # rows keep their index; dropped walkers stay as gaps
        counts += torch.bincount(w.reshape(-1).long(), minlength=graph.n_vertices)
    ids = torch.nonzero(counts >= max(int(p["min_count"]), 1)).reshape(-1)
    if ids.numel() == 0:
        raise RuntimeError("you must first build vocabulary before training the model")
    cnt = counts[ids]
    order = torch.sort(cnt, descending=True, stable=True).indices  # ties: ascending id
    ids, cnt = ids[order], cnt[order]
    index_of = torch.full((graph.n_vertices,), -1, dtype=torch.int32, device=dev)
    index_of[ids] = torch.arange(ids.numel(), dtype=torch.int32, device=dev)
    vocab = sgns.Vocab(ids, cnt, index_of)
    model = sgns.SgnsModel(vocab, int(p["size"]), int(p["window"]), negative, int(p["seed"] or seed),
                           sample=float(p["sample"] or 0.0), ns_exponent=float(p["ns_exponent"]),
                           device=dev)
    logging.info("fit_streaming: %d rows, vocabulary %d", rows_total, len(vocab))

    # ---- pass 2: per epoch, walk a batch and train on it while it is resident
    epochs = max(int(p["iter"]), 1)
    alpha, min_alpha = float(p["alpha"]), float(p["min_alpha"])
    total, done = rows_total * epochs, 0
    submitted = 0  # SGNS sentence ids: rows handed to the kernel so far (after split_rows)
    for ep in range(epochs):
        for lo, b in batches():
            walks, valid = walk(b)
            idx = index_of[walks.long().clamp(min=0)]
            idx = torch.where(valid.bool().unsqueeze(1) & (walks >= 0), idx, torch.full_like(idx, -1))
            a = max(min_alpha, alpha - (alpha - min_alpha) * (done / max(total, 1)))
            for part in torch.split(sgns.split_rows(idx), 1 << 22):
                model.train_block(part, a, submitted)
                submitted += part.shape[0]
            done += idx.shape[0]
    torch.cuda.synchronize(dev)
    tokens = [str(int(i)) for i in vocab.ids.cpu().numpy()]
    p["negative"] = negative
    out = HipW2V(KeyedVectors(tokens, model.syn0.cpu().numpy()), model.syn1neg.cpu().numpy(), p,
                 int(model.pairs.item()))
    return (out, model) if return_model else out
