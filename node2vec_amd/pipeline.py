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

Under an initialised torch.distributed process group (one process per GPU, graph replicated)
each rank streams its contiguous range of the start vertices (shard.shard_range): token counts
and row counts are summed over the ranks, so every rank builds the same vocabulary and model;
the replicas are averaged by sgns.DeltaSync (RCCL all-reduce of the model deltas) every few
launches and once more at the end.  No collective is on the walk path.
"""
import logging
from typing import Any, Dict, Optional

import torch

from node2vec_amd import randomwalk as rw
from node2vec_amd import sgns
from node2vec_amd.constants import GENSIM_PARAMS, HIP_SGNS_PARAMS, NODE2VEC_PARAMS
from node2vec_amd.embedding import HipW2V, KeyedVectors
from node2vec_amd.graph import DeviceGraph


def corpus_vocabulary(graph: DeviceGraph, walk, n_batches: int, min_count: int, in_ranks: bool,
                      multi: bool = False, after_count=None):
    """Pass 1 of fit_streaming: the vocabulary of the virtual corpus `walk(0) ... walk(n_batches - 1)`
    -- what gensim's vocabulary scan does with the token strings of embedding.py:125: token counts,
    min_count, descending-count order (ties: ascending id).  No host synchronisation per batch:
    invalid rows are masked out, not compacted.  Returns (Vocab, per-token lookup): the lookup takes
    the tokens as the walks carry them (degree ranks when `in_ranks`) to vocabulary indices.
    bench.py builds its SGNS leg's vocabulary with this same function."""
    import torch.distributed as dist

    from node2vec_amd.shard import all_reduce

    dev = graph.device
    counts = torch.zeros(graph.n_vertices, dtype=torch.int64, device=dev)
    for k in range(n_batches):
        walks, valid = walk(k)
        if walks.numel() == 0:
            continue
        sgns.corpus_count(walks, valid, counts)  # one pass, no widening / clamping / scatter ops
    if in_ranks:
        counts = counts[graph.rank_of.long()]  # counted per rank: back to vertex ids
    if multi:
        all_reduce(counts, dist.ReduceOp.SUM)
    if after_count is not None:
        after_count()  # (the nonzero() below synchronises anyway)
    ids = torch.nonzero(counts >= max(int(min_count), 1)).reshape(-1)
    if ids.numel() == 0:
        raise RuntimeError("you must first build vocabulary before training the model")
    cnt = counts[ids]
    order = torch.sort(cnt, descending=True, stable=True).indices  # ties: ascending id
    ids, cnt = ids[order], cnt[order]
    del counts, order
    index_of = torch.full((graph.n_vertices,), -1, dtype=torch.int32, device=dev)
    index_of[ids] = torch.arange(ids.numel(), dtype=torch.int32, device=dev)
    vocab = sgns.Vocab(ids, cnt, index_of)
    token_index = index_of[graph.rank_vertex.long()].contiguous() if in_ranks else index_of
    return vocab, token_index


def fit_streaming(graph: DeviceGraph, n2v_params: Dict[str, Any], w2v_params: Dict[str, Any],
                  random_seed: int, batch_vertices: int = 65536, walk_seed_ids=None,
                  mode: str = "exact", return_model: bool = False, timings: Optional[dict] = None):
    """node2vec end to end on the GPU: returns a HipW2V (and the SgnsModel when asked).

    n2v_params / w2v_params take the reference's keys (NODE2VEC_PARAMS, GENSIM_PARAMS plus
    the pass-through names of HIP_SGNS_PARAMS); missing keys are filled in the caller's
    dicts as the reference does (fugue.py:120-122, embedding.py:105-107).  w2v_params["batched"]
    selects the opt-in shared-negative trainer; "sync_every" / "sync_wire" the multi-GPU exchange;
    "deterministic" the one-wave reproducible mode (tests).
    `timings` (optional dict) receives the seconds spent walking and training."""
    import time

    import torch.distributed as dist

    from node2vec_amd.shard import shard_range

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
    dev = graph.device
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank, world = (dist.get_rank(), dist.get_world_size()) if multi else (0, 1)
    start_all = rw.start_vertices(graph, walk_seed_ids)
    lo_r, hi_r = shard_range(start_all.numel(), rank, world)
    start = start_all[lo_r:hi_r]
    n_start = start.numel()
    batch_vertices = max(1, int(batch_vertices))
    # every rank lays the same batch grid over the LARGEST shard, so that ranks with fewer
    # start vertices still take part in every exchange
    n_start_max = shard_range(start_all.numel(), 0, world)[1]
    n_batches = max(1, -(-n_start_max // batch_vertices))
    t_walk = t_train = 0.0

    def clock():
        if timings is not None:
            torch.cuda.synchronize(dev)
        return time.perf_counter()

    # the walk status word (the reference's ZeroDivisionError on a visited row whose biased weights
    # sum to 0, randomwalk.py:172-173; an id out of range) is OR-ed on the device over all batches
    # and raised at the two host synchronisations the function has anyway -- never dropped
    walk_status = torch.zeros(1, dtype=torch.int32, device=dev)
    # p = q = 1 on unit weights: the walks come out in degree RANKS (4-byte table entries, the
    # fastest walk there is: graph.build_ranked) -- nothing downstream needs vertex ids per token,
    # only counts per vertex (permuted once) and the per-token vocabulary lookup (composed once)
    in_ranks = False
    if graph.unit_weights and pp == 1.0 and qq == 1.0 and mode == "exact" and graph.rowptr.is_cuda:
        if graph.rank_hops is None and not graph.rank_tried:
            graph.build_ranked()
        in_ranks = graph.rank_hops is not None

    # every batch is walked into the same pair of buffers, chosen once by audition (where an output
    # buffer lies decides up to 7 % of the walk kernel's time: randomwalk.audition_buffers)
    buffers = [None]

    def walk(k):
        st = {}
        sk = start[k * batch_vertices:(k + 1) * batch_vertices].contiguous()
        if buffers[0] is None and sk.numel() == batch_vertices and n_batches > 2 and p.get("audition", True):
            rep = {}
            buffers[0] = rw.audition_buffers(graph, sk, W, L, pp, qq, seed, mode, report=rep, rank_ids=in_ranks)
            if timings is not None:
                timings["audition"] = rep
        out = None
        if buffers[0] is not None:
            rows = sk.numel() * W
            out = (buffers[0][0][:rows], buffers[0][1][:rows])
        walks, valid = rw.walk(graph, sk, W, L, pp, qq, seed, mode, out=out, check=False, stats=st,
                               rank_ids=in_ranks)
        walk_status.bitwise_or_(st["status"][:1])
        return walks, valid.bool()

    def raise_walk_status():
        from node2vec_amd import _lib
        _lib.check_status_word(int(walk_status.item()), "n2v_walk")

    # ---- pass 1: token counts of the virtual corpus (vocabulary, cum_table, subsampling)
    t0 = clock()
    vocab, token_index = corpus_vocabulary(graph, walk, n_batches, int(p["min_count"]), in_ranks, multi,
                                           raise_walk_status)
    t_walk += clock() - t0
    rows_rank_max = n_start_max * W  # rows of the largest shard: the sentence-id stride of a rank
    model = sgns.SgnsModel(vocab, int(p["size"]), int(p["window"]), negative, int(p["seed"] or seed),
                           sample=float(p["sample"] or 0.0), ns_exponent=float(p["ns_exponent"]),
                           device=dev)
    model.batched = bool(p.get("batched", False))
    model.hub_rows = None if p.get("hub_rows") is None else int(p["hub_rows"])
    deterministic = bool(p.get("deterministic", False))
    logging.info("fit_streaming: %d start vertices on this rank, vocabulary %d", n_start, len(vocab))
    sync = None
    if multi:
        sync = sgns.DeltaSync(model, sync_every=p.get("sync_every"), wire=p.get("sync_wire", "fp32"))
        sync.max_every = n_batches

    # ---- pass 2: per epoch, walk a batch and train on it while it is resident
    epochs = max(int(p["iter"]), 1)
    alpha, min_alpha = float(p["alpha"]), float(p["min_alpha"])
    parts_per_row = -(-(L + 1) // sgns.MAX_SENTENCE)
    batch_words = int(p.get("batch_words") or 0)
    for ep in range(epochs):
        # gensim's schedule: the rate of a job of batch_words words (constants.py:58); rows of the
        # virtual corpus are sentences in order.  A dropped walker (a sink was reached: valid = 0) keeps
        # its row and trains nothing, where gensim's corpus would not contain it at all: on a graph WITH
        # sinks the job boundaries and `pushed / total` therefore count those empty rows and the rates
        # drift from gensim's by the share of dropped rows; on graphs without sinks (every symmetrised
        # graph: all BASELINE configs) the schedule is gensim's exactly.
        sched = None
        if batch_words and parts_per_row == 1:
            sched = sgns.JobSchedule.for_corpus(batch_words, L + 1, rows_rank_max, ep, epochs, alpha, min_alpha)
        for k in range(n_batches):
            t0 = clock()
            walks, valid = walk(k)
            t1 = clock()
            t_walk += t1 - t0
            if walks.numel():
                idx = sgns.corpus_index(walks, valid, token_index)
                done = (ep * n_batches + k) / (epochs * n_batches)
                a = max(min_alpha, alpha - (alpha - min_alpha) * done)
                # sentence id = (epoch, rank, row of the rank's virtual corpus): never repeats
                base = ((ep * world + rank) * rows_rank_max + k * batch_vertices * W) * parts_per_row
                for j, part in enumerate(torch.split(sgns.split_rows(idx), 1 << 22)):
                    model.train_block(part, a, base + j * (1 << 22), deterministic, sched,
                                      k * batch_vertices * W + j * (1 << 22))
            if sync is not None:
                sync.step()
            t2 = clock()
            t_train += t2 - t1
            if timings is not None:  # (per batch: walking, training)
                timings.setdefault("batch_s", []).append((t1 - t0, t2 - t1))
    if sync is not None:
        sync.finish()
    torch.cuda.synchronize(dev)
    raise_walk_status()
    if timings is not None:
        timings.update(walk_s=t_walk, train_s=t_train, batches=n_batches, epochs=epochs,
                       rows_this_rank=n_start * W, world=world)
    p["negative"] = negative
    # what the trainer really ran with (hub_rows None = chosen from the corpus: recorded)
    p["hub_rows"], p["hub_rows_auto"], p["hub_waves"] = model.hub_rows, model.hub_rows_auto, model.hub_waves
    if timings is not None:
        timings.update(hub_rows=model.hub_rows, hub_rows_auto=model.hub_rows_auto, hub_waves=model.hub_waves)
    # device-backed result: no host copy of the matrices, integer ids instead of token strings
    out = HipW2V(KeyedVectors(vocab.ids, model.syn0), model.syn1neg, p, int(model.pairs.item()))
    return (out, model) if return_model else out
