"""The multi-GPU paths with TWO real ranks on real kernels.

One MI355X box has one GPU and RCCL refuses two ranks on one device, so the two ranks are two
processes on cuda:0 joined by gloo (device tensors are staged through the host by
shard.all_reduce; everything else -- the walk sharding, the globally summed vocabulary, the block
grid over the largest shard, DeltaSync's pack / apply kernels, its side-stream overlap, the
period autotune -- is the code that runs under RCCL).  Checked:
  * Node2VecHIP.fit() under the group with uneven shards, and with one EMPTY shard; fp32 and
    bf16 wire; overlap on; sync_every fixed and autotuned: every rank returns bit-identical
    matrices, and each rank trained exactly the pairs a single process trains on that shard with
    that rank's sentence ids;
  * pipeline.fit_streaming under the group: identical replicas, the vocabulary of the whole
    corpus, the pair count of the two shards;
  * partitioned.walk_partitioned under the group: each rank holds HALF of the CSR, steps its
    resident walkers with n2v_partition_step and exchanges the migrating ones (headers + the
    travelling wedge lists) through all_to_all_single; the two halves of the output are the rows
    of n2v_walk on the whole graph, bit for bit -- also when one rank lacks the per-edge tables
    (both then fall back to travelling rows).
"""
import hashlib
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

N2V = {"num_walks": 4, "walk_length": 20, "return_param": 0.5, "inout_param": 2.0}
W2V = {"min_count": 1, "iter": 2, "size": 64, "negative": 5, "sample": 0.0, "seed": 5, "window": 5}
SCENARIOS = [  # (name, share of the walks on rank 0, wire, sync_every)
    ("uneven_fp32", 0.7, "fp32", 2),
    ("uneven_bf16_autotune", 0.7, "bf16", None),
    ("empty_shard_fp32_autotune", 1.0, "fp32", None),
    ("empty_shard_bf16", 1.0, "bf16", 1),
]


PART_PQ = [(0.5, 2.0), (2.0, 1.0)]  # the row of the previous vertex travels / does not


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _digest(t):
    return hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()


def _corpus():
    from node2vec_amd import randomwalk as rw
    from node2vec_amd import synthetic

    g = synthetic.rmat(11, 12000, seed=3, device="cuda:0")
    start = rw.start_vertices(g)
    walks, valid = rw.walk(g, start, N2V["num_walks"], N2V["walk_length"], N2V["return_param"],
                           N2V["inout_param"], 17)
    return g, walks[valid]


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from node2vec_amd.embedding import Node2VecHIP
        from node2vec_amd.pipeline import fit_streaming

        g, walks = _corpus()
        out = {}
        for name, share0, wire, every in SCENARIOS:
            cut = int(round(share0 * walks.shape[0]))
            mine = walks[:cut] if rank == 0 else walks[cut:]
            params = dict(W2V, sync_wire=wire)
            if every is not None:
                params["sync_every"] = every
            n2v = Node2VecHIP(mine.clone(), params, random_seed=5)
            model = n2v.fit()
            out[name] = {"rows": int(mine.shape[0]), "pairs": int(model.pairs_trained),
                         "syn0": _digest(model.wv._vectors), "syn1neg": _digest(model._syn1neg),
                         "n_vocab": len(model.wv), "finite": bool(torch.isfinite(model.wv._vectors).all())}
        t = {}
        sm, raw = fit_streaming(g, dict(N2V), dict(W2V, sync_wire="fp32", sync_every=3), random_seed=17,
                                batch_vertices=200, return_model=True, timings=t)
        out["streaming"] = {"pairs": int(sm.pairs_trained), "syn0": _digest(raw.syn0),
                            "syn1neg": _digest(raw.syn1neg), "n_vocab": len(sm.wv),
                            "ids": _digest(raw.vocab.ids), "counts": _digest(raw.vocab.counts),
                            "world": t["world"], "rows": t["rows_this_rank"]}
        # p = q = 1: the walks run on the degree-ranked form and come out in ranks on every rank
        n11 = dict(N2V, return_param=1.0, inout_param=1.0)
        sm1, raw1 = fit_streaming(g, n11, dict(W2V, sync_wire="fp32", sync_every=3), random_seed=17,
                                  batch_vertices=200, return_model=True)
        out["streaming_ranks"] = {"pairs": int(sm1.pairs_trained), "syn0": _digest(raw1.syn0),
                                  "syn1neg": _digest(raw1.syn1neg), "ids": _digest(raw1.vocab.ids),
                                  "counts": _digest(raw1.vocab.counts), "ranked": g.rank_hops is not None}
        from node2vec_amd import partitioned as P
        from node2vec_amd import randomwalk as rw

        part = P.partition_graph(g, world)[rank]
        start = rw.start_vertices(g)
        for pq in PART_PQ:
            tp = {}
            pw, pv, prow = P.walk_partitioned(part, start, 2, 15, pq[0], pq[1], 31, timings=tp)
            out["partitioned_%g_%g" % pq] = {"walks": pw.cpu().numpy(), "valid": pv.cpu().numpy(),
                                             "rows": prow.cpu().numpy(), "edges": int(part.col.numel()),
                                             "bounded_attempts": len(tp.get("bounded_caps", []))}
        # one rank WITHOUT the per-edge tables: both must fall back to rows travelling
        import dataclasses

        assert part.wedge_off is not None
        bare = part if rank == 0 else dataclasses.replace(part, edge_classes=None, wedge_off=None,
                                                          wedge_pos=None)
        pw, pv, prow = P.walk_partitioned(bare, start, 2, 15, 0.5, 2.0, 31)
        out["partitioned_mixed"] = {"walks": pw.cpu().numpy(), "valid": pv.cpu().numpy(),
                                    "rows": prow.cpu().numpy(), "edges": int(part.col.numel())}
        ret[rank] = out
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_fit_and_fit_streaming():
    import torch.multiprocessing as mp

    from node2vec_amd import sgns
    from node2vec_amd.pipeline import fit_streaming
    from node2vec_amd.shard import sentence_base, shard_range

    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]

    g, walks = _corpus()
    vocab = sgns.build_vocab(walks, 1)
    idx_all = vocab.index_of[walks.long()]
    for name, share0, wire, every in SCENARIOS:
        a, b = r0[name], r1[name]
        assert a["finite"] and b["finite"], name
        # identical replicas after the final blocking exchange
        assert a["syn0"] == b["syn0"] and a["syn1neg"] == b["syn1neg"], name
        assert a["n_vocab"] == b["n_vocab"] == len(vocab), name
        cut = int(round(share0 * walks.shape[0]))
        assert (a["rows"], b["rows"]) == (cut, walks.shape[0] - cut), name
        # each rank trained the pairs a single process trains on that shard with that rank's
        # sentence ids (the block grid is laid over the largest shard)
        rows_max = max(cut, walks.shape[0] - cut)
        for rank, got in ((0, a), (1, b)):
            shard = idx_all[:cut] if rank == 0 else idx_all[cut:]
            m = sgns.SgnsModel(vocab, 64, 5, 5, seed=5, sample=0.0)
            m.train(shard, 2, 0.025, 1e-4, rows_global_max=rows_max,
                    sentence_base=sentence_base(rank, world, rows_max * 2))
            torch.cuda.synchronize()
            assert int(m.pairs.item()) == got["pairs"], (name, rank)
        if share0 == 1.0:
            assert b["pairs"] == 0 and a["pairs"] > 0

    from node2vec_amd import randomwalk as rw

    start = rw.start_vertices(g)
    for pq, key in [(pq, "partitioned_%g_%g" % pq) for pq in PART_PQ] + [((0.5, 2.0), "partitioned_mixed")]:
        want, wv = rw.walk(g, start, 2, 15, pq[0], pq[1], 31)
        want, wv = want.cpu().numpy(), wv.cpu().numpy().astype(bool)
        pa, pb = r0[key], r1[key]
        assert 0 < pa["edges"] < g.n_edges and pa["edges"] + pb["edges"] == g.n_edges
        if key != "partitioned_mixed":  # the steps after the calibration ran with capacity-bounded mailboxes
            assert pa["bounded_attempts"] == pb["bounded_attempts"] >= 1
        rows = np.concatenate([pa["rows"], pb["rows"]])
        assert np.array_equal(np.sort(rows), np.arange(want.shape[0]))  # every row emitted once
        for part in (pa, pb):
            assert len(part["rows"]) > 0
            assert np.array_equal(part["valid"].astype(bool), wv[part["rows"]]), pq
            assert np.array_equal(part["walks"], want[part["rows"]]), pq

    sa, sb = r0["streaming"], r1["streaming"]
    assert sa["world"] == sb["world"] == 2
    assert sa["syn0"] == sb["syn0"] and sa["syn1neg"] == sb["syn1neg"]
    assert sa["ids"] == sb["ids"] == _digest(vocab.ids) and sa["counts"] == sb["counts"] == _digest(vocab.counts)
    assert sa["n_vocab"] == len(vocab)
    from node2vec_amd import randomwalk as rw

    n_start = int(rw.start_vertices(g).numel())
    lo0, hi0 = shard_range(n_start, 0, 2)
    assert sa["rows"] == (hi0 - lo0) * N2V["num_walks"] and sa["rows"] + sb["rows"] == n_start * N2V["num_walks"]
    # a single process streaming the whole graph trains a different schedule (other sentence
    # ids), but the same corpus: pair counts agree to a fraction of a per cent
    # the same at p = q = 1 (walks in degree ranks): the vocabulary of the whole corpus in vertex ids
    ra, rb = r0["streaming_ranks"], r1["streaming_ranks"]
    assert ra["ranked"] and rb["ranked"]
    assert ra["syn0"] == rb["syn0"] and ra["syn1neg"] == rb["syn1neg"]
    w11, v11 = rw.walk(g, rw.start_vertices(g), N2V["num_walks"], N2V["walk_length"], 1.0, 1.0, 17)
    vocab11 = sgns.build_vocab(w11[v11], 1)
    assert ra["ids"] == rb["ids"] == _digest(vocab11.ids) and ra["counts"] == rb["counts"] == _digest(vocab11.counts)
    single = fit_streaming(g, dict(N2V), dict(W2V), random_seed=17, batch_vertices=200)
    tot = sa["pairs"] + sb["pairs"]
    assert abs(tot - single.pairs_trained) / single.pairs_trained < 0.01, (tot, single.pairs_trained)
    assert np.isfinite(single.wv.vectors).all()


def test_bench_under_torchrun_rehearses_the_rccl_exchange():
    """`bench.py --gpus 1` launched as the driver launches the N > 1 case (torch.distributed.run,
    one rank): the process group is RCCL ("nccl"), and the exchange step of the SGNS path -- pack,
    all_to_all / all_gather of bytes over RCCL, the rank-ordered fp32 sum, apply -- runs on device
    tensors on a group of one rank.  The N = 8 run on the 8-GPU node is the driver's; this is the
    same code with world = 1."""
    import json
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "cfg2", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--no-fast", "--no-regimes", "--no-biased",
           "--no-batched", "--no-hub"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    run = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0
    api = out["api"]  # random_walk() -> DataFrame -> Node2VecGensim.fit() -> embedding(), wall clock per call
    assert api["rows"] > 4_000_000 and api["random_walk_s"] > api["random_walk_device_s"] > 0
    assert api["fit_s"] > 0 and api["embedding_s"] > 0 and api["with_arrow_backed_columns"]["pairs"] == api["with_arrow_backed_columns"]["pairs"]
    ex = out["sgns"]["exchange"]
    assert ex["world"] == 1 and ex["backend"] == "nccl (RCCL)" and ex["wire_dtype"] == "bf16"
    assert ex["tensors_on_device"] is True and ex["blocks_exchanged"] >= 2  # syn0 and syn1neg
    assert ex["delta_allreduce_s"] > 0


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus N` started as ONE process (no RANK in the environment) starts the N
    ranks itself: a child `python -m torch.distributed.run --nproc-per-node N bench.py ...` (never an
    exec; the parent makes no GPU call), relays the one JSON line and returns the child's exit code.
    One GPU here, so N = 1 through the same path (`--spawn` forces it at N = 1; at N > 1 it is what
    `--gpus N` does): `n_gpus` == --gpus and the exchange ran on a group of that size."""
    import json
    import subprocess

    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--config", "cfg2",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-fast", "--no-regimes", "--no-biased",
           "--no-batched", "--no-hub", "--no-api"]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    run = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0
    ex = out["sgns"]["exchange"]  # only a run under torch.distributed has it: the ranks were spawned
    assert ex["world"] == 1 and ex["backend"] == "nccl (RCCL)"
    assert ex["hbm_peak_allocated_GB_with_exchange_live"] > 0
    assert out["summary"]["sgns_exchange_world"] == 1
    assert list(out)[-1] == "summary" and "roofline" in out and "api" not in out
    # a rank count that contradicts --gpus is refused before anything touches the GPU
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], cwd=ROOT,
                         env=dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"), capture_output=True,
                         text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr
