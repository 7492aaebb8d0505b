"""BASELINE cfg 4 end to end on ONE MI355X through pipeline.fit_streaming:
10^8 vertices, 10 walks x 80 steps per vertex, 1 epoch, sample = 0, min_count = 0, dim 128.
usage: e2e_cfg4.py [default|batched] [fraction of the start vertices, default 1.0]   (E2E_PQ="0.5,2": p, q)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from node2vec_amd import randomwalk as rw  # noqa: E402
from node2vec_amd import synthetic  # noqa: E402
from node2vec_amd.pipeline import fit_streaming  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "default"
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
n = int(os.environ.get("E2E_VERTICES", 100_000_000))
P_, Q_ = (float(x) for x in os.environ.get("E2E_PQ", "1,1").split(","))
t0 = time.perf_counter()
g = synthetic.chung_lu(n, 5 * n, seed=42, device="cuda").trimmed(10_000, 42)
torch.cuda.synchronize()
t_graph = time.perf_counter() - t0
start = rw.start_vertices(g)
seed_ids = None
if frac < 1.0:
    seed_ids = start[: int(frac * start.numel())]
timings = {}
import threading


def heartbeat():  # gpurun kills a command that is silent for 7 minutes
    while True:
        time.sleep(60)
        print(f"[e2e] {time.perf_counter() - t0:.0f} s", file=sys.stderr, flush=True)


t0 = time.perf_counter()
threading.Thread(target=heartbeat, daemon=True).start()
model = fit_streaming(g, {"num_walks": 10, "walk_length": 80, "return_param": P_, "inout_param": Q_},
                      {"size": 128, "window": 5, "negative": 5, "iter": 1, "sample": 0, "min_count": 0,
                       "seed": 1, "batched": mode == "batched"},
                      random_seed=42, batch_vertices=1 << 20, walk_seed_ids=seed_ids, timings=timings)
torch.cuda.synchronize()
total = time.perf_counter() - t0
out = {"config": f"cfg4 chung_lu {g.n_vertices} vertices / {g.n_edges} edges, fraction {frac} of "
                 f"{int(start.numel())} start vertices, p={P_} q={Q_}, W=10, L=80, dim 128, 1 epoch, mode {mode}",
       "graph_build_s": t_graph, "fit_streaming_s": total, "walk_s": timings["walk_s"],
       "train_s": timings["train_s"], "other_s": total - timings["walk_s"] - timings["train_s"],
       "pairs": model.pairs_trained, "pairs_per_s_train_only": model.pairs_trained / timings["train_s"],
       "walk_steps": timings["rows_this_rank"] * 80 * 2,  # counting pass + training pass
       "vocabulary": len(model.wv), "hbm_peak_GB": torch.cuda.max_memory_allocated() / 1e9,
       "train_s_per_batch": [round(b[1], 3) for b in timings.get("batch_s", [])]}
print(json.dumps(out), flush=True)
