"""weighted exact walks step by step (n2v_walk_weighted_step): the loop issued from the host against the same loop
captured into one hipGraph, by batch size; weighted cfg 2, (0.5, 2):   python scripts/r6/time_weighted_graph.py"""
import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from node2vec_amd import synthetic, randomwalk as rw
g = synthetic.rmat(20, 5_000_000, device="cuda", weights="uniform")
start_all = rw.start_vertices(g)
rw.weighted_lanes_tables(g); rw.weighted_row_sums(g); rw.weighted_hub_summaries(g)
p, q = 0.5, 2.0
ref = {}
for batch in [int(x) for x in os.environ.get("BATCHES", "1024,4710,47104,%d" % start_all.numel()).split(",")]:
    start = start_all[:batch].contiguous()
    for graph in (False, True):
        rw.WEIGHTED_LANES_GRAPH = graph
        best, st = 1e9, {}
        for it in range(3):
            st = {}
            torch.cuda.synchronize(); t = time.time()
            walks, valid = rw.walk(g, start, 10, 80, p, q, 42, use_weighted_lanes=True, stats=st)
            torch.cuda.synchronize(); best = min(best, time.time() - t)
        steps = int(valid.sum()) * 80
        same = ""
        if graph:
            same = f" identical to the eager loop: {bool(torch.equal(walks, ref['w']) and torch.equal(valid, ref['v']))}"
        else:
            ref = {"w": walks, "v": valid}
        print(f"{batch * 10:8d} walkers, graph={st.get('graph')}: {best * 1e3:8.1f} ms = {steps / best / 1e6:8.1f} M steps/s"
              f" undecided {int(st['undecided'])}{same}", flush=True)
    if batch == 47104:
        torch.cuda.synchronize(); t = time.time()
        w2, v2 = rw.walk(g, start, 10, 80, p, q, 42, use_weighted_lanes=False)
        torch.cuda.synchronize(); dt = time.time() - t
        print(f"{batch * 10:8d} walkers, one-launch wave kernel: {dt * 1e3:8.1f} ms = {steps / dt / 1e6:8.1f} M steps/s "
              f"identical={bool(torch.equal(walks, w2))}", flush=True)
