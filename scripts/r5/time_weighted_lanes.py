"""weighted exact walks at p, q != 1: the lane-per-walker step kernel (n2v_walk_weighted_step) against the
wave-per-walker kernel (walk_exact_kernel), weighted cfg 2 (R-MAT 1 M / 10 M, fp32 weights U[0.1, 2]):
  BATCH=47104 PQ="0.5,2.0;3.0,0.7" OLD=1 python scripts/r5/time_weighted_lanes.py"""
import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from node2vec_amd import synthetic, randomwalk as rw
from node2vec_amd.graph import DeviceGraph
base = synthetic.rmat(20, 5_000_000, device="cuda")
gen = torch.Generator(device="cuda").manual_seed(1)
kinds = {"int": torch.randint(1, 5, (base.n_edges,), generator=gen, device="cuda").float(),  # few values: sums tie
         "fp32": (torch.rand(base.n_edges, generator=gen, device="cuda") * 1.9 + 0.1),
         "fp64": (torch.rand(base.n_edges, generator=gen, device="cuda", dtype=torch.float64) * 1.9 + 0.1)}
B = int(os.environ.get("BATCH", 47104))
if os.environ.get("HUB_SLOTS"):
    rw.WEIGHTED_HUB_SLOTS = int(os.environ["HUB_SLOTS"])  # rows with block summaries (0: none)
for kind in os.environ.get("KINDS", "fp32").split(","):
    g = DeviceGraph(base.rowptr, base.col, kinds[kind])
    start_all = rw.start_vertices(g)
    t0 = time.time(); rw.weighted_lanes_tables(g); torch.cuda.synchronize()
    print(f"{kind}: per-edge tables of the weighted graph built in {time.time() - t0:.2f} s "
          f"(wedge lists {g.wedge_pos.numel() * 2 / 1e6:.0f} MB)", flush=True)
    for pq in os.environ.get("PQ", "0.5,2.0").split(";"):
        p, q = (float(x) for x in pq.split(","))
        for batch in sorted({B, min(start_all.numel(), int(os.environ.get("BIG", start_all.numel())))}):
            start = start_all[:batch].contiguous()
            for margins in ([True, False] if os.environ.get("BOTH") else [rw.WEIGHTED_LANES_MARGINS]):
                rw.WEIGHTED_LANES_MARGINS = margins
                best, st = 1e9, {}
                for it in range(2):
                    st = {}
                    torch.cuda.synchronize(); t = time.time()
                    walks, valid = rw.walk(g, start, 10, 80, p, q, 42, use_weighted_lanes=True, stats=st)
                    torch.cuda.synchronize(); best = min(best, time.time() - t)
                steps = int(valid.sum()) * 80
                print(f"{kind} p={p} q={q} {batch} start vertices: {'margin kernels (a lane per walker, a wave per walker on the long rows)' if margins else 'exact lane kernel'} "
                      f"{best * 1e3:8.1f} ms = {steps / best / 1e6:8.1f} M steps/s"
                      + (f"; walker-steps left to the exact wave kernel: {int(st['undecided'])} of {steps} ({int(st['second_chance'])} "
                         f"decided on the reference-order row sum)" if margins else ""), flush=True)
            rw.WEIGHTED_LANES_MARGINS = True
            if os.environ.get("OLD", "1") == "1" and batch == B:
                torch.cuda.synchronize(); t = time.time()
                w2, v2 = rw.walk(g, start, 10, 80, p, q, 42, use_weighted_lanes=False)
                torch.cuda.synchronize(); dt = time.time() - t
                print(f"{kind} p={p} q={q} {batch} start vertices: wave  {dt * 1e3:8.1f} ms = {steps / dt / 1e6:8.1f} M steps/s "
                      f"identical={bool(torch.equal(walks, w2) and torch.equal(valid, v2))}", flush=True)
