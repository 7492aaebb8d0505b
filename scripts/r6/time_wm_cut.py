"""weighted exact walks step by step: the cut between the lane-per-walker and the wave-per-walker margin kernels
(n2v_weighted_hubs.lane_cut; 0 = the library's choice from the batch) by batch size; weighted cfg 2, (0.5, 2):
   python scripts/r6/time_wm_cut.py"""
import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from node2vec_amd import synthetic, randomwalk as rw
kind = os.environ.get("KIND", "fp32")
g = synthetic.rmat(20, 5_000_000, device="cuda", weights="uniform")
if kind == "int":
    from node2vec_amd.graph import DeviceGraph
    gen = torch.Generator(device="cuda").manual_seed(1)
    g = DeviceGraph(g.rowptr, g.col, torch.randint(1, 5, (g.n_edges,), generator=gen, device="cuda").float())
start_all = rw.start_vertices(g)
rw.weighted_lanes_tables(g); rw.weighted_row_sums(g); rw.weighted_hub_summaries(g)
p, q = (float(x) for x in os.environ.get("PQ", "0.5,2.0").split(","))
for batch in [int(x) for x in os.environ.get("BATCHES", "1024,4710,47104,%d" % start_all.numel()).split(",")]:
    start = start_all[:batch].contiguous()
    ref = None
    for cut in [int(x) for x in os.environ.get("CUTS", "768,0,32,64,128,256,512").split(",")]:
        rw.WEIGHTED_LANE_CUT = cut
        best, st = 1e9, {}
        for it in range(3):
            st = {}
            torch.cuda.synchronize(); t = time.time()
            walks, valid = rw.walk(g, start, 10, 80, p, q, 42, use_weighted_lanes=True, stats=st)
            torch.cuda.synchronize(); best = min(best, time.time() - t)
        steps = int(valid.sum()) * 80
        if ref is None:
            ref = (walks, valid)
        same = bool(torch.equal(walks, ref[0]) and torch.equal(valid, ref[1]))
        print(f"{kind} ({p:g}, {q:g}) {int(valid.numel()):8d} walkers, cut {cut:4d}: {best * 1e3:8.1f} ms = {steps / best / 1e6:8.1f} M steps/s "
              f"undecided {int(st['undecided'])} identical to the first cut: {same}", flush=True)
    if os.environ.get("OLD"):
        torch.cuda.synchronize(); t = time.time()
        w2, v2 = rw.walk(g, start, 10, 80, p, q, 42, use_weighted_lanes=False)
        torch.cuda.synchronize(); dt = time.time() - t
        print(f"{kind} ({p:g}, {q:g}) {int(v2.numel()):8d} walkers, one-launch wave kernel: {dt * 1e3:8.1f} ms = {steps / dt / 1e6:8.1f} M steps/s "
              f"identical={bool(torch.equal(ref[0], w2))}", flush=True)
