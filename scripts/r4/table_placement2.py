"""placement sensitivity of the p = q = 1 kernels, one library variant per process (N2V_HIP_LIB): table as built,
table cloned, output buffer moved -- ranked (3 GB) and hop table (12 GB)"""
import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import bench
from node2vec_amd import synthetic, randomwalk as rw
tag = os.path.basename(os.environ.get("N2V_HIP_LIB", "default"))
g = synthetic.chung_lu(100_000_000, 500_000_000, device="cuda").trimmed(10_000, 42)
start = rw.start_vertices(g)
g.build_ranked(); g.build_hops()
def t(leg):
    for k in range(2): leg.step(k)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for k in range(2, 8): leg.step(k)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 6
leg = bench.WalkLeg(torch, rw, g, start, 10, 80, 1.0, 1.0, "exact", 1 << 20, 0, 1, rank_ids=True)
legv = bench.WalkLeg(torch, rw, g, start, 10, 80, 1.0, 1.0, "exact", 1 << 20, 0, 1)
res = {"ranked": [t(leg)], "hops": [t(legv)]}
old, oldh = g.rank_hops, g.hops
keep = []
for rep in range(3):
    g.rank_hops = old.clone(); keep.append(g.rank_hops)
    res["ranked"].append(t(leg))
    leg.walks = torch.empty_like(leg.walks); keep.append(leg.walks)
    res["ranked"].append(t(leg))
g.rank_hops = old
for rep in range(2):
    g.hops = oldh.clone()
    res["hops"].append(t(legv))
    legv.walks = torch.empty_like(legv.walks); keep.append(legv.walks)
    res["hops"].append(t(legv))
    g.hops = oldh
print(tag, "ranked ms over placements:", [round(x, 2) for x in res["ranked"]], " hop table ms:", [round(x, 2) for x in res["hops"]], flush=True)
