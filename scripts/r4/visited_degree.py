"""how long the row is that an exact step on a WEIGHTED graph must read: the degree of the vertex a
walker stands on, averaged over walk steps (a walk on a symmetric graph visits vertex v with
probability ~ deg(v), so the mean is sum d^2 / sum d), and the bytes that implies per step.
GRAPH=cfg2|cfg3|cfg4 python scripts/r4/visited_degree.py"""
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from node2vec_amd import synthetic, randomwalk as rw
cfg = os.environ.get("GRAPH", "cfg2")
if cfg == "cfg4":
    g = synthetic.chung_lu(100_000_000, 500_000_000, device="cuda").trimmed(10_000, 42)
elif cfg == "cfg3":
    g = synthetic.chung_lu(10_000_000, 100_000_000, device="cuda").trimmed(10_000, 42)
else:
    g = synthetic.rmat(20, 5_000_000, device="cuda")
deg = g.degrees().double()
print(cfg, "vertices", g.n_vertices, "edges", g.n_edges, "max degree", int(deg.max()),
      "mean degree", float(deg[deg > 0].mean()), "edge-weighted mean degree (sum d^2 / sum d)",
      float((deg * deg).sum() / deg.sum()))
start = rw.start_vertices(g)[: 1 << 16].contiguous()
walks, valid = rw.walk(g, start, 10, 80, 0.5, 2.0, 42)
d = g.degrees()[walks[valid][:, :-1].long()].double()
print("  measured over", d.numel(), "steps of exact walks at (0.5, 2): mean degree of the vertex stood on",
      float(d.mean()), "median", float(d.flatten().median()), "p90", float(d.flatten().kthvalue(int(0.9 * d.numel())).values))
print("  bytes of one pass over that row: weights fp32", 4 * float(d.mean()), "fp64", 8 * float(d.mean()))
