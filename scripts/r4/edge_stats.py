"""what a biased step finds, by edge: distribution of the shared count nM and of the degree of
the vertex stood on, over the directed edges of a config (a walk on a symmetric graph visits
directed edges ~uniformly).  GRAPH=cfg4|cfg3|cfg2 python scripts/r4/edge_stats.py"""
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from node2vec_amd import synthetic, randomwalk as rw
cfg = os.environ.get("GRAPH", "cfg4")
if cfg == "cfg4":
    g = synthetic.chung_lu(100_000_000, 500_000_000, device="cuda").trimmed(10_000, 42)
elif cfg == "cfg3":
    g = synthetic.chung_lu(10_000_000, 100_000_000, device="cuda").trimmed(10_000, 42)
else:
    g = synthetic.rmat(20, 5_000_000, device="cuda")
g.build_edge_classes()
ec = g.edge_classes.view(torch.int32).long() & 0xffffffff
nM = ec & 0xffffff
nR = ec >> 24
deg = g.degrees()
dv = deg[g.col.long()]          # degree of the vertex the walker stands on after walking e
E = g.n_edges
def share(mask): return float(mask.sum()) / E
print(cfg, "edges", E, "mean nM", float(nM.float().mean()), "nR==1", share(nR == 1), "nR==0", share(nR == 0))
for lo, hi in ((0, 0), (1, 3), (4, 12), (13, 28), (29, 60), (61, 256), (257, 1 << 30)):
    print(f"  nM in [{lo},{hi}]: {share((nM >= lo) & (nM <= hi)):.4f}")
for d in (8, 16, 24, 32, 56, 64, 128, 1024):
    m = dv <= d
    print(f"  deg(v) <= {d}: {share(m):.4f}   of which nM > 0: {share(m & (nM > 0)):.4f}")
m = nM > 0
print("  nM > 0:", share(m), " and deg(v) > 24:", share(m & (dv > 24)), " and deg(v) > 56:", share(m & (dv > 56)))
# list bytes if lists of <= 3 entries were inline
print("  entries total", int(nM.sum()), " in lists <= 3:", int(nM[nM <= 3].sum()), " <= 28:", int(nM[nM <= 28].sum()))
