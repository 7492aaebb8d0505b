"""How large would per-edge shared-position lists be?  sum over edges (s -> v) of the number of
neighbours of v that are neighbours of s (edge_classes low 24 bits), by graph.
GRAPH=cfg2|cfg3|cfg4 python scripts/wedge_size.py"""
import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from node2vec_amd import synthetic
for name in os.environ.get("GRAPH", "cfg2,cfg3,cfg4").split(","):
    if name == "cfg4":
        g = synthetic.chung_lu(100_000_000, 500_000_000, device="cuda").trimmed(10_000, 42)
    elif name == "cfg3":
        g = synthetic.chung_lu(10_000_000, 100_000_000, device="cuda").trimmed(10_000, 42)
    else:
        g = synthetic.rmat(20, 5_000_000, device="cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g.build_edge_classes(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ec = g.edge_classes
    nm = (ec & 0xffffff).long()
    nr = ((ec >> 24) & 0xff).long()
    sat = int(((ec & 0xffffff) == 0xffffff).sum())
    deg = g.degrees()
    src = torch.repeat_interleave(torch.arange(g.n_vertices, device="cuda"), deg)
    # visit-weighted: a walker crosses edge (s -> v) with probability ~ 1/E at stationarity (unit weights)
    q = torch.tensor([0.5, 0.9, 0.99, 0.999], device="cuda", dtype=torch.float64)
    samp = nm[torch.randint(0, nm.numel(), (5_000_000,), device="cuda")].double()
    print(f"{name}: V={g.n_vertices} E={g.n_edges} classes_build={dt:.2f}s sum_nM={int(nm.sum())} "
          f"({int(nm.sum())/g.n_edges:.2f} per edge) max_nM={int(nm.max())} saturated={sat} "
          f"edges_with_nM>0={int((nm>0).sum())/g.n_edges:.3f} nR_max={int(nr.max())} "
          f"quantiles(50,90,99,99.9)={[int(x) for x in torch.quantile(samp, q)]}", flush=True)
    dv = deg[g.col.long()]
    for lo, hi in ((0, 64), (64, 1024), (1024, 4096), (4096, 1 << 30)):
        sel = (dv > lo) & (dv <= hi)
        print(f"   edges into deg(v) in ({lo},{hi}]: {int(sel.sum())/g.n_edges:.4f} of edges, sum_nM {int(nm[sel].sum())}", flush=True)
    del g, ec, nm, nr, src, dv
    torch.cuda.empty_cache()
