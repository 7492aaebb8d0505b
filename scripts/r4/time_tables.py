"""one-off tables of exact biased walks on a BASELINE graph, timed: edge classes, wedge table (+ slots),
hop table.  GRAPH=cfg4|cfg3|cfg2 python scripts/r4/time_tables.py"""
import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from node2vec_amd import synthetic
cfg = os.environ.get("GRAPH", "cfg4")
if cfg == "cfg4":
    g = synthetic.chung_lu(100_000_000, 500_000_000, device="cuda").trimmed(10_000, 42)
elif cfg == "cfg3":
    g = synthetic.chung_lu(10_000_000, 100_000_000, device="cuda").trimmed(10_000, 42)
else:
    g = synthetic.rmat(20, 5_000_000, device="cuda")


def timed(what, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    print(f"{cfg} {what}: {time.perf_counter() - t0:.3f} s", flush=True)


timed("edge classes", g.build_edge_classes)
timed("wedge table + slots", g.build_wedges)
timed("hop table (inline return positions)", lambda: g.build_hops(inline_rpos=True))
print(cfg, "checksums", int(g.edge_classes.long().sum()), int(g.wedge_off.sum()), int(g.wedge_pos.long().sum()),
      int(g.wedge_slots.long().sum()))
