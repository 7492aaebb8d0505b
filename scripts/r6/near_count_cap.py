"""diagnostic (build_variants/libn2v_wedge_nearcount.so, -DN2V_NEAR_COUNT=1): of the steps of an exact walk, how many run
the pairing (dyadic p, q) / pass the quick accept (other values), and how many of those the closed forms decline --
on the graph trimmed at TRIM (default: the reference's cap of 100 000).  GRAPH=cfg4|cfg3 TRIM= PQ="3,0.7;0.5,2" """
import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from node2vec_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "build_variants", "libn2v_wedge_nearcount.so")
from node2vec_amd import synthetic, randomwalk as rw
TRIM = int(os.environ.get("TRIM", 100_000))
if os.environ.get("GRAPH", "cfg4") == "cfg4":
    g = synthetic.chung_lu(100_000_000, 500_000_000, device="cuda").trimmed(TRIM, 42)
else:
    g = synthetic.chung_lu(10_000_000, 100_000_000, device="cuda").trimmed(TRIM, 42)
start = rw.start_vertices(g)[:1 << 18].contiguous()
for pq in os.environ.get("PQ", "3,0.7;0.5,2;4,0.25").split(";"):
    p, q = (float(x) for x in pq.split(","))
    st = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    walks, valid = rw.walk(g, start, 10, 80, p, q, 42, stats=st)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    steps = int(valid.sum()) * 80
    s = st["status"].cpu().numpy().astype("uint32")
    dyadic = all((1.0 / x) == 2.0 ** round(__import__("math").log2(1.0 / x)) for x in (p, q))
    what = "pairings" if dyadic else "past the quick accept"
    print(f"trim {TRIM} p={p} q={q}: steps {steps} in {dt * 1e3:.0f} ms (first call, tables included); {what} {int(s[2])} "
          f"({s[2] / steps:.4f}); declined by the closed forms {int(s[3])} ({s[3] / max(int(s[2]), 1):.5f} of those)", flush=True)
cnt = (g.edge_classes & 0xffffff).long()
E = g.n_edges
edges = [0, 1, 15, 65, 577, 2049, 1 << 30]
print("share of edges by length of their shared list: " + ", ".join(
    f"[{a}, {b}): {float(((cnt >= a) & (cnt < b)).sum()) / E:.4f}" for a, b in zip(edges[:-1], edges[1:])), flush=True)
print(f"slots {g.wedge_slots is not None}, index {g.wedge_index is not None}, mean list {float(cnt.sum()) / E:.1f}", flush=True)
