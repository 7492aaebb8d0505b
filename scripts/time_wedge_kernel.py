"""time the biased exact leg of bench.py (cfg 4, p = 0.5, q = 2, 131072 start vertices x 10 x 80)
with a variant library loaded by path: python scripts/time_wedge_kernel.py <label>"""
import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from node2vec_amd import _lib
if os.environ.get("N2V_VARIANT_LIB"):
    _lib.LIB_PATH = os.environ["N2V_VARIANT_LIB"]  # developer build, loaded by path
from node2vec_amd import synthetic, randomwalk as rw
cfg = os.environ.get("GRAPH", "cfg4")
if cfg == "cfg4":
    g = synthetic.chung_lu(100_000_000, 500_000_000, device="cuda").trimmed(10_000, 42)
elif cfg == "cfg3":
    g = synthetic.chung_lu(10_000_000, 100_000_000, device="cuda").trimmed(10_000, 42)
elif cfg == "cfg5":
    g = synthetic.hub_bipartite(50_000_000, 5000, 10_000, device="cuda")
else:
    g = synthetic.rmat(20, 5_000_000, device="cuda")
start = rw.start_vertices(g)
b = min(131072, start.numel())
walks = torch.empty((b * 10, 81), dtype=torch.int32, device="cuda")
valid = torch.empty(b * 10, dtype=torch.uint8, device="cuda")
for pq in os.environ.get("PQ", "0.5,2.0").split(";"):  # PQ="0.5,2;4,2": several on one graph
    P_, Q_ = (float(x) for x in pq.split(","))
    def run(k):
        rw.walk(g, start[k * b:(k + 1) * b], 10, 80, P_, Q_, 42, out=(walks, valid), check=False)
    run(0); run(1); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(2, 12): run(k % max(1, start.numel() // b))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{sys.argv[1] if len(sys.argv) > 1 else ''}: {cfg} p={P_} q={Q_} exact {b * 800 / dt / 1e9:.2f} G steps/s ({dt * 1e3:.2f} ms per launch), checksum {int(walks.long().sum())}", flush=True)
