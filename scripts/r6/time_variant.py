"""one library build (N2V_HIP_LIB), one graph, several (p, q): the exact biased walk timed (timing-only ablation
builds of scripts/r4/build_wedge_variants.sh give different walks -- nothing is compared).
  GRAPH=cfg4|cfg3 TRIM=100000 PQ="0.5,2;3,0.7;4,0.25" REPS=4 python scripts/r6/time_variant.py <label>"""
import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from node2vec_amd import synthetic, randomwalk as rw
label = sys.argv[1] if len(sys.argv) > 1 else ""
TRIM = int(os.environ.get("TRIM", 100_000))
if os.environ.get("GRAPH", "cfg4") == "cfg4":
    g = synthetic.chung_lu(100_000_000, 500_000_000, device="cuda").trimmed(TRIM, 42)
else:
    g = synthetic.chung_lu(10_000_000, 100_000_000, device="cuda").trimmed(TRIM, 42)
start = rw.start_vertices(g)
b = min(int(os.environ.get("BATCH", 1 << 20)), start.numel())
nb = max(1, start.numel() // b)
walks = torch.empty((b * 10, 81), dtype=torch.int32, device="cuda")
valid = torch.empty(b * 10, dtype=torch.uint8, device="cuda")
reps = int(os.environ.get("REPS", 4))
for pq in os.environ.get("PQ", "0.5,2.0;3,0.7;4,0.25").split(";"):
    P_, Q_ = (float(x) for x in pq.split(","))

    def run(k):
        rw.walk(g, start[(k % nb) * b:(k % nb + 1) * b], 10, 80, P_, Q_, 42, out=(walks, valid), check=False,
                mode=os.environ.get("MODE", "exact"),
                use_workspace=bool(os.environ.get("USE_WS")))  # (USE_WS=1 + a `make WEDGE2=1` library: the passes of n2v_walk_ws)

    run(0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(1, 1 + reps): run(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{label}: trim {TRIM} p={P_} q={Q_}: {b * 800 / dt / 1e9:.2f} G steps/s ({dt * 1e3:.2f} ms) slots {g.wedge_slots is not None}",
          flush=True)
