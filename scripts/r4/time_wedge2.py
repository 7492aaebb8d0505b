"""exact biased walks: the one-launch kernel (n2v_walk) against the passes over a workspace
(n2v_walk_ws, csrc/n2v_walk_wedge2.hip) -- same walks, timed on one graph for several (p, q).
  GRAPH=cfg4|cfg3|cfg5|cfg2  PQ="0.5,2;4,0.25"  BATCH=1048576  ROUNDS="4;2;6"  python scripts/r4/time_wedge2.py <label>"""
import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from node2vec_amd import _lib
if os.environ.get("N2V_VARIANT_LIB"):
    _lib.LIB_PATH = os.environ["N2V_VARIANT_LIB"]  # developer build, loaded by path
from node2vec_amd import synthetic, randomwalk as rw
label = sys.argv[1] if len(sys.argv) > 1 else ""
cfg = os.environ.get("GRAPH", "cfg4")
TRIM = int(os.environ.get("TRIM", 10_000))  # 100000 = the reference's default cap (constants.py:6)
if cfg == "cfg4":
    g = synthetic.chung_lu(100_000_000, 500_000_000, device="cuda").trimmed(TRIM, 42)
elif cfg == "cfg3":
    g = synthetic.chung_lu(10_000_000, 100_000_000, device="cuda").trimmed(TRIM, 42)
elif cfg == "cfg5":
    g = synthetic.hub_bipartite(50_000_000, 5000, 10_000, device="cuda")
else:
    g = synthetic.rmat(20, 5_000_000, device="cuda")
start = rw.start_vertices(g)
print(f"{label}: {cfg} trim {TRIM}: {g.n_edges} edges, max degree {int(g.degrees().max())}", flush=True)
b = min(int(os.environ.get("BATCH", 1 << 20)), start.numel())
nb = max(1, start.numel() // b)
walks = torch.empty((b * 10, 81), dtype=torch.int32, device="cuda")
valid = torch.empty(b * 10, dtype=torch.uint8, device="cuda")
ref = torch.empty_like(walks)
for pq in os.environ.get("PQ", "0.5,2.0").split(";"):
    P_, Q_ = (float(x) for x in pq.split(","))

    def run(k, ws, out=walks, slots=False):
        rw.walk(g, start[(k % nb) * b:(k % nb + 1) * b], 10, 80, P_, Q_, 42, out=(out, valid), check=False,
                use_workspace=ws, use_wedge_slots=slots)

    def timed(ws, reps=6, slots=False):
        run(0, ws, slots=slots); run(1, ws, slots=slots); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(2, 2 + reps): run(k, ws, slots=slots)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    dt = timed(False)
    if pq == os.environ.get("PQ", "0.5,2.0").split(";")[0]:
        deg = g.degrees()
        print(f"{label}: wedge_mode {g.wedge_mode}, slots {g.wedge_slots is not None}, share of edges into rows >= 65536: "
              f"{float(deg[deg >= 65536].sum()) / g.n_edges:.4f}", flush=True)
    print(f"{label}: {cfg} p={P_} q={Q_} batch {b} one-launch {b * 800 / dt / 1e9:.2f} G steps/s ({dt * 1e3:.2f} ms)", flush=True)
    run(3, False, ref); torch.cuda.synchronize()
    if g.wedge_slots is not None:
        dt = timed(False, slots=True)
        run(3, False, slots=True); torch.cuda.synchronize()
        print(f"{label}: {cfg} p={P_} q={Q_} batch {b} one-launch + slots {b * 800 / dt / 1e9:.2f} G steps/s "
              f"({dt * 1e3:.2f} ms) identical={bool(torch.equal(walks, ref))}", flush=True)
    for rounds in [x for x in os.environ.get("ROUNDS", "4").split(";") if x]:
        os.environ["N2V_WEDGE2_ROUNDS"] = rounds
        dt = timed(True)
        run(3, True); torch.cuda.synchronize()
        same = bool(torch.equal(walks, ref))
        print(f"{label}: {cfg} p={P_} q={Q_} batch {b} passes (rounds {rounds}) {b * 800 / dt / 1e9:.2f} G steps/s "
              f"({dt * 1e3:.2f} ms) identical={same}", flush=True)
        if not same:
            bad = torch.nonzero((walks != ref).any(1)).reshape(-1)
            print("  rows differing:", bad.numel(), bad[:5].tolist(), flush=True)
