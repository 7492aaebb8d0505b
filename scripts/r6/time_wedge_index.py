"""MEASURED AND NOT KEPT (round 6; the tree it ran on is scripts/r6/index_park_experiment.patch: `git apply` it first --
the flags use_wedge_index / use_park exist only there).  Exact biased walks with the long lists searched through a
sample index of wedge_pos (every 32nd / 1024th entry, aligned to the array), and the slots kernel with the lanes whose
next step is slow set aside and stepped in groups -- against the kernels of the tree, timed on one graph for several
(p, q): profiles/r11a_*, r11b_*, r11e_*.
  GRAPH=cfg4|cfg3 TRIM=100000 PQ="0.5,2;3,0.7;4,0.25" BATCH=1048576 python scripts/r6/time_wedge_index.py <label>"""
import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from node2vec_amd import synthetic, randomwalk as rw
label = sys.argv[1] if len(sys.argv) > 1 else ""
cfg = os.environ.get("GRAPH", "cfg4")
TRIM = int(os.environ.get("TRIM", 100_000))  # the reference's default cap (constants.py:6)
if cfg == "cfg4":
    g = synthetic.chung_lu(100_000_000, 500_000_000, device="cuda").trimmed(TRIM, 42)
else:
    g = synthetic.chung_lu(10_000_000, 100_000_000, device="cuda").trimmed(TRIM, 42)
start = rw.start_vertices(g)
print(f"{label}: {cfg} trim {TRIM}: {g.n_edges} edges, max degree {int(g.degrees().max())}", flush=True)
b = min(int(os.environ.get("BATCH", 1 << 20)), start.numel())
nb = max(1, start.numel() // b)
walks = torch.empty((b * 10, 81), dtype=torch.int32, device="cuda")
valid = torch.empty(b * 10, dtype=torch.uint8, device="cuda")
ref = torch.empty_like(walks)
first = True
for pq in os.environ.get("PQ", "0.5,2.0;3,0.7;4,0.25").split(";"):
    P_, Q_ = (float(x) for x in pq.split(","))

    def run(k, out=walks, **kw):
        rw.walk(g, start[(k % nb) * b:(k % nb + 1) * b], 10, 80, P_, Q_, 42, out=(out, valid), check=False, **kw)

    def timed(reps=5, **kw):
        run(0, **kw); run(1, **kw); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(2, 2 + reps): run(k, **kw)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    dt0 = timed(use_wedge_index=False)
    if first:
        first = False
        cnt = (g.edge_classes & 0xffffff)
        deg = g.degrees()
        print(f"{label}: wedge_mode {g.wedge_mode}, slots {g.wedge_slots is not None}, index {g.wedge_index is not None} "
              f"({0 if g.wedge_index is None else g.wedge_index[0].numel() / 1e9:.2f} GB; wedge_pos "
              f"{g.wedge_pos.numel() * g.wedge_pos.element_size() / 1e9:.1f} GB), longest list {int(cnt.max())}, lists > 64: "
              f"{int((cnt > 64).sum())}, > 2048: {int((cnt > 2048).sum())}; share of edges into rows >= 65536: "
              f"{float(deg[deg >= 65536].sum()) / g.n_edges:.4f}, >= 10000: {float(deg[deg >= 10000].sum()) / g.n_edges:.4f}",
              flush=True)
    run(3, ref, use_wedge_index=False); torch.cuda.synchronize()
    dt1 = timed()
    run(3); torch.cuda.synchronize()
    print(f"{label}: {cfg} trim {TRIM} p={P_} q={Q_} batch {b}: lists searched {b * 800 / dt0 / 1e9:.2f} G steps/s "
          f"({dt0 * 1e3:.2f} ms), through the index {b * 800 / dt1 / 1e9:.2f} G ({dt1 * 1e3:.2f} ms) "
          f"identical={bool(torch.equal(walks, ref))}", flush=True)
    if os.environ.get("PARK"):
        dt4 = timed(use_park=False)
        run(3, use_park=False); torch.cuda.synchronize()
        print(f"{label}:   lock step (no lane set aside): {b * 800 / dt4 / 1e9:.2f} G ({dt4 * 1e3:.2f} ms) "
              f"identical={bool(torch.equal(walks, ref))}", flush=True)
    if os.environ.get("NOSLOTS"):
        dt2 = timed(use_wedge_slots=False, use_wedge_index=False)
        dt3 = timed(use_wedge_slots=False)
        run(3, use_wedge_slots=False); torch.cuda.synchronize()
        print(f"{label}:   without the slots: lists searched {b * 800 / dt2 / 1e9:.2f} G, through the index "
              f"{b * 800 / dt3 / 1e9:.2f} G identical={bool(torch.equal(walks, ref))}", flush=True)
