"""diagnostic (build_variants/libn2v_wedge_declines.so, -DN2V_DECLINE_STATS=1): why the closed forms of the pairing
loop decline on rows of more than 64 slots (codes: n2v_unit_core.h, N2V_DECLINE), cfg 4 trimmed at TRIM"""
import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
os.environ["N2V_DIAG_STATUS_WORDS"] = "512"
from node2vec_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "build_variants", "libn2v_wedge_declines.so")
from node2vec_amd import synthetic, randomwalk as rw
TRIM = int(os.environ.get("TRIM", 100_000))  # (cycle counts of this build are NOT usable: its counters are contended atomics)
g = synthetic.chung_lu(100_000_000, 500_000_000, device="cuda").trimmed(TRIM, 42)
start = rw.start_vertices(g)[:1 << 18].contiguous()
for pq in os.environ.get("PQ", "0.5,2;4,0.25").split(";"):
    p, q = (float(x) for x in pq.split(","))
    st = {}
    walks, valid = rw.walk(g, start, 10, 80, p, q, 42, stats=st)
    torch.cuda.synchronize()
    steps = int(valid.sum()) * 80
    s = st["status"].cpu().numpy().astype("uint32")[8:]
    rep = st["status"].cpu().numpy().astype("uint32")[8 + 64:8 + 74]
    print(f"trim {TRIM} p={p} q={q} replays on rows > 64 by list length (<= 64, <= 256, <= 1024, <= 4096, more): "
          + ", ".join(f"{int(rep[2 * b + 1])} x {256.0 * rep[2 * b] / max(int(rep[2 * b + 1]), 1) / 1e3:.0f} k cycles"
                      for b in range(5))
          + f"; in all {256.0 * float(rep[0::2].sum()) / 2.4e9 * 1e3:.0f} ms of lane time at 2.4 GHz", flush=True)
    ex = st["status"].cpu().numpy().astype("uint32")[8 + 80:8 + 83]
    print(f"   lane_case_b: one-by-one counts of other_pos {int(ex[0])}, loop iterations {int(ex[1])}, corrections of the "
          f"cascade length {int(ex[2])}", flush=True)
    allw = st["status"].cpu().numpy().astype("uint32")[8:]
    print("   declines by reason, rows of any length: " + str({c: int(allw[256 + c]) for c in range(1, 100) if allw[256 + c]})
          + "; rows > 64, codes 40+: " + str({c: int(allw[96 + c]) for c in range(32, 100) if allw[96 + c]}), flush=True)
    for name, o in (("rows > 64", 0), ("rows >= 4096", 32)):
        codes = {c: int(s[o + c]) for c in range(1, 24) if s[o + c]}
        arrs = {a: int(s[o + 24 + a]) for a in range(6) if s[o + 24 + a]}
        print(f"trim {TRIM} p={p} q={q} {name}: steps {steps}, pairings {int(s[o + 30])}, declined by arrangement {arrs}, "
              f"by reason {codes}", flush=True)
