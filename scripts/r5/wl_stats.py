"""diagnostic (build_variants/libn2v_wlanes_stats.so): per step launch of the weighted lane kernel, the longest
sum pass and the longest pairing of any lane (ticks of the cycle counter) and the row it stood on"""
import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from node2vec_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "build_variants", os.environ.get("WL_LIB", "libn2v_wlanes_stats.so"))
from node2vec_amd import synthetic, randomwalk as rw
from node2vec_amd.graph import DeviceGraph
base = synthetic.rmat(20, 5_000_000, device="cuda")
gen = torch.Generator(device="cuda").manual_seed(1)
g = DeviceGraph(base.rowptr, base.col, torch.rand(base.n_edges, generator=gen, device="cuda") * 1.9 + 0.1)
print("max degree", int(g.degrees().max()), flush=True)
start = rw.start_vertices(g)[:47104].contiguous()
for L in (2, 6):
    st = {}
    torch.cuda.synchronize(); t = time.time()
    walks, valid = rw.walk(g, start, 10, L, 0.5, 2.0, 42, use_weighted_lanes=True, stats=st)
    torch.cuda.synchronize(); dt = time.time() - t
    s = st["status"].cpu().numpy().astype("uint32")
    print(f"walk length {L}: {dt * 1e3:.1f} ms; longest sum pass {int(s[2]) * 256} ticks, longest pairing {int(s[3]) * 256} ticks on a row of {int(s[1])} slots", flush=True)
