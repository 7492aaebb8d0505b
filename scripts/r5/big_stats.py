"""diagnostic (build_variants/libn2v_wedge_big20k.so / big65k.so): pairings on rows of >= N slots and the cycles
they take (status[2], status[3] x 256), cfg 3 trimmed at 100 000"""
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from node2vec_amd import _lib
_lib.LIB_PATH = os.environ["N2V_VARIANT_LIB"]
from node2vec_amd import synthetic, randomwalk as rw
g = synthetic.chung_lu(10_000_000, 100_000_000, device="cuda").trimmed(100_000, 42)
start = rw.start_vertices(g)[:1 << 20].contiguous()
for p, q in ((0.5, 2.0), (4.0, 0.25), (3.0, 0.7)):
    st = {}
    walks, valid = rw.walk(g, start, 10, 80, p, q, 42, stats=st)
    torch.cuda.synchronize()
    steps = int(valid.sum()) * 80
    s = st["status"].cpu().numpy().astype("uint32")
    print(f"{os.path.basename(_lib.LIB_PATH)} p={p} q={q}: steps {steps}; pairings on big rows {int(s[2])} "
          f"({s[2] / steps:.2e} of the steps), {int(s[3]) * 256 / max(int(s[2]), 1):.0f} cycles each, "
          f"{int(s[3]) * 256 / 2.4e9 * 1e3:.1f} ms of lane time in all", flush=True)
