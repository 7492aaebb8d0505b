"""diagnostic (build_variants/libn2v_wedge_nearcount.so): of the steps of an exact walk with values that are not dyadic,
how many pass the quick accept, how many reach the closed forms with margins, how many of those are declined"""
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from node2vec_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "build_variants", "libn2v_wedge_nearcount.so")
from node2vec_amd import synthetic, randomwalk as rw
g = synthetic.chung_lu(100_000_000, 500_000_000, device="cuda").trimmed(10_000, 42)
start = rw.start_vertices(g)[:1 << 18].contiguous()
for p, q in ((0.5, 2.0), (4.0, 0.25), (4.0, 2.0), (0.25, 0.5), (0.7, 3.0), (1.3, 1.3), (3.0, 0.7), (0.3, 0.7)):
    st = {}
    walks, valid = rw.walk(g, start, 10, 80, p, q, 42, stats=st)
    torch.cuda.synchronize()
    steps = int(valid.sum()) * 80
    s = st["status"].cpu().numpy().astype("uint32")
    what = "pairings" if all((1.0 / x) == 2.0 ** round(__import__("math").log2(1.0 / x)) for x in (p, q)) else "past the quick accept"
    print(f"p={p} q={q}: steps {steps}; {what} {int(s[2])} ({s[2] / steps:.3f}); "
          f"declined by the closed forms {int(s[3])} ({s[3] / max(int(s[2]), 1):.4f} of those)", flush=True)
