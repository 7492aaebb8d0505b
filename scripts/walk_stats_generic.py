import ctypes as C, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from node2vec_amd import _lib
_lib.LIB_PATH = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "build_stats/libn2v_stats.so")  # scripts/build_stats.sh
from node2vec_amd import synthetic, randomwalk as rw
from node2vec_amd.graph import DeviceGraph
base = synthetic.rmat(20, 5_000_000, device="cuda")
gen = torch.Generator(device="cuda").manual_seed(1)
if os.environ.get("WEIGHTS", "integer") == "integer":
    w = torch.randint(1, 6, (base.n_edges,), generator=gen, device="cuda").float()
else:
    w = torch.rand(base.n_edges, generator=gen, device="cuda") * 1.9 + 0.1
g = DeviceGraph(base.rowptr, base.col, w)
start = rw.start_vertices(g)[:47104].contiguous()
L = _lib.load()
names = {0: "steps", 1: "filter_steps", 2: "direct_steps", 3: "maybes", 5: "past_quick_exit", 6: "pair_invocations", 7: "pair_inv_n<=64", 8: "pair_iterations", 9: "overfull_consumed", 10: "pair_uncached_inv"}
for p, q in ((0.5, 2.0), (4.0, 0.25)):
    buf = (C.c_ulonglong * 32)()
    L.n2v_debug_stats(buf, 1)
    walks, valid = rw.walk(g, start, 10, 80, p, q, 42); torch.cuda.synchronize()
    L.n2v_debug_stats(buf, 1)
    b = list(buf); steps = max(b[0], 1)
    print(p, q, {v: (b[k], round(b[k] / steps, 3)) for k, v in names.items()})
    ph = ["P0 filter", "P1 stream", "P2 verify", "sum+avg", "minmax", "pair cached", "pair uncached"]
    print("  wave-cycles/step", round(b[23] / steps), {n: (round(c / steps), f"{100*c/max(b[23],1):.1f}%") for n, c in zip(ph, b[16:23])})
