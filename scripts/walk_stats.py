"""Where the exact unit-weight walk kernels spend their cycles (diagnostic build, -DN2V_STATS):
GRAPH=cfg2|cfg3 PQ=0.5,2.0 KERNEL=lanes|wave python scripts/walk_stats.py   (bash scripts/build_stats.sh first)"""
import ctypes as C, os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from node2vec_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "build_stats", "libn2v_stats.so")  # developer build, loaded by path
from node2vec_amd import synthetic, randomwalk as rw
if os.environ.get("GRAPH") == "cfg4":
    g = synthetic.chung_lu(100_000_000, 500_000_000, device="cuda").trimmed(10_000, 42)
    sv = rw.start_vertices(g)
    start = sv[torch.randperm(sv.numel(), device="cuda")[:131072]].contiguous()
elif os.environ.get("GRAPH") == "cfg3":
    g = synthetic.chung_lu(10_000_000, 100_000_000, device="cuda").trimmed(10_000, 42)
    sv = rw.start_vertices(g)
    start = sv[torch.randperm(sv.numel(), device="cuda")[:131072]].contiguous()
else:
    g = synthetic.rmat(20, 5_000_000, device="cuda")
    start = rw.start_vertices(g)[:131072].contiguous()
L = _lib.load()
L.n2v_debug_stats_unit.argtypes = [C.c_void_p, C.c_int]
names = ["draw_steps", "staged+filter", "direct_search", "maybes", "verify_rounds", "merge_steps",
         "pair_invocations", "engine_chunk_loads_uncached", "absorbed_slots", "cascades", "engine_chunk_loads",
         "staged_nofilter", "big_filter", "reverse"]
p, q = (float(x) for x in os.environ.get("PQ", "0.5,2.0").split(","))
lanes = os.environ.get("KERNEL", "lanes") == "lanes"
buf = (C.c_ulonglong * 40)()
rw.walk(g, start[:1024], 10, 80, p, q, 42, use_edge_classes=lanes); torch.cuda.synchronize()
L.n2v_debug_stats_unit(buf, 1)
walks, valid = rw.walk(g, start, 10, 80, p, q, 42, use_edge_classes=lanes); torch.cuda.synchronize()
L.n2v_debug_stats_unit(buf, 1)
b = list(buf)
st = dict(zip(names, b))
draws = max(st["draw_steps"], 1)
print("p,q", p, q, "kernel", "lanes" if lanes else "wave", {k: (v, round(v / draws, 3)) for k, v in st.items()})
ph = ["P0 stage/filter", "P1 staged/filter", "P2 verify", "sum+avg", "P1 direct", "pairing", "P1 merge", "reverse classify"]
cyc = b[16:23] + [b[24]]
tot = b[23]
print("  unit_draw cycles per draw by phase:", {n: round(c / draws) for n, c in zip(ph, cyc)}, " sum", round(sum(cyc) / draws))
if lanes:
    steps = max(b[28], 1)
    print(f"  lanes kernel: walker-steps {b[28]}, fallback draws {b[29]} ({100*b[29]/steps:.1f} %), wave-cycles total {tot}")
    print(f"  wave-cycles: quick phase {100*b[26]/tot:.1f} %, fallback phase {100*b[27]/tot:.1f} %; per fallback draw {round(b[27]/max(b[29],1))} cycles")
    bk = ["n<=64", "n<=1024", "n<=4096", "n>4096"]
    print("  fallback draws by deg(v):", {k: (b[32 + i], f"{100*b[36+i]/max(b[27],1):.1f}% of fallback cycles", round(b[36 + i] / max(b[32 + i], 1))) for i, k in enumerate(bk)})
else:
    print("  wave kernel: wave-cycles total", tot, "per step", round(tot / draws))
