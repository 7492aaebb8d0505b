import ctypes as C, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
os.environ["N2V_HIP_LIB"] = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "build_variants/libn2v_stats.so")
from node2vec_amd import synthetic, randomwalk as rw, _lib
if os.environ.get("GRAPH") == "cfg3":  # Chung-Lu 10 M, trimmed at 10 000 (scripts/scale_check.py)
    from node2vec_amd.fugue import trim_hotspot_edges
    from node2vec_amd.graph import DeviceGraph
    g = synthetic.chung_lu(10_000_000, 100_000_000, device="cuda")
    src = torch.repeat_interleave(torch.arange(g.n_vertices, device="cuda"), g.degrees())
    keep = trim_hotspot_edges(src, 10_000, 42)
    g = DeviceGraph.from_edges(src[keep], g.col[keep].long(), g.w[keep], n_vertices=g.n_vertices, device="cuda")
    sv = rw.start_vertices(g)
    start = sv[torch.randperm(sv.numel(), device="cuda")[:47104]].contiguous()
else:
    g = synthetic.rmat(20, 5_000_000, device="cuda")
    start = rw.start_vertices(g)[:47104].contiguous()
L = _lib.load()
names = ["draw_steps", "staged+filter", "direct_search", "maybes", "verify_rounds", "merge_steps",
         "pair_invocations", "engine_chunk_loads_uncached", "absorbed_slots", "cascades", "engine_chunk_loads", "staged_nofilter", "big_filter", "reverse"]
PQ = tuple(float(x) for x in os.environ.get("PQ", "0.5,2.0").split(","))
for p, q in (PQ,):
    buf = (C.c_ulonglong * 32)()
    L.n2v_debug_stats_unit(buf, 1)
    walks, valid = rw.walk(g, start, 10, 80, p, q, 42); torch.cuda.synchronize()
    L.n2v_debug_stats_unit(buf, 1)
    st = dict(zip(names, list(buf)))
    print(p, q, {k: (v, round(v / max(st["draw_steps"], 1), 4)) for k, v in st.items()})
    ph = ["P0 stage/filter", "P1 staged/filter", "P2 verify", "sum+avg", "P1 direct", "pairing", "P1 merge", "reverse classify"]
    cyc = list(buf)[16:23] + [list(buf)[24]]; tot = list(buf)[23]
    print("  wave-cycles total", tot, "per step", round(tot/st["draw_steps"]));    print("  cycles/step by phase:", {n: (round(c / st["draw_steps"]), f"{100*c/tot:.1f}%") for n, c in zip(ph, cyc)})
    b = list(buf)
    tot_draw = sum(b[25:30])
    print("  draw cycles by deg(v) bucket [<=64, <=1024, <=4096, <=8192, >8192]:", [f"{100*x/tot_draw:.1f}%" for x in b[25:30]],
          " steps 4096<n<=8192:", b[14], " n>8192:", b[15])
    print("  walkers: mean wave-cycles", round(b[31] / (walks.shape[0])), " slowest", b[30], " kernel wave-cycles per wave", round(tot / 8192))
