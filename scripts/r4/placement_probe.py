"""does n2v_mem_probe (random 4-byte reads of ONE buffer) see what the walk kernel sees?  The ranked table at several
places (as built + clones); per place: the walk kernel's time (same output buffer) and the probe's rate on it."""
import ctypes as C, os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import bench
from node2vec_amd import synthetic, randomwalk as rw, _lib
L = _lib.load()
g = synthetic.chung_lu(100_000_000, 500_000_000, device="cuda").trimmed(10_000, 42)
start = rw.start_vertices(g)
g.build_ranked()
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
def probe(buf, mode=1, arg=4):
    n = C.c_int64(0); best = None
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(L.n2v_mem_probe(buf.data_ptr(), buf.numel() * buf.element_size(), mode, 256, arg, C.byref(n),
                                   sink.data_ptr(), _lib.current_stream_ptr()), "probe")
        b.record(); torch.cuda.synchronize()
        dt = 1e-3 * a.elapsed_time(b)
        if rep: best = dt if best is None else min(best, dt)
    return n.value / best / 1e9
def t(leg):
    for k in range(2): leg.step(k)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for k in range(2, 8): leg.step(k)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 6
leg = bench.WalkLeg(torch, rw, g, start, 10, 80, 1.0, 1.0, "exact", 1 << 20, 0, 1, rank_ids=True)
old = g.rank_hops
keep = [old]
for rep in range(6):
    tab = keep[-1] if rep == 0 else old.clone()
    keep.append(tab)
    g.rank_hops = tab
    print(f"table at {tab.data_ptr():#x}: walk {t(leg):.2f} ms; probe 4-byte chain {probe(tab):.1f} G/s, with class search {probe(tab, 4, 8192):.1f} G/s", flush=True)
outs = [leg.walks] + [torch.empty_like(leg.walks) for _ in range(4)]
g.rank_hops = old
for o in outs:
    leg.walks = o
    print(f"output at {o.data_ptr():#x}: walk {t(leg):.2f} ms; probe on it {probe(o):.1f} G/s", flush=True)
