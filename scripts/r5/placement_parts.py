"""Is the placement effect a property of EACH buffer or of the pair?  In one process, three copies of the
ranked table x three output buffers; for every pair the full kernel, the kernel without path stores
(build_variants/libn2v_uniform_nostore.so) and the kernel without table reads (..._noread.so): 27 timings
on the same memory.  Then the AUDITION: can a short launch (2^17 start vertices) tell the fast copy from
the slow one?"""
import ctypes as C, os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from node2vec_amd import _lib, synthetic, randomwalk as rw
g = synthetic.chung_lu(100_000_000, 500_000_000, device="cuda").trimmed(10_000, 42)
start = rw.start_vertices(g)
g.build_ranked()
L0 = _lib.load()
libs = {"full": L0}
for name in ("nostore", "noread"):
    path = os.path.join(ROOT, "build_variants", f"libn2v_uniform_{name}.so")
    if os.path.exists(path):
        lib = C.CDLL(path)
        lib.n2v_walk_ws.restype = C.c_int
        lib.n2v_walk_ws.argtypes = L0.n2v_walk_ws.argtypes
        libs[name] = lib
B, W, L = 1 << 20, 10, 80
valid = torch.empty(B * W, dtype=torch.uint8, device="cuda")
status = torch.zeros(4, dtype=torch.int32, device="cuda")

def launch(lib, table, out, k, b=B):
    g.rank_hops = table
    cs = g.c_struct(); cs.hops = 0; cs.hops8 = 0; cs.rank_emit = 1
    st = start[k * b:(k + 1) * b]
    rc = lib.n2v_walk_ws(cs, st.data_ptr(), st.numel(), W, L, 1.0, 1.0, 42, 0, out.data_ptr(), valid.data_ptr(),
                         status.data_ptr(), None, 0, _lib.current_stream_ptr())
    assert rc == 0, rc

def ms(lib, table, out, reps=3, b=B):
    launch(lib, table, out, 0, b)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for k in range(1, 1 + reps): launch(lib, table, out, k, b)
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps

t0 = g.rank_hops
tables = [t0, t0.clone(), t0.clone()]
outs = [torch.empty((B * W, L + 1), dtype=torch.int32, device="cuda") for _ in range(3)]
for name, lib in libs.items():
    for ti, t in enumerate(tables):
        print(f"PARTS {name:8s} table {ti}: " + "  ".join(f"out {oi}: {ms(lib, t, o):6.2f} ms" for oi, o in enumerate(outs)), flush=True)
# audition: a launch of 2^17 start vertices (1/8 of a batch) per pair
for ti, t in enumerate(tables):
    print(f"AUDITION (2^17 start vertices) table {ti}: " + "  ".join(f"out {oi}: {ms(L0, t, o, 5, 1 << 17):6.3f} ms" for oi, o in enumerate(outs)), flush=True)
# and once more the full kernel: are the figures stable over the life of the process?
for ti, t in enumerate(tables):
    print(f"AGAIN full     table {ti}: " + "  ".join(f"out {oi}: {ms(L0, t, o):6.2f} ms" for oi, o in enumerate(outs)), flush=True)
