"""go / no-go of 8-byte {rank, classes} hop entries for the biased kernels: n2v_mem_probe
mode 5 -- a dependent chain of hop entries + the wedge slot of 49 % of the steps -- with 16-byte entries (today)
and 8-byte entries over tables of cfg 4's size (E = 7.56e8 edges), and the hop chain alone (mode 1)."""
import ctypes as C, os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from node2vec_amd import _lib
L = _lib.load()
E = int(os.environ.get("EDGES", 756_466_949))
buf = torch.empty(E * 48 // 4 + 1024, dtype=torch.int32, device="cuda"); buf.random_()
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
def run(mode, iters, rb, nbytes):
    n = C.c_int64(0); best = None
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(L.n2v_mem_probe(buf.data_ptr(), nbytes, mode, iters, rb, C.byref(n), sink.data_ptr(), _lib.current_stream_ptr()), "probe")
        b.record(); torch.cuda.synchronize()
        dt = 1e-3 * a.elapsed_time(b)
        if rep: best = dt if best is None else min(best, dt)
    return n.value / best / 1e9
for w in (16, 8):
    print(f"PROBE biased-step shape, {w}-byte hop entries ({E * w / 1e9:.1f} GB) + 32-byte slots ({E * 32 / 1e9:.1f} GB) at 49 %: "
          f"{run(5, 256, w, E * (w + 32)):.1f} G steps/s;  the hop chain alone: {run(1, 256, w, E * w):.1f} G/s", flush=True)
