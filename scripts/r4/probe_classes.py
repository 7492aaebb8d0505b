"""Go / no-go for 4-byte hop entries on a graph numbered by descending degree:
n2v_mem_probe mode 1 at 16 / 8 / 4 bytes against mode 4 (4-byte chain + LDS class search) on tables of
the cfg 4 sizes (7.56e8 entries).  Run on the GPU box."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from node2vec_amd import _lib
L = _lib.load()
E = 756_000_000
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
def run(buf, mode, iters, arg):
    n = C.c_int64(0); best = None
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(L.n2v_mem_probe(buf.data_ptr(), buf.numel() * buf.element_size(), mode, iters, arg, C.byref(n),
                                   sink.data_ptr(), _lib.current_stream_ptr()), "probe")
        b.record(); torch.cuda.synchronize()
        dt = 1e-3 * a.elapsed_time(b)
        if rep: best = dt if best is None else min(best, dt)
    return n.value / best / 1e9
for width in (16, 8, 4):
    buf = torch.zeros(E * width // 8, dtype=torch.int64, device="cuda")
    print(f"chain of {width}-byte gathers over {buf.numel()*8/1e9:.1f} GB: {run(buf, 1, 256, width):.1f} G/s", flush=True)
    if width == 4:
        for classes in (256, 1024, 4096, 8192):
            print(f"  + class search over {classes} classes in LDS: {run(buf, 4, 256, classes):.1f} G/s", flush=True)
    del buf
