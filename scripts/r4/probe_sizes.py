"""Is the random-gather rate a function of the entry WIDTH or of the table SIZE (TLB reach)?  n2v_mem_probe
mode 1 (one dependent chain per lane) for 4- and 16-byte entries over buffers of 0.75 .. 24 GB; then the same
3 GB buffer allocated after 40 GB of other allocations were made and freed in pieces (fragmented pool)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from node2vec_amd import _lib
L = _lib.load()
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
first = torch.zeros(3 * (1 << 30) // 8, dtype=torch.int64, device="cuda")
print(f"3 GB allocated first in the process: 4-byte chain {run(first, 1, 256, 4):.1f} G/s, 16-byte chain {run(first, 1, 256, 16):.1f} G/s", flush=True)
for gb in (0.75, 1.5, 3, 6, 12, 24, 48):
    buf = torch.zeros(int(gb * (1 << 30)) // 8, dtype=torch.int64, device="cuda")
    print(f"{gb:5.2f} GB: 4-byte chain {run(buf, 1, 256, 4):.1f} G/s   8-byte {run(buf, 1, 256, 8):.1f}   16-byte chain {run(buf, 1, 256, 16):.1f} G/s   "
          f"16-byte independent {run(buf, 0, 256, 16):.1f} G/s", flush=True)
    del buf; torch.cuda.empty_cache()
# fragment the pool: many 64 MB blocks, free every other one, then allocate 3 GB
blocks = [torch.empty(64 << 20, dtype=torch.uint8, device="cuda") for _ in range(640)]
del blocks[::2]
late = torch.zeros(3 * (1 << 30) // 8, dtype=torch.int64, device="cuda")
print(f"3 GB allocated with 20 GB of 64 MB blocks live and 20 GB freed around them: 4-byte chain {run(late, 1, 256, 4):.1f} G/s, "
      f"16-byte {run(late, 1, 256, 16):.1f} G/s", flush=True)
print(f"the first 3 GB again: 4-byte chain {run(first, 1, 256, 4):.1f} G/s", flush=True)
