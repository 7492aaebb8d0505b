"""random-gather rates by element width and table size (n2v_mem_probe modes 0 / 1)"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from node2vec_amd import _lib
L = _lib.load()
buf = torch.zeros((12 << 30) // 4, dtype=torch.int32, device="cuda")
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
for gb, width in ((12, 16), (6, 8), (12, 8), (3, 4), (6, 16)):
    for mode in (0, 1):
        n = C.c_int64(0)
        best = 1e9
        for rep in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _lib.check(L.n2v_mem_probe(buf.data_ptr(), gb << 30, mode, 256, width, C.byref(n), sink.data_ptr(), _lib.current_stream_ptr()), "probe")
            b.record(); torch.cuda.synchronize()
            if rep: best = min(best, 1e-3 * a.elapsed_time(b))
        print(f"{width:2d}-byte gathers over {gb:2d} GB, {'chain' if mode else 'independent'}: {n.value / best / 1e9:6.2f} G/s", flush=True)
