"""would an 8-byte hop entry + a cached second lookup beat the 16-byte entry?  steps/s of a chain
{8-byte gather over 6 GB} + {16-byte gather over a small table for X % of the steps}"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from node2vec_amd import _lib
L = _lib.load()
L.n2v_probe_hop8_experiment.restype = C.c_int
L.n2v_probe_hop8_experiment.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.c_void_p, C.c_void_p]
BIG = int(float(os.environ.get("BIG_GB", "6")) * (1 << 30)) // 8 * 8
big = torch.zeros(BIG // 4, dtype=torch.int32, device="cuda")
small = torch.zeros((64 << 20) // 4, dtype=torch.int32, device="cuda")
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
for small_mb in (1, 2, 4):
    for share in (0, 40, 53, 60):
        n = C.c_int64(0); best = 1e9
        for rep in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            L.n2v_probe_hop8_experiment(big.data_ptr(), BIG, small.data_ptr(), small_mb << 20, 256, share, C.byref(n), sink.data_ptr(), _lib.current_stream_ptr())
            b.record(); torch.cuda.synchronize()
            if rep: best = min(best, 1e-3 * a.elapsed_time(b))
        print(f"small table {small_mb:3d} MB, {share:3d} % of the steps look it up: {n.value / best / 1e9:6.2f} G steps/s", flush=True)
