"""n2v_mem_probe over a 16 GB buffer: known byte counts for calibrating FETCH_SIZE / WRITE_SIZE on
the access shape of K3 (one wave reads / rewrites one 512-byte row, 8 bytes per lane).  Run under
rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (scripts/r3/pmc_calibrate.sh)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from node2vec_amd import _lib  # noqa: E402

L = _lib.load()
buf = torch.zeros((16 << 30) // 4, dtype=torch.float32, device="cuda")
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
for mode, row_bytes, iters in ((0, 0, 256), (2, 512, 512), (3, 512, 512), (2, 1024, 512), (3, 1024, 512)):
    n = C.c_int64(0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    _lib.check(L.n2v_mem_probe(buf.data_ptr(), buf.numel() * 4, mode, iters, row_bytes, C.byref(n),
                               sink.data_ptr(), _lib.current_stream_ptr()), "n2v_mem_probe")
    b.record()
    torch.cuda.synchronize()
    unit = 16 if mode < 2 else row_bytes
    print(f"mode {mode} row_bytes {row_bytes}: {n.value} accesses = {n.value * unit} bytes read"
          f"{' and written' if mode == 3 else ''}; {1e-3 * a.elapsed_time(b):.4f} s "
          f"-> {n.value * unit * (2 if mode == 3 else 1) / (1e-3 * a.elapsed_time(b)) / 1e9:.1f} GB/s", flush=True)
