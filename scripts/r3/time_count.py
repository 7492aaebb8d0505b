"""how fast can a batch of walk tokens be counted?  64-bit atomics (n2v_corpus_count), 32-bit
atomics, torch.bincount, sort + unique -- on cfg-4-like token streams (10^8 vertices)."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from node2vec_amd import _lib, sgns  # noqa: E402
from node2vec_amd import randomwalk as rw  # noqa: E402
from node2vec_amd import synthetic  # noqa: E402

n = int(os.environ.get("VERTICES", 100_000_000))
g = synthetic.chung_lu(n, 5 * n, seed=42, device="cuda").trimmed(10_000, 42)
start = rw.start_vertices(g)[: 1 << 20]
walks, valid = rw.walk(g, start, 10, 80, 1.0, 1.0, 42)
L = _lib.load()
tok = walks.numel()


def t(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


c64 = torch.zeros(g.n_vertices, dtype=torch.int64, device="cuda")
c32 = torch.zeros(g.n_vertices, dtype=torch.int32, device="cuda")
a = t(lambda: sgns.corpus_count(walks, valid, c64))
print(f"64-bit atomics: {a * 1e3:.1f} ms for {tok} tokens = {tok / a / 1e9:.2f} G/s", flush=True)


b = t(lambda: sgns.corpus_count(walks, valid, c64, sort_above=1))
print(f"sort + run-length + add: {b * 1e3:.1f} ms = {tok / b / 1e9:.2f} G/s", flush=True)
d = t(lambda: torch.sort(walks.reshape(-1)), reps=1)
print(f"torch.sort of the tokens: {d * 1e3:.1f} ms", flush=True)
k = t(lambda: rw.walk(g, start, 10, 80, 1.0, 1.0, 42))
print(f"the walk itself (rw.walk incl. host side): {k * 1e3:.1f} ms", flush=True)
