"""pairs/s of the default and the batched SGNS kernel on the walks of a BASELINE graph
(cfg3 by default: 10 M vertices, model 2 x 5 GB -- far beyond L2 / Infinity Cache)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from node2vec_amd import randomwalk as rw  # noqa: E402
from node2vec_amd import sgns, synthetic  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
if which == "cfg2":
    g = synthetic.rmat(20, 5_000_000, seed=42, device="cuda")
    nv = 47_104
else:
    n = 10_000_000 if which == "cfg3" else 100_000_000
    g = synthetic.chung_lu(n, 10 * n if which == "cfg3" else 5 * n, seed=42, device="cuda").trimmed(10_000, 42)
    nv = 1 << 16
start = rw.start_vertices(g)[:nv]
walks, valid = rw.walk(g, start, 10, 80, 1.0, 1.0, 42)
walks = walks[valid]
deg = g.degrees().clamp(min=1)
order = torch.sort(deg, descending=True, stable=True).indices
index_of = torch.empty(g.n_vertices, dtype=torch.int32, device="cuda")
index_of[order] = torch.arange(g.n_vertices, dtype=torch.int32, device="cuda")
FOLD = int(os.environ.get("FOLD", "0"))  # experiment: fold the vocabulary to FOLD rows (cache-resident model)
if FOLD:
    order, index_of = order[:FOLD], (index_of % FOLD).to(torch.int32)
vocab = sgns.Vocab(order, deg[order], index_of)
idx = index_of[walks.long()].contiguous()
g.hops = None
torch.cuda.empty_cache()
print(f"{which}: fold={FOLD} {g.n_vertices} vertices, corpus {tuple(idx.shape)}, dim {dim}", flush=True)
HUB = int(os.environ.get('HUB_ROWS', '0'))
for batched, cache in ((False, 0), (True, 0)):
    m = sgns.SgnsModel(vocab, dim, 5, 5, seed=1, sample=0.0)
    m.batched = batched
    m.window_cache = cache
    m.hub_rows = HUB
    for k in range(2):
        m.train_block(idx, 0.025, k * idx.shape[0])
    m.pairs.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 5
    for k in range(steps):
        m.train_block(idx, 0.025, (2 + k) * idx.shape[0])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pairs = int(m.pairs.item())
    print(f"batched={batched} window_cache={cache} hub_rows={HUB}: {pairs / dt / 1e6:9.1f} M pairs/s  {1e3 * dt / steps:8.2f} ms/launch  "
          f"pairs/launch {pairs // steps}", flush=True)
    del m
    torch.cuda.empty_cache()
