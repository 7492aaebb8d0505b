"""Round 3, VERDICT item 1b/1c: the distribution of the statistics tests/test_sgns_parity_gpu.py
asserts on the R-MAT scale-20 corpus (BASELINE cfg 2), over >= 20 hogwild runs, and as a function
of the hogwild concurrency (waves training at once).  Prints one line per run; the tolerances of
the test are derived from this log (profiles/r3*_hogwild_auc_runs.log)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from node2vec_amd import randomwalk as rw  # noqa: E402
from node2vec_amd import sgns, synthetic  # noqa: E402

n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
BATCHED = len(sys.argv) > 2 and sys.argv[2] == "batched"
WAVES = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 2048, 512, 64]
g = synthetic.rmat(20, 5_000_000, device="cuda")
start = rw.start_vertices(g)
walks, valid = rw.walk(g, start, 10, 80, 0.5, 2.0, 42)
vocab = sgns.build_vocab(walks, 10)
idx = vocab.index_of[walks.long()]
del walks
index_of = vocab.index_of
deg = g.degrees()
gen = torch.Generator(device="cuda").manual_seed(1)
e = torch.randint(0, g.n_edges, (200_000,), generator=gen, device="cuda")
src = torch.repeat_interleave(torch.arange(g.n_vertices, device="cuda"), deg)
pa, pb = index_of[src[e]].long(), index_of[g.col[e].long()].long()
del src
ok = (pa >= 0) & (pb >= 0)
pa, pb = pa[ok], pb[ok]
na = torch.randint(0, len(vocab), (200_000,), generator=gen, device="cuda")
nb = torch.randint(0, len(vocab), (200_000,), generator=gen, device="cuda")
cand = torch.nonzero(deg[vocab.ids] >= 20).reshape(-1)
probe = cand[torch.randperm(cand.numel(), generator=gen, device="cuda")[:2000]]
print(f"corpus: {idx.shape[0]} rows, vocabulary {len(vocab)}", flush=True)


def run(seed, max_waves=0):
    m = sgns.SgnsModel(vocab, 128, 5, 5, seed=seed, sample=1e-3)
    m.max_waves = max_waves
    m.batched = BATCHED
    hr = os.environ.get('HUB_ROWS', '0')
    m.hub_rows = None if hr == 'auto' else int(hr)  # auto: SgnsModel.auto_hub_rows (the default)
    t0 = time.perf_counter()
    m.train(idx, epochs=1, alpha=0.025, min_alpha=1e-4)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    u = m.syn0 - m.syn0.mean(0)
    u = u / (u.norm(dim=1, keepdim=True) + 1e-30)
    sp, sn = (u[pa] * u[pb]).sum(1), (u[na] * u[nb]).sum(1)
    k = min(sp.numel(), 20000)
    auc = float((sp[:k, None] > sn[None, :2000]).float().mean())
    s = u[probe] @ u.T
    s[torch.arange(probe.numel(), device="cuda"), probe] = -1e9
    nbrs = torch.topk(s, 10, dim=1).indices.cpu().numpy()
    # hub rows: norms of the 100 most frequent words (where concurrent writers collide)
    hub = float(m.syn0[:100].norm(dim=1).mean())
    hub1 = float(m.syn1neg[:100].norm(dim=1).mean())
    global LAST_HUB_ROWS
    LAST_HUB_ROWS = m.hub_rows
    return auc, nbrs, dt, hub, hub1, int(m.pairs.item())


def overlap(a, b):
    return float(np.mean([len(set(x) & set(y)) / len(x) for x, y in zip(a, b)]))


print(f'batched={BATCHED} hub_rows={os.environ.get("HUB_ROWS", "0")}', flush=True)
for waves in WAVES:
    aucs, first = [], None
    n = n_runs if waves == 0 else 5
    for seed in range(n):
        auc, nbrs, dt, hub, hub1, pairs = run(seed, waves)
        first = nbrs if first is None else first
        ov = overlap(first, nbrs) if seed else 1.0
        aucs.append(auc)
        print(f"max_waves={waves} seed={seed} auc={auc:.4f} knn10_overlap_vs_seed0={ov:.3f} "
              f"train_s={dt:.2f} pairs={pairs} hub_norm_syn0={hub:.3f} hub_norm_syn1neg={hub1:.3f} "
              f"hub_rows={LAST_HUB_ROWS}",
              flush=True)
    a = np.array(aucs)
    print(f"max_waves={waves}: n={len(a)} mean={a.mean():.4f} sd={a.std(ddof=1):.4f} "
          f"min={a.min():.4f} max={a.max():.4f} spread={a.max() - a.min():.4f}", flush=True)
