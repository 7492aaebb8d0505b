"""SGNS launch time on the cfg 2 corpus block (run on the GPU box): python scripts/time_sgns.py [dim ...]"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from node2vec_amd import synthetic, randomwalk as rw, sgns
dims = [int(x) for x in sys.argv[1:]] or [128]
g = synthetic.rmat(20, 5_000_000, device="cuda")
s = rw.start_vertices(g)[:47104].contiguous()
walks, valid = rw.walk(g, s, 10, 80, 0.5, 2.0, 42)
deg = g.degrees().clamp(min=1)
order = torch.sort(deg, descending=True, stable=True).indices
index_of = torch.empty(g.n_vertices, dtype=torch.int32, device="cuda"); index_of[order] = torch.arange(g.n_vertices, dtype=torch.int32, device="cuda")
vocab = sgns.Vocab(order, deg[order], index_of)
idx = index_of[walks[valid.bool()].long()].contiguous()
for dim in dims:
    m = sgns.SgnsModel(vocab, dim, 5, 5, seed=1, sample=0.0)
    m.train_block(idx, 0.025, 0); torch.cuda.synchronize()
    best = 1e9
    for it in range(3):
        m.pairs.zero_(); torch.cuda.synchronize(); t = time.time()
        m.train_block(idx, 0.025, (it + 1) * idx.shape[0]); torch.cuda.synchronize(); best = min(best, time.time() - t)
    pairs = int(m.pairs.item())
    print(f"{os.environ.get('N2V_SGNS_BLOCKS_PER_CU', '-')} dim {dim}: {best*1e3:.1f} ms {pairs/best/1e6:.1f} Mpairs/s frac {pairs/best*8*dim*7/8e12:.3f}", flush=True)
