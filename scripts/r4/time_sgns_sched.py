"""SGNS launch time with and without gensim's per-job rates (n2v_sgns_params.row_alpha), cfg 2 corpus
block, default trainer (hub rows by the automatic rule) and batched; run on the GPU box."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from node2vec_amd import synthetic, randomwalk as rw, sgns
g = synthetic.rmat(20, 5_000_000, device="cuda")
s = rw.start_vertices(g)[:47104].contiguous()
walks, valid = rw.walk(g, s, 10, 80, 0.5, 2.0, 42)
deg = g.degrees().clamp(min=1)
order = torch.sort(deg, descending=True, stable=True).indices
index_of = torch.empty(g.n_vertices, dtype=torch.int32, device="cuda"); index_of[order] = torch.arange(g.n_vertices, dtype=torch.int32, device="cuda")
vocab = sgns.Vocab(order, deg[order], index_of)
idx = index_of[walks[valid.bool()].long()].contiguous()
rows = idx.shape[0]
for batched in (False, True):
    for hub in (0, None):
        m = sgns.SgnsModel(vocab, 128, 5, 5, seed=1, sample=0.0)
        m.batched, m.hub_rows = batched, hub
        for name, sched in (("launch rate", None), ("job rates", sgns.JobSchedule.for_corpus(1000, 81, rows * 4, 0, 1, 0.025, 1e-4))):
            m.train_block(idx, 0.025, 0, False, sched, 0); torch.cuda.synchronize()
            best = 1e9
            for it in range(4):
                m.pairs.zero_(); torch.cuda.synchronize(); t = time.time()
                m.train_block(idx, 0.025, (it + 1) * rows, False, sched, (it % 4) * rows); torch.cuda.synchronize()
                best = min(best, time.time() - t)
            print(f"batched={batched} hub_rows={hub} {name}: {best*1e3:.1f} ms {int(m.pairs.item())/best/1e6:.1f} Mpairs/s", flush=True)
