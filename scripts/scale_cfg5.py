"""BASELINE cfg 5 at full size on one GPU: skewed bipartite graph, 50 M vertices, 5 000 hubs of
10 000 leaves + one hub per leaf; p=4, q=0.25; SGNS dim 256."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from node2vec_amd import synthetic, randomwalk as rw, sgns
t0 = time.time(); g = synthetic.hub_bipartite(50_000_000, 5000, 10_000, device="cuda"); torch.cuda.synchronize()
deg = g.degrees(); print(f"cfg5 graph V={g.n_vertices} E={g.n_edges} maxdeg={int(deg.max())} hubs mean deg={deg[:5000].double().mean().item():.0f}: {time.time()-t0:.1f} s", flush=True)
t0 = time.time(); g.build_alias(); torch.cuda.synchronize(); print(f"alias build: {time.time()-t0:.3f} s", flush=True)
start = rw.start_vertices(g)
sample = torch.cat([start[:5000], start[5000:][torch.randperm(start.numel() - 5000, device='cuda')[:95_000]]]).contiguous()
for mode in ("exact", "fast"):
    rw.walk(g, sample[:1000], 10, 80, 4.0, 0.25, 42, mode=mode); torch.cuda.synchronize()
    t0 = time.time(); walks, valid = rw.walk(g, sample, 10, 80, 4.0, 0.25, 42, mode=mode); torch.cuda.synchronize(); dt = time.time() - t0
    w = walks[valid].long(); dv = deg[w[:, :-1]].double()
    alg = (16 + 8 * dv + 4).sum() + (16 + 4 * dv[:, :-1]).sum()
    print(f"{mode} p=4 q=0.25: {dt*1e3:.1f} ms {int(valid.sum())*80/dt/1e6:.1f} Msteps/s, mean visited deg {dv.mean().item():.0f}, "
          f"algorithmic {alg.item()/dt/1e12:.2f} TB/s (exact-mode formula)", flush=True)
clampdeg = deg.clamp(min=1)
order = torch.sort(clampdeg, descending=True, stable=True).indices
index_of = torch.empty(g.n_vertices, dtype=torch.int32, device="cuda"); index_of[order] = torch.arange(g.n_vertices, dtype=torch.int32, device="cuda")
m = sgns.SgnsModel(sgns.Vocab(order, clampdeg[order], index_of), 256, 5, 5, seed=1, sample=0.0)
idx = index_of[walks[valid].long()]
for it in range(2):
    m.pairs.zero_(); torch.cuda.synchronize(); t0 = time.time(); m.train_block(idx, 0.025, 0); torch.cuda.synchronize(); dt = time.time() - t0
    print(f"sgns dim=256 n_vocab={g.n_vertices}: {int(m.pairs.item())/dt/1e6:.1f} Mpairs/s = {int(m.pairs.item())/dt*8*256*7/1e12:.2f} TB/s algorithmic", flush=True)
print("max HBM allocated GB", torch.cuda.max_memory_allocated() / 1e9)
