"""cfg 3 scale check on one GPU: Chung-Lu 10 M vertices / ~200 M directed edges,
trim at 10 000, alias build, exact + fast walks on a sample, SGNS on the walks."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from node2vec_amd import synthetic, randomwalk as rw, sgns
from node2vec_amd.fugue import trim_hotspot_edges
from node2vec_amd.graph import DeviceGraph

def t(msg, t0):
    torch.cuda.synchronize(); print(f"{msg}: {time.time()-t0:.2f} s", flush=True)

N, DRAWS = int(os.environ.get("N", 10_000_000)), int(os.environ.get("DRAWS", 100_000_000))
t0 = time.time(); g = synthetic.chung_lu(N, DRAWS, device="cuda"); t(f"chung_lu V={g.n_vertices} E={g.n_edges} maxdeg={int(g.degrees().max())}", t0)
# trim_hotspot_vertices at cap 10 000 (examples/fugue_spark.py:47)
t0 = time.time()
src = torch.repeat_interleave(torch.arange(g.n_vertices, device="cuda"), g.degrees())
keep = trim_hotspot_edges(src, 10_000, 42)
g = DeviceGraph.from_edges(src[keep], g.col[keep].long(), g.w[keep], n_vertices=g.n_vertices, device="cuda")
del src, keep
t(f"trim -> E={g.n_edges} maxdeg={int(g.degrees().max())}", t0)
t0 = time.time(); g.build_alias(); t("alias build (K1)", t0)
start_all = rw.start_vertices(g); print("start vertices", start_all.numel())
sample = start_all[torch.randperm(start_all.numel(), device="cuda")[:100_000]].contiguous()
for mode in ("exact", "fast"):
    for p, q in ((1.0, 1.0), (0.5, 2.0)):
        rw.walk(g, sample[:1000], 10, 80, p, q, 42, mode=mode); torch.cuda.synchronize()
        t0 = time.time(); walks, valid = rw.walk(g, sample, 10, 80, p, q, 42, mode=mode); torch.cuda.synchronize(); dt = time.time() - t0
        deg = g.degrees(); dv = deg[walks[valid][:, :-1].long()].double().mean().item()
        print(f"{mode} p={p} q={q}: {dt*1e3:.1f} ms {int(valid.sum())*80/dt/1e6:.1f} Msteps/s, mean visited deg {dv:.0f}", flush=True)
deg = g.degrees().clamp(min=1)
order = torch.sort(deg, descending=True, stable=True).indices
index_of = torch.empty(g.n_vertices, dtype=torch.int32, device="cuda"); index_of[order] = torch.arange(g.n_vertices, dtype=torch.int32, device="cuda")
m = sgns.SgnsModel(sgns.Vocab(order, deg[order], index_of), 128, 5, 5, seed=1, sample=0.0)
idx = index_of[walks[valid].long()]
for it in range(2):
    m.pairs.zero_(); torch.cuda.synchronize(); t0 = time.time(); m.train_block(idx, 0.025, 0); torch.cuda.synchronize(); dt = time.time() - t0
    print(f"sgns dim=128 n_vocab={g.n_vertices} sample=0: {int(m.pairs.item())/dt/1e6:.1f} Mpairs/s = {int(m.pairs.item())/dt*7168/1e12:.2f} TB/s algorithmic", flush=True)
print("max HBM allocated GB", torch.cuda.max_memory_allocated() / 1e9)
