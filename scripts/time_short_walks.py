import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from node2vec_amd import synthetic, randomwalk as rw
g = synthetic.rmat(20, 5_000_000, device="cuda")
start = rw.start_vertices(g).contiguous()
for L, p, q in ((80, 0.5, 2.0), (5, 0.5, 2.0), (2, 0.5, 2.0)):
    s = start[:47104] if L == 80 else start
    best = 1e9
    for it in range(3):
        torch.cuda.synchronize(); t = time.time()
        walks, valid = rw.walk(g, s, 10, L, p, q, 42, use_index=False) if False else rw.walk(g, s, 10, L, p, q, 42)
        torch.cuda.synchronize(); best = min(best, time.time() - t)
    print(f"L={L}: {best*1e3:.2f} ms {int(valid.sum())*L/best/1e6:.1f} Msteps/s walkers {walks.shape[0]}", flush=True)
