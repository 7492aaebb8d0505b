import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from node2vec_amd import synthetic, randomwalk as rw
g = synthetic.rmat(20, 5_000_000, device="cuda")
start = rw.start_vertices(g)[:47104].contiguous()
for p, q in ((0.7, 1.3), (3.0, 0.7), (1.0, 3.0)):
    best = 1e9
    for it in range(2):
        torch.cuda.synchronize(); t = time.time()
        walks, valid = rw.walk(g, start, 10, 80, p, q, 42)
        torch.cuda.synchronize(); best = min(best, time.time() - t)
    print(f"exact p={p} q={q}: {best*1e3:8.1f} ms {int(valid.sum())*80/best/1e6:7.1f} Msteps/s", flush=True)
