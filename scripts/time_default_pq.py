import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from node2vec_amd import synthetic, randomwalk as rw
for weights in (None, "uniform"):
    g = synthetic.rmat(20, 5_000_000, device="cuda", weights=weights)
    start = rw.start_vertices(g)[:47104].contiguous()
    for it in range(3):
        torch.cuda.synchronize(); t = time.time()
        walks, valid = rw.walk(g, start, 10, 80, 1.0, 1.0, 42)
        torch.cuda.synchronize(); dt = time.time() - t
        print(f"exact p=q=1 weights={weights} run {it}: {dt*1e3:.1f} ms {int(valid.sum())*80/dt/1e6:.1f} Msteps/s", flush=True)
