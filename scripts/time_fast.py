import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from node2vec_amd import synthetic, randomwalk as rw
for weights in (None, "uniform"):
    g = synthetic.rmat(20, 5_000_000, device="cuda", weights=weights).build_alias()
    start = rw.start_vertices(g)[:47104].contiguous()
    for p, q in ((0.5, 2.0), (4.0, 0.25), (1.0, 1.0)):
        best = 1e9; st = {}
        for it in range(3):
            torch.cuda.synchronize(); t = time.time()
            walks, valid = rw.walk(g, start, 10, 80, p, q, 42, mode="fast", stats=st)
            torch.cuda.synchronize(); best = min(best, time.time() - t)
        n = int(valid.sum()) * 80
        chk = int((walks.long() * torch.arange(1, walks.shape[1] + 1, device='cuda')).sum().item()) & 0xffffffffffff
        print(f"fast weights={weights} p={p} q={q}: {best*1e3:7.1f} ms {n/best/1e6:8.1f} Msteps/s trials/step {int(st['trials'].item())/n:.2f} checksum {chk:x}", flush=True)
