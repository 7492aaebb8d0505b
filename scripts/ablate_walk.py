import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from node2vec_amd import synthetic, randomwalk as rw
g = synthetic.rmat(20, 5_000_000, device="cuda")
start = rw.start_vertices(g)[:47104].contiguous()
PQ = ((0.5, 2.0),) if os.environ.get('N2V_HIP_LIB') else ((0.5, 2.0), (1.0, 1.0), (1.0, 2.0), (0.5, 1.0))
for p, q in PQ:
    best = 1e9
    for it in range(3):
        torch.cuda.synchronize(); t = time.time()
        walks, valid = rw.walk(g, start, 10, 80, p, q, 42)
        torch.cuda.synchronize(); best = min(best, time.time() - t)
    print(f"{os.environ.get('N2V_HIP_LIB','prod'):40s} p={p} q={q}: {best*1e3:7.1f} ms {int(valid.sum())*80/best/1e6:7.1f} Msteps/s", flush=True)

if not os.environ.get('N2V_HIP_LIB'):
    deg = g.degrees()
    wl = walks[valid].long()
    dv = deg[wl[:, 1:-1]].reshape(-1).double(); ds = deg[wl[:, :-2]].reshape(-1).double()
    qs = torch.tensor([0.1, 0.25, 0.5, 0.75, 0.9, 0.99], device="cuda", dtype=torch.float64)
    idx = torch.randperm(dv.numel(), device="cuda")[:2_000_000]
    print("deg(v) quantiles", torch.quantile(dv[idx], qs).tolist(), "mean", dv.mean().item())
    print("frac deg(v)<=64:", (dv <= 64).double().mean().item(), " <=16:", (dv <= 16).double().mean().item(), " >1024:", (dv > 1024).double().mean().item())
    print("bytes share deg(v)>1024:", (dv[dv > 1024].sum() / dv.sum()).item(), " deg(v)<=64:", (dv[dv <= 64].sum() / dv.sum()).item())
    print("frac m>8n+64:", (ds > 8 * dv + 64).double().mean().item(), " m>8192:", (ds > 8192).double().mean().item())
