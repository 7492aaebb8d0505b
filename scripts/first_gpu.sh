set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_walk_gpu.py -x -q 2>&1 | tail -30
timeout 600 python - <<'PY' 2>&1 | tee gpurun_out/first_timing.log
import time, torch, sys
sys.path.insert(0, "oracle")
from node2vec_amd import synthetic, randomwalk as rw
t=time.time(); g = synthetic.rmat(20, 5_000_000, device="cuda"); torch.cuda.synchronize()
print("rmat V", g.n_vertices, "E", g.n_edges, "gen s", time.time()-t, "maxdeg", int(g.degrees().max()))
start = rw.start_vertices(g); print("start", start.numel())
for (p,q) in ((1.0,1.0),(0.5,2.0)):
  for n_start in (20000, 200000):
    s = start[torch.randperm(start.numel(), device="cuda")[:n_start]].contiguous()
    for L in (80,):
        torch.cuda.synchronize(); t=time.time()
        walks, valid = rw.walk(g, s, 10, L, p, q, 42)
        torch.cuda.synchronize(); dt=time.time()-t
        steps = int(valid.sum())*L
        deg = g.degrees()
        dv = deg[walks[:, :-1].long()].double().mean().item()
        print(f"p={p} q={q} n_start={n_start} L={L}: {dt:.3f}s  {steps/dt/1e6:.2f} Msteps/s  mean deg visited {dv:.0f}")
PY
