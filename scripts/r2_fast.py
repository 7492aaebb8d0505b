"""fast-mode timing on cfg 2 (and the chi-square parity tests are in tests/): python scripts/r2_fast.py"""
import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from node2vec_amd import synthetic, randomwalk as rw
for weights in (None, "uniform"):
    g = synthetic.rmat(20, 5_000_000, device="cuda", weights=weights)
    start = rw.start_vertices(g)[:47104 * 4].contiguous()
    for p, q in ((0.5, 2.0), (4.0, 0.25), (1.0, 1.0)):
        for uec in ((True, False) if weights is None else (True,)):
            st = {}
            rw.walk(g, start[:1000], 10, 80, p, q, 42, mode="fast", use_edge_classes=uec); torch.cuda.synchronize()
            t0 = time.time(); a, va = rw.walk(g, start, 10, 80, p, q, 42, mode="fast", stats=st, use_edge_classes=uec); torch.cuda.synchronize(); dt = time.time() - t0
            steps = int(va.sum()) * 80
            print(f"fast weights={weights} p={p} q={q} edge_classes={uec}: {dt*1e3:.1f} ms {steps/dt/1e9:.2f} Gsteps/s trials/step {int(st['trials'].item())/steps:.2f}", flush=True)
