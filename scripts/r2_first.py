"""Round-2 first GPU check: lanes kernel vs wave kernel (bit-equal + timing), edge classes vs oracle."""
import os, sys, time, torch, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import n2v_oracle
from node2vec_amd import synthetic, randomwalk as rw

dev = "cuda"
g = synthetic.rmat(12, 40000, device=dev)
t0 = time.time(); g.build_edge_classes(); torch.cuda.synchronize(); print("ec small", time.time() - t0)
want = n2v_oracle.edge_classes(g.rowptr.cpu().numpy(), g.col.cpu().numpy())
got = g.edge_classes.cpu().numpy().view(np.uint32)
print("edge classes equal to oracle:", np.array_equal(want, got), (want != got).sum())
start = rw.start_vertices(g)
for p, q in ((0.5, 2.0), (4.0, 0.25), (2.0, 1.0), (1.0, 1.0), (0.25, 0.25)):
    a, va = rw.walk(g, start, 4, 40, p, q, 7, mode="exact")
    b, vb = rw.walk(g, start, 4, 40, p, q, 7, mode="exact", use_edge_classes=False)
    torch.cuda.synchronize()
    print(p, q, "lanes == wave:", bool(torch.equal(a, b)) and bool(torch.equal(va, vb)))
    w, wv = n2v_oracle.random_walk(g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.w.cpu().numpy(), start[:300].cpu().numpy(), 4, 40, p, q, 7, n_threads=8)
    print("   == oracle:", np.array_equal(a[:1200].cpu().numpy()[wv], w[wv]))

g = synthetic.rmat(20, 5_000_000, device=dev)
torch.cuda.synchronize(); t0 = time.time(); g.build_edge_classes(); torch.cuda.synchronize(); print("edge classes cfg2: %.3f s" % (time.time() - t0))
start = rw.start_vertices(g)[:47104].contiguous()
for p, q in ((0.5, 2.0), (4.0, 0.25), (1.0, 1.0), (2.0, 1.0)):
    for uec in (True, False):
        rw.walk(g, start[:1000], 10, 80, p, q, 42, use_edge_classes=uec); torch.cuda.synchronize()
        t0 = time.time(); a, va = rw.walk(g, start, 10, 80, p, q, 42, use_edge_classes=uec); torch.cuda.synchronize(); dt = time.time() - t0
        print(f"cfg2 p={p} q={q} edge_classes={uec}: {dt*1e3:.1f} ms  {int(va.sum())*80/dt/1e6:.0f} Msteps/s", flush=True)
        if uec: ref = a
        else: print("   equal:", bool(torch.equal(ref, a)))
