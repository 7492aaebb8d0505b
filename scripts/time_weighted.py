import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from node2vec_amd import synthetic, randomwalk as rw
from node2vec_amd.graph import DeviceGraph
base = synthetic.rmat(20, 5_000_000, device="cuda")
gen = torch.Generator(device="cuda").manual_seed(1)
for name, w in (("integer weights 1..5", torch.randint(1, 6, (base.n_edges,), generator=gen, device="cuda").float()),
                ("arbitrary fp32 weights", torch.rand(base.n_edges, generator=gen, device="cuda") * 1.9 + 0.1)):
    g = DeviceGraph(base.rowptr, base.col, w)
    start = rw.start_vertices(g)[:47104].contiguous()
    for p, q in ((0.5, 2.0), (1.0, 1.0)):
        best = 1e9
        for it in range(2):
            torch.cuda.synchronize(); t = time.time()
            walks, valid = rw.walk(g, start, 10, 80, p, q, 42)
            torch.cuda.synchronize(); best = min(best, time.time() - t)
        print(f"generic kernel, {name}, p={p} q={q}: {best*1e3:7.1f} ms {int(valid.sum())*80/best/1e6:7.1f} Msteps/s", flush=True)
