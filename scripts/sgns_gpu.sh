set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sgns_gpu.py -x -q 2>&1 | tail -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python - <<'PY' 2>&1 | tee gpurun_out/sgns_timing.log
import time, torch
from node2vec_amd import synthetic, randomwalk as rw, sgns
g = synthetic.rmat(20, 5_000_000, device="cuda")
start = rw.start_vertices(g)
s = start[:47104].contiguous()
walks, valid = rw.walk(g, s, 10, 80, 1.0, 1.0, 42)
torch.cuda.synchronize()
# vocabulary = all vertices (ids as indices), counts from degrees (stationary distribution)
deg = g.degrees().clamp(min=1)
order = torch.sort(deg, descending=True, stable=True).indices
index_of = torch.empty(g.n_vertices, dtype=torch.int32, device="cuda"); index_of[order] = torch.arange(g.n_vertices, dtype=torch.int32, device="cuda")
vocab = sgns.Vocab(order, deg[order], index_of)
for dim in (128, 256, 512):
    m = sgns.SgnsModel(vocab, dim, 5, 5, seed=1, sample=0.0)
    idx = index_of[walks.long()]
    for it in range(3):
        m.pairs.zero_(); torch.cuda.synchronize(); t=time.time()
        m.train_block(idx, 0.025, 0)
        torch.cuda.synchronize(); dt=time.time()-t
        pairs = int(m.pairs.item())
        print(f"dim={dim} rows={idx.shape[0]} pairs={pairs} {dt*1e3:.1f} ms  {pairs/dt/1e6:.1f} Mpairs/s  {pairs/dt*8*dim*7/1e12:.2f} TB/s algorithmic")
PY
