import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from node2vec_amd import randomwalk as rw, sgns
from node2vec_amd.graph import DeviceGraph
rng = np.random.default_rng(0)
nc, sz = 50, 40; nv = nc * sz
comm = np.repeat(np.arange(nc), sz)
src, dst = [], []
for v in range(nv):
    inside = rng.choice(np.nonzero(comm == comm[v])[0], 8)
    outside = rng.integers(0, nv, 2)
    for u in list(inside) + list(outside):
        if u != v: src += [v, u]; dst += [u, v]
g = DeviceGraph.from_edges(src, dst, np.ones(len(src), np.float32), n_vertices=nv, device="cuda")
walks, valid = rw.walk(g, rw.start_vertices(g), 10, 40, 1.0, 1.0, 1)
vocab = sgns.build_vocab(walks, 1); idx = vocab.index_of[walks.long()]
ids = vocab.ids.cpu().numpy()
def train(det, seed):
    m = sgns.SgnsModel(vocab, 64, 5, 5, seed=seed, sample=0.0)
    m.train(idx, epochs=3, alpha=0.025, deterministic=det); torch.cuda.synchronize()
    return m.syn0.cpu().numpy()
def auc(v):
    v = v - v.mean(0); v /= np.linalg.norm(v, axis=1, keepdims=True)
    a, b = rng.integers(0, len(v), 200000), rng.integers(0, len(v), 200000)
    s = (v[a] * v[b]).sum(1); same = comm[ids[a]] == comm[ids[b]]
    pos, neg = s[same], s[~same]
    return float((pos[:, None] > neg[None, :3000]).mean())
def procrustes_cos(x, y):
    x = x - x.mean(0); y = y - y.mean(0)
    u, _, vt = np.linalg.svd(x.T @ y); r = u @ vt
    xr = x @ r
    return float(np.mean((xr * y).sum(1) / (np.linalg.norm(xr, axis=1) * np.linalg.norm(y, axis=1))))
d = train(True, 7); h1 = train(False, 7); h2 = train(False, 7); d2 = train(True, 8)
print("AUC deterministic", auc(d), "hogwild", auc(h1), auc(h2), "deterministic other seed", auc(d2))
print("procrustes cos det vs hogwild", procrustes_cos(d, h1), "hogwild vs hogwild", procrustes_cos(h1, h2), "det seed7 vs det seed8", procrustes_cos(d, d2))
