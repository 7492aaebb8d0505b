"""Link AUC after ONE epoch of the streamed pipeline on BASELINE cfg 3 (Chung-Lu 10^7 vertices / 10^8 draws, trimmed at
10 000, p = q = 1, 10 x 80, dim 128) by concurrency regime of the SGNS trainer: hub_rows chosen from the corpus (the
default) against plain stores everywhere (hub_rows = 0).  The evidence for the default regime was cfg 2 only (VERDICT r5,
weak 8).   python scripts/r6/auc_cfg3.py [auto,0]"""
import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from node2vec_amd import synthetic
from node2vec_amd.pipeline import fit_streaming

g = synthetic.chung_lu(10_000_000, 100_000_000, seed=42, device="cuda").trimmed(10_000, 42)
deg = g.degrees()
gen = torch.Generator(device="cuda").manual_seed(1)
e = torch.randint(0, g.n_edges, (400_000,), generator=gen, device="cuda")
src = torch.repeat_interleave(torch.arange(g.n_vertices, device="cuda"), deg)
pa, pb = src[e], g.col[e].long()
del src
have = torch.nonzero(deg > 0).reshape(-1)
na = have[torch.randint(0, have.numel(), (400_000,), generator=gen, device="cuda")]
nb = have[torch.randint(0, have.numel(), (400_000,), generator=gen, device="cuda")]
# the rows the regime is about: edges of the 1 000 vertices of highest degree against (hub, random vertex) pairs
hubs = torch.topk(deg, 1000).indices
hmask = torch.zeros(g.n_vertices, dtype=torch.bool, device="cuda")
hmask[hubs] = True
he = torch.nonzero(hmask[torch.repeat_interleave(torch.arange(g.n_vertices, device="cuda"), deg)]).reshape(-1)
he = he[torch.randint(0, he.numel(), (40_000,), generator=gen, device="cuda")]
hsrc = torch.searchsorted(g.rowptr, he, right=True) - 1
hdst = g.col[he].long()
hneg = have[torch.randint(0, have.numel(), (4_000,), generator=gen, device="cuda")]
print(f"cfg3: {g.n_vertices} vertices, {g.n_edges} edges", flush=True)
for hub in (sys.argv[1] if len(sys.argv) > 1 else "auto,0").split(","):
    w2v = {"size": 128, "iter": 1, "min_count": 0, "sample": 0.0, "negative": 5, "window": 5}
    if hub != "auto":
        w2v["hub_rows"] = int(hub)
    t = {}
    t0 = time.perf_counter()
    model, m = fit_streaming(g, {"num_walks": 10, "walk_length": 80, "return_param": 1.0, "inout_param": 1.0}, w2v, 42,
                             batch_vertices=1 << 18, return_model=True, timings=t)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    index_of = m.vocab.index_of.long()
    u = m.syn0 - m.syn0.mean(0)
    u = u / (u.norm(dim=1, keepdim=True) + 1e-30)
    sp = (u[index_of[pa]] * u[index_of[pb]]).sum(1)
    sn = (u[index_of[na]] * u[index_of[nb]]).sum(1)
    k = 40000
    auc = float((sp[:k, None] > sn[None, :4000]).float().mean())
    auc2 = float((sp[k:2 * k, None] > sn[None, 4000:8000]).float().mean())
    hp = (u[index_of[hsrc]] * u[index_of[hdst]]).sum(1)
    hn = u[index_of[hsrc[:4000]]] @ u[index_of[hneg]].T  # every sampled hub against 4 000 random vertices
    hauc = float((hp[:4000, None] > hn).float().mean())
    hub_norm = float(m.syn0[:100].norm(dim=1).mean()), float(m.syn1neg[:100].norm(dim=1).mean())
    print(f"hub_rows={hub}: edges of the 1 000 biggest hubs: link AUC {hauc:.4f}; norms of the 100 most frequent rows "
          f"syn0 {hub_norm[0]:.3f} syn1neg {hub_norm[1]:.3f}", flush=True)
    print(f"hub_rows={hub} (used {m.hub_rows}, waves {m.hub_waves}): link AUC {auc:.4f} / {auc2:.4f} (two disjoint samples), "
          f"epoch {dt:.1f} s, pairs {int(m.pairs.item())}, timings { {k_: round(v, 1) for k_, v in t.items() if isinstance(v, float)} }",
          flush=True)
    del model, m, u
    torch.cuda.empty_cache()
