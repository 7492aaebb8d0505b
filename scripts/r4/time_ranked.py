"""p = q = 1 exact walks: the 16-byte hop table against the degree-ranked 4-byte table, emitting vertex
ids (one more gather per token) or ranks.  GRAPH=cfg4|cfg3|cfg5|cfg2  BATCH=1048576  python scripts/r4/time_ranked.py"""
import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from node2vec_amd import synthetic, randomwalk as rw
cfg = os.environ.get("GRAPH", "cfg4")
if cfg == "cfg4":
    g = synthetic.chung_lu(100_000_000, 500_000_000, device="cuda").trimmed(10_000, 42)
elif cfg == "cfg3":
    g = synthetic.chung_lu(10_000_000, 100_000_000, device="cuda").trimmed(10_000, 42)
elif cfg == "cfg5":
    g = synthetic.hub_bipartite(50_000_000, 5000, 10_000, device="cuda")
else:
    g = synthetic.rmat(20, 5_000_000, device="cuda")
start = rw.start_vertices(g)
b = min(int(os.environ.get("BATCH", 1 << 20)), start.numel())
nb = max(1, start.numel() // b)
walks = torch.empty((b * 10, 81), dtype=torch.int32, device="cuda")
valid = torch.empty(b * 10, dtype=torch.uint8, device="cuda")
ref = torch.empty_like(walks)
t0 = time.perf_counter(); g.build_ranked(); torch.cuda.synchronize()
print(f"{cfg}: build_ranked {time.perf_counter() - t0:.2f} s; ranked form "
      f"{'declined' if g.rank_hops is None else 'built'}; head {0 if g.rank_head is None else g.rank_head.numel()} "
      f"classes {None if g.rank_class_first is None else int((g.rank_class_first != -1).sum())} "
      f"of {None if g.rank_class_first is None else g.rank_class_first.numel()}; distinct degrees "
      f"{torch.unique(g.degrees()).numel()}; max degree {int(g.degrees().max())}", flush=True)
if g.rank_head is not None:
    H = g.rank_head.numel()
    print(f"  share of edge ends at head vertices: {float(g.degrees()[g.rank_vertex[:H].long()].sum()) / g.n_edges:.4f}", flush=True)

def run(k, out=walks, **kw):
    rw.walk(g, start[(k % nb) * b:(k % nb + 1) * b], 10, 80, 1.0, 1.0, 42, out=(out, valid), check=False, **kw)

def timed(reps=6, **kw):
    run(0, **kw); run(1, **kw); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(2, 2 + reps): run(k, **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

for name, kw in (("hop table 16 B", dict(use_hops8=False)), ("default (8-byte table where it pays)", {}),
                 ("ranked 4 B, vertex ids out", dict(use_ranked=True)), ("ranked 4 B, ranks out", dict(rank_ids=True))):
    dt = timed(**kw)
    run(3, **kw); torch.cuda.synchronize()
    if name.startswith("hop"):
        ref.copy_(walks); same = True
    elif "ranks out" in name:
        same = bool(torch.equal(torch.where(walks >= 0, g.rank_vertex[walks.clamp(min=0).long()], walks), ref))
    else:
        same = bool(torch.equal(walks, ref))
    print(f"{cfg} p=q=1 batch {b}: {name}: {b * 800 / dt / 1e9:.2f} G steps/s ({dt * 1e3:.2f} ms) identical={same}", flush=True)
# the per-token lookup that follows in fit_streaming (corpus_index): with ranks the table is composed once
from node2vec_amd import sgns
index_of = torch.randperm(g.n_vertices, device="cuda", dtype=torch.int64).to(torch.int32)
composed = index_of[g.rank_vertex.long()].contiguous()
for name, kw, tab in (("vertex ids + corpus_index", dict(use_hops8=False), index_of), ("ranks + corpus_index(composed)", dict(rank_ids=True), composed)):
    outs = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run(4 + rep, **kw)
        idx = sgns.corpus_index(walks, valid, tab)
        torch.cuda.synchronize(); outs.append(time.perf_counter() - t0)
    if "composed" in name:
        print(f"  {name}: {min(outs) * 1e3:.2f} ms; same indices: {bool(torch.equal(idx, ref_idx))}", flush=True)
    else:
        run(4 + 2, **kw); ref_idx = sgns.corpus_index(walks, valid, tab)
        print(f"  {name}: {min(outs) * 1e3:.2f} ms", flush=True)
