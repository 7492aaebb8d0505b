"""fit_streaming trains a batch in 6.2 s, the same batch in a fresh process in 5.75 s (r5f, r5g).  What in the
process state does it?  MODE=early: the two matrices allocated before anything else; MODE=late: after ten batches of
pass 1 (walk + count) as the pipeline does; then the pipeline's own first training batch, timed."""
import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from node2vec_amd import synthetic, randomwalk as rw, sgns
mode = os.environ.get("MODE", "late")
torch.cuda.init()
pre = None
if mode == "early":
    pre = (torch.empty((86_699_303, 128), dtype=torch.float32, device="cuda"),
           torch.empty((86_699_303, 128), dtype=torch.float32, device="cuda"))
g = synthetic.chung_lu(100_000_000, 500_000_000, device="cuda").trimmed(10_000, 42)
start = rw.start_vertices(g)
g.build_ranked()
counts = torch.zeros(g.n_vertices, dtype=torch.int64, device="cuda")
for k in range(10):
    walks, valid = rw.walk(g, start[k << 20:(k + 1) << 20].contiguous(), 10, 80, 1.0, 1.0, 42, rank_ids=True, check=False)
    sgns.corpus_count(walks, valid, counts)
counts = counts[g.rank_of.long()]
deg = g.degrees()
counts = torch.where(deg > 0, torch.maximum(counts, torch.ones_like(counts)), counts)  # every start vertex in the vocabulary
ids = torch.nonzero(counts >= 1).reshape(-1)
cnt = counts[ids]
o = torch.sort(cnt, descending=True, stable=True).indices
ids, cnt = ids[o], cnt[o]
del counts, o
index_of = torch.full((g.n_vertices,), -1, dtype=torch.int32, device="cuda")
index_of[ids] = torch.arange(ids.numel(), dtype=torch.int32, device="cuda")
vocab = sgns.Vocab(ids, cnt, index_of)
token_index = index_of[g.rank_vertex.long()].contiguous()
if pre is not None:
    n = len(vocab)
    class M(sgns.SgnsModel):
        pass
    m = sgns.SgnsModel.__new__(sgns.SgnsModel)
    sgns.SgnsModel.__init__(m, vocab, 128, 5, 5, 1, sample=0.0)
    a, b = pre[0][:n], pre[1][:n]
    a.copy_(m.syn0); b.copy_(m.syn1neg)
    m.syn0, m.syn1neg = a, b
    torch.cuda.empty_cache()
else:
    m = sgns.SgnsModel(vocab, 128, 5, 5, 1, sample=0.0)
print(f"MODE={mode}: vocabulary {len(vocab)}, syn0 at {m.syn0.data_ptr():#x}, syn1neg at {m.syn1neg.data_ptr():#x}", flush=True)
sched = sgns.JobSchedule.for_corpus(1000, 81, 86_699_303 * 10, 0, 1, 0.025, 1e-4)
for k in range(3):
    walks, valid = rw.walk(g, start[k << 20:(k + 1) << 20].contiguous(), 10, 80, 1.0, 1.0, 42, rank_ids=True, check=False)
    idx = sgns.corpus_index(walks, valid, token_index)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for j, part in enumerate(torch.split(idx, 1 << 22)):
        m.train_block(part, 0.025, k * 10_485_760 + j * (1 << 22), False, sched, k * 10_485_760 + j * (1 << 22))
    torch.cuda.synchronize()
    print(f"  batch {k}: {time.perf_counter() - t0:.3f} s", flush=True)
