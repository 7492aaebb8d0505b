"""why does fit_streaming's trainer run at 772 M pairs/s on cfg 4 when bench.py's SGNS leg shows 884 M?  One batch of
the real pipeline (2^20 start vertices x 10 walks), the 10^8 x 128 model, launches timed with HIP events:
with / without per-job rates, chunks of 2^22 rows or the whole batch, vocabulary by count or by degree."""
import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from node2vec_amd import synthetic, randomwalk as rw, sgns
g = synthetic.chung_lu(100_000_000, 500_000_000, device="cuda").trimmed(10_000, 42)
start = rw.start_vertices(g)
walks, valid = rw.walk(g, start[:1 << 20].contiguous(), 10, 80, 1.0, 1.0, 42)
deg = g.degrees()
for order_name in ("degree", "count"):
    if order_name == "degree":
        cnt = deg.clamp(min=1)
    else:  # the pipeline's vocabulary: token counts of the corpus (here: expected counts = 81 x 10 x degree share), ties by id
        cnt = (deg * 10).clamp(min=1)
    ids = torch.nonzero(deg > 0).reshape(-1)
    c = cnt[ids]
    o = torch.sort(c, descending=True, stable=True).indices
    ids, c = ids[o], c[o]
    index_of = torch.full((g.n_vertices,), -1, dtype=torch.int32, device="cuda")
    index_of[ids] = torch.arange(ids.numel(), dtype=torch.int32, device="cuda")
    vocab = sgns.Vocab(ids, c, index_of)
    m = sgns.SgnsModel(vocab, 128, 5, 5, seed=1, sample=0.0)
    idx = sgns.corpus_index(walks, valid, index_of)
    rows = idx.shape[0]
    print(f"vocabulary by {order_name}: {len(vocab)} words, auto hub rows {m.auto_hub_rows()}", flush=True)
    sched = sgns.JobSchedule.for_corpus(1000, 81, 86_699_303 * 10, 0, 1, 0.025, 1e-4)
    for name, sc, chunk in (("launch rate, 2^22-row chunks", None, 1 << 22), ("job rates, 2^22-row chunks", sched, 1 << 22),
                            ("job rates, whole batch", sched, rows), ("launch rate, whole batch", None, rows)):
        for rep in range(2):
            m.pairs.zero_()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for j, part in enumerate(torch.split(idx, chunk)):
                m.train_block(part, 0.025, (rep + 1) * rows + j * chunk, False, sc, j * chunk)
            b.record(); torch.cuda.synchronize()
        dt = a.elapsed_time(b) * 1e-3
        print(f"  {name}: {int(m.pairs.item()) / dt / 1e6:.0f} M pairs/s ({dt * 1e3:.0f} ms)", flush=True)
    del m, vocab
    torch.cuda.empty_cache()
