"""vertex-id output at the ranked kernel's rate? (VERDICT r4 missing #6).  The ranked walk writes degree ranks;
random_walk() returns vertex ids.  Times, on a cfg 4 batch (2^20 x 10 x 81 tokens): the ranked kernel, the
hop-table kernel (vertex ids, today's default), the ranked kernel translating in-kernel (one gather per token),
and a streaming pass rank -> vertex id over the finished batch (torch gather through the 0.35 GB rank_vertex
table: what any out-of-kernel translation costs at best)."""
import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from node2vec_amd import synthetic, randomwalk as rw
g = synthetic.chung_lu(100_000_000, 500_000_000, device="cuda").trimmed(10_000, 42)
start = rw.start_vertices(g)
g.build_ranked(); g.build_hops()
B = 1 << 20
walks = torch.empty((B * 10, 81), dtype=torch.int32, device="cuda"); valid = torch.empty(B * 10, dtype=torch.uint8, device="cuda")
def ev(fn, reps=5):
    fn(0); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for k in range(1, 1 + reps): fn(k)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
t_rank = ev(lambda k: rw.walk(g, start[k * B:(k + 1) * B], 10, 80, 1.0, 1.0, 42, out=(walks, valid), check=False, rank_ids=True))
t_hops = ev(lambda k: rw.walk(g, start[k * B:(k + 1) * B], 10, 80, 1.0, 1.0, 42, out=(walks, valid), check=False))
t_inker = ev(lambda k: rw.walk(g, start[k * B:(k + 1) * B], 10, 80, 1.0, 1.0, 42, out=(walks, valid), check=False, use_ranked=True))
rw.walk(g, start[:B], 10, 80, 1.0, 1.0, 42, out=(walks, valid), check=False, rank_ids=True)
flat = walks.view(-1)
out = torch.empty_like(flat)
idx64 = flat.long()
t_gather = ev(lambda k: torch.index_select(g.rank_vertex, 0, idx64, out=out))
t_copy = ev(lambda k: out.copy_(flat))
print(f"cfg4 batch: ranked (ranks out) {t_rank:.2f} ms | hop table (vertex ids) {t_hops:.2f} ms | ranked, translated in-kernel {t_inker:.2f} ms | "
      f"streaming rank->vertex gather over the batch {t_gather:.2f} ms (plain copy of the batch {t_copy:.2f} ms) -> ranked + pass = {t_rank + t_gather:.2f} ms", flush=True)
