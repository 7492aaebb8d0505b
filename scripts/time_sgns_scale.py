"""SGNS launch time on synthetic sentences over a large vocabulary (no graph): uniform random
tokens, count^0.75 table from Zipf-like counts.  python scripts/time_sgns_scale.py n_vocab [dim]"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from node2vec_amd import _lib
if os.environ.get("N2V_VARIANT_LIB"):
    _lib.LIB_PATH = os.environ["N2V_VARIANT_LIB"]  # developer build, loaded by path
from node2vec_amd import sgns
n_vocab = int(float(sys.argv[1])); dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = "cuda"
counts = (1e7 / torch.arange(1, n_vocab + 1, device=dev, dtype=torch.float64) ** 0.8).clamp(min=1).long()
ids = torch.arange(n_vocab, device=dev)
vocab = sgns.Vocab(ids, counts, ids.to(torch.int32))
# sentences: tokens drawn proportionally to the counts (what walks look like), 471 040 x 81
probs = counts.double() / counts.sum()
cdf = torch.cumsum(probs, 0)  # inverse-CDF sampling (torch.multinomial stops at 2^24 categories)
idx = torch.searchsorted(cdf, torch.rand(471040 * 81, device=dev, dtype=torch.float64)).clamp_(max=n_vocab - 1)
idx = idx.to(torch.int32).view(471040, 81).contiguous()
del cdf
m = sgns.SgnsModel(vocab, dim, 5, 5, seed=1, sample=0.0)
m.train_block(idx, 0.025, 0); torch.cuda.synchronize()
best = 1e9
for it in range(3):
    m.pairs.zero_(); torch.cuda.synchronize(); t = time.time()
    m.train_block(idx, 0.025, (it + 1) * idx.shape[0]); torch.cuda.synchronize(); best = min(best, time.time() - t)
pairs = int(m.pairs.item())
print(f"{os.path.basename(os.environ.get('N2V_VARIANT_LIB', 'in-tree'))} n_vocab {n_vocab} dim {dim}: {best*1e3:.1f} ms {pairs/best/1e6:.1f} Mpairs/s = {pairs/best*8*dim*7/1e12:.2f} TB/s algorithmic", flush=True)
