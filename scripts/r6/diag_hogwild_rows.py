"""Which rows differ between the ordered (deterministic) run and the default (hogwild) launch of ONE block on a
large model, and by how much -- the question behind the first version of tests/test_sgns_model_size_gpu.py, whose
"syn1neg is order-independent to first order" bound (5 %) failed at 14 - 44 %.  Run on the GPU box:
    python scripts/r6/diag_hogwild_rows.py [n_vocab] [kind] [max_waves]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import test_sgns_model_size_gpu as T  # noqa: E402

from node2vec_amd import sgns  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "top"
max_waves = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dim, rows, length, base = 128, 768, 81, 123_456_789
gen = torch.Generator().manual_seed(n % 1000 + dim + len(kind))
m = T._model(n, dim, kind, 20 + dim, 0.0)
m.max_waves = max_waves
idx = T._corpus(kind, n, rows, length, gen).cuda()
init0 = sgns.init_syn0(n, dim, m.seed, m.syn0.device)
m.train_block(idx, 0.025, base, deterministic=True)
torch.cuda.synchronize()
ch0, ch1 = T._changed_rows(m.syn0, init0), T._changed_rows(m.syn1neg, None)
d0, d1 = m.syn0[ch0].clone(), m.syn1neg[ch1].clone()
m.syn0[ch0] = init0[ch0]
m.syn1neg[ch1] = 0.0
m.pairs.zero_()
m.train_block(idx, 0.025, base)
torch.cuda.synchronize()
print("hub_rows", m.hub_rows, "waves", m.hub_waves, "pairs", int(m.pairs.item()))
h1 = m.syn1neg[ch1]
tok = idx.reshape(-1)
tok = tok[tok >= 0].long()
cnt = torch.bincount(tok, minlength=n)
is_tok = cnt[ch1] > 0
rel = (h1 - d1).norm(dim=1) / d1.norm(dim=1)
print("rows changed in syn1neg:", ch1.numel(), "of which tokens:", int(is_tok.sum()))
for name, sel in (("token rows (centres)", is_tok), ("negative-only rows", ~is_tok)):
    r = rel[sel]
    q = torch.quantile(r.double(), torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], dtype=torch.float64, device=r.device))
    print(name, "n =", r.numel(), "rel quantiles 50/90/99/99.9/max:", [f"{x:.3g}" for x in q.tolist()],
          "share > 0.01:", float((r > 0.01).double().mean()),
          "energy share of the difference:", float(((h1 - d1)[sel].norm() / (h1 - d1).norm()) ** 2))
print("whole: rel", float((h1 - d1).norm() / d1.norm()))
worst = torch.topk(rel, 12).indices
for w in worst.tolist():
    print("row", int(ch1[w]), "tokens", int(cnt[ch1[w]]), "|det|", float(d1[w].norm()), "|hog|", float(h1[w].norm()),
          "cos", float(torch.nn.functional.cosine_similarity(d1[w], h1[w], dim=0)))
h0 = m.syn0[ch0]
rel0 = (h0 - d0).norm(dim=1) / (d0 - init0[ch0]).norm(dim=1)
print("syn0: median drift / step", float(rel0.median()))
