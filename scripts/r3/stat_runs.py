"""Every statistic that a hogwild (racy) test asserts, measured over N runs on this box, so that the
tolerances in tests/ are mean +- 5 sigma of a logged sample and not one lucky measurement
(VERDICT r2 item 1b).  python scripts/r3/stat_runs.py [N]   -> one line per statistic."""
import os
import sys

import numpy as np
import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import n2v_oracle as oracle  # noqa: E402  (test infrastructure: the serial reference point)
import test_sgns_batched_gpu as TB  # noqa: E402
import test_sgns_gpu as TS  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 25


def report(name, xs):
    xs = np.asarray(xs, np.float64)
    print(f"{name}: n={len(xs)} mean={xs.mean():.4f} sd={xs.std(ddof=1):.4f} min={xs.min():.4f} "
          f"max={xs.max():.4f} mean-5sd={xs.mean() - 5 * xs.std(ddof=1):.4f} "
          f"mean+5sd={xs.mean() + 5 * xs.std(ddof=1):.4f}", flush=True)


# ---- planted partition, default trainer (tests/test_sgns_gpu.py) -------------------------------
case = TS.planted_case()
det = case["train"](True)
a_det = case["auc"](det)
print(f"planted partition: deterministic AUC {a_det:.4f}", flush=True)
runs = [case["train"](False) for _ in range(N)]
aucs = [case["auc"](h) for h in runs]
report("planted default hogwild AUC", aucs)
report("planted default |AUC - det AUC|", [abs(a - a_det) for a in aucs])
report("planted default procrustes(raw) hogwild vs hogwild[0]", [case["procrustes_raw"](runs[0], h) for h in runs[1:]])
report("planted default procrustes(unit rows) hogwild vs hogwild[0]", [case["procrustes"](runs[0], h) for h in runs[1:]])
report("planted default procrustes(raw) det vs hogwild", [case["procrustes_raw"](det, h) for h in runs])
report("planted default procrustes(unit rows) det vs hogwild", [case["procrustes"](det, h) for h in runs])
report("planted default max row norm / median row norm", [float(np.linalg.norm(h, axis=1).max() / np.median(np.linalg.norm(h, axis=1))) for h in runs])

# ---- two cliques (tests/test_sgns_gpu.py) ----------------------------------------------------------
cl = TS.two_clique_case()
report("two cliques hogwild gap", [cl["gap_gpu"]() for _ in range(N)])
print(f"two cliques oracle gap {cl['gap_cpu'](oracle):.4f}", flush=True)

# ---- norm ratios on a 500-row vocabulary -----------------------------------------------------------
for name, hub in (("default", 0), ("hub_rows=500", 500)):
    r = []
    for _ in range(N):
        sg, m, idx = TS._setup(500, 3000, 41, 128, seed=1, sample=1e-3)
        m.hub_rows = hub
        if not r:
            s0, s1 = m.syn0.cpu().numpy().copy(), m.syn1neg.cpu().numpy().copy()
            oracle.sgns_train(idx.cpu().numpy(), s0, s1, m.cum_table.cpu().numpy(), m.sample_int.cpu().numpy(),
                              sg.exp_table(), len(m.vocab), 0, m.seed, 128, 5, 5, 0.025)
        m.train_block(idx, 0.025, 0)
        torch.cuda.synchronize()
        r.append(float(np.linalg.norm(m.syn0.cpu().numpy()) / np.linalg.norm(s0)))
    report(f"500-row vocabulary, {name}: |syn0| hogwild / serial", r)
r = []
walks = TB._corpus(500, 3000, 41, 1, True)
for _ in range(N):
    sg, m, idx = TB._model(walks, 128, 5, 5, 1, 1e-3)
    if not r:
        s0, s1 = m.syn0.cpu().numpy().copy(), m.syn1neg.cpu().numpy().copy()
        oracle.sgns_train(idx.cpu().numpy(), s0, s1, m.cum_table.cpu().numpy(), m.sample_int.cpu().numpy(),
                          sg.exp_table(), len(m.vocab), 0, m.seed, 128, 5, 5, 0.025, batched=True)
    m.train_block(idx, 0.025, 0)
    torch.cuda.synchronize()
    r.append(float(np.linalg.norm(m.syn0.cpu().numpy()) / np.linalg.norm(s0)))
report("500-row vocabulary, batched: |syn0| hogwild / serial", r)

# ---- planted partition, batched trainer (tests/test_sgns_batched_gpu.py) ---------------------------
pc = TB.planted_auc_case()
a_bdet = pc["auc"](True, True)
print(f"planted partition: batched deterministic AUC {a_bdet:.4f}", flush=True)
d = [pc["auc"](False, False) for _ in range(N)]
b = [pc["auc"](True, False) for _ in range(N)]
report("planted (batched test) default hogwild AUC", d)
report("planted (batched test) batched hogwild AUC", b)
report("planted (batched test) batched - default", [x - y for x, y in zip(b, d)])
report("planted (batched test) batched - batched det", [x - a_bdet for x in b])
