"""Randomised differential test of the SGNS kernels (deterministic mode) against the CPU
oracle: python scripts/fuzz_sgns.py [seconds] [seed].  Round 3: a third of the cases run the
batched kernel against ITS oracle, a fifth the window-cache variant of the default kernel, and half
of the sentences are walk-like (immediate returns: the same word several times in a window)."""
import os, sys, time
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import n2v_oracle
from node2vec_amd import sgns

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); n = 0; tot_pairs = 0
while time.time() - t0 < budget:
    n_tok = int(rng.choice([3, 8, 40, 300, 5000]))
    rows, ln = int(rng.integers(1, 40)), int(rng.choice([1, 2, 7, 30, 81, 200, 256]))
    dim = int(rng.choice([1, 16, 33, 64, 100, 128, 200, 256, 300, 512, 1024]))
    window, K = int(rng.choice([1, 2, 5, 10, 30])), int(rng.choice([1, 3, 5, 6, 11, 20]))
    sample = float(rng.choice([0.0, 1e-3, 1e-1]))
    mode = str(rng.choice(["default", "default", "cache", "batched", "batched"]))
    if mode == "batched":
        dim, window, K = int(rng.choice([64, 128, 256])), int(rng.choice([1, 2, 5, 7])), int(rng.choice([1, 3, 5, 7, 11, 15]))
    elif mode == "cache":
        dim, window = int(rng.choice([64, 128])), int(rng.choice([1, 2, 5, 7]))
    p = 1.0 / np.arange(1, n_tok + 1) ** rng.uniform(0.0, 1.5); p /= p.sum()
    walks = torch.from_numpy(rng.choice(n_tok, size=(rows, ln), p=p).astype(np.int32))
    if ln > 2 and rng.random() < 0.5:  # a b a: frequent immediate returns
        back = torch.from_numpy(rng.random((rows, ln)) < 0.35)
        walks[:, 2:] = torch.where(back[:, 2:], walks[:, :-2], walks[:, 2:])
    if rng.random() < 0.4:
        walks[torch.from_numpy(rng.random((rows, ln)) < 0.15)] = -1  # out-of-vocabulary tokens
    walks = walks.cuda()
    if int((walks >= 0).sum()) == 0:
        continue
    vocab = sgns.build_vocab(walks, int(rng.choice([1, 1, 2])))
    if len(vocab) == 0:
        continue
    m = sgns.SgnsModel(vocab, dim, window, K, seed=int(rng.integers(0, 2 ** 62)), sample=sample,
                       use_cum_index=bool(rng.random() < 0.7))
    m.batched, m.window_cache = mode == "batched", int(mode == "cache")
    idx = torch.where(walks >= 0, vocab.index_of[walks.clamp(min=0).long()], torch.full_like(walks, -1))
    s0, s1 = m.syn0.cpu().numpy().copy(), m.syn1neg.cpu().numpy().copy()
    base, alpha = int(rng.integers(0, 10 ** 9)), float(rng.choice([0.025, 0.1, 0.5]))
    for rep in range(2):
        m.train_block(idx, alpha, base + rep * rows, deterministic=True)  # every drawn shape is supported
        pairs = n2v_oracle.sgns_train(idx.cpu().numpy(), s0, s1, m.cum_table.cpu().numpy(),
                                      None if m.sample_int is None else m.sample_int.cpu().numpy(),
                                      sgns.exp_table(), len(vocab), base + rep * rows, m.seed, dim, window, K, alpha,
                                      batched=m.batched)
    torch.cuda.synchronize()
    ok = np.array_equal(m.syn0.cpu().numpy(), s0) and np.array_equal(m.syn1neg.cpu().numpy(), s1)
    n += 1; tot_pairs += pairs
    if not ok:
        print("MISMATCH", dict(mode=mode, n_tok=n_tok, rows=rows, ln=ln, dim=dim, window=window, K=K, sample=sample, alpha=alpha))
        sys.exit(1)
print(f"sgns fuzz ok: {n} cases bit-identical to the oracle ({tot_pairs} pairs in the last reps) in {time.time()-t0:.0f} s")
