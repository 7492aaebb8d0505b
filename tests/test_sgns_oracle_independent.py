"""Independent check of the SGNS oracle (oracle/n2v_oracle_sgns.c).

The C oracle sums its dot products in the wave64 tree order the kernel uses, so that
the deterministic GPU mode can be compared bit for bit.  To show that this choice is
immaterial and that the oracle really is the published algorithm, this test restates
word2vec's skip-gram negative-sampling update a second time -- plain Python/numpy,
float64, ordinary sequential dot products, written from the algorithm description
(DESIGN.md "SGNS"), sharing only the counter-based draw function -- and requires the
two to agree within fp32 rounding."""
import numpy as np
import torch

M64 = (1 << 64) - 1


def mix64(z):
    z &= M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def reference_sgns(walks, syn0, syn1, cum, sample_int, table, seed, base, window, k, alpha):
    """float64 restatement: filter -> reduced windows -> (centre, context) pairs ->
    label-1 target + k negatives via bisect_left on the cumulative table."""
    syn0, syn1 = syn0.astype(np.float64), syn1.astype(np.float64)
    n_vocab, ln = len(cum), walks.shape[1]
    pairs = 0
    for r in range(walks.shape[0]):
        hs = mix64(seed ^ mix64(base + r + 0xA0761D6478BD642F))
        draw = lambda idx: mix64(hs + (idx + 1) * 0xE7037ED1A0B428DB)  # noqa: E731
        sent, red = [], []
        for t in range(ln):
            tok = int(walks[r, t])
            if tok < 0 or tok >= n_vocab:
                continue
            if sample_int is not None and int(sample_int[tok]) < (draw(2 * t) >> 32):
                continue
            sent.append(tok)
            red.append((draw(2 * t + 1) >> 32) % window)
        for i, centre in enumerate(sent):
            lo, hi = max(0, i - window + red[i]), min(len(sent), i + window + 1 - red[i])
            for j in range(lo, hi):
                if j == i:
                    continue
                rel = j - i + window - (1 if j > i else 0)
                ctx = sent[j]
                work = np.zeros(syn0.shape[1])
                for d in range(k + 1):
                    if d == 0:
                        target, label = centre, 1.0
                    else:
                        idx = 2 * ln + ((i * 2 * window + rel) * k + (d - 1))
                        target = int(np.searchsorted(cum, (draw(idx) >> 16) % int(cum[-1]), side="left"))
                        if target == centre:
                            continue
                        label = 0.0
                    f = float(syn0[ctx] @ syn1[target])
                    if f <= -6.0 or f >= 6.0:
                        continue
                    g = (label - float(table[int((f + 6.0) * 83.0)])) * alpha
                    work += g * syn1[target]
                    syn1[target] += g * syn0[ctx]
                syn0[ctx] += work
                pairs += 1
    return syn0, syn1, pairs


def test_c_oracle_matches_independent_float64_restatement(oracle):
    from node2vec_amd import sgns

    rng = np.random.default_rng(4)
    for dim, window, k, sample in ((16, 5, 5, 0.0), (48, 3, 7, 1e-2), (128, 5, 5, 1e-3)):
        walks = torch.from_numpy(rng.integers(0, 30, size=(12, 15)).astype(np.int32))
        walks[rng.random((12, 15)) < 0.1] = -1
        vocab = sgns.build_vocab(walks, 1)
        idx = torch.where(walks >= 0, vocab.index_of[walks.clamp(min=0).long()],
                          torch.full_like(walks, -1)).numpy()
        cum = sgns.make_cum_table(vocab.counts).numpy().view(np.uint32)
        si = sgns.make_sample_int(vocab.counts, sample)
        si_np = None if si is None else si.numpy().view(np.uint32)
        s0 = sgns.init_syn0(len(vocab), dim, 7, "cpu").numpy()
        s1 = (rng.normal(size=s0.shape) * 0.05).astype(np.float32)  # non-zero outputs: f matters
        table = sgns.exp_table()
        r0, r1, rp = reference_sgns(idx, s0, s1, cum, si_np, table, 99, 1000, window, k, 0.05)
        c0, c1 = s0.copy(), s1.copy()
        cp = oracle.sgns_train(idx, c0, c1, cum, si_np, table, len(vocab), 1000, 99, dim, window, k, 0.05)
        assert cp == rp and cp > 20
        # fp32 vs fp64 arithmetic and a different summation order: rounding-level agreement
        np.testing.assert_allclose(c0, r0, rtol=2e-4, atol=2e-6)
        np.testing.assert_allclose(c1, r1, rtol=2e-4, atol=2e-6)
        assert np.abs(c0 - s0).max() > 1e-4  # and the pass did change the vectors
