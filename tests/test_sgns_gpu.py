"""GPU parity tests for K3 (n2v_sgns_train) through the C ABI.

Deterministic mode (one wave, sentences in order) must equal the CPU restatement
oracle/n2v_oracle_sgns.c BIT FOR BIT (the oracle sums dot products in wave64 order).
Hogwild mode is checked through order-independent properties.  Note the oracle is
"parity unpinned" against gensim itself: no gensim here, and the
reference's tests assert shapes only.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# TOLERANCES of the hogwild (racy) tests, from 30 runs of every statistic on one MI355X
# (scripts/r3/stat_runs.py -> profiles/r3ad_stat_runs.log) and a 400-run search for rare outliers
# (scripts/r3/stat_outliers.py -> profiles/r3ad_stat_outliers.log).  Each bound is looser than
# mean -+ 5 sd of the sample it comes from:
#   two cliques: gap 0.8727 in all 30 runs (= the oracle's; 40 words -> one wave), bounds 0.2 / 0.15
#   planted partition: AUC 1.0000 +- 0.0000 hogwild, 0.9994 deterministic -> > 0.99, |diff| < 0.01
#   Procrustes cosine of the row directions: hogwild vs hogwild 0.9588 +- 0.0010, deterministic vs
#     hogwild 0.7764 +- 0.0013 -- BUT one hogwild run in 400 (and one pair in ~8 suite runs) lands
#     at 0.80 from the others, the distance of the deterministic order, with AUC 1.0000 and
#     ordinary row norms: a different, equally good solution (sentences are handed out from a
#     counter, so a scheduling hiccup changes the order at large).  The cosine between two runs is
#     therefore only held above the level of "another order": 0.6 for both comparisons.
#   500-word vocabulary, |syn0| hogwild / serial: 0.990 +- 0.009 default, 0.998 +- 0.001 with
#     hub_rows = 500 -> (0.9, 1.1) and (0.95, 1.05)
GAP_MIN, GAP_DIFF_MAX = 0.2, 0.15
PLANTED_AUC_MIN, PLANTED_AUC_DIFF_MAX = 0.99, 0.01
PLANTED_COS_HH_MIN, PLANTED_COS_DH_MIN = 0.6, 0.6
NORM_RATIO_DEFAULT, NORM_RATIO_HUB = (0.9, 1.1), (0.95, 1.05)


def _setup(n_tok, rows, ln, dim, seed, sample, min_count=1, oov=False):
    from node2vec_amd import sgns

    gen = torch.Generator().manual_seed(seed)
    # zipf-ish token frequencies
    p = 1.0 / torch.arange(1, n_tok + 1, dtype=torch.float64)
    walks = torch.multinomial(p, rows * ln, replacement=True, generator=gen).reshape(rows, ln).to(torch.int32)
    walks = walks.cuda()
    vocab = sgns.build_vocab(walks, min_count)
    m = sgns.SgnsModel(vocab, dim, 5, 5, seed=seed, sample=sample)
    idx = vocab.index_of[walks.long()]
    if oov:
        assert int((idx < 0).sum()) > 0
    return sgns, m, idx


@pytest.mark.parametrize("dim", [16, 64, 100, 128, 256, 512])
@pytest.mark.parametrize("sample", [0.0, 1e-2])
def test_deterministic_mode_bit_identical_to_oracle(oracle, dim, sample):
    sgns, m, idx = _setup(60, 40, 21, dim, seed=5 + dim, sample=sample)
    s0, s1 = m.syn0.cpu().numpy().copy(), m.syn1neg.cpu().numpy().copy()
    for blk, alpha in ((0, 0.025), (1, 0.02)):  # two launches: sentence_base moves on
        m.train_block(idx, alpha, blk * idx.shape[0], deterministic=True)
        n = oracle.sgns_train(idx.cpu().numpy(), s0, s1, m.cum_table.cpu().numpy(),
                              None if m.sample_int is None else m.sample_int.cpu().numpy(),
                              sgns.exp_table(), len(m.vocab), blk * idx.shape[0], m.seed, dim, 5, 5, alpha)
    torch.cuda.synchronize()
    assert n > 0
    assert np.array_equal(m.syn0.cpu().numpy(), s0)
    assert np.array_equal(m.syn1neg.cpu().numpy(), s1)
    assert np.abs(s1).max() > 0


def test_oov_tokens_and_long_window(oracle):
    """min_count drops rare tokens: they are removed BEFORE windowing; window 30, k 20"""
    sgns, m, idx = _setup(200, 30, 40, 32, seed=9, sample=1e-3, min_count=4, oov=True)
    m.window, m.negative = 30, 20
    s0, s1 = m.syn0.cpu().numpy().copy(), m.syn1neg.cpu().numpy().copy()
    m.train_block(idx, 0.025, 7, deterministic=True)
    n = oracle.sgns_train(idx.cpu().numpy(), s0, s1, m.cum_table.cpu().numpy(),
                          m.sample_int.cpu().numpy(), sgns.exp_table(), len(m.vocab), 7, m.seed,
                          32, 30, 20, 0.025)
    torch.cuda.synchronize()
    assert int(m.pairs.item()) == n
    assert np.array_equal(m.syn0.cpu().numpy(), s0) and np.array_equal(m.syn1neg.cpu().numpy(), s1)


def test_hogwild_pair_count_equals_oracle(oracle):
    """the set of trained pairs does not depend on launch geometry"""
    sgns, m, idx = _setup(500, 3000, 41, 128, seed=1, sample=1e-3)
    s0, s1 = m.syn0.cpu().numpy().copy(), m.syn1neg.cpu().numpy().copy()
    m.train_block(idx, 0.025, 0, deterministic=False)
    torch.cuda.synchronize()
    n = oracle.sgns_train(idx.cpu().numpy(), s0, s1, m.cum_table.cpu().numpy(),
                          m.sample_int.cpu().numpy(), sgns.exp_table(), len(m.vocab), 0, m.seed,
                          128, 5, 5, 0.025)
    assert int(m.pairs.item()) == n
    # hogwild races on a 500-row vocabulary make the vectors differ from the serial
    # order; what must hold: finite values, every touched row moved, same scale
    got = m.syn0.cpu().numpy()
    assert np.isfinite(got).all() and np.isfinite(m.syn1neg.cpu().numpy()).all()
    ratio = float(np.linalg.norm(got) / np.linalg.norm(s0))
    assert NORM_RATIO_DEFAULT[0] < ratio < NORM_RATIO_DEFAULT[1], ratio


def two_clique_case():
    """two 20-vertex cliques joined by one edge (shared with scripts/r3/stat_runs.py)"""
    from node2vec_amd import randomwalk as rw
    from node2vec_amd import sgns
    from node2vec_amd.graph import DeviceGraph

    src, dst = [], []
    for c in (0, 20):
        for a in range(c, c + 20):
            for b in range(c, c + 20):
                if a != b:
                    src.append(a)
                    dst.append(b)
    src += [0, 20]
    dst += [20, 0]
    g = DeviceGraph.from_edges(src, dst, np.ones(len(src), np.float32), device="cuda")
    walks, valid = rw.walk(g, rw.start_vertices(g), 20, 30, 1.0, 1.0, 3)
    vocab = sgns.build_vocab(walks, 1)
    idx = vocab.index_of[walks.long()]
    ids = vocab.ids.cpu().numpy()

    def gap(vecs):
        v = vecs / np.linalg.norm(vecs, axis=1, keepdims=True)
        side = ids < 20
        s = v @ v.T
        intra = (s[np.ix_(side, side)].mean() + s[np.ix_(~side, ~side)].mean()) / 2
        return float(intra - s[np.ix_(side, ~side)].mean())

    def gap_gpu():
        m = sgns.SgnsModel(vocab, 32, 5, 5, seed=2, sample=0.0)
        m.train(idx, epochs=3, alpha=0.025)
        torch.cuda.synchronize()
        return gap(m.syn0.cpu().numpy())

    def gap_cpu(oracle):
        m = sgns.SgnsModel(vocab, 32, 5, 5, seed=2, sample=0.0)
        s0, s1 = m.syn0.cpu().numpy().copy(), m.syn1neg.cpu().numpy().copy()
        rows = idx.shape[0]
        for ep in range(3):  # same schedule on the oracle (one block per epoch here)
            a = max(1e-4, 0.025 - (0.025 - 1e-4) * (ep * rows / (3 * rows)))
            oracle.sgns_train(idx.cpu().numpy(), s0, s1, m.cum_table.cpu().numpy(), None,
                              sgns.exp_table(), len(vocab), ep * rows, m.seed, 32, 5, 5, a)
        return gap(s0)

    return {"gap_gpu": gap_gpu, "gap_cpu": gap_cpu}


@pytest.mark.statistical
def test_hogwild_learns_community_structure(oracle):
    """two 20-vertex cliques joined by one edge: after training, vectors of the
    same clique are closer than vectors of different cliques (GPU and oracle).
    Tolerances: TOLERANCES below."""
    case = two_clique_case()
    g_gpu, g_cpu = case["gap_gpu"](), case["gap_cpu"](oracle)
    assert g_gpu > GAP_MIN and g_cpu > GAP_MIN, (g_gpu, g_cpu)
    assert abs(g_gpu - g_cpu) < GAP_DIFF_MAX, (g_gpu, g_cpu)


def planted_case():
    """planted partition, 50 communities x 40 vertices (shared with scripts/r3/stat_runs.py)"""
    from node2vec_amd import randomwalk as rw
    from node2vec_amd import sgns
    from node2vec_amd.graph import DeviceGraph

    rng = np.random.default_rng(0)
    nc, sz = 50, 40
    nv = nc * sz
    comm = np.repeat(np.arange(nc), sz)
    src, dst = [], []
    for v in range(nv):
        for u in list(rng.choice(np.nonzero(comm == comm[v])[0], 8)) + list(rng.integers(0, nv, 2)):
            if u != v:
                src += [v, u]
                dst += [u, v]
    g = DeviceGraph.from_edges(src, dst, np.ones(len(src), np.float32), n_vertices=nv, device="cuda")
    walks, _ = rw.walk(g, rw.start_vertices(g), 10, 40, 1.0, 1.0, 1)
    vocab = sgns.build_vocab(walks, 1)
    idx = vocab.index_of[walks.long()]
    ids = vocab.ids.cpu().numpy()
    prng = np.random.default_rng(1)
    a, b = prng.integers(0, len(ids), 100000), prng.integers(0, len(ids), 100000)
    same = comm[ids[a]] == comm[ids[b]]

    def train(det):
        m = sgns.SgnsModel(vocab, 64, 5, 5, seed=7, sample=0.0)
        m.train(idx, epochs=3, alpha=0.025, deterministic=det)
        torch.cuda.synchronize()
        return m.syn0.cpu().numpy()

    def auc(v):
        v = v - v.mean(0)
        v = v / np.linalg.norm(v, axis=1, keepdims=True)
        s = (v[a] * v[b]).sum(1)
        return float((s[same][:, None] > s[~same][None, :2000]).mean())

    def procrustes_raw(x, y):
        x, y = x - x.mean(0), y - y.mean(0)
        u, _, vt = np.linalg.svd(x.T @ y)
        xr = x @ (u @ vt)
        return float(np.mean((xr * y).sum(1) / (np.linalg.norm(xr, axis=1) * np.linalg.norm(y, axis=1))))

    def procrustes(x, y):
        """rows scaled to unit length first: the rotation is then fitted to directions, and a
        few long rows (the race amplifies some norms) cannot dominate it"""
        x, y = x - x.mean(0), y - y.mean(0)
        x = x / np.linalg.norm(x, axis=1, keepdims=True)
        y = y / np.linalg.norm(y, axis=1, keepdims=True)
        return procrustes_raw(x, y)

    return {"train": train, "auc": auc, "procrustes": procrustes, "procrustes_raw": procrustes_raw}


@pytest.mark.statistical
def test_hogwild_vs_deterministic_statistical_parity():
    """Full-speed (hogwild) mode cannot be bit-compared; the claim is statistical
    (SURVEY.md 8c): on a planted-partition graph (50 communities x 40 vertices) the
    community-separation AUC of the hogwild embedding equals the deterministic one and, after
    Procrustes alignment of the row directions, two hogwild runs agree with each other and with
    the deterministic run.  Tolerances: TOLERANCES below."""
    case = planted_case()
    det, h1, h2 = case["train"](True), case["train"](False), case["train"](False)
    a_det, a_h1, a_h2 = case["auc"](det), case["auc"](h1), case["auc"](h2)
    assert a_det > PLANTED_AUC_MIN and min(a_h1, a_h2) > PLANTED_AUC_MIN, (a_det, a_h1, a_h2)
    assert max(abs(a_det - a_h1), abs(a_det - a_h2)) < PLANTED_AUC_DIFF_MAX, (a_det, a_h1, a_h2)
    assert case["procrustes"](h1, h2) > PLANTED_COS_HH_MIN
    assert case["procrustes"](det, h1) > PLANTED_COS_DH_MIN


def test_cum_index_is_bisect_left_and_changes_no_draw(oracle):
    """n2v_cum_index_build: index[b] == bisect_left(cum_table, b << (31 - bits)) for every bucket
    (numpy searchsorted on the same table), and training through it is bit-identical to training
    through the LDS buckets (and to the oracle, which bisects the whole table) on a vocabulary
    large enough for buckets to hold many words"""
    from node2vec_amd import sgns

    gen = torch.Generator().manual_seed(8)
    n_tok = 200_000
    p = 1.0 / torch.arange(1, n_tok + 1, dtype=torch.float64) ** 0.9
    walks = torch.multinomial(p, 3000 * 41, replacement=True, generator=gen).reshape(3000, 41).to(torch.int32).cuda()
    vocab = sgns.build_vocab(walks, 1)
    a = sgns.SgnsModel(vocab, 64, 5, 5, seed=3)
    b = sgns.SgnsModel(vocab, 64, 5, 5, seed=3, use_cum_index=False)
    assert a.cum_index is not None and b.cum_index is None
    cum = a.cum_table.cpu().numpy().astype(np.uint32)
    bits = a.cum_index_bits
    edges = (np.arange((1 << bits) + 1, dtype=np.uint64) << np.uint64(31 - bits))
    want = np.searchsorted(cum, np.minimum(edges, 2 ** 32 - 1).astype(np.uint32), side="left")
    want[edges > 0x7fffffff] = len(cum)
    assert np.array_equal(a.cum_index.cpu().numpy(), want.astype(np.int32))
    idx = vocab.index_of[walks.long()]
    s0, s1 = a.syn0.cpu().numpy().copy(), a.syn1neg.cpu().numpy().copy()
    a.train_block(idx[:64], 0.025, 5, deterministic=True)
    b.train_block(idx[:64], 0.025, 5, deterministic=True)
    torch.cuda.synchronize()
    assert torch.equal(a.syn0, b.syn0) and torch.equal(a.syn1neg, b.syn1neg)
    n = oracle.sgns_train(idx[:64].cpu().numpy(), s0, s1, cum, None, sgns.exp_table(), len(vocab), 5, 3,
                          64, 5, 5, 0.025)
    assert n == int(a.pairs.item()) == int(b.pairs.item())
    assert np.array_equal(a.syn0.cpu().numpy(), s0) and np.array_equal(a.syn1neg.cpu().numpy(), s1)


@pytest.mark.parametrize("dim,window", [(64, 5), (128, 5), (128, 7), (128, 2)])
def test_window_cache_changes_no_bit(oracle, dim, window):
    """the default kernel with the syn0 rows of the window kept in an LDS ring (window_cache = 1:
    one read and one write per position instead of per pair) against the same kernel without it
    (0) and against the oracle, on walk-like rows where the window holds the same word at
    several positions (shared ring rows): identical bits in deterministic mode"""
    from node2vec_amd import sgns

    gen = torch.Generator().manual_seed(dim + window)
    steps = torch.randint(-2, 3, (48, 33), generator=gen)
    walks = (torch.cumsum(steps, 1) + torch.randint(0, 70, (48, 1), generator=gen)) % 70
    back = torch.rand((48, 33), generator=gen) < 0.3
    walks[:, 2:] = torch.where(back[:, 2:], walks[:, :-2], walks[:, 2:])
    walks = walks.to(torch.int32).cuda()
    walks[5, 1:] = -1  # a one-token sentence
    walks[6, :] = -1   # an empty one
    vocab = sgns.build_vocab(walks, 1)
    idx = torch.where(walks >= 0, vocab.index_of[walks.clamp(min=0).long()], torch.full_like(walks, -1))
    out = {}
    for mode in (1, 0):
        m = sgns.SgnsModel(vocab, dim, window, 5, seed=3, sample=1e-2)
        m.window_cache = mode
        s0, s1 = m.syn0.cpu().numpy().copy(), m.syn1neg.cpu().numpy().copy()
        for blk, alpha in ((0, 0.025), (1, 0.02)):
            m.train_block(idx, alpha, blk * idx.shape[0], deterministic=True)
        torch.cuda.synchronize()
        out[mode] = (m.syn0.cpu().numpy(), m.syn1neg.cpu().numpy(), int(m.pairs.item()))
    n = 0
    for blk, alpha in ((0, 0.025), (1, 0.02)):
        n += oracle.sgns_train(idx.cpu().numpy(), s0, s1, m.cum_table.cpu().numpy(),
                               m.sample_int.cpu().numpy(), sgns.exp_table(), len(vocab),
                               blk * idx.shape[0], m.seed, dim, window, 5, alpha)
    assert out[1][2] == out[0][2] == n
    assert np.array_equal(out[1][0], out[0][0]) and np.array_equal(out[1][1], out[0][1])
    assert np.array_equal(out[1][0], s0) and np.array_equal(out[1][1], s1)


def test_window_cache_is_refused_where_it_does_not_fit():
    from node2vec_amd import sgns

    walks = torch.randint(0, 30, (8, 12), dtype=torch.int32).cuda()
    vocab = sgns.build_vocab(walks, 1)
    for dim, window in ((100, 5), (256, 5), (128, 8)):
        m = sgns.SgnsModel(vocab, dim, window, 5, seed=1)
        m.window_cache = 1
        with pytest.raises(ValueError):
            m.train_block(vocab.index_of[walks.long()], 0.025, 0)
        m.window_cache = 0
        m.train_block(vocab.index_of[walks.long()], 0.025, 0)
    torch.cuda.synchronize()


def test_hub_rows_atomic_updates(oracle):
    """hub_rows > 0 (hogwild only): the top rows receive atomic adds instead of read-modify-write
    stores.  Deterministic mode ignores it (bit-identical to the oracle); in hogwild mode the pairs
    trained are the same and, on a 500-row vocabulary where every row is contended, the scale of
    the vectors stays that of the serial run"""
    sgns, m, idx = _setup(500, 3000, 41, 128, seed=1, sample=1e-3)
    m.hub_rows = 500
    s0, s1 = m.syn0.cpu().numpy().copy(), m.syn1neg.cpu().numpy().copy()
    n = oracle.sgns_train(idx.cpu().numpy(), s0, s1, m.cum_table.cpu().numpy(),
                          m.sample_int.cpu().numpy(), sgns.exp_table(), len(m.vocab), 0, m.seed,
                          128, 5, 5, 0.025)
    keep0, keep1 = m.syn0.clone(), m.syn1neg.clone()
    m.train_block(idx, 0.025, 0, deterministic=True)
    torch.cuda.synchronize()
    assert np.array_equal(m.syn0.cpu().numpy(), s0) and np.array_equal(m.syn1neg.cpu().numpy(), s1)
    m.syn0.copy_(keep0)
    m.syn1neg.copy_(keep1)
    m.pairs.zero_()
    m.train_block(idx, 0.025, 0)
    torch.cuda.synchronize()
    assert int(m.pairs.item()) == n
    got = m.syn0.cpu().numpy()
    assert np.isfinite(got).all() and np.isfinite(m.syn1neg.cpu().numpy()).all()
    ratio = float(np.linalg.norm(got) / np.linalg.norm(s0))
    print("hub_rows hogwild / serial norm of syn0:", ratio)
    assert NORM_RATIO_HUB[0] < ratio < NORM_RATIO_HUB[1], ratio


@pytest.mark.parametrize("batched", [False, True])
def test_gensim_job_schedule_in_the_kernel_equals_the_oracle_job_by_job(oracle, batched):
    """n2v_sgns_params.row_alpha / n2v_sgns_job_alpha: the learning rate of a row is that of its gensim JOB (word2vec.py
    _job_producer / _get_next_alpha: max(end, start - (start - end) * (epoch + pushed / total) /
    epochs) per batch of batch_words words).  Deterministic mode with the schedule == the oracle
    called job by job with the rate restated in plain Python floats, bit for bit -- both kernels,
    launches that start in the middle of a job, two epochs."""
    from node2vec_amd import sgns

    rng = np.random.default_rng(11)
    n_vocab, rows, length, dim = 300, 230, 21, 64
    walks = rng.integers(0, n_vocab, (rows, length)).astype(np.int32)
    walks[rng.random((rows, length)) < 0.05] = -1
    counts = np.maximum(np.bincount(walks[walks >= 0], minlength=n_vocab), 1)
    order = np.argsort(-counts, kind="stable")
    vocab = sgns.Vocab(torch.arange(n_vocab).cuda(), torch.from_numpy(counts[order]).cuda(),
                       torch.arange(n_vocab, dtype=torch.int32).cuda())
    m = sgns.SgnsModel(vocab, dim, 5, 5, seed=3, sample=0.0)
    m.batched = batched
    s0, s1 = m.syn0.cpu().numpy().copy(), m.syn1neg.cpu().numpy().copy()
    idx = torch.from_numpy(walks).cuda()
    batch_words, epochs, a0, amin = 100, 2, 0.025, 1e-4
    job_rows = batch_words // length  # 4 sentences per job
    m.train(idx, epochs, a0, amin, block_rows=70, deterministic=True, batch_words=batch_words)
    torch.cuda.synchronize()
    cum = m.cum_table.cpu().numpy().astype(np.uint32)
    pairs = 0
    for ep in range(epochs):
        for j0 in range(0, rows, job_rows):
            progress = (ep + 1.0 * j0 / rows) / epochs
            alpha = float(np.float32(max(amin, a0 - (a0 - amin) * progress)))
            pairs += oracle.sgns_train(walks[j0:j0 + job_rows], s0, s1, cum, None, sgns.exp_table(), n_vocab,
                                       ep * rows + j0, 3, dim, 5, 5, alpha, batched=batched)
    assert pairs == int(m.pairs.item()) > 0
    assert np.array_equal(m.syn0.cpu().numpy(), s0) and np.array_equal(m.syn1neg.cpu().numpy(), s1)
    # and the host restatement the tests of other modules use
    sch = sgns.JobSchedule.for_corpus(batch_words, length, rows, 1, epochs, a0, amin)
    want = [np.float32(max(amin, a0 - (a0 - amin) * ((1 + 1.0 * ((r // job_rows) * job_rows) / rows) / epochs)))
            for r in range(rows)]
    assert sch.job_rows == job_rows and np.array_equal(sch.alpha_of_rows(0, rows), np.array(want, np.float32))


def test_job_alpha_entry_point_equals_python_floats():
    """n2v_sgns_job_alpha == gensim's expression in Python floats, bit for bit, at the row counts of
    configuration 4 (jobs of 12 sentences in a 1.13e9-sentence corpus, launches starting mid-job)
    and at the clamp to min_alpha; bad arguments are refused."""
    from node2vec_amd import _lib, sgns

    L = _lib.load()
    for job_rows, epoch, epochs, row0, rows, a0, amin, n in (
            (12, 0, 1, 1_130_000_000 - 70_001, 1_130_000_000, 0.025, 1e-4, 70_001),
            (12, 2, 5, 999_999_937, 1_130_000_000, 0.025, 1e-4, 4099),
            (1, 0, 1, 0, 7, 0.025, 0.025, 7),
            (47, 0, 3, 5, 100_000, 0.05, 0.04, 99_995)):
        out = torch.empty(n, dtype=torch.float32, device="cuda")
        _lib.check(L.n2v_sgns_job_alpha(job_rows, epoch, epochs, row0, rows, a0, amin, n, out.data_ptr(),
                                        _lib.current_stream_ptr()), "n2v_sgns_job_alpha")
        got = out.cpu().numpy()
        assert np.array_equal(got, sgns.JobSchedule(job_rows, rows, epoch, epochs, a0, amin).alpha_of_rows(row0, n))
        for i in (0, 1, n // 2, n - 1):  # plain Python floats, as gensim computes them
            pushed = ((row0 + i) // job_rows) * job_rows
            want = max(amin, a0 - (a0 - amin) * ((epoch + 1.0 * pushed / rows) / epochs))
            assert got[i] == np.float32(want)
    out = torch.empty(4, dtype=torch.float32, device="cuda")
    for bad in ((0, 0, 1, 0, 4, 0.1, 0.01, 4), (1, 0, 0, 0, 4, 0.1, 0.01, 4), (1, 0, 1, -1, 4, 0.1, 0.01, 4),
                (1, 0, 1, 0, 0, 0.1, 0.01, 4)):
        assert L.n2v_sgns_job_alpha(*bad, out.data_ptr(), _lib.current_stream_ptr()) == _lib.EINVAL
