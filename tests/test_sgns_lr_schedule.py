"""gensim lowers the learning rate once per JOB (batch_words words of the corpus).  Since round 4
the HIP trainer does the same when it is given batch_words (n2v_sgns_job_alpha,
sgns.JobSchedule: Node2VecHIP.fit and fit_streaming pass the reference's value, constants.py:58);
called without it (SgnsModel.train(batch_words=None), a corpus of split rows) the rate falls once
per LAUNCH (max(65 536 rows, rows / 64) per launch), which is what this file quantifies on the
oracle (CPU, no GPU).  Both are the same linear ramp sampled at different granularity.  On a planted-partition corpus trained
with the oracle under (a) a per-launch ramp of 8 launches per epoch -- coarser than any real
run -- and (b) a per-job ramp (10 000 words), the two embeddings agree: Procrustes cosine >= 0.97,
community AUC equal within 0.01."""
import numpy as np


def test_per_launch_and_per_job_learning_rate_ramps_agree(oracle):
    rng = np.random.default_rng(0)
    nc, sz, L = 12, 25, 21
    nv = nc * sz
    comm = np.repeat(np.arange(nc), sz)
    nbrs = [np.concatenate([rng.choice(np.nonzero(comm == comm[v])[0], 6), rng.integers(0, nv, 2)]) for v in range(nv)]
    walks = np.zeros((nv * 6, L), np.int32)
    for r in range(len(walks)):
        v = r % nv
        for t in range(L):
            walks[r, t] = v
            v = int(rng.choice(nbrs[v]))
    counts = np.bincount(walks.reshape(-1), minlength=nv).astype(np.float64)
    order = np.argsort(-counts, kind="stable")
    index_of = np.empty(nv, np.int32)
    index_of[order] = np.arange(nv, dtype=np.int32)
    idx = index_of[walks]
    p = counts[order] ** 0.75
    cum = np.round(np.cumsum(p) / p.sum() * (2 ** 31 - 1)).astype(np.int64)
    cum[-1] = 2 ** 31 - 1
    cum = cum.astype(np.uint32)
    x = (np.arange(1000, dtype=np.float32) / np.float32(1000) * np.float32(2) - np.float32(1)) * np.float32(6)
    e = np.exp(x.astype(np.float64)).astype(np.float32)
    exp_table = (e / (e + np.float32(1))).astype(np.float32)
    dim, epochs, a0, a1 = 32, 5, 0.025, 1e-4
    init = ((np.random.default_rng(1).random((nv, dim), dtype=np.float32) - 0.5) / dim).astype(np.float32)

    def train(rows_per_step):
        s0, s1 = init.copy(), np.zeros((nv, dim), np.float32)
        rows = len(idx)
        total, done = rows * epochs, 0
        for ep in range(epochs):
            for lo in range(0, rows, rows_per_step):
                hi = min(rows, lo + rows_per_step)
                a = max(a1, a0 - (a0 - a1) * done / total)
                oracle.sgns_train(idx[lo:hi], s0, s1, cum, None, exp_table, nv, ep * rows + lo, 7, dim, 5, 5, a)
                done += hi - lo
        return s0[index_of]  # row = vertex id

    per_launch = train(-(-len(idx) // 8))           # 8 launches per epoch
    per_job = train(max(1, 10_000 // L))            # gensim's job: 10 000 words
    x, y = per_launch - per_launch.mean(0), per_job - per_job.mean(0)
    u, _, vt = np.linalg.svd(x.T @ y)
    xr = x @ (u @ vt)
    cos = float(np.mean((xr * y).sum(1) / (np.linalg.norm(xr, axis=1) * np.linalg.norm(y, axis=1))))

    def auc(v):
        v = v - v.mean(0)
        v = v / np.linalg.norm(v, axis=1, keepdims=True)
        a, b = rng.integers(0, nv, 40000), rng.integers(0, nv, 40000)
        s = (v[a] * v[b]).sum(1)
        same = comm[a] == comm[b]
        return float((s[same][:, None] > s[~same][None, :1500]).mean())

    a_l, a_j = auc(per_launch), auc(per_job)
    print("procrustes cosine", cos, "AUC per-launch", a_l, "per-job", a_j)
    assert cos >= 0.97
    assert a_l > 0.9 and a_j > 0.9 and abs(a_l - a_j) <= 0.01


def test_job_schedule_is_gensims_expression():
    """sgns.JobSchedule.alpha_of_rows == word2vec.py _get_next_alpha evaluated at the sentences
    pushed before each job, in plain Python floats; jobs of batch_words // sentence length rows"""
    from node2vec_amd.sgns import JobSchedule

    for batch_words, length, rows, epochs in ((1000, 81, 5000, 10), (10_000, 21, 777, 1), (50, 81, 40, 3)):
        k = max(1, batch_words // length)
        for ep in range(epochs):
            sch = JobSchedule.for_corpus(batch_words, length, rows, ep, epochs, 0.025, 1e-4)
            got = sch.alpha_of_rows(3, rows - 3)
            for r in (3, 4, k - 1, k, k + 1, rows // 2, rows - 1):
                if r < 3 or r >= rows:
                    continue
                pushed = (r // k) * k
                progress = (ep + 1.0 * pushed / rows) / epochs
                want = np.float32(max(1e-4, 0.025 - (0.025 - 1e-4) * progress))
                assert got[r - 3] == want, (batch_words, length, ep, r)
            assert (np.diff(got.astype(np.float64)) <= 0).all()  # never rises inside an epoch
