"""Skip-gram negative-sampling trainer: host logic around the K3 kernel (n2v_sgns_train).

What gensim.models.Word2Vec(sentences, sg=1, hs=0, negative=k) does around its
Cython inner loop (the reference's call site is embedding.py:126; gensim itself is
third-party and absent from the reference tree -- DESIGN.md "SGNS" restates it):
vocabulary with min_count, descending-count order, frequent-word subsampling
thresholds, the cumulative count^0.75 table, (rand - 0.5) / dim initialisation,
linear learning-rate decay, `iter` epochs.  Tensors live on the GPU; torch is
plumbing (sort / unique / gather), the arithmetic of training is the HIP kernel.
"""
import math
from typing import Optional

import numpy as np
import torch

from node2vec_amd import _lib
from node2vec_amd.shard import all_reduce, ordered_sum, ordered_sum_shard

EXP_TABLE_SIZE = 1000
MAX_EXP = 6
CUM_DOMAIN = 2 ** 31 - 1
MAX_SENTENCE = 256  # N2V_SGNS_MAX_SENTENCE


def exp_table() -> np.ndarray:
    """word2vec's EXP_TABLE: sigma(x) at x = (i / 1000 * 2 - 1) * 6, fp32."""
    i = np.arange(EXP_TABLE_SIZE, dtype=np.float32)
    x = (i / np.float32(EXP_TABLE_SIZE) * np.float32(2) - np.float32(1)) * np.float32(MAX_EXP)
    e = np.exp(x.astype(np.float64)).astype(np.float32)
    return (e / (e + np.float32(1))).astype(np.float32)


class Vocab:
    """index -> vertex id (descending count, ties by ascending id), counts, and the
    dense lookup vertex id -> index (-1 = below min_count)."""

    def __init__(self, ids: torch.Tensor, counts: torch.Tensor, index_of: torch.Tensor):
        self.ids, self.counts, self.index_of = ids, counts, index_of

    def __len__(self):
        return self.ids.numel()


def build_vocab(walks: torch.Tensor, min_count: int, group=None) -> Vocab:
    """Vocabulary of the walk corpus.  Under torch.distributed every rank holds only its
    shard of the walks, so token counts are summed over the ranks first (one all-reduce of
    a dense count vector): all ranks then build the SAME index, which the replicated model
    and its delta all-reduce rely on."""
    flat = walks.reshape(-1)
    flat = flat[flat >= 0].long()
    import torch.distributed as dist

    n_ids = int(flat.max()) + 1 if flat.numel() else 0  # size of the id -> index lookup

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        size = torch.tensor([int(flat.max()) + 1 if flat.numel() else 0], device=walks.device)
        all_reduce(size, dist.ReduceOp.MAX, group)
        n_ids = int(size.item())  # the same on every rank
        dense = torch.bincount(flat, minlength=n_ids)
        all_reduce(dense, dist.ReduceOp.SUM, group)
        ids = torch.nonzero(dense).reshape(-1)
        counts = dense[ids]
    else:
        ids, counts = torch.unique(flat, return_counts=True)  # ids ascending
    keep = counts >= max(int(min_count), 0)
    ids, counts = ids[keep], counts[keep]
    order = torch.sort(counts, descending=True, stable=True).indices
    ids, counts = ids[order], counts[order]
    index_of = torch.full((n_ids,), -1, dtype=torch.int32, device=walks.device)
    index_of[ids] = torch.arange(ids.numel(), dtype=torch.int32, device=walks.device)
    return Vocab(ids, counts, index_of)


def make_cum_table(counts: torch.Tensor, ns_exponent: float = 0.75) -> torch.Tensor:
    """cum_table[i] = round(sum_{j<=i} count_j^p / sum count^p * (2^31 - 1)), uint32
    stored in an int64->int32-compatible tensor (values < 2^31)."""
    p = counts.double() ** ns_exponent
    cum = torch.cumsum(p, 0) / p.sum() * CUM_DOMAIN
    tab = torch.round(cum).to(torch.int64)
    if tab.numel():
        tab[-1] = CUM_DOMAIN
    return tab.to(torch.int32)  # < 2^31: same bits as uint32


def make_sample_int(counts: torch.Tensor, sample: float) -> Optional[torch.Tensor]:
    """keep-threshold per word: keep iff sample_int >= random uint32.  None when
    sample == 0 (no subsampling)."""
    if not sample or sample <= 0:
        return None
    v = counts.double()
    retain_total = float(v.sum())
    threshold = sample if sample >= 1.0 else sample * retain_total
    prob = (torch.sqrt(v / threshold) + 1.0) * (threshold / v)
    prob = torch.clamp(prob, max=1.0)
    si = torch.round(prob * 2.0 ** 32).clamp_(max=2.0 ** 32 - 1)
    # uint32 bit pattern inside an int32 tensor
    si = si.to(torch.int64)
    return torch.where(si >= 2 ** 31, si - 2 ** 32, si).to(torch.int32)


def init_syn0(n_vocab: int, dim: int, seed: int, device) -> torch.Tensor:
    """(U[0,1) - 0.5) / dim, fp32 (gensim seeds each row from hash(word + str(seed)),
    which is not reproducible across processes; a seeded device generator is)."""
    gen = torch.Generator(device=device).manual_seed(seed & (2 ** 63 - 1))
    out = torch.rand((n_vocab, dim), generator=gen, device=device, dtype=torch.float32)
    return out.sub_(0.5).div_(dim)


class JobSchedule:
    """gensim's learning-rate schedule (word2vec.py `_job_producer`, `_get_next_alpha`, called from
    embedding.py:126): sentences are queued in jobs of at most `batch_words` raw words, and the rate
    of a job is  max(end, start - (start - end) * (epoch + pushed / total) / epochs)  with `pushed`
    the sentences queued before it.  For the equal-length sentences of a walk corpus a job is
    `job_rows` = max(1, batch_words // sentence length) consecutive rows; n2v_sgns_job_alpha writes
    the rate of every row of a launch (n2v_sgns_params.row_alpha), `alpha_of_rows` is the same on the
    host."""

    def __init__(self, job_rows: int, rows: int, epoch: int, epochs: int, alpha0: float, alpha_min: float):
        self.job_rows, self.rows = max(1, int(job_rows)), max(1, int(rows))
        self.epoch, self.epochs = int(epoch), max(1, int(epochs))
        self.alpha0, self.alpha_min = float(alpha0), float(alpha_min)

    @classmethod
    def for_corpus(cls, batch_words: int, sentence_len: int, rows: int, epoch: int, epochs: int,
                   alpha0: float, alpha_min: float) -> "JobSchedule":
        return cls(max(1, int(batch_words) // max(1, int(sentence_len))), rows, epoch, epochs,
                   alpha0, alpha_min)

    def alpha_of_rows(self, row0: int, n: int) -> np.ndarray:
        """fp32 rate of rows row0 .. row0 + n - 1 (numpy float64 arithmetic, as Python's)"""
        job = (row0 + np.arange(n, dtype=np.int64)) // self.job_rows
        pushed = (job * self.job_rows).astype(np.float64)
        epoch_progress = 1.0 * pushed / float(self.rows)
        progress = (float(self.epoch) + epoch_progress) / float(self.epochs)
        nxt = self.alpha0 - (self.alpha0 - self.alpha_min) * progress
        return np.maximum(self.alpha_min, nxt).astype(np.float32)


class SgnsModel:
    """The trained state: what gensim keeps in model.wv.vectors / trainables.syn1neg."""

    def __init__(self, vocab: Vocab, dim: int, window: int, negative: int, seed: int,
                 sample: float = 0.0, ns_exponent: float = 0.75, device=None,
                 use_cum_index: bool = True):
        device = device or vocab.ids.device
        self.vocab, self.dim, self.window, self.negative = vocab, int(dim), int(window), int(negative)
        self.seed = int(seed) & (2 ** 64 - 1)
        n = len(vocab)
        if n == 0:
            raise RuntimeError("you must first build vocabulary before training the model")
        self.syn0 = init_syn0(n, self.dim, self.seed, device)
        self.syn1neg = torch.zeros((n, self.dim), dtype=torch.float32, device=device)
        self.cum_table = make_cum_table(vocab.counts, ns_exponent).to(device)
        self.sample_int = make_sample_int(vocab.counts, sample)
        if self.sample_int is not None:
            self.sample_int = self.sample_int.to(device)
        self.exp_table = torch.from_numpy(exp_table()).to(device)
        # fine index over cum_table (n2v_cum_index_build): ~16 words per bucket, so a negative
        # draw is one index sector + one table sector; built on the GPU, skipped on CPU tensors
        self.cum_index, self.cum_index_bits = None, 0
        if self.cum_table.is_cuda and use_cum_index:
            self.cum_index_bits = int(min(24, max(10, math.ceil(math.log2(max(n, 2))) - 4)))
            self.cum_index = torch.empty((1 << self.cum_index_bits) + 1, dtype=torch.int32, device=device)
            with torch.cuda.device(device):
                _lib.check(_lib.load().n2v_cum_index_build(
                    self.cum_table.data_ptr(), n, self.cum_index_bits, self.cum_index.data_ptr(),
                    _lib.current_stream_ptr()), "n2v_cum_index_build")
        # include/n2v_hip.h: pairs_out is two uint64, [0] the pair counter, [1] kernel scratch
        self.max_waves = 0  # hogwild concurrency cap (0 = the library's rule, n2v_sgns_params)
        self.batched = False  # opt-in: negatives shared by the pairs of a centre position
        self.window_cache = 0  # default kernel: syn0 rows of the window in LDS (measured slower: off)
        # hogwild: atomic adds instead of stores on rows [0, hub_rows), the most frequent words.
        # None = chosen from the corpus (auto_hub_rows); 0 = plain stores everywhere (gensim's code)
        self.hub_rows: Optional[int] = None
        self.hub_rows_auto = False  # True once auto_hub_rows chose hub_rows (with hub_waves waves in flight)
        self.hub_waves: Optional[int] = None
        self.hub_share: Optional[float] = None      # share of all row-holds on the rows the lambda rule selects
        self.hub_candidates: Optional[int] = None   # how many rows it selects (hub_rows = that, or 0 below HUB_MIN_SHARE)
        self.ns_exponent = float(ns_exponent)
        self._counters = torch.zeros(2, dtype=torch.int64, device=device)
        self.pairs = self._counters[:1]
        self.sentences_seen = 0

    HUB_LAMBDA = 1.5      # rows held by at least this many waves on average are updated atomically ...
    HUB_MIN_SHARE = 0.10  # ... when together they carry at least this share of all row-holds (auto_hub_rows)

    def hogwild_waves(self, rows: int, length: int) -> int:
        """the waves n2v_sgns_train keeps in flight for a launch of `rows` sentences of `length`
        tokens on THIS device with this model's parameters -- asked of the library
        (n2v_sgns_hogwild_waves: its concurrency rule, max_waves and the kernel's occupancy), not
        assumed (8 192 on an MI355X at dim 128)"""
        L = _lib.load()
        P = _lib.SgnsParams(len(self.vocab), 0, self.seed, self.dim, self.window, self.negative, 0.025, 0,
                            self.cum_index_bits, 0 if self.cum_index is None else self.cum_index.data_ptr(),
                            int(self.max_waves), 0, int(self.window_cache), 0, 0)
        with torch.cuda.device(self.syn0.device):
            w = int(L.n2v_sgns_hogwild_waves(P, int(rows), int(length)))
        if w < 0:
            _lib.check(w, "n2v_sgns_hogwild_waves")
        return max(w, 1)

    def auto_hub_rows(self, rows: int = 1 << 30, length: int = 81) -> int:
        """How many of the most frequent rows to update by atomic adds so that the trainer's
        concurrency regime is the reference's.  gensim runs <= 16 unsynchronised threads
        (constants.py:67 `workers`, embedding.py:126): a row is practically never held by two of
        them.  The GPU runs up to 8 192 waves; each holds one syn0 row (its context word) and
        1 + k syn1neg rows (centre word and negatives), so row i is held by
            lambda_i = waves x (f_i + k n_i)        f_i token share, n_i negative-draw share
        waves at a time on average.  Rows held by more than one wave more often than not are where
        read-modify-write stores overwrite what other waves learned (measured on cfg 2: link AUC
        0.897 against 0.908 - 0.914 at <= 64 waves, profiles/r3a_hogwild_auc_runs.log); they get
        atomic adds.  The vocabulary is in descending count order, so they are a prefix [0, H).
        The threshold lambda_i >= 1.5 is the knee of the measured curve on cfg 2
        (profiles/r4n_hogwild_auc_hub_rows_knee.log, r4l_*: H = 512 / 1024 / 2048 / 4096 / 6196 ->
        AUC 0.9015 / 0.9041 / 0.9093 / 0.9094 / 0.9119 at +9 / +12 / +17 / +30 / +56 % of the epoch
        time): the level of the <= 64-wave runs is reached around 2 000 - 3 000 rows, more rows only
        cost.  Round 6: only when those rows carry >= HUB_MIN_SHARE of all row-holds (below).  Opt-in kernels
        (batched) keep 0."""
        if self.batched:
            return 0
        waves = self.hogwild_waves(rows, length)  # (of the launch the rule is applied to: the first)
        self.hub_waves = waves
        c = self.vocab.counts.to(torch.float64)
        pw = c.pow(self.ns_exponent)
        held = c / c.sum() + self.negative * pw / pw.sum()
        h = int((waves * held >= self.HUB_LAMBDA).sum().item())
        # ... and only where those rows carry a real share of the training (round 6).  On cfg 2 they take 18 % of all
        # row-holds (31 % of the tokens) and the atomics are worth + 0.009 link AUC for + 17 % of the epoch; on cfg 3
        # (5 % of the row-holds, 561 rows) one epoch ends at the same AUC with or without them -- 0.9148 / 0.9150, on the
        # edges of the 1 000 biggest hubs 0.863 / 0.866 -- and they cost 12 % (profiles/r10r_auc_cfg3.log,
        # r10s_hub_share.log); cfg 4: 2.6 %, 284 rows, 5 % of the rate.  Below HUB_MIN_SHARE the trainer keeps gensim's
        # plain stores.
        self.hub_share = float(held[:h].sum().item()) / (1.0 + self.negative) if h else 0.0
        self.hub_candidates = h
        return h if self.hub_share >= self.HUB_MIN_SHARE else 0

    def _hub_rows(self, rows: int, length: int) -> int:
        if self.hub_rows is None:
            self.hub_rows = self.auto_hub_rows(rows, length)
            self.hub_rows_auto = True
        return int(self.hub_rows)

    # -- one kernel launch ----------------------------------------------------
    def train_block(self, walks_idx: torch.Tensor, alpha: float, sentence_base: int,
                    deterministic: bool = False, sched: Optional["JobSchedule"] = None, row0: int = 0):
        """walks_idx: int32 [rows, len] vocabulary indices (-1 = out of vocabulary).  `alpha` is
        the rate of every row of the launch unless `sched` (gensim's per-job schedule) is given;
        row 0 of the block is then sentence `row0` of the schedule's epoch."""
        L = _lib.load()
        _lib.require_gpu()
        if walks_idx.dtype != torch.int32 or walks_idx.dim() != 2 or not walks_idx.is_cuda:
            raise TypeError("train_block wants a CUDA int32 [rows, len] tensor")
        if walks_idx.shape[1] > MAX_SENTENCE:
            raise ValueError(f"walks longer than {MAX_SENTENCE}: split rows first (split_rows)")
        walks_idx = walks_idx.contiguous()
        row_alpha = None
        if sched is not None and walks_idx.shape[0] > 0:  # the rate of every row: that of its gensim job
            row_alpha = torch.empty(walks_idx.shape[0], dtype=torch.float32, device=walks_idx.device)
            with torch.cuda.device(walks_idx.device):
                _lib.check(L.n2v_sgns_job_alpha(sched.job_rows, sched.epoch, sched.epochs, int(row0),
                                                sched.rows, sched.alpha0, sched.alpha_min,
                                                row_alpha.numel(), row_alpha.data_ptr(),
                                                _lib.current_stream_ptr()), "n2v_sgns_job_alpha")
        P = _lib.SgnsParams(len(self.vocab), int(sentence_base), self.seed, self.dim, self.window,
                            self.negative, float(alpha), int(bool(deterministic)),
                            self.cum_index_bits, 0 if self.cum_index is None else self.cum_index.data_ptr(),
                            int(self.max_waves), int(bool(self.batched)), int(self.window_cache),
                            self._hub_rows(walks_idx.shape[0], walks_idx.shape[1]),
                            0 if row_alpha is None else row_alpha.data_ptr())
        with torch.cuda.device(walks_idx.device):
            rc = L.n2v_sgns_train(walks_idx.data_ptr(), walks_idx.shape[0], walks_idx.shape[1],
                                  self.syn0.data_ptr(), self.syn1neg.data_ptr(),
                                  self.cum_table.data_ptr(),
                                  0 if self.sample_int is None else self.sample_int.data_ptr(),
                                  self.exp_table.data_ptr(), P, self.pairs.data_ptr(),
                                  _lib.current_stream_ptr())
        _lib.check(rc, "n2v_sgns_train")

    # -- epochs with linear decay ----------------------------------------------
    def train(self, walks_idx: torch.Tensor, epochs: int, alpha: float = 0.025,
              min_alpha: float = 1e-4, block_rows: Optional[int] = None,
              sentence_base: int = 0, deterministic: bool = False, sync=None,
              rows_global_max: Optional[int] = None, batch_words: Optional[int] = None):
        """`epochs` passes over walks_idx; the learning rate falls linearly from alpha to
        min_alpha.  With `batch_words` (gensim's parameter; the reference passes 1000,
        constants.py:58) it falls as gensim lowers it (exactly so when every row is a sentence; rows
        of dropped walkers count as empty sentences here, gensim's corpus would not hold them): per job of
        max(1, batch_words // sentence length) consecutive sentences (JobSchedule); without, once
        per launch with the fraction of rows trained.

        `sync`: the multi-GPU exchange (DeltaSync, or any object with step()/finish()).
        Every rank must then make the SAME number of calls to it, whatever its own row
        count, so the block grid is laid over `rows_global_max` (the largest row count of
        any rank, all-reduced by the caller): a rank whose shard is shorter -- or empty --
        trains nothing in its last blocks but still takes part in every collective."""
        rows = walks_idx.shape[0]
        grid_rows = max(rows, int(rows_global_max or 0))
        if block_rows is None:
            block_rows = max(1, min(max(grid_rows, 1), max(65536, math.ceil(grid_rows / 64))))
        total = max(1, grid_rows * max(epochs, 1))
        done = 0
        if sync is not None and not hasattr(sync, "step"):
            sync = _CallableSync(sync)  # the round-1 protocol: a plain callable, called per launch
        if sync is not None and getattr(sync, "max_every", 0) is None:
            sync.max_every = max(1, math.ceil(max(grid_rows, 1) / block_rows))  # launches per epoch
        for ep in range(epochs):
            sched = None
            if batch_words:
                sched = JobSchedule.for_corpus(batch_words, walks_idx.shape[1], grid_rows, ep, epochs,
                                               alpha, min_alpha)
            for lo in range(0, max(grid_rows, 1), block_rows):
                hi = min(grid_rows, lo + block_rows)
                a = max(min_alpha, alpha - (alpha - min_alpha) * (done / total))
                if lo < rows:
                    extra = {} if sched is None else {"sched": sched, "row0": lo}
                    self.train_block(walks_idx[lo:min(hi, rows)], a,
                                     sentence_base + ep * grid_rows + lo, deterministic, **extra)
                done += hi - lo
                if sync is not None:
                    sync.step()
        if sync is not None:
            sync.finish()
        self.sentences_seen += rows * epochs
        return self


SORT_COUNT_MIN_TOKENS = 1 << 24


def corpus_count(walks: torch.Tensor, valid: Optional[torch.Tensor], counts: torch.Tensor,
                 sort_above: int = SORT_COUNT_MIN_TOKENS) -> None:
    """counts[v] += occurrences of v in the valid rows of `walks`: int32 [rows, len] walks,
    uint8 / bool [rows] valid (or None), int64 [n_vertices] counts, all on the GPU.

    Small batches: one atomic add per token (n2v_corpus_count).  Large batches: device-wide atomics
    on a count vector far beyond the caches run at ~3.6 G/s whatever their width (measured on a
    cfg 4 batch of 8.5 x 10^8 tokens, profiles/r3o_time_count.log: 237 ms), while a radix sort of
    the same tokens takes 35 ms -- so the tokens are sorted, run-length encoded and the distinct
    ones added (plumbing ops; each distinct vertex is touched once)."""
    L = _lib.load()
    if walks.numel() == 0:
        return
    walks = walks.contiguous()
    if walks.numel() >= sort_above:
        flat = walks if valid is None else torch.where(valid.bool().unsqueeze(1), walks,
                                                       torch.full_like(walks, -1))
        srt = torch.sort(flat.reshape(-1)).values
        del flat
        uniq, cnt = torch.unique_consecutive(srt, return_counts=True)
        del srt
        keep = (uniq >= 0) & (uniq < counts.numel())
        counts.index_add_(0, uniq[keep].long(), cnt[keep])
        return
    v = None if valid is None else valid.to(torch.uint8).contiguous()
    with torch.cuda.device(walks.device):
        _lib.check(L.n2v_corpus_count(walks.data_ptr(), 0 if v is None else v.data_ptr(), walks.shape[0],
                                      walks.shape[1], counts.numel(), counts.data_ptr(),
                                      _lib.current_stream_ptr()), "n2v_corpus_count")


def corpus_index(walks: torch.Tensor, valid: Optional[torch.Tensor], index_of: torch.Tensor) -> torch.Tensor:
    """int32 vocabulary indices of a batch of walks (n2v_corpus_index): -1 for tokens of dropped
    rows, for negative tokens and for vertices below min_count"""
    L = _lib.load()
    out = torch.empty_like(walks, dtype=torch.int32)
    if walks.numel() == 0:
        return out
    walks = walks.contiguous()
    v = None if valid is None else valid.to(torch.uint8).contiguous()
    with torch.cuda.device(walks.device):
        _lib.check(L.n2v_corpus_index(walks.data_ptr(), 0 if v is None else v.data_ptr(),
                                      index_of.data_ptr(), walks.shape[0], walks.shape[1],
                                      index_of.numel(), out.data_ptr(), _lib.current_stream_ptr()),
                   "n2v_corpus_index")
    return out


class _CallableSync:
    """adapter: SgnsModel.train(sync=f) with a plain callable f (called after every launch)"""

    def __init__(self, fn):
        self.fn = fn

    def step(self):
        self.fn()

    def finish(self):
        pass


def split_rows(walks_idx: torch.Tensor, max_len: int = MAX_SENTENCE) -> torch.Tensor:
    """Cut walks longer than the kernel's sentence buffer into rows of max_len
    (padding with -1), the way gensim cuts sentences at MAX_SENTENCE_LEN."""
    n, ln = walks_idx.shape
    if ln <= max_len:
        return walks_idx
    parts = math.ceil(ln / max_len)
    pad = parts * max_len - ln
    w = torch.nn.functional.pad(walks_idx, (0, pad), value=-1)
    return w.reshape(n * parts, max_len).contiguous()


class DeltaSync:
    """Multi-GPU exchange step of the SGNS path.  Every rank trains its
    own walks on a full replica of syn0 / syn1neg; every `sync_every` launches the replicas
    are averaged: new = mean over ranks (= the synchronised state + the mean of the deltas
    trained since).  RCCL over xGMI with backend "nccl"; gloo on CPU tensors in the tests.

    * `sync_every` None = chosen after the first exchange so that the collective takes at
      most `comm_share` (10 %) of the time: ceil(t_sync * (1 - share) / (share * t_launch)),
      agreed by a MAX all-reduce.  At cfg 4 (2 x 51 GB replicas) one exchange moves
      2 x 51 GB (fp32) or 2 x 26 GB (bf16) per rank.
    * wire "fp32": the rows themselves are summed -- no reference copy of the model exists
      at all; wire "bf16": bf16(row - ref) against a bf16 reference shared by all ranks
      (half the bytes on the links; + 2 bytes per element of HBM for the reference).
    * No full clone: rows go through two reusable buffers of `block_rows` rows.
    * overlap: on the GPU the whole exchange runs on a side stream, block by block (pack,
      all-reduce, apply), while the main stream keeps training; `apply` adds mean - snapshot,
      so what was trained meanwhile is kept.  finish() drains it and ends with one blocking
      exchange that leaves all replicas bit-identical.
    Every rank must call step() the same number of times (SgnsModel.train sees to that)."""

    def __init__(self, model_or_tensors, group=None, sync_every: Optional[int] = None,
                 wire: str = "fp32", block_rows: int = 1 << 20, overlap: bool = True,
                 comm_share: float = 0.10, rehearse: bool = False):
        import torch.distributed as dist

        if wire not in ("fp32", "bf16"):
            raise ValueError(f"unknown wire format {wire!r}")
        tensors = ([model_or_tensors.syn0, model_or_tensors.syn1neg]
                   if hasattr(model_or_tensors, "syn0") else list(model_or_tensors))
        self.dist, self.group = dist, group
        self.tensors = [t for t in tensors]
        for t in self.tensors:
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise TypeError("DeltaSync wants contiguous float32 matrices")
        self.active = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.active else 1
        self.wire, self.block_rows, self.comm_share = wire, int(block_rows), float(comm_share)
        self.sync_every = None if sync_every is None else max(1, int(sync_every))
        self.calls = 0
        self.syncs = 0
        self.max_every: Optional[int] = None  # ceiling of the autotuned period (set by train())
        self.on_gpu = all(t.is_cuda for t in self.tensors)
        self.side = torch.cuda.Stream(self.tensors[0].device) if (overlap and self.on_gpu) else None
        self._pending = None
        self._t_sync = None
        self._t_mark = None
        self._before = self._wire = None
        self._scratch = {}
        # rehearse: a group of ONE rank still runs pack -> collectives -> apply (a mean of one);
        # `exchanged_blocks` counts the blocks that went through them
        self.rehearse = bool(rehearse) and self.active
        self.exchanged_blocks = 0
        self.refs = None
        if wire == "bf16" and (self.world > 1 or self.rehearse):
            self.refs = [self._ref_init(t) for t in self.tensors]

    # -- sizes reported by bench.py ------------------------------------------------------
    @property
    def wire_dtype_name(self) -> str:
        return self.wire

    @property
    def wire_bytes(self) -> int:
        return sum(t.numel() for t in self.tensors) * (4 if self.wire == "fp32" else 2)

    # -- elementwise passes: HIP kernels on the GPU, torch on CPU tensors (gloo tests) ------
    def _ref_init(self, t):
        ref = torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)
        if t.is_cuda:
            L = _lib.load()
            with torch.cuda.device(t.device):
                _lib.check(L.n2v_delta_ref_init(t.data_ptr(), t.numel(), ref.data_ptr(),
                                                _lib.current_stream_ptr()), "n2v_delta_ref_init")
        else:
            ref.copy_(t)
        return ref

    @staticmethod
    def block_elems(shapes, block_rows: int) -> int:
        """elements of the largest block that goes through the exchange at once"""
        return min(int(block_rows), max(int(sh[0]) for sh in shapes)) * max(
            int(sh[1]) if len(sh) > 1 else 1 for sh in shapes)

    def _buffers(self, like):
        n = self.block_elems([tuple(t.shape) for t in self.tensors], self.block_rows)
        if self._before is None or self._before.numel() < n or self._before.device != like.device:
            self._before = torch.empty(n, dtype=torch.float32, device=like.device)
            self._wire = torch.empty(n, dtype=torch.float32 if self.wire == "fp32" else torch.bfloat16,
                                     device=like.device)
        return self._before, self._wire

    def _pack(self, cur, ref, before, wire):
        if cur.is_cuda:
            L = _lib.load()
            _lib.check(L.n2v_delta_pack(cur.data_ptr(), 0 if ref is None else ref.data_ptr(),
                                        cur.numel(), 0 if before is None else before.data_ptr(),
                                        wire.data_ptr(),
                                        _lib.WIRE_F32 if self.wire == "fp32" else _lib.WIRE_BF16,
                                        _lib.current_stream_ptr()), "n2v_delta_pack")
        else:
            if before is not None:
                before.copy_(cur.reshape(-1))
            wire.copy_(cur.reshape(-1) if ref is None else cur.reshape(-1) - ref.reshape(-1).float())

    def _apply(self, cur, ref, before, wire):
        if cur.is_cuda:
            L = _lib.load()
            _lib.check(L.n2v_delta_apply(cur.data_ptr(), 0 if ref is None else ref.data_ptr(),
                                         0 if before is None else before.data_ptr(), wire.data_ptr(),
                                         _lib.WIRE_F32 if self.wire == "fp32" else _lib.WIRE_BF16,
                                         self.world, cur.numel(), _lib.current_stream_ptr()),
                       "n2v_delta_apply")
        else:
            flat = cur.reshape(-1)
            if ref is None:
                mean = wire / self.world
            else:
                mean = ref.reshape(-1).float() + wire.float() / self.world
                ref.reshape(-1).copy_(mean)
            if before is None:
                flat.copy_(mean)
            else:
                flat.add_(mean - before)

    def _exchange(self, exact: bool = False):
        """pack / all-reduce / apply over all matrices, block by block, on the current stream;
        `exact` (nothing trains meanwhile): rows are SET to the mean, no snapshot"""
        for k, t in enumerate(self.tensors):
            ref = None if self.refs is None else self.refs[k]
            before, wire = self._buffers(t)
            for lo in range(0, t.shape[0], self.block_rows):
                hi = min(t.shape[0], lo + self.block_rows)
                cur = t[lo:hi]
                n = cur.numel()
                snap = None if exact else before[:n]
                self._pack(cur, None if ref is None else ref[lo:hi], snap, wire[:n])
                # the sum over the ranks: bytes through the collective library, the additions here,
                # in fp32 and in rank order (shard.ordered_sum) -- the same bits under RCCL and gloo
                ordered_sum(wire[:n], self.group, self.dist, self._scratch, force=self.rehearse)
                self.exchanged_blocks += 1
                self._apply(cur, None if ref is None else ref[lo:hi], snap, wire[:n])

    # -- the protocol -------------------------------------------------------------------------
    def sync(self, blocking: bool = False):
        if not self.active or (self.world == 1 and not self.rehearse):
            return
        self.syncs += 1
        if self.side is None or blocking:
            self._drain()
            self._exchange(exact=True)
            return
        main = torch.cuda.current_stream(self.tensors[0].device)
        ev = torch.cuda.Event()
        ev.record(main)
        self.side.wait_event(ev)  # everything trained so far is visible to the exchange
        with torch.cuda.stream(self.side):
            self._exchange()
            self._pending = torch.cuda.Event()
            self._pending.record(self.side)

    def _drain(self):
        if self._pending is not None:
            torch.cuda.current_stream(self.tensors[0].device).wait_event(self._pending)
            self._pending = None

    def step(self):
        """call after every training launch, on every rank"""
        self.calls += 1
        if not self.active or self.world == 1:
            return
        if self.sync_every is None:
            self._autotune()
        elif self.calls % self.sync_every == 0:
            self.sync()

    def _autotune(self):
        import time

        def now():
            if self.on_gpu:
                torch.cuda.synchronize(self.tensors[0].device)
            return time.perf_counter()

        if self._t_sync is None:  # first call: one timed blocking exchange
            t0 = now()
            self.sync(blocking=True)
            self._t_mark = now()
            self._t_sync = self._t_mark - t0
            return
        # The MEASUREMENTS are agreed (MAX over ranks), not the derived period: a rank whose
        # shard is empty or shorter than a block measures t_launch ~ 0 and would otherwise push
        # an astronomically long period onto everybody -- the replicas would then
        # train independently and only be averaged once, at finish().
        t_launch = max(now() - self._t_mark, 0.0)  # one training launch since then
        dev = self.tensors[0].device
        v = torch.tensor([self._t_sync, t_launch], dtype=torch.float64, device=dev)
        all_reduce(v, self.dist.ReduceOp.MAX, self.group, self.dist)
        self.sync_every = self.period_for(float(v[0]), float(v[1]))
        self.calls = 0  # periods count from here

    def period_for(self, t_sync: float, t_launch: float) -> int:
        """launches between two exchanges so that the collective takes <= comm_share of the time;
        never longer than `max_every` (SgnsModel.train: the launches of one epoch), so that the
        replicas are averaged at least once per epoch whatever the timings say"""
        share = min(max(self.comm_share, 1e-3), 0.999)
        every = max(1, math.ceil(t_sync * (1.0 - share) / (share * max(t_launch, 1e-6))))
        if self.max_every is not None:
            every = min(every, max(1, int(self.max_every)))
        return every

    def finish(self):
        """drain the side stream and average once more, blocking: all replicas identical"""
        if not self.active or self.world == 1:
            return
        self._drain()
        self.sync(blocking=True)

    __call__ = sync  # the old DeltaAllReduce was called like a function


DeltaAllReduce = DeltaSync

HBM_BYTES = 288e9          # MI355X (MI355X_MICROARCH.md)
XGMI_LINK_GBPS = 153.0     # per link, 7 links per GPU, full mesh of 8 (the guide's figure; per direction: half)


def exchange_plan(shapes, world: int, wire: str = "bf16", block_rows: int = 1 << 20, resident_bytes: int = 0,
                  hbm_bytes: float = HBM_BYTES, link_GBps: float = XGMI_LINK_GBPS) -> dict:
    """What one rank of `world` holds and moves when DeltaSync averages matrices of `shapes` (fp32): from the very
    size rules the exchange allocates by (DeltaSync.block_elems, shard.ordered_sum_shard; a two-rank gloo test
    compares them with the live buffers), so it can be evaluated for a world nobody can run here -- BASELINE cfg 4
    on 8 GPUs (bench.py prints it in sgns.exchange_plan_world8).

    HBM: the replicas, the bf16 reference of the wire format, the snapshot / wire block, ordered_sum's send / recv /
    shard buffers, + `resident_bytes` (graph, walk tables, corpus batch).  Links: the all-to-all hands every peer one
    shard of every block and the all-gather returns one: per sync, 2 x wire_bytes / world in EACH direction of EVERY
    link of the full mesh, all links busy at once (shard.ordered_sum)."""
    if wire not in ("fp32", "bf16"):
        raise ValueError(f"unknown wire format {wire!r}")
    world = int(world)
    wb = 4 if wire == "fp32" else 2
    elems = sum(int(np.prod(sh)) for sh in shapes)
    model = 4 * elems
    refs = 2 * elems if wire == "bf16" and world > 1 else 0
    n = DeltaSync.block_elems(shapes, block_rows)
    m = ordered_sum_shard(n, world)
    buffers = 4 * n + wb * n
    scratch = wb * (2 * world * m + m)
    link = 0
    blocks = 0
    for sh in shapes:
        rows, cols = int(sh[0]), (int(sh[1]) if len(sh) > 1 else 1)
        for lo in range(0, rows, int(block_rows)):
            nb = (min(rows, lo + int(block_rows)) - lo) * cols
            link += 2 * wb * ordered_sum_shard(nb, world)
            blocks += 1
    total = model + refs + buffers + scratch + int(resident_bytes)
    per_dir = link_GBps * 1e9 / 2.0
    return {"world": world, "wire": wire, "block_rows": int(block_rows), "blocks_per_sync": blocks,
            "model_bytes": model, "bf16_reference_bytes": refs, "block_buffers_bytes": buffers,
            "ordered_sum_buffers_bytes": scratch, "resident_bytes": int(resident_bytes),
            "hbm_bytes_per_rank": total, "hbm_share": total / hbm_bytes, "fits": total <= hbm_bytes,
            "wire_bytes_per_rank_per_sync": wb * elems,
            "bytes_per_link_per_direction_per_sync": 0 if world == 1 else link,
            "links_used": max(world - 1, 0),
            "link_seconds_per_sync_at_peak": 0.0 if world == 1 else link / per_dir,
            "link_GBps_assumed": {"per_link": link_GBps, "per_direction": link_GBps / 2.0}}
