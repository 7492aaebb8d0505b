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
        dist.all_reduce(size, op=dist.ReduceOp.MAX, group=group)
        n_ids = int(size.item())  # the same on every rank
        dense = torch.bincount(flat, minlength=n_ids)
        dist.all_reduce(dense, op=dist.ReduceOp.SUM, group=group)
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


class SgnsModel:
    """The trained state: what gensim keeps in model.wv.vectors / trainables.syn1neg."""

    def __init__(self, vocab: Vocab, dim: int, window: int, negative: int, seed: int,
                 sample: float = 0.0, ns_exponent: float = 0.75, device=None):
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
        # include/n2v_hip.h: pairs_out is two uint64, [0] the pair counter, [1] kernel scratch
        self._counters = torch.zeros(2, dtype=torch.int64, device=device)
        self.pairs = self._counters[:1]
        self.sentences_seen = 0

    # -- one kernel launch ----------------------------------------------------
    def train_block(self, walks_idx: torch.Tensor, alpha: float, sentence_base: int,
                    deterministic: bool = False):
        """walks_idx: int32 [rows, len] vocabulary indices (-1 = out of vocabulary)."""
        L = _lib.load()
        _lib.require_gpu()
        if walks_idx.dtype != torch.int32 or walks_idx.dim() != 2 or not walks_idx.is_cuda:
            raise TypeError("train_block wants a CUDA int32 [rows, len] tensor")
        if walks_idx.shape[1] > MAX_SENTENCE:
            raise ValueError(f"walks longer than {MAX_SENTENCE}: split rows first (split_rows)")
        walks_idx = walks_idx.contiguous()
        P = _lib.SgnsParams(len(self.vocab), int(sentence_base), self.seed, self.dim, self.window,
                            self.negative, float(alpha), int(bool(deterministic)), 0)
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
              sentence_base: int = 0, deterministic: bool = False, sync=None):
        """`epochs` passes over walks_idx; the learning rate falls linearly from alpha
        to min_alpha with the fraction of rows trained (gensim: by words, per job).
        `sync`: optional callable run after every block (multi-GPU delta all-reduce)."""
        rows = walks_idx.shape[0]
        if block_rows is None:
            block_rows = max(1, min(rows, max(65536, math.ceil(rows / 64))))
        total = max(1, rows * max(epochs, 1))
        done = 0
        for ep in range(epochs):
            for lo in range(0, rows, block_rows):
                hi = min(rows, lo + block_rows)
                a = max(min_alpha, alpha - (alpha - min_alpha) * (done / total))
                self.train_block(walks_idx[lo:hi], a, sentence_base + ep * rows + lo, deterministic)
                done += hi - lo
                if sync is not None:
                    sync(self)
        self.sentences_seen += rows * epochs
        return self


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


class DeltaAllReduce:
    """Multi-GPU exchange step of the SGNS path: every rank trains
    its own walks on a full replica; `sync` all-reduces the model deltas accumulated
    since the last sync (RCCL over xGMI with backend "nccl"; gloo on CPU in tests)
    and applies their mean (or sum) to the synchronised copy.  Row blocks bound the
    temporary to `block_rows * dim` floats."""

    def __init__(self, model_tensors, group=None, mean: bool = True, block_rows: int = 1 << 20):
        import torch.distributed as dist

        self.dist, self.group, self.mean, self.block_rows = dist, group, mean, block_rows
        self.tensors = list(model_tensors)
        self.synced = [t.clone() for t in self.tensors]
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = dist.is_initialized()

    def __call__(self, _model=None):
        if not self.active:
            return
        for t, s in zip(self.tensors, self.synced):
            for lo in range(0, t.shape[0], self.block_rows):
                hi = min(t.shape[0], lo + self.block_rows)
                d = t[lo:hi] - s[lo:hi]
                self.dist.all_reduce(d, op=self.dist.ReduceOp.SUM, group=self.group)
                if self.mean:
                    d.div_(self.world)
                s[lo:hi].add_(d)
                t[lo:hi].copy_(s[lo:hi])
