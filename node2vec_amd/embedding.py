"""Embedding plugin API: the reference's node2vec/embedding.py with a HIP trainer.

`Node2VecBase` is the reference's abstract plugin interface (embedding.py:22-66:
fit / embedding / get_vector / save_model / load_model, all NotImplementedError).
`Node2VecHIP` mirrors `Node2VecGensim` (embedding.py:70-178): same constructor
arguments, same in-place filling of the caller's w2v_params from GENSIM_PARAMS,
same ValueErrors, same DataFrame shapes -- with the gensim.models.Word2Vec call of
embedding.py:126 replaced by the SGNS kernel.  `Node2VecGensim` is exported as an
alias so `from node2vec.embedding import Node2VecGensim` ports by changing the
package name only.

The one behavioural decision (SURVEY.md finding 4): the reference's defaults
(negative=0 with gensim's sg=0, hs=0) perform no weight updates at all; this
trainer is skip-gram with negative sampling, so sg=1 and a missing/zero `negative`
becomes 5 (constants.HIP_SGNS_PARAMS).  hs=1 / sg=0 are rejected with ValueError.
"""
import logging
import os
import time
from typing import Any, Dict, List, Optional, Union

import numpy as np
import pandas as pd
import torch

from node2vec_amd import corpus, sgns
from node2vec_amd.constants import GENSIM_PARAMS, HIP_SGNS_PARAMS


class Node2VecBase(object):
    """Base class for conducting Node2Vec in various computing frameworks
    (embedding.py:22-66)."""

    def __init__(self):
        pass

    def fit(self):
        raise NotImplementedError()

    def embedding(self):
        raise NotImplementedError()

    def get_vector(self, vertex_id: Union[str, int]):
        raise NotImplementedError()

    def save_model(self, file_path: str, file_name: str):
        raise NotImplementedError()

    def load_model(self, file_path: str, file_name: str):
        raise NotImplementedError()


class _Tokens:
    """index -> token, lazily: the tokens of a fitted model are the decimal strings of its vertex
    ids (embedding.py:125); at BASELINE cfg 4 a list of 10^8 Python strings is ~6 GB of objects
    nobody reads, so the ids stay an int64 array and strings are made on access."""

    def __init__(self, ids: np.ndarray):
        self.ids = ids

    def __len__(self):
        return int(self.ids.shape[0])

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [str(int(x)) for x in self.ids[i]]
        return str(int(self.ids[i]))

    def __iter__(self):
        for lo in range(0, len(self), 1 << 16):
            yield from map(str, self.ids[lo:lo + (1 << 16)].tolist())

    def __eq__(self, other):
        return len(self) == len(other) and all(a == b for a, b in zip(self, other))


class _Vocab:
    """token -> row, lazily (what callers of the reference do with model.wv.vocab: `in`, `[]`,
    `len`, iteration in index order -- embedding.py:135-136).  Lookup through a sorted copy of the
    id array, built on first use (searchsorted): no 10^8-entry dict."""

    def __init__(self, ids: np.ndarray):
        self.ids = ids
        self._sorted = None

    def _row(self, token) -> int:
        try:
            v = int(token)
        except (TypeError, ValueError):
            return -1
        if str(v) != str(token):  # "07", " 7": not a token this vocabulary ever produced
            return -1
        if self._sorted is None:
            order = np.argsort(self.ids, kind="stable")
            self._sorted = (self.ids[order], order)
        keys, order = self._sorted
        k = int(np.searchsorted(keys, v))
        return int(order[k]) if k < keys.shape[0] and keys[k] == v else -1

    def __contains__(self, token):
        return self._row(token) >= 0

    def __getitem__(self, token):
        r = self._row(token)
        if r < 0:
            raise KeyError(token)
        return r

    def get(self, token, default=None):
        r = self._row(token)
        return default if r < 0 else r

    def __len__(self):
        return int(self.ids.shape[0])

    def __iter__(self):
        return iter(_Tokens(self.ids))

    def keys(self):
        return iter(self)

    def items(self):
        return ((t, i) for i, t in enumerate(self))


class KeyedVectors:
    """What callers of the reference touch on model.wv: `vocab` (token -> row; tokens
    are decimal strings of vertex ids, embedding.py:125), `wv[token]`, `index2word`, `vectors`
    and the word2vec text format.

    `tokens`: a list of strings (a loaded text file), or an integer id array / tensor (a fitted
    model: tokens and the token -> row map are then lazy).  `vectors`: numpy [n, dim], or the
    trainer's device tensor -- it stays in HBM; `wv[token]` copies one row, `.vectors` converts
    the whole matrix to numpy on first access (51 GB at cfg 4: ask for rows or chunks instead,
    `rows(lo, hi)`)."""

    def __init__(self, tokens, vectors):
        if isinstance(tokens, torch.Tensor):
            tokens = tokens.cpu().numpy()
        if isinstance(tokens, np.ndarray) and tokens.dtype.kind in "iu":
            self.ids: Optional[np.ndarray] = tokens.astype(np.int64, copy=False)
            self.index2word = _Tokens(self.ids)
            self.vocab = _Vocab(self.ids)
        else:
            self.ids = None
            self.index2word = list(tokens)
            self.vocab = {t: i for i, t in enumerate(self.index2word)}
        self._vectors = vectors
        self.vector_size = int(vectors.shape[1]) if vectors.ndim == 2 else 0

    @property
    def vectors(self) -> np.ndarray:
        if isinstance(self._vectors, torch.Tensor):
            self._vectors = self._vectors.cpu().numpy()
        return self._vectors

    def rows(self, lo: int, hi: int) -> np.ndarray:
        """vectors[lo:hi] as numpy without converting the whole matrix"""
        v = self._vectors[lo:hi]
        return v.cpu().numpy() if isinstance(v, torch.Tensor) else v

    def __len__(self):
        return len(self.index2word)

    def __getitem__(self, token: str) -> np.ndarray:
        r = self.vocab[token]
        return self.rows(r, r + 1)[0]

    def __contains__(self, token: str) -> bool:
        return token in self.vocab

    def save_word2vec_format(self, fname: str, chunk_rows: int = 1 << 16) -> None:
        with open(fname, "w") as f:
            f.write(f"{len(self.index2word)} {self.vector_size}\n")
            for lo in range(0, len(self), chunk_rows):
                block = self.rows(lo, lo + chunk_rows).astype(np.float64).tolist()
                toks = self.index2word[lo:lo + chunk_rows]
                f.writelines(t + " " + " ".join(map(repr, v)) + "\n" for t, v in zip(toks, block))

    @classmethod
    def load_word2vec_format(cls, fname: str) -> "KeyedVectors":
        with open(fname) as f:
            n, dim = (int(x) for x in f.readline().split())
            tokens, rows = [], np.zeros((n, dim), np.float32)
            for i in range(n):
                parts = f.readline().rstrip("\n").split(" ")
                tokens.append(parts[0])
                rows[i] = np.asarray(parts[1:1 + dim], dtype=np.float32)
        return cls(tokens, rows)


class HipW2V:
    """The fitted model object returned by fit() (stands where gensim's Word2Vec
    object stands): .wv plus the output matrix and the training parameters.  The matrices may be
    the trainer's device tensors (they are converted when saved or read as numpy)."""

    def __init__(self, wv: KeyedVectors, syn1neg, params: Dict[str, Any], pairs: int):
        self.wv, self._syn1neg, self.params, self.pairs_trained = wv, syn1neg, dict(params), pairs

    @property
    def syn1neg(self) -> np.ndarray:
        if isinstance(self._syn1neg, torch.Tensor):
            self._syn1neg = self._syn1neg.cpu().numpy()
        return self._syn1neg

    def save(self, fname: str) -> None:
        tokens = self.wv.ids if self.wv.ids is not None else self.wv.index2word
        torch.save({"tokens": tokens, "vectors": self.wv.vectors,
                    "syn1neg": self.syn1neg, "params": self.params,
                    "pairs": self.pairs_trained}, fname)

    @classmethod
    def load(cls, fname: str) -> "HipW2V":
        d = torch.load(fname, weights_only=False)
        return cls(KeyedVectors(d["tokens"], d["vectors"]), d["syn1neg"], d["params"], d["pairs"])


class Node2VecHIP(Node2VecBase):
    """Drop-in for Node2VecGensim (embedding.py:70-178) on one MI355X."""

    def __init__(
        self,
        df_walks: pd.DataFrame,
        w2v_params: Dict[str, Any],
        name_id: Optional[pd.DataFrame] = None,
        window_size: Optional[int] = None,
        vector_size: Optional[int] = None,
        random_seed: Optional[int] = None,
    ) -> None:
        super().__init__()
        self.walks = df_walks
        self.name_id = name_id
        self.model: Optional[HipW2V] = None

        for param in GENSIM_PARAMS:  # embedding.py:105-107: fills the caller's dict
            if param not in w2v_params:
                w2v_params[param] = GENSIM_PARAMS[param]
        w2v_params["seed"] = random_seed if random_seed else int(time.time()) // 60  # :108
        if window_size is not None:
            if window_size < 5 or window_size > 30:  # :110-111
                raise ValueError(f"Inappropriate context window size {window_size}!")
            w2v_params["window"] = window_size
        if vector_size is not None:
            if vector_size < 32 or vector_size > 1024:  # :114-115
                raise ValueError(f"Inappropriate vector dimension {vector_size}!")
            w2v_params["size"] = vector_size
        if w2v_params.get("hs", 0) or not w2v_params.get("sg", 1):
            raise ValueError("the HIP trainer implements sg=1, hs=0 (skip-gram, negative sampling)")
        logging.info(f"__init__(): w2v params: {w2v_params}")
        self.w2v_params = w2v_params

    # -- training ---------------------------------------------------------------
    def _walk_tensor(self, device) -> torch.Tensor:
        if isinstance(self.walks, torch.Tensor):
            return self.walks.to(device=device, dtype=torch.int32)
        from node2vec_amd import corpus

        dev_walks = corpus.lookup(self.walks)  # the frame random_walk() returned, unchanged
        if dev_walks is not None:
            return dev_walks.to(device=device, dtype=torch.int32)
        arr = corpus.arrow_rows(self.walks["walk"])  # an Arrow-backed column: no Python object per vertex
        if arr is not None:
            return torch.from_numpy(np.array(arr, dtype=np.int32)).to(device)  # (a copy: the Arrow buffer is read-only)
        # embedding.py:125 requires equal-length walks (np.array(walks.tolist()))
        arr = np.array(self.walks["walk"].tolist())
        if arr.ndim != 2:
            raise ValueError("walks must all have the same length")
        return torch.from_numpy(arr.astype(np.int32)).to(device)

    def fit(self, device=None, sync=None, sentence_base: int = 0) -> HipW2V:
        """Trains and returns the model (embedding.py:120-127).

        Under an initialised torch.distributed process group (one process per GPU) the
        walks held by this object are this rank's shard: the vocabulary is built from
        globally summed counts, every rank starts from the same seeded model, trains on its
        own walks with disjoint sentence ids, and the replicas are averaged (RCCL) every
        `sync_every` launches (sgns.DeltaSync; w2v_params["sync_every"], default: chosen so that
        the exchange takes <= 10 % of the time; w2v_params["sync_wire"] "fp32" | "bf16"), with a
        final blocking exchange, so all ranks return the same vectors.  Ranks may hold
        different numbers of walks (or none): the block grid is laid over the largest shard."""
        import torch.distributed as dist

        from node2vec_amd import _lib

        dev = device or _lib.require_gpu()
        p = dict(HIP_SGNS_PARAMS)
        p.update(self.w2v_params)
        negative = int(p["negative"]) if p["negative"] else int(HIP_SGNS_PARAMS["negative"])
        walks = self._walk_tensor(dev)
        vocab = sgns.build_vocab(walks, int(p["min_count"]))
        if len(vocab) == 0:
            raise RuntimeError("you must first build vocabulary before training the model")
        m = sgns.SgnsModel(vocab, int(p["size"]), int(p["window"]), negative, int(p["seed"]),
                           sample=float(p["sample"] or 0.0), ns_exponent=float(p["ns_exponent"]),
                           device=dev)
        # opt-in, not gensim's sampling: w2v_params["batched"] = True shares the k negatives of a
        # centre position among its pairs (csrc/n2v_sgns_batched.hip; dim 64 / 128 / 256,
        # window <= 7, negative <= 15)
        m.batched = bool(p.get("batched", False))
        m.hub_rows = None if p.get("hub_rows") is None else int(p["hub_rows"])  # hogwild: atomic adds on the top rows
        # tokens < 0 (rows of dropped walkers in an on-device corpus, fugue.random_walk_tensors)
        # stay outside the vocabulary; a negative index must not wrap around
        idx = torch.where(walks >= 0, vocab.index_of[walks.clamp(min=0).long()],
                          torch.full_like(walks, -1))
        idx = sgns.split_rows(idx)
        rows_max = None
        if sync is None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from node2vec_amd.shard import all_reduce, sentence_base as rank_base

            rows = torch.tensor([idx.shape[0]], device=dev)
            all_reduce(rows, dist.ReduceOp.MAX)
            rows_max = int(rows.item())  # every rank lays the same block grid over this
            sentence_base = rank_base(dist.get_rank(), dist.get_world_size(),
                                      rows_max * max(int(p["iter"]), 1))
            sync = sgns.DeltaSync(m, sync_every=p.get("sync_every"), wire=p.get("sync_wire", "fp32"))
        # the rate falls per job of batch_words words as in gensim (constants.py:58; a corpus whose
        # walks had to be split into sentences keeps one rate per launch)
        split = idx.shape[0] != walks.shape[0]
        m.train(idx, int(p["iter"]), float(p["alpha"]), float(p["min_alpha"]),
                sentence_base=sentence_base, sync=sync, rows_global_max=rows_max,
                deterministic=bool(p.get("deterministic", False)),
                batch_words=None if split else int(p.get("batch_words") or 0) or None)
        torch.cuda.synchronize(dev)
        p["negative"] = negative
        # what the trainer really ran with (hub_rows None = chosen from the corpus: recorded, so that a
        # parity run can pin it -- hub_rows = 0 is gensim's code as written)
        p["hub_rows"], p["hub_rows_auto"], p["hub_waves"] = m.hub_rows, m.hub_rows_auto, m.hub_waves
        # the matrices stay in HBM and the tokens stay integer ids (lazy strings): at cfg 4 the
        # host copies would be 2 x 51 GB + 10^8 Python strings that most callers never read
        self.model = HipW2V(KeyedVectors(vocab.ids, m.syn0), m.syn1neg, p, int(m.pairs.item()))
        return self.model

    # -- results ------------------------------------------------------------------
    def iter_embedding(self, chunk_rows: int = 1 << 18, vector_column: str = "auto"):
        """embedding() in chunks of `chunk_rows` rows: DataFrames ["id" | "name", "vector"] of the
        reference's shape, made without ever holding the whole model as Python objects.  `vector_column`:
        "list" = Python lists of floats, "rows" = one read-only ndarray view per row, "arrow" = an Arrow-backed
        list<float> column, "auto" = lists up to corpus.LIST_COLUMN_MAX_VALUES values per chunk, rows beyond
        (corpus.list_column)"""
        from node2vec_amd import corpus
        if self.model is None:
            raise ValueError("Model is not available. Please run fit()")
        wv = self.model.wv
        names = None
        if self.name_id is not None:
            # embedding.py:139-140 builds a dict: a repeated id keeps its LAST name, and an id
            # of the vocabulary that name_id does not list is a KeyError
            names = self.name_id.drop_duplicates("id", keep="last").set_index("id")["name"]
        for lo in range(0, len(wv), chunk_rows):
            hi = min(len(wv), lo + chunk_rows)
            if wv.ids is not None:
                ids = wv.ids[lo:hi]
            else:
                ids = np.array([int(t) for t in wv.index2word[lo:hi]], dtype=np.int64)
            vectors = corpus.list_column(wv.rows(lo, hi), vector_column)
            if names is not None:
                missing = ~pd.Index(ids).isin(names.index)
                if missing.any():
                    raise KeyError(int(np.asarray(ids)[missing][0]))
                yield pd.DataFrame({"name": names.reindex(ids).to_numpy(), "vector": vectors})
            else:
                yield pd.DataFrame({"id": ids, "vector": vectors})

    def embedding(self) -> pd.DataFrame:
        """embedding.py:129-143.  One DataFrame of Python lists, as the reference returns: fine
        up to a few million vertices; beyond 2^31 vector elements use iter_embedding()."""
        if self.model is None:
            raise ValueError("Model is not available. Please run fit()")
        wv = self.model.wv
        if len(wv) * max(wv.vector_size, 1) >= 2 ** 31:
            raise MemoryError(f"embedding(): {len(wv)} x {wv.vector_size} values as Python lists do "
                              "not fit a DataFrame; use iter_embedding(chunk_rows) or model.wv.rows()")
        big = len(wv) * max(wv.vector_size, 1) > corpus.LIST_COLUMN_MAX_VALUES
        parts = list(self.iter_embedding(len(wv) if big else 1 << 18, str(self.w2v_params.get("vector_column", "auto"))))
        if not parts:
            return pd.DataFrame({("name" if self.name_id is not None else "id"): [], "vector": []})
        return parts[0] if len(parts) == 1 else pd.concat(parts, ignore_index=True)

    def get_vector(self, vertex_id: Union[str, int]) -> List[float]:
        """embedding.py:145-151"""
        if isinstance(vertex_id, int):
            vertex_id = str(vertex_id)
        return list(self.model.wv[vertex_id])  # type: ignore

    def save_model(self, file_path: str, file_name: str) -> None:
        """embedding.py:153-157: "<path>/<name>.model" """
        self.model.save(os.path.join(file_path, file_name + ".model"))  # type: ignore

    def load_model(self, file_path: str, file_name: str) -> HipW2V:
        """embedding.py:159-164"""
        self.model = HipW2V.load(os.path.join(file_path, file_name + ".model"))
        return self.model

    def save_vectors(self, file_path: str, file_name: str) -> None:
        """embedding.py:166-170: word2vec text format"""
        self.model.wv.save_word2vec_format(os.path.join(file_path, file_name))  # type: ignore

    @staticmethod
    def load_vectors(file_path: str, file_name: str) -> KeyedVectors:
        """embedding.py:172-178"""
        return KeyedVectors.load_word2vec_format(os.path.join(file_path, file_name))


# import-compatible names
Node2VecGensim = Node2VecHIP
GensimW2V = HipW2V
