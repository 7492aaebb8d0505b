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

from node2vec_amd import sgns
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


class KeyedVectors:
    """What callers of the reference touch on model.wv: `vocab` (token -> row; tokens
    are decimal strings of vertex ids, embedding.py:125), `wv[token]`, and the
    word2vec text format."""

    def __init__(self, tokens: List[str], vectors: np.ndarray):
        self.index2word = list(tokens)
        self.vocab = {t: i for i, t in enumerate(self.index2word)}
        self.vectors = vectors
        self.vector_size = vectors.shape[1] if vectors.ndim == 2 else 0

    def __getitem__(self, token: str) -> np.ndarray:
        return self.vectors[self.vocab[token]]

    def __contains__(self, token: str) -> bool:
        return token in self.vocab

    def save_word2vec_format(self, fname: str) -> None:
        with open(fname, "w") as f:
            f.write(f"{len(self.index2word)} {self.vector_size}\n")
            for t, v in zip(self.index2word, self.vectors):
                f.write(t + " " + " ".join(repr(float(x)) for x in v) + "\n")

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
    object stands): .wv plus the output matrix and the training parameters."""

    def __init__(self, wv: KeyedVectors, syn1neg: np.ndarray, params: Dict[str, Any], pairs: int):
        self.wv, self.syn1neg, self.params, self.pairs_trained = wv, syn1neg, dict(params), pairs

    def save(self, fname: str) -> None:
        torch.save({"tokens": self.wv.index2word, "vectors": self.wv.vectors,
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
        # tokens < 0 (rows of dropped walkers in an on-device corpus, fugue.random_walk_tensors)
        # stay outside the vocabulary; a negative index must not wrap around
        idx = torch.where(walks >= 0, vocab.index_of[walks.clamp(min=0).long()],
                          torch.full_like(walks, -1))
        idx = sgns.split_rows(idx)
        rows_max = None
        if sync is None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from node2vec_amd.shard import sentence_base as rank_base

            rows = torch.tensor([idx.shape[0]], device=dev)
            dist.all_reduce(rows, op=dist.ReduceOp.MAX)
            rows_max = int(rows.item())  # every rank lays the same block grid over this
            sentence_base = rank_base(dist.get_rank(), dist.get_world_size(),
                                      rows_max * max(int(p["iter"]), 1))
            sync = sgns.DeltaSync(m, sync_every=p.get("sync_every"), wire=p.get("sync_wire", "fp32"))
        m.train(idx, int(p["iter"]), float(p["alpha"]), float(p["min_alpha"]),
                sentence_base=sentence_base, sync=sync, rows_global_max=rows_max)
        torch.cuda.synchronize(dev)
        tokens = [str(int(i)) for i in vocab.ids.cpu().numpy()]
        p["negative"] = negative
        self.model = HipW2V(KeyedVectors(tokens, m.syn0.cpu().numpy()), m.syn1neg.cpu().numpy(),
                            p, int(m.pairs.item()))
        return self.model

    # -- results ------------------------------------------------------------------
    def embedding(self) -> pd.DataFrame:
        """embedding.py:129-143"""
        if self.model is None:
            raise ValueError("Model is not available. Please run fit()")
        ids = [int(t) for t in self.model.wv.vocab]
        vectors = [list(self.model.wv[t]) for t in self.model.wv.vocab]
        if self.name_id is not None:
            dic = self.name_id.set_index("id").to_dict()["name"]
            names = [dic[i] for i in ids]
            return pd.DataFrame.from_dict({"name": names, "vector": vectors})
        return pd.DataFrame.from_dict({"id": ids, "vector": vectors})

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
