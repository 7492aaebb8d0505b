"""node2vec_amd -- MI355X-native node2vec hot path (walk sampler + SGNS trainer).

Host-side mirror of the reference's Python interface for that path
(graph-embedding/node2vec: node2vec/fugue.py, randomwalk.py, embedding.py,
indexer.py, constants.py) over the C ABI of include/n2v_hip.h.  All compute runs
in hand-written HIP kernels for gfx950 (node2vec_amd/csrc); there is no CPU
fallback: importing the kernels without libn2v_hip.so, or calling them without
a GPU, raises.
"""
from node2vec_amd.constants import (GENSIM_PARAMS, MAX_OUT_DEGREES, NODE2VEC_PARAMS,
                                    NUM_PARTITIONS, WORD2VEC_PARAMS)

__all__ = ["GENSIM_PARAMS", "MAX_OUT_DEGREES", "NODE2VEC_PARAMS", "NUM_PARTITIONS",
           "WORD2VEC_PARAMS"]
__version__ = "0.1.0"
