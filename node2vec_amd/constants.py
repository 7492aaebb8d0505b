"""Default parameters, same keys and values as the reference's node2vec/constants.py
(:6-68) so that caller dicts behave identically.  One deliberate difference is
documented at HIP_SGNS_PARAMS: the reference's gensim default negative=0 with
sg=0/hs=0 trains nothing (SURVEY.md finding 4); the HIP trainer is skip-gram with
negative sampling and defaults to k=5.
"""
from typing import Any, Dict

# constants.py:6 -- default cap used by trim_hotspot_vertices when max_out_degree <= 0
MAX_OUT_DEGREES: int = 100000

# constants.py:10 -- Spark partition count; accepted for compatibility, unused on device
NUM_PARTITIONS: int = 3000

# constants.py:14-27
NODE2VEC_PARAMS: Dict[str, Any] = {
    "num_walks": 10,        # walks started from every vertex
    "walk_length": 20,      # steps per walk (a walk has walk_length + 1 vertices)
    "return_param": 1.0,    # p of the node2vec paper
    "inout_param": 1.0,     # q of the node2vec paper
}

# constants.py:31-46 (Spark ML Word2Vec names; kept for callers that pass them)
WORD2VEC_PARAMS: Dict[str, Any] = {
    "minCount": 10,
    "numPartitions": 100,
    "stepSize": 0.025,
    "maxIter": 10,
    "seed": None,
    "maxSentenceLength": 10000,
    "windowSize": 5,
    "vectorSize": 128,
}

# constants.py:50-68 (gensim Word2Vec names)
GENSIM_PARAMS: Dict[str, Any] = {
    "min_count": 10,
    "alpha": 0.025,
    "iter": 10,
    "seed": None,
    "batch_words": 1000,
    "window": 5,
    "size": 128,
    "negative": 0,
    "workers": 16,
}

# What the HIP trainer adds on top of GENSIM_PARAMS when the caller leaves them
# out: gensim's own defaults for the pass-through names, except sg/negative which
# select the north-star's skip-gram negative-sampling objective.
HIP_SGNS_PARAMS: Dict[str, Any] = {
    "sg": 1,
    "hs": 0,
    "negative": 5,
    "sample": 1e-3,
    "min_alpha": 1e-4,
    "ns_exponent": 0.75,
    # options of the HIP trainer that gensim does not have (all off = gensim's semantics):
    "batched": False,   # True: negatives shared by the pairs of a centre position (MFMA kernel)
    "hub_rows": None,   # hogwild mode: rows [0, hub_rows) -- the most frequent words -- are updated by
                        # atomic adds.  None = as many as are held by >= 1.5 waves at a time on average
                        # (SgnsModel.auto_hub_rows: gensim's <= 16 threads never share a row, 8 192
                        # waves do); 0 = plain stores everywhere, gensim's code as written
    "deterministic": False,  # True: one wave, sentences in order -- reproducible bit for bit
                             # (gensim's workers=1), orders of magnitude slower: for tests
}
