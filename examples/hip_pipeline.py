#!/usr/bin/env python3
"""The reference's three-stage example (examples/fugue_spark.py: `index | walk | embed`,
stage files in parquet) on one MI355X with node2vec_amd.

    python examples/hip_pipeline.py index  <workdir> <edges.csv|parquet>   # name -> id, trim
    python examples/hip_pipeline.py walk   <workdir>                       # random walks
    python examples/hip_pipeline.py embed  <workdir>                       # SGNS vectors

Stage files under <workdir>: graph_indexed.parquet, graph_name_id.parquet,
graph_walks.parquet, graph_vectors.parquet, vectors.w2v (word2vec text format).
"""
import logging
import os
import sys

import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from node2vec_amd import io as n2v_io  # noqa: E402
from node2vec_amd.embedding import Node2VecHIP  # noqa: E402
from node2vec_amd.fugue import random_walk_tensors, trim_index  # noqa: E402
from node2vec_amd.graph import DeviceGraph  # noqa: E402

N2V_PARAMS = {"num_walks": 10, "walk_length": 20, "return_param": 0.5, "inout_param": 2.0}
W2V_PARAMS = {"min_count": 1, "iter": 5, "negative": 5, "sample": 1e-3}


def main(argv):
    logging.basicConfig(format="%(asctime)s %(levelname)s %(message)s", level=logging.INFO)
    stage, work = argv[1], argv[2]
    p = lambda name: os.path.join(work, name)  # noqa: E731
    if stage == "index":
        src = argv[3]
        df = pd.read_parquet(src) if src.endswith(".parquet") else pd.read_csv(src)
        # examples/fugue_spark.py:47 trims at 10 000 out-edges
        edges, name_id = trim_index(None, df, indexed=False, directed=False, max_out_deg=10000,
                                    random_seed=42)
        n2v_io.write_table(edges, p("graph_indexed.parquet"))
        n2v_io.write_table(name_id, p("graph_name_id.parquet"))
        logging.info("indexed %d edges, %d vertices", len(edges), len(name_id))
    elif stage == "walk":
        g = DeviceGraph.from_pandas(n2v_io.read_table(p("graph_indexed.parquet")), device="cuda")
        walks, valid = random_walk_tensors(g, dict(N2V_PARAMS), random_seed=42)
        n = n2v_io.write_walks(p("graph_walks.parquet"), walks, valid)
        logging.info("wrote %d walks", n)
    else:  # embed
        walks = n2v_io.read_walks(p("graph_walks.parquet"), device="cuda")
        name_id = n2v_io.read_table(p("graph_name_id.parquet"))
        n2v = Node2VecHIP(walks, dict(W2V_PARAMS), name_id=name_id, window_size=5, vector_size=128,
                          random_seed=42)
        n2v.fit()
        n2v_io.write_vectors(p("graph_vectors.parquet"), n2v.embedding())
        n2v.save_vectors(work, "vectors.w2v")
        logging.info("wrote %d vectors", len(n2v.model.wv.vocab))


if __name__ == "__main__":
    main(sys.argv)
