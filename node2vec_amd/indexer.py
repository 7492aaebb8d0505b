"""Vertex indexing: name -> dense int id (reference node2vec/indexer.py).

`index_graph_pandas` keeps the reference's contract (indexer.py:9-49): it needs
columns src and dst (ValueError otherwise, :18-19), adds weight 1.0 when missing
(:20-21), casts weight to float, and symmetrises + de-duplicates when the graph is
undirected (:45-48).  Ids are dense, numbered in sorted-name order like the
reference's Spark twin (indexer.py:69-70), and name_id has the columns
["name", "id"] that Node2Vec*.embedding() consumes (embedding.py:138).  (The
reference's pandas twin numbers by position of first appearance through a
reset_index quirk and needs pandas < 2; its tests pin only counts.)
"""
from typing import Tuple

import numpy as np
import pandas as pd


def index_graph_pandas(df_graph: pd.DataFrame, directed: bool) -> Tuple[pd.DataFrame, pd.DataFrame]:
    if "src" not in df_graph.columns or "dst" not in df_graph.columns:
        raise ValueError(f"Input graph NOT in the right format: {df_graph.columns}")
    if "weight" not in df_graph.columns:
        df_graph = df_graph.assign(weight=1.0)
    df_graph = df_graph[["src", "dst", "weight"]].astype({"weight": float})
    names = np.unique(np.concatenate([df_graph["src"].to_numpy(), df_graph["dst"].to_numpy()]))
    name_id = pd.DataFrame({"name": names, "id": np.arange(len(names), dtype=np.int64)})
    src = np.searchsorted(names, df_graph["src"].to_numpy())
    dst = np.searchsorted(names, df_graph["dst"].to_numpy())
    df_edge = pd.DataFrame({"src": src.astype(np.int64), "dst": dst.astype(np.int64),
                            "weight": df_graph["weight"].to_numpy()})
    if directed is not True:
        rev = df_edge.rename(columns={"src": "dst", "dst": "src"})[["src", "dst", "weight"]]
        df_edge = pd.concat([df_edge, rev], ignore_index=True).drop_duplicates(ignore_index=True)
    return df_edge, name_id
