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
from typing import Optional, Tuple

import numpy as np
import pandas as pd
import torch


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


def index_graph_tensors(src: torch.Tensor, dst: torch.Tensor, weight: Optional[torch.Tensor] = None,
                        directed: bool = True, device=None):
    """The same indexing for integer-named edge lists that already live in (or fit) device
    memory: sort/unique on the GPU instead of pandas merges, for graphs of 10^8-10^9 edges.

    Returns (src_id int64, dst_id int64, weight float32, names int64) with
    names[id] = original name, ids numbered in sorted-name order exactly like
    index_graph_pandas; undirected graphs are symmetrised and de-duplicated on
    (src, dst, weight) like indexer.py:45-48.  The edge order is by (src, dst, weight)."""
    src = torch.as_tensor(src).to(device=device, dtype=torch.int64).reshape(-1)
    dst = torch.as_tensor(dst).to(device=device, dtype=torch.int64).reshape(-1)
    if src.numel() != dst.numel():
        raise ValueError("src and dst differ in length")
    if weight is None:
        w = torch.ones(src.numel(), dtype=torch.float32, device=src.device)  # indexer.py:20-21
    else:
        w = torch.as_tensor(weight).to(device=src.device, dtype=torch.float32).reshape(-1)
        if w.numel() != src.numel():
            raise ValueError("weight differs in length from src")
    names, inverse = torch.unique(torch.cat([src, dst]), sorted=True, return_inverse=True)
    s_id, d_id = inverse[: src.numel()], inverse[src.numel():]
    if directed is not True:
        s_id, d_id, w = torch.cat([s_id, d_id]), torch.cat([d_id, s_id]), torch.cat([w, w])
        # drop_duplicates over the three columns: sort by (src, dst, weight), keep run heads
        order = torch.argsort(w, stable=True)
        s_id, d_id, w = s_id[order], d_id[order], w[order]
        order = torch.argsort(s_id * max(int(names.numel()), 1) + d_id, stable=True)
        s_id, d_id, w = s_id[order], d_id[order], w[order]
        head = torch.ones(s_id.numel(), dtype=torch.bool, device=s_id.device)
        if s_id.numel() > 1:
            head[1:] = (s_id[1:] != s_id[:-1]) | (d_id[1:] != d_id[:-1]) | (w[1:] != w[:-1])
        s_id, d_id, w = s_id[head], d_id[head], w[head]
    return s_id, d_id, w, names
