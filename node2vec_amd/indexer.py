"""Vertex indexing: name -> dense int id (reference node2vec/indexer.py).

`index_graph_pandas` keeps the reference's contract (indexer.py:9-49): it needs
columns src and dst (ValueError otherwise, :18-19), adds weight 1.0 when missing
(:20-21), casts weight to float, and symmetrises + de-duplicates when the graph is
undirected (:45-48).  The reference has two id rules and both are built:

* id_rule="sorted" (default): dense ids in sorted-name order, name_id columns
  ["name", "id"] -- the Spark twin (indexer.py:69-70), the form
  Node2Vec*.embedding() consumes (embedding.py:138);
* id_rule="first_appearance": the pandas twin (indexer.py:26-35).  There
  `src.append(dst, ignore_index=True).drop_duplicates().reset_index()` makes the id
  of a name the POSITION of its first appearance in the concatenation
  [src column, dst column] -- ids are unique but not dense (they range below
  2 * n_edges) -- and name_id has the columns ["vertex_id", "vertex_name"].
  (That code needs pandas < 2, DataFrame.append; its tests pin only counts.)
"""
from typing import Optional, Tuple

import numpy as np
import pandas as pd
import torch


ID_RULES = ("sorted", "first_appearance")


def index_graph_pandas(df_graph: pd.DataFrame, directed: bool,
                       id_rule: str = "sorted") -> Tuple[pd.DataFrame, pd.DataFrame]:
    if "src" not in df_graph.columns or "dst" not in df_graph.columns:
        raise ValueError(f"Input graph NOT in the right format: {df_graph.columns}")
    if id_rule not in ID_RULES:
        raise ValueError(f"unknown id_rule {id_rule!r}")
    if "weight" not in df_graph.columns:
        df_graph = df_graph.assign(weight=1.0)
    df_graph = df_graph[["src", "dst", "weight"]].astype({"weight": float})
    both = np.concatenate([df_graph["src"].to_numpy(), df_graph["dst"].to_numpy()])
    if id_rule == "first_appearance":  # indexer.py:26-35
        names, first = np.unique(both, return_index=True)
        ids = first.astype(np.int64)  # position of the first appearance in [src..., dst...]
        order = np.argsort(ids)
        name_id = pd.DataFrame({"vertex_id": ids[order], "vertex_name": names[order]})
    else:  # indexer.py:69-70
        names = np.unique(both)
        ids = np.arange(len(names), dtype=np.int64)
        name_id = pd.DataFrame({"name": names, "id": ids})
    src = ids[np.searchsorted(names, df_graph["src"].to_numpy())]
    dst = ids[np.searchsorted(names, df_graph["dst"].to_numpy())]
    df_edge = pd.DataFrame({"src": src.astype(np.int64), "dst": dst.astype(np.int64),
                            "weight": df_graph["weight"].to_numpy()})
    if directed is not True:
        rev = df_edge.rename(columns={"src": "dst", "dst": "src"})[["src", "dst", "weight"]]
        df_edge = pd.concat([df_edge, rev], ignore_index=True).drop_duplicates(ignore_index=True)
    return df_edge, name_id


def index_graph_tensors(src: torch.Tensor, dst: torch.Tensor, weight: Optional[torch.Tensor] = None,
                        directed: bool = True, device=None, id_rule: str = "sorted",
                        weight_dtype: torch.dtype = torch.float32):
    """The same indexing for integer-named edge lists that already live in (or fit) device
    memory: sort/unique on the GPU instead of pandas merges, for graphs of 10^8-10^9 edges.

    Returns (src_id int64, dst_id int64, weight float32, names int64) with
    names[id] = original name, ids numbered exactly like index_graph_pandas with the same
    id_rule ("sorted": dense, sorted-name order; "first_appearance": position of the first
    appearance in [src..., dst...], names[id] = -1 for the unused ids in between); undirected
    graphs are symmetrised and de-duplicated on (src, dst, weight) like indexer.py:45-48.  The
    edge order is by (src, dst, weight).

    Weights: the reference's two twins differ here as well -- index_graph_spark casts the column
    with cast("float") (indexer.py:65, 32 bits), index_graph_pandas with astype(float) (:24, 64
    bits).  `weight_dtype` selects the twin: torch.float32 (default, the Spark twin and the 4-byte
    storage form of the kernels) or torch.float64 (the pandas twin: weights like 0.1 keep all
    their bits, and DeviceGraph.from_edges stores them as fp64); de-duplication compares the
    weights in that type."""
    if id_rule not in ID_RULES:
        raise ValueError(f"unknown id_rule {id_rule!r}")
    if weight_dtype not in (torch.float32, torch.float64):
        raise ValueError("weight_dtype is torch.float32 (Spark twin) or torch.float64 (pandas twin)")
    src = torch.as_tensor(src).to(device=device, dtype=torch.int64).reshape(-1)
    dst = torch.as_tensor(dst).to(device=device, dtype=torch.int64).reshape(-1)
    if src.numel() != dst.numel():
        raise ValueError("src and dst differ in length")
    if weight is None:
        w = torch.ones(src.numel(), dtype=weight_dtype, device=src.device)  # indexer.py:20-21
    else:
        if not torch.is_tensor(weight):  # (torch would read a list of Python floats as fp32)
            weight = np.asarray(weight.to_numpy() if hasattr(weight, "to_numpy") else weight, dtype=np.float64)
        w = torch.as_tensor(weight).to(device=src.device, dtype=weight_dtype).reshape(-1)
        if w.numel() != src.numel():
            raise ValueError("weight differs in length from src")
    names, inverse = torch.unique(torch.cat([src, dst]), sorted=True, return_inverse=True)
    if id_rule == "first_appearance":
        pos = torch.arange(inverse.numel(), dtype=torch.int64, device=inverse.device)
        first = torch.full((names.numel(),), inverse.numel(), dtype=torch.int64, device=inverse.device)
        first.scatter_reduce_(0, inverse, pos, reduce="amin")
        inverse = first[inverse]
        table = torch.full((int(first.max()) + 1 if first.numel() else 0,), -1, dtype=torch.int64,
                           device=inverse.device)
        table[first] = names
        names = table
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


def index_graph_names(src, dst, weight=None, directed: bool = True, device=None,
                      id_rule: str = "sorted", weight_dtype: torch.dtype = torch.float32,
                      chunk_rows: int = 1 << 24):
    """index_graph_tensors for vertex NAMES of any type -- strings as in the reference's own test
    input ('a1', 'a2', ...: tests/test_indexer.py:14-16) -- at sizes where the pandas twin's
    merges (indexer.py:36-41) do not finish.  The names are dictionary-encoded on the host by
    pyarrow's hash table, `chunk_rows` at a time with one unified dictionary (an exact code per
    distinct name: nothing is hashed down to a key that could collide); everything per EDGE -- id
    lookup, symmetrising, de-duplication -- then runs on the device on the integer codes.

    Returns (src_id int64, dst_id int64, weight, names) with names a numpy object array,
    names[id] = the vertex name (id_rule "first_appearance": None at the unused ids in between),
    ids numbered exactly like index_graph_pandas with the same id_rule."""
    import pyarrow as pa

    if id_rule not in ID_RULES:
        raise ValueError(f"unknown id_rule {id_rule!r}")

    def chunks(col):
        col = col.to_numpy() if hasattr(col, "to_numpy") else np.asarray(col)
        return [pa.array(col[lo:lo + chunk_rows]) for lo in range(0, max(len(col), 1), chunk_rows)]

    n_src = len(src)
    if n_src != len(dst):
        raise ValueError("src and dst differ in length")
    enc = pa.chunked_array(chunks(src) + chunks(dst)).dictionary_encode().unify_dictionaries()
    dictionary = enc.chunk(0).dictionary.to_numpy(zero_copy_only=False)
    codes = np.concatenate([c.indices.to_numpy(zero_copy_only=False) for c in enc.chunks]).astype(np.int64)
    if id_rule == "sorted":
        # dense ids in sorted-name order (index_graph_pandas above / the Spark twin): rank of every
        # distinct name, on the host -- V names, not 2 E
        order = np.argsort(dictionary, kind="stable")
        rank = np.empty(len(order), dtype=np.int64)
        rank[order] = np.arange(len(order), dtype=np.int64)
        codes = rank[codes]
        by_id = dictionary[order]
    codes = torch.from_numpy(codes).to(device)
    s_id, d_id, w, code_of_id = index_graph_tensors(codes[:n_src], codes[n_src:], weight, directed,
                                                    device, id_rule, weight_dtype)
    if id_rule == "sorted":
        names = by_id  # code_of_id is 0 .. V-1
    else:
        c = code_of_id.cpu().numpy()
        names = np.full(len(c), None, dtype=object)
        names[c >= 0] = dictionary[c[c >= 0]]
    return s_id, d_id, w, names
