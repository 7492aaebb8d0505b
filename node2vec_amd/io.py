"""Stage files between the three steps of the reference's example pipeline
(examples/fugue_spark.py:35-75: `index | walk | embed`, each stage reading and writing
parquet): the indexed edge list, the name_id map, the walks [src, walk] and the
vectors [id|name, vector].  Walk tensors can be written straight from the device
corpus without building Python lists row by row."""
import os
from typing import Optional, Tuple

import numpy as np
import pandas as pd
import pyarrow as pa
import pyarrow.parquet as pq
import torch


def write_table(df: pd.DataFrame, path: str) -> None:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    df.to_parquet(path, index=False)


def read_table(path: str) -> pd.DataFrame:
    return pd.read_parquet(path)


def write_walks(path: str, walks: torch.Tensor, valid: Optional[torch.Tensor] = None) -> int:
    """walks int32 [n, L+1] (+ optional validity mask) -> parquet with the reference's
    schema src:int, walk:[int] (randomwalk.py:342).  Returns the number of rows written."""
    if valid is not None:
        walks = walks[valid.bool()]
    w = walks.cpu().numpy().astype(np.int64, copy=False)
    n, ln = w.shape
    offsets = pa.array(np.arange(0, (n + 1) * ln, ln, dtype=np.int32))
    col = pa.ListArray.from_arrays(offsets, pa.array(w.reshape(-1)))
    table = pa.table({"src": pa.array(w[:, 0] if n else np.zeros(0, np.int64)), "walk": col})
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    pq.write_table(table, path)
    return n


def read_walks(path: str, device=None) -> torch.Tensor:
    """parquet [src, walk] with equal-length walks -> int32 tensor [n, L+1]
    (embedding.py:125 needs equal lengths too)."""
    col = pq.read_table(path, columns=["walk"]).column("walk").combine_chunks()
    flat = col.flatten().to_numpy(zero_copy_only=False)
    n = len(col)
    if n == 0:
        return torch.zeros((0, 0), dtype=torch.int32, device=device)
    if len(flat) % n:
        raise ValueError("walks must all have the same length")
    return torch.from_numpy(flat.astype(np.int32).reshape(n, -1)).to(device)


def write_vectors(path: str, df_vectors: pd.DataFrame) -> None:
    """[id|name, vector] as produced by Node2Vec*.embedding() (embedding.py:137-143)"""
    write_table(df_vectors, path)


def read_edges(path: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    df = read_table(path)
    return df["src"].to_numpy(), df["dst"].to_numpy(), df["weight"].to_numpy()
