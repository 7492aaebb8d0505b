"""Public walk API: the reference's node2vec/fugue.py on one MI355X node.

    trim_index(compute_engine, df_graph, indexed, directed, max_out_deg, random_seed)
    random_walk(compute_engine, df_graph, n2v_params, walk_seed, random_seed, checkpoint_dir)

Same signatures, defaults, dict-filling side effects and errors as fugue.py:24-31 /
:81-88.  `compute_engine` is kept for signature compatibility: None, "hip",
a torch.device or a GPU index select the device (the reference passes a Fugue
ExecutionEngine, whose join/transform loop is replaced by one kernel launch).
DataFrames are pandas in, pandas out.
"""
import logging
from typing import Any, Dict, Optional, Tuple

import numpy as np
import pandas as pd
import torch

from node2vec_amd import _lib, corpus
from node2vec_amd import randomwalk as rw
from node2vec_amd.constants import MAX_OUT_DEGREES, NODE2VEC_PARAMS
from node2vec_amd.graph import DeviceGraph
from node2vec_amd.indexer import index_graph_pandas


def _device(compute_engine) -> torch.device:
    _lib.require_gpu()
    if compute_engine is None or (isinstance(compute_engine, str) and compute_engine in ("hip", "cuda")):
        return torch.device("cuda", torch.cuda.current_device())
    if isinstance(compute_engine, int):
        return torch.device("cuda", compute_engine)
    return torch.device(compute_engine)


def _as_pandas(df) -> pd.DataFrame:
    if isinstance(df, pd.DataFrame):
        return df
    if hasattr(df, "as_pandas"):  # Fugue DataFrame
        return df.as_pandas()
    if hasattr(df, "to_pandas"):  # pyarrow Table
        return df.to_pandas()
    return pd.DataFrame(df)


def trim_hotspot_edges(src: torch.Tensor, max_out_degree: int, seed: int) -> torch.Tensor:
    """Device form of trim_hotspot_vertices (randomwalk.py:238-262): returns a bool
    mask over edges; rows above the cap keep a uniform sample without replacement of
    exactly `cap` edges (n2v_trim_mark), other rows keep everything."""
    L = _lib.load()
    if max_out_degree <= 0:
        max_out_degree = MAX_OUT_DEGREES  # randomwalk.py:252-253
    order = torch.sort(src, stable=True).indices
    n_rows = int(src.max()) + 1 if src.numel() else 0
    counts = torch.bincount(src, minlength=n_rows)
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=src.device)
    torch.cumsum(counts, 0, out=rowptr[1:])
    keep_sorted = torch.ones(src.numel(), dtype=torch.uint8, device=src.device)
    if src.numel() and int(counts.max()) > max_out_degree:
        with torch.cuda.device(src.device):
            rc = L.n2v_trim_mark(rowptr.data_ptr(), n_rows, int(max_out_degree),
                                 seed & (2 ** 64 - 1), keep_sorted.data_ptr(),
                                 _lib.current_stream_ptr())
        _lib.check(rc, "n2v_trim_mark")
    keep = torch.empty_like(keep_sorted)
    keep[order] = keep_sorted
    return keep.bool()


def trim_index(
    compute_engine,
    df_graph,
    indexed: bool = False,
    directed: bool = True,
    max_out_deg: int = 0,
    random_seed: Optional[int] = None,
    id_rule: str = "sorted",
) -> Tuple[pd.DataFrame, Optional[pd.DataFrame]]:
    """fugue.py:24-77: validate, trim hotspot vertices, index.  Returns
    (edges[src:int, dst:int, weight:float], name_id[name, id]) or (trimmed df, None)
    when the graph is already indexed.  `id_rule` (not in the reference's signature) selects
    the numbering of indexer.index_graph_pandas: "sorted" (Spark twin) or "first_appearance"
    (pandas twin)."""
    logging.info("trim_index(): start validating, trimming, and indexing ...")
    df = _as_pandas(df_graph)
    if "src" not in df.columns or "dst" not in df.columns:
        raise ValueError(f"Input graph NOT in the right format: {list(df.columns)}")
    dev = _device(compute_engine)
    seed = rw.fresh_seed() if random_seed is None else int(random_seed)
    # the sampling is per source vertex and independent of its label: trim on a
    # dense relabelling of src, keep the caller's columns
    codes = torch.from_numpy(np.unique(df["src"].to_numpy(), return_inverse=True)[1]).to(dev)
    keep = trim_hotspot_edges(codes.long(), max_out_deg, seed).cpu().numpy()
    df = df[keep].reset_index(drop=True)
    if indexed is True:
        return df, None  # fugue.py:70-71
    return index_graph_pandas(df, directed, id_rule=id_rule)


def random_walk_tensors(graph: DeviceGraph, n2v_params: Dict[str, Any], walk_seed_ids=None,
                        random_seed: Optional[int] = None, mode: str = "exact", shard: bool = True):
    """The on-device corpus: (walks int32 [n, L+1], valid bool [n])
    stay in HBM, ready for the SGNS kernel; no DataFrame is materialised.

    Under an initialised torch.distributed process group (one process per GPU, graph
    replicated) each rank walks its contiguous range of the start vertices
    (shard.shard_range).  The walker RNG is keyed by (seed, start vertex, ordinal), so the
    union of the ranks' walks equals the single-GPU result and no collective is needed;
    pass the SAME random_seed on every rank.  shard=False walks everything on this rank."""
    for param in NODE2VEC_PARAMS:  # fugue.py:120-122: fills the caller's dict
        if param not in n2v_params:
            n2v_params[param] = NODE2VEC_PARAMS[param]
    seed = rw.fresh_seed() if random_seed is None else int(random_seed)
    start = rw.start_vertices(graph, walk_seed_ids)
    import torch.distributed as dist

    if shard and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if random_seed is None:
            raise ValueError("random_seed must be given (the same on every rank) when walks are sharded")
        from node2vec_amd.shard import shard_range

        lo, hi = shard_range(start.numel(), dist.get_rank(), dist.get_world_size())
        start = start[lo:hi].contiguous()
    return rw.walk(graph, start, int(n2v_params["num_walks"]), int(n2v_params["walk_length"]),
                   float(n2v_params["return_param"]), float(n2v_params["inout_param"]), seed, mode)


def random_walk(
    compute_engine,
    df_graph,
    n2v_params: Dict[str, Any],
    walk_seed=None,
    random_seed: Optional[int] = None,
    checkpoint_dir: Optional[str] = "/tmp",
    mode: str = "exact",
) -> pd.DataFrame:
    """fugue.py:81-155.  df_graph: indexed edges [src:int, dst:int, weight:float].
    Returns DataFrame ["src", "walk"], one row per surviving walker, every walk with
    walk_length + 1 vertices.  `checkpoint_dir` is accepted and unused: the walker
    state never leaves the GPU, so there is no lineage to checkpoint (fugue.py:149)."""
    logging.info("random_walk(): start random walking ...")
    for param in NODE2VEC_PARAMS:
        if param not in n2v_params:
            n2v_params[param] = NODE2VEC_PARAMS[param]
    seed_ids = None
    if walk_seed is not None:
        ws = _as_pandas(walk_seed)
        if "id" not in ws.columns:  # fugue.py:123-124
            raise ValueError(f"walk_seed has no column of 'id': {list(ws.columns)}!")
        seed_ids = ws["id"].to_numpy()
    dev = _device(compute_engine)
    graph = df_graph if isinstance(df_graph, DeviceGraph) else DeviceGraph.from_pandas(
        _as_pandas(df_graph), device=dev)
    walks, valid = random_walk_tensors(graph, n2v_params, seed_ids, random_seed, mode)
    kept = walks[valid]
    w = kept.cpu().numpy()
    logging.info("random_walk(): random walking done ...")
    # to_path, randomwalk.py:343-349: {"src": path[0], "walk": path}.  Up to corpus.LIST_COLUMN_MAX_VALUES
    # vertices the column holds Python lists (ndarray.tolist(): one C call); beyond, one read-only ndarray view
    # per row into the D2H buffer (no Python object per vertex -- 14 GB and ~20 s at BASELINE cfg 2; what
    # pandas.read_parquet yields for a list column).  n2v_params["walk_column"] = "list" | "rows" | "arrow"
    # forces a form (corpus.list_column).
    df = pd.DataFrame({"src": w[:, 0].astype(np.int64) if len(w) else np.zeros(0, np.int64),
                       "walk": corpus.list_column(w, str(n2v_params.get("walk_column", "auto")))})
    # the same walks as an on-device corpus: Node2VecHIP.fit() trains from it when THIS frame
    # reaches it unchanged, instead of converting the list column back (embedding.py:125).  The
    # frame carries a plain integer token only (pandas copies / compares / serialises attrs).
    corpus.attach(df, kept)
    return df
