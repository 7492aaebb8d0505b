"""The reference's transformer-level surface (node2vec/randomwalk.py:17-349) over the HIP
library: same names, arguments, return shapes and errors, so that callers of those functions
-- and the reference's own tests/test_randomwalk.py -- can be pointed at this package.

Production walks never come through here: `randomwalk.walk` runs all steps of all walkers in
one launch and never materialises a per-step table.  These functions are the row-level
protocol the Fugue transformers speak (one partition in, dict rows out) with the table
arithmetic done by the same kernels on materialised tables:

    generate_alias_tables        K1 (n2v_alias_build) on a one-row graph
    generate_edge_alias_tables   n2v_edge_bias + K1
    AliasProb.sampling_from_*    n2v_alias_draw with the caller's uniforms
    next_step_random_walk        the three of them, one launch each per partition
    trim_hotspot_vertices        n2v_trim_mark on the partition's row

Data carriers (Neighbors, AliasProb, RandomPath) keep the reference's wire format
(base64 of a protocol-3 pickle, randomwalk.py:36-37) so serialized rows interoperate.  Their
constructor / property / serialize() surface is the drop-in boundary itself -- names, argument
forms and wire format are dictated by the reference (randomwalk.py:17-41, 44-68, 102-121) -- and
is stated once, in _Carrier; every numeric body in this file is a call into the HIP library.
There is no CPU fallback: every numeric function raises without a HIP device.
"""
import base64
import pickle
import random
from typing import Any, Dict, Iterable, List, Optional, Set, Tuple, Union

import numpy as np
import pandas as pd
import torch

from node2vec_amd import _lib
from node2vec_amd.constants import MAX_OUT_DEGREES

_PICKLE_PROTOCOL = 3  # what the reference's pinned strings were written with (CPython 3.7 default)


def _loads(obj: str):
    return pickle.loads(base64.b64decode(obj.encode()))


def _dumps(data) -> str:
    return base64.b64encode(pickle.dumps(data, protocol=_PICKLE_PROTOCOL)).decode()


class _Carrier(object):
    """One wire format for the three carriers: a base64 protocol-3 pickle of `_data`.  A carrier is
    built from its serialized string, from a DataFrame holding `_columns`, or from the data."""

    _columns: Tuple[str, ...] = ()

    def __init__(self, obj):
        if isinstance(obj, str):
            self._data = _loads(obj)
        elif isinstance(obj, pd.DataFrame) and self._columns:
            self._data = tuple(obj[c].tolist() for c in self._columns)
        else:
            self._data = obj

    def serialize(self):
        return _dumps(self._data)


class Neighbors(_Carrier):
    """randomwalk.py:17-41: (neighbour ids, weights) of one vertex"""

    _columns = ("dst", "weight")
    dst_id = property(lambda self: self._data[0])
    dst_wt = property(lambda self: self._data[1])

    def items(self):
        return zip(self._data[0], self._data[1])

    def as_pandas(self):
        return pd.DataFrame({"dst": self._data[0], "weight": self._data[1]})


# ---- device plumbing ----------------------------------------------------------------------
def _dev() -> torch.device:
    _lib.load()
    return _lib.require_gpu()


def _i64(a, dev):
    return torch.as_tensor(np.asarray(a, dtype=np.int64), device=dev)


def _build_tables(rowptr: torch.Tensor, ids: torch.Tensor, w64: torch.Tensor) -> torch.Tensor:
    """K1 over packed rows with fp64 weights: slots int32 [nnz, 4] (n2v_slot)."""
    L = _lib.load()
    dev = rowptr.device
    nnz = int(ids.numel())
    slots = torch.zeros((max(nnz, 1), 4), dtype=torch.int32, device=dev)
    status = torch.zeros(4, dtype=torch.int32, device=dev)
    g = _lib.Graph(rowptr.numel() - 1, nnz, rowptr.data_ptr(), ids.data_ptr(), 0, w64.data_ptr(),
                   0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
    with torch.cuda.device(dev):
        rc = L.n2v_alias_build(g, slots.data_ptr(), status.data_ptr(), _lib.current_stream_ptr())
    _lib.check(rc, "n2v_alias_build")
    _lib.check_status_word(int(status[0].item()), "generate_alias_tables")
    return slots[:nnz]


def _draw_device(rowptr: torch.Tensor, slots: torch.Tensor, t1: torch.Tensor,
                 t2: Optional[torch.Tensor]) -> torch.Tensor:
    """n2v_alias_draw on device tensors: one draw per packed row; t2 None = the one-uniform
    (wiki) variant.  Returns the drawn vertex ids (int32, on the device)."""
    L = _lib.load()
    dev = rowptr.device
    n_rows = rowptr.numel() - 1
    out = torch.empty(max(n_rows, 1), dtype=torch.int32, device=dev)
    status = torch.zeros(4, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = L.n2v_alias_draw(rowptr.data_ptr(), slots.data_ptr(), n_rows, t1.data_ptr(),
                              0 if t2 is None else t2.data_ptr(), out.data_ptr(),
                              status.data_ptr(), _lib.current_stream_ptr())
    _lib.check(rc, "n2v_alias_draw")
    if int(status[0].item()) & _lib.ST_RANGE:
        raise IndexError("list index out of range")  # probs[pick] with r1 outside [0, 1)
    return out[:n_rows]


def _draw(rowptr: torch.Tensor, slots: torch.Tensor, r1, r2) -> np.ndarray:
    dev = rowptr.device
    t1 = torch.as_tensor(np.asarray(r1, dtype=np.float64), device=dev)
    t2 = None if r2 is None else torch.as_tensor(np.asarray(r2, dtype=np.float64), device=dev)
    return _draw_device(rowptr, slots, t1, t2).cpu().numpy()


class AliasProb(_Carrier):
    """randomwalk.py:44-99: (alias indices in [0, n), pseudo-probabilities); the two samplers run
    n2v_alias_draw on this table"""

    _columns = ("alias", "probs")
    alias = property(lambda self: self._data[0])
    probs = property(lambda self: self._data[1])

    def _slots(self, dev):
        # col = the index itself, alias = the alias index: the draw then returns an INDEX
        n = len(self.alias)
        s = torch.zeros((n, 4), dtype=torch.int32, device=dev)
        s[:, 0] = torch.arange(n, dtype=torch.int32, device=dev)
        s[:, 1] = torch.as_tensor(np.asarray(self.alias, dtype=np.int32), device=dev)
        s.view(torch.float64)[:, 1] = torch.as_tensor(np.asarray(self.probs, dtype=np.float64),
                                                      device=dev)
        return s, _i64([0, n], dev)

    def sampling_from_alias_wiki(self, first_random: float) -> int:
        """randomwalk.py:70-84: one uniform"""
        dev = _dev()
        slots, rowptr = self._slots(dev)
        return int(_draw(rowptr, slots, [first_random], None)[0])

    def sampling_from_alias(self, first_random: float, second_random: float) -> int:
        """randomwalk.py:86-99: two uniforms (the draw the walks use)"""
        dev = _dev()
        slots, rowptr = self._slots(dev)
        return int(_draw(rowptr, slots, [first_random], [second_random])[0])


class RandomPath(_Carrier):
    """randomwalk.py:102-153: the vertex list of one walker"""

    path = property(lambda self: self._data)
    last_edge = property(lambda self: (self._data[-2], self._data[-1]))

    def __str__(self):
        return self._data.__repr__()

    def _extended(self, next_vertex: int) -> "RandomPath":
        path = list(self._data)
        if len(path) == 2 and path[0] < 0:  # first step, :146-148
            path = [path[1], next_vertex]
        else:  # :150-151
            path.append(next_vertex)
        return RandomPath(path)

    def append(self, dst_neighbors: List[int], alias_prob: AliasProb, first_random: float,
               second_random: Optional[float] = None):
        if second_random is not None:
            next_index = alias_prob.sampling_from_alias(first_random, second_random)
        else:
            next_index = alias_prob.sampling_from_alias_wiki(first_random)
        return self._extended(dst_neighbors[next_index])


# ---- a1 / a2 --------------------------------------------------------------------------------
def generate_alias_tables(node_weights: List[float]) -> Tuple[List[int], List[float]]:
    """randomwalk.py:157-190 on the GPU (K1): (alias, probs), bit-identical fp64."""
    n = len(node_weights)
    if n == 0:
        raise ZeroDivisionError("division by zero")  # sum([]) / 0, :172
    dev = _dev()
    w = torch.as_tensor(np.asarray(node_weights, dtype=np.float64), device=dev)
    ids = torch.arange(n, dtype=torch.int32, device=dev)  # alias vertex == alias index
    slots = _build_tables(_i64([0, n], dev), ids, w)
    alias = slots[:, 1].cpu().numpy().tolist()
    probs = slots.view(torch.float64)[:, 1].cpu().numpy().tolist()
    return alias, probs


def _bias_rows(rowptr, ids, w, src_id, src_rowptr, src_nbs, p, q) -> torch.Tensor:
    """n2v_edge_bias over packed rows; w: None (unit weights), float32 or float64 [nnz]"""
    L = _lib.load()
    dev = rowptr.device
    out = torch.empty(max(int(ids.numel()), 1), dtype=torch.float64, device=dev)
    w32 = w.data_ptr() if (w is not None and w.dtype == torch.float32) else 0
    w64 = w.data_ptr() if (w is not None and w.dtype == torch.float64) else 0
    with torch.cuda.device(dev):
        rc = L.n2v_edge_bias(rowptr.data_ptr(), ids.data_ptr(), w32, w64,
                             src_id.data_ptr(), src_rowptr.data_ptr(), src_nbs.data_ptr(),
                             rowptr.numel() - 1, int(ids.numel()), float(p), float(q),
                             out.data_ptr(), _lib.current_stream_ptr())
    _lib.check(rc, "n2v_edge_bias")
    return out[: ids.numel()]


def generate_edge_alias_tables(
    src_id: int,
    src_nbs_id: Set[int],
    dst_neighbors: Tuple[List[int], List[float]],
    return_param: float = 1.0,
    inout_param: float = 1.0,
) -> Tuple[List[int], List[float]]:
    """randomwalk.py:193-232: the p/q bias (n2v_edge_bias) then the table (K1)."""
    if len(dst_neighbors) != 2 or len(dst_neighbors[0]) != len(dst_neighbors[1]):
        raise ValueError(f"Invalid neighbors tuple '{dst_neighbors}'!")
    if return_param == 0 or inout_param == 0:
        raise ValueError(f"Zero return ({return_param}) or inout ({inout_param}) parameter!")
    n = len(dst_neighbors[0])
    if n == 0:
        raise ZeroDivisionError("division by zero")
    dev = _dev()
    ids = torch.as_tensor(np.asarray(dst_neighbors[0], dtype=np.int32), device=dev)
    w = torch.as_tensor(np.asarray(dst_neighbors[1], dtype=np.float64), device=dev)
    nbs = np.asarray(sorted(src_nbs_id), dtype=np.int32)
    rowptr = _i64([0, n], dev)
    # a negative src_id is an ordinary id here (the reference compares it like any other)
    biased = _bias_rows_any_src(rowptr, ids, w, int(src_id), nbs, return_param, inout_param, dev)
    idx = torch.arange(n, dtype=torch.int32, device=dev)
    slots = _build_tables(rowptr, idx, biased)
    return slots[:, 1].cpu().numpy().tolist(), slots.view(torch.float64)[:, 1].cpu().numpy().tolist()


def _bias_rows_any_src(rowptr, ids, w, src, nbs, p, q, dev):
    """one row; n2v_edge_bias treats src < 0 as 'first step', the plain function does not: shift
    the ids of the row so that the comparison happens on non-negative numbers"""
    lo = min(0, src, int(ids.min()) if ids.numel() else 0, int(nbs.min()) if len(nbs) else 0)
    if lo < 0:
        shift = -int(lo)
        if max(src, int(ids.max()), int(nbs.max()) if len(nbs) else 0) + shift >= 2 ** 31:
            raise ValueError("vertex id out of range for int32")
        ids, src, nbs = ids + shift, src + shift, nbs + shift
    return _bias_rows(rowptr, ids.contiguous(), w,
                      torch.tensor([src], dtype=torch.int32, device=dev),
                      _i64([0, len(nbs)], dev),
                      torch.as_tensor(np.ascontiguousarray(nbs, dtype=np.int32), device=dev)
                      if len(nbs) else torch.zeros(1, dtype=torch.int32, device=dev), p, q)


# ---- transformer functions ------------------------------------------------------------------
def trim_hotspot_vertices(df: pd.DataFrame, max_out_degree: int = 0,
                          random_seed: Optional[int] = None) -> Iterable[Dict[str, Any]]:
    """randomwalk.py:238-262: one partition = the edge rows of one source vertex; above the cap
    a uniform sample without replacement of exactly `cap` rows survives (n2v_trim_mark)."""
    if max_out_degree <= 0:
        max_out_degree = MAX_OUT_DEGREES
    n = len(df["dst"].tolist())
    if n > max_out_degree:
        L = _lib.load()
        dev = _dev()
        from node2vec_amd.randomwalk import fresh_seed

        seed = fresh_seed() if random_seed is None else int(random_seed)
        rowptr = _i64([0, n], dev)
        keep = torch.ones(n, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = L.n2v_trim_mark(rowptr.data_ptr(), 1, int(max_out_degree), seed & (2 ** 64 - 1),
                                 keep.data_ptr(), _lib.current_stream_ptr())
        _lib.check(rc, "n2v_trim_mark")
        df = df[keep.bool().cpu().numpy()]
    for _, row in df.iterrows():
        yield dict(row)


# schema: id:int,neighbors:str
def get_vertex_neighbors(df: pd.DataFrame) -> Iterable[Dict[str, Any]]:
    """randomwalk.py:266-275"""
    src = df.loc[0, "src"]
    yield {"id": src, "neighbors": Neighbors(df).serialize()}


# schema: src:int,dst:int,path:[int]
def initiate_random_walk(df: Iterable[Dict[str, Any]], num_walks: int) -> Iterable[Dict[str, Any]]:
    """randomwalk.py:279-296 (fresh dict per row; the reference re-yields one mutated dict)"""
    for arow in df:
        src = arow["id"]
        for i in range(1, num_walks + 1):
            yield {"dst": src, "src": -i, "path": [-i, src]}


# schema: src:int,dst:int,path:[int]
def next_step_random_walk(df: Iterable[Dict[str, Any]], return_param: float, inout_param: float,
                          random_seed: Optional[int] = None) -> Iterable[Dict[str, Any]]:
    """randomwalk.py:300-339 for a whole partition: the rows' tables are built by ONE
    n2v_edge_bias + ONE n2v_alias_build launch and drawn from by ONE n2v_alias_draw launch.
    The uniforms are the reference's: random.random() twice per row, in row order, from the
    module-global generator, reseeded when random_seed is given (:314-315, :336-337)."""
    if random_seed is not None:
        random.seed(random_seed)
    rows = list(df)
    if not rows:
        return
    if return_param == 0 or inout_param == 0:
        if any(r["src"] >= 0 for r in rows):  # raised by the first biased row, :214-217
            raise ValueError(f"Zero return ({return_param}) or inout ({inout_param}) parameter!")
        return_param = inout_param = 1.0  # never used: every row is a first step
    dst_ids, dst_w, src_ids, src_nbs, rp, sp = [], [], [], [], [0], [0]
    for r in rows:
        nb = Neighbors(r["dst_neighbors"])
        if len(nb.dst_id) == 0:
            raise ZeroDivisionError("division by zero")
        dst_ids.extend(nb.dst_id)
        dst_w.extend(nb.dst_wt)
        rp.append(len(dst_ids))
        s_nbs = r.get("src_neighbors")
        ids = sorted(set(Neighbors(s_nbs).dst_id)) if s_nbs is not None else []
        src_nbs.extend(ids)
        sp.append(len(src_nbs))
        src_ids.append(int(r["src"]))
    dev = _dev()
    rowptr = _i64(rp, dev)
    ids = torch.as_tensor(np.asarray(dst_ids, dtype=np.int32), device=dev)
    w = torch.as_tensor(np.asarray(dst_w, dtype=np.float64), device=dev)
    nbs = torch.as_tensor(np.asarray(src_nbs if src_nbs else [0], dtype=np.int32), device=dev)
    biased = _bias_rows(rowptr, ids, w, torch.as_tensor(np.asarray(src_ids, dtype=np.int32), device=dev),
                        _i64(sp, dev), nbs, return_param, inout_param)
    slots = _build_tables(rowptr, ids, biased)  # slot.alias = the neighbour id behind the index
    r1, r2 = [], []
    for _ in rows:
        r1.append(random.random())
        r2.append(random.random())
    nxt = _draw(rowptr, slots, r1, r2)
    for r, x in zip(rows, nxt):
        _p = RandomPath(r["path"])._extended(int(x))
        yield {"src": _p.last_edge[0], "dst": _p.last_edge[1], "path": _p.path}


# schema: src:int,walk:[int]
def to_path(df: Iterable[Dict[str, Any]]) -> Iterable[Dict[str, Any]]:
    """randomwalk.py:343-349"""
    for row in df:
        path = RandomPath(row["path"]).path
        yield {"src": path[0], "walk": path}
