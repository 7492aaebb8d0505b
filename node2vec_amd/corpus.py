"""The on-device walk corpus behind a DataFrame of walks.

random_walk() returns the reference's DataFrame [src, walk] (fugue.py:155); the same walks are
still in HBM as an int32 tensor, and Node2VecHIP.fit() trains from that tensor instead of
converting the list column back (embedding.py:125).  The tensor is NOT stored on the frame:
pandas treats DataFrame.attrs as plain metadata (it is deep-copied into every derived frame,
compared with == by concat / merge, and serialised as JSON by to_parquet).  The frame carries an
opaque integer token; this module maps token -> (weak reference to the frame, tensor) and hands
the tensor out only to the very frame object random_walk() returned, after checking that its
"walk" column still is what the tensor says: every row must still be the very list object
random_walk() put there -- the registry keeps a reference to each of them, so an address can never
be reused by another object while the entry lives, and `df.at[i, "walk"] = ...`, masking,
reordering or any other replacement of rows is always seen --, and 64 sampled rows + the ends are
compared element by element (an edit
INSIDE one of the original list objects is only caught on those; set `corpus.STRICT = True` to
compare every element, at the cost of the host conversion the device corpus exists to avoid, or
pass `df.copy()` / drop `df.attrs` to train on the frame's contents).  The entry dies with the frame.
"""
import itertools
import weakref
from typing import Optional

import numpy as np
import pandas as pd
import pyarrow as pa
import torch

# Large results.  A Python list per row of Python ints per vertex is what the reference's frame holds, and at BASELINE
# cfg 2 (4.7 M walks of 81 vertices) that is 3.8 x 10^8 int objects: 14 GB and ~20 s of `ndarray.tolist()` behind a
# 40 ms kernel.  Beyond LIST_COLUMN_MAX_VALUES values the column therefore holds one READ-ONLY ndarray VIEW per row into
# the one host buffer of the D2H copy ("rows": 2 s, 0.5 GB at cfg 2) -- what pandas.read_parquet gives for a list
# column anyway.  `np.array(col.tolist())` (the reference's consumer, embedding.py:125), iteration, len(), indexing and
# to_parquet (schema randomwalk.py:342) work as on lists; an edit inside a row raises (read-only), a replaced row is
# another object: the device corpus behind the frame stays valid exactly while every row is the original object.
# "arrow": an Arrow-backed list<int32> column over the same buffer, O(1) to build and immutable; opt-in, because
# pandas < 3 cannot read back the parquet file it writes from such a column (its own dtype string in the metadata).
LIST_COLUMN_MAX_VALUES = 1 << 22  # rows x length up to which random_walk() / embedding() build Python lists
LIST_COLUMN_KINDS = ("auto", "list", "rows", "arrow")


def list_column(values: np.ndarray, kind: str = "auto"):
    """[n, k] numpy -> the column of n rows.  kind "list": Python lists (ndarray.tolist(), one C call); "rows": an
    object array of read-only ndarray views; "arrow": pandas ArrowExtensionArray over the flat buffer; "auto": lists
    up to LIST_COLUMN_MAX_VALUES values, rows beyond."""
    if kind not in LIST_COLUMN_KINDS:
        raise ValueError(f"list column kind {kind!r}: " + " | ".join(LIST_COLUMN_KINDS))
    n, k = values.shape
    if kind == "auto":
        kind = "list" if n * k <= LIST_COLUMN_MAX_VALUES else "rows"
    if kind == "list":
        return values.tolist()
    values = np.ascontiguousarray(values)
    if kind == "rows":
        values.setflags(write=False)  # (the views inherit it)
        rows = np.empty(n, dtype=object)
        rows[:] = list(values)
        return rows
    flat = pa.array(values.reshape(-1))
    if (n + 1) * k < 2 ** 31:
        arr = pa.ListArray.from_arrays(pa.array(np.arange(0, (n + 1) * k, k, dtype=np.int32)), flat)
    else:
        arr = pa.LargeListArray.from_arrays(pa.array(np.arange(0, (n + 1) * k, k, dtype=np.int64)), flat)
    return pd.arrays.ArrowExtensionArray(arr)


def arrow_rows(col) -> Optional[np.ndarray]:
    """an Arrow-backed list column of equal-length rows as a [n, k] numpy array without touching Python objects;
    None when the column is not Arrow-backed or ragged"""
    if not isinstance(getattr(col, "dtype", None), pd.ArrowDtype):
        return None
    arr = pa.chunked_array(col.array.__arrow_array__()).combine_chunks()
    if not (pa.types.is_list(arr.type) or pa.types.is_large_list(arr.type) or pa.types.is_fixed_size_list(arr.type)):
        return None
    n = len(arr)
    if arr.null_count or n == 0:
        return None
    flat = arr.flatten().to_numpy(zero_copy_only=False)
    if flat.shape[0] % n:
        return None
    k = flat.shape[0] // n
    if not pa.types.is_fixed_size_list(arr.type):
        off = arr.offsets.to_numpy()
        if not np.array_equal(off - off[0], np.arange(n + 1, dtype=off.dtype) * k):
            return None
    return flat.reshape(n, k)


def _arrow_identity(col):
    """(address, offset, length) of the value buffer of every chunk of an Arrow-backed column, or None"""
    if not isinstance(getattr(col, "dtype", None), pd.ArrowDtype):
        return None
    chunks = pa.chunked_array(col.array.__arrow_array__()).chunks
    out = []
    for c in chunks:
        if not (pa.types.is_list(c.type) or pa.types.is_large_list(c.type)):
            return None
        out.append((c.values.buffers()[1].address, c.offset, len(c), c.offsets.buffers()[1].address))
    return tuple(out)

ATTR = "n2v_device_walks"
_tokens = itertools.count(1)
_registry = {}  # token -> (weakref to the frame, device tensor, ids of the row objects, the row objects)
STRICT = False  # True: lookup() compares the whole column with the tensor


def _row_ids(col) -> np.ndarray:
    return np.fromiter(map(id, col.to_numpy()), dtype=np.int64, count=len(col))


def attach(frame, walks: torch.Tensor) -> None:
    token = next(_tokens)
    ident = _arrow_identity(frame["walk"])
    if ident is not None:
        # Arrow-backed column: immutable, so "the same buffers at the same offsets" IS "unchanged"; the registry keeps
        # the array alive, so the addresses cannot be reused while the entry lives
        _registry[token] = (weakref.ref(frame), walks, ident, frame["walk"].array)
        weakref.finalize(frame, _registry.pop, token, None)
        frame.attrs[ATTR] = token
        return
    rows = frame["walk"].to_numpy()
    # (the row objects themselves are kept: CPython reuses the address of a freed list, so bare ids
    # could match a row that was dropped and replaced)
    _registry[token] = (weakref.ref(frame), walks, _row_ids(frame["walk"]), list(rows))
    weakref.finalize(frame, _registry.pop, token, None)
    frame.attrs[ATTR] = token


def lookup(frame) -> Optional[torch.Tensor]:
    """the device walks of `frame`, or None when it is not the frame random_walk() returned or
    its rows no longer say what the tensor says (64 sampled rows + the ends are compared)"""
    token = getattr(frame, "attrs", {}).get(ATTR)
    entry = _registry.get(token) if isinstance(token, int) else None
    if entry is None or entry[0]() is not frame:
        return None
    walks = entry[1]
    n = walks.shape[0]
    if n != len(frame) or n == 0 or "walk" not in frame.columns:
        return None
    col = frame["walk"]
    if isinstance(entry[2], tuple):  # Arrow-backed: the column must still be the very buffers random_walk() made
        return walks if _arrow_identity(col) == entry[2] else None
    if isinstance(col.dtype, pd.ArrowDtype):
        return None
    if not np.array_equal(_row_ids(col), entry[2]):  # some row is no longer the original object
        return None
    if STRICT:
        try:
            host = np.asarray(col.tolist(), dtype=np.int64)
        except ValueError:  # ragged rows
            return None
        if host.shape != tuple(walks.shape):
            return None
        for lo in range(0, n, 1 << 20):
            if not bool((torch.as_tensor(host[lo:lo + (1 << 20)], device=walks.device) == walks[lo:lo + (1 << 20)]).all()):
                return None
        return walks
    rows = np.unique(np.r_[0, n - 1, np.random.default_rng(token).integers(0, n, 64)])
    want = walks[torch.as_tensor(rows, device=walks.device)].cpu().numpy().tolist()
    if [list(col.iloc[int(r)]) for r in rows] != want:
        return None
    return walks
