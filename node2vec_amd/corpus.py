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
import torch

ATTR = "n2v_device_walks"
_tokens = itertools.count(1)
_registry = {}  # token -> (weakref to the frame, device tensor, ids of the row objects, the row objects)
STRICT = False  # True: lookup() compares the whole column with the tensor


def _row_ids(col) -> np.ndarray:
    return np.fromiter(map(id, col.to_numpy()), dtype=np.int64, count=len(col))


def attach(frame, walks: torch.Tensor) -> None:
    token = next(_tokens)
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
