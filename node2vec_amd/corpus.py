"""The on-device walk corpus behind a DataFrame of walks.

random_walk() returns the reference's DataFrame [src, walk] (fugue.py:155); the same walks are
still in HBM as an int32 tensor, and Node2VecHIP.fit() trains from that tensor instead of
converting the list column back (embedding.py:125).  The tensor is NOT stored on the frame:
pandas treats DataFrame.attrs as plain metadata (it is deep-copied into every derived frame,
compared with == by concat / merge, and serialised as JSON by to_parquet).  The frame carries an
opaque integer token; this module maps token -> (weak reference to the frame, tensor) and hands
the tensor out only to the very frame object random_walk() returned, with a sampled content check.
The entry dies with the frame.
"""
import itertools
import weakref
from typing import Optional

import numpy as np
import torch

ATTR = "n2v_device_walks"
_tokens = itertools.count(1)
_registry = {}  # token -> (weakref to the frame, device tensor)


def attach(frame, walks: torch.Tensor) -> None:
    token = next(_tokens)
    _registry[token] = (weakref.ref(frame), walks)
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
    rows = np.unique(np.r_[0, n - 1, np.random.default_rng(token).integers(0, n, 64)])
    want = walks[torch.as_tensor(rows, device=walks.device)].cpu().numpy().tolist()
    col = frame["walk"]
    if [list(col.iloc[int(r)]) for r in rows] != want:
        return None
    return walks
