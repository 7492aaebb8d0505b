"""Looks for rare outlier runs of the hogwild trainer on the planted-partition case of
tests/test_sgns_gpu.py: N runs, Procrustes cosine (unit rows) against the first one, largest
component and row norm; prints every run below 0.93 and the summary."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_sgns_gpu as TS  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
case = TS.planted_case()
ref = case["train"](False)
cos, t0 = [], time.time()
for i in range(N):
    if i % 50 == 25:  # an idle gap, as after a long single-wave launch: clocks fall back
        time.sleep(3.0)
    h = case["train"](False)
    c = case["procrustes"](ref, h)
    cos.append(c)
    if c < 0.93:
        nr = np.linalg.norm(h, axis=1)
        print(f"run {i}: cos {c:.4f} AUC {case['auc'](h):.4f} max|x| {np.abs(h).max():.3f} "
              f"max row norm {nr.max():.3f} median {np.median(nr):.3f} "
              f"worst dims {np.argsort(-np.abs(h).max(0))[:4].tolist()}", flush=True)
    if i % 50 == 0:
        print(f"... {i} runs, {time.time() - t0:.0f} s", flush=True)
cos = np.asarray(cos)
print(f"n={N} mean={cos.mean():.4f} sd={cos.std(ddof=1):.4f} min={cos.min():.4f} max={cos.max():.4f} "
      f"below 0.93: {(cos < 0.93).sum()}")
