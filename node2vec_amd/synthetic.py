"""Synthetic graphs of BASELINE.md section 4, generated with torch on
whatever device is asked for.  Inputs for benchmarks and tests only."""
import math
from typing import Optional

import torch

from node2vec_amd.graph import DeviceGraph


def _symmetrize_dedupe_keys(src, dst, n):
    """sorted unique keys src * n + dst of the symmetrised graph without self-loops"""
    keep = src != dst
    src, dst = src[keep], dst[keep]
    key = torch.cat([src * n + dst, dst * n + src])
    del src, dst
    return torch.unique(key)


def _symmetrize_dedupe(src, dst, n):
    key = _symmetrize_dedupe_keys(src, dst, n)
    return key // n, key % n


def rmat(scale: int, n_draws: int, a=0.57, b=0.19, c=0.19, seed=42, device=None,
         weights: Optional[str] = None) -> DeviceGraph:
    """R-MAT, `n_draws` directed pairs on 2**scale ids, self-loops dropped,
    symmetrised and de-duplicated (cfg 2: scale 20, 5 M draws -> ~10 M edges)."""
    gen = torch.Generator(device=device or "cpu").manual_seed(seed)
    n = 1 << scale
    src = torch.zeros(n_draws, dtype=torch.int64, device=device)
    dst = torch.zeros(n_draws, dtype=torch.int64, device=device)
    ab, abc = a + b, a + b + c
    for _ in range(scale):
        r = torch.rand(n_draws, generator=gen, device=device)
        src = src * 2 + (r >= ab).long()
        dst = dst * 2 + (((r >= a) & (r < ab)) | (r >= abc)).long()
    if weights != "uniform":
        return DeviceGraph.from_sorted_keys(_symmetrize_dedupe_keys(src, dst, n), n)
    s, d = _symmetrize_dedupe(src, dst, n)
    w = torch.rand(s.numel(), generator=gen, device=device) * 1.9 + 0.1
    return DeviceGraph.from_edges(s, d, w, n_vertices=n, device=device)


def chung_lu(n: int, n_draws: int, gamma=2.1, seed=42, device=None) -> DeviceGraph:
    """Power-law graph: endpoints drawn with probability ~ rank^(-1/(gamma-1))
    (cfg 3/4), symmetrised and de-duplicated."""
    gen = torch.Generator(device=device or "cpu").manual_seed(seed)
    alpha = 1.0 / (gamma - 1.0)
    # inverse-CDF sampling of a continuous power law over ranks [1, n+1)
    e = 1.0 - alpha

    def draw(k):
        u = torch.rand(k, generator=gen, device=device, dtype=torch.float64)
        x = ((n + 1.0) ** e - 1.0) * u + 1.0
        return (x ** (1.0 / e)).long().clamp_(1, n) - 1

    return DeviceGraph.from_sorted_keys(_symmetrize_dedupe_keys(draw(n_draws), draw(n_draws), n), n)


def hub_bipartite(n: int, n_hubs: int, hub_degree: int, seed=42, device=None) -> DeviceGraph:
    """cfg 5: `n_hubs` hubs each linked to `hub_degree` uniform leaves, every leaf
    also attached to one uniform hub; symmetrised."""
    gen = torch.Generator(device=device or "cpu").manual_seed(seed)
    n_leaf = n - n_hubs
    hs = torch.arange(n_hubs, device=device).repeat_interleave(hub_degree)
    ls = torch.randint(0, n_leaf, (n_hubs * hub_degree,), generator=gen, device=device) + n_hubs
    l2 = torch.arange(n_leaf, device=device) + n_hubs
    h2 = torch.randint(0, n_hubs, (n_leaf,), generator=gen, device=device)
    return DeviceGraph.from_sorted_keys(_symmetrize_dedupe_keys(torch.cat([hs, h2]), torch.cat([ls, l2]), n), n)
