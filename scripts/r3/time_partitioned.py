"""steps/s of graph-partitioned walking (node2vec_amd/partitioned.py) with a batch large enough
that the per-step launch overhead of its torch plumbing is amortised: cfg 2 graph, 8 parts in one
process, every start vertex, 2 walks, 20 steps; checked against n2v_walk."""
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from node2vec_amd import partitioned as P  # noqa: E402
from node2vec_amd import randomwalk as rw  # noqa: E402
from node2vec_amd import synthetic  # noqa: E402

W = int(os.environ.get("WALKS", "2"))  # walks per start vertex: 2 -> 9.4e5 walkers, 10 -> 4.7e6
g = synthetic.rmat(20, 5_000_000, device="cuda")
start = rw.start_vertices(g)
for wedges in (True, False):
    parts = P.partition_graph(g, 8, wedges=wedges)
    print("walkers carry", "wedge lists" if parts[0].wedge_off is not None else "whole rows",
          "when q != 1; bytes of the largest part:", max(pt.nbytes() for pt in parts), flush=True)
    P.walk_partitioned_local(parts, start[::50].contiguous(), 1, 3, 0.5, 2.0, 1)  # warm-up (first launches)
    for p, q in ((1.0, 1.0), (0.5, 2.0), (0.5, 1.0), (4.0, 0.25), (0.7, 1.3)):
        want, wv = rw.walk(g, start, W, 20, p, q, 42)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        walks, valid = P.walk_partitioned_local(parts, start, W, 20, p, q, 42)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ok = torch.equal(valid, wv) and torch.equal(walks, want)
        print(f"  p={p} q={q}: {int(valid.sum())} walkers x 20 steps in {dt:.2f} s = "
              f"{int(valid.sum()) * 20 / dt / 1e6:.2f} M steps/s, bit-identical to n2v_walk: {ok}", flush=True)
