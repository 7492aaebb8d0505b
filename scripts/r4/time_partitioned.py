"""steps/s of graph-partitioned walking (node2vec_amd/partitioned.py), cfg 2 graph, 8 parts in one process, every
start vertex, WALKS walks, STEPS steps; the forwarding form (n2v_partition_step + n2v_partition_forward per part and
step, one host read per step) against the launch-per-stage routing; checked against n2v_walk."""
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from node2vec_amd import partitioned as P  # noqa: E402
from node2vec_amd import randomwalk as rw  # noqa: E402
from node2vec_amd import synthetic  # noqa: E402

P.FORWARD_STREAMS = os.environ.get("STREAMS", "1") == "1"
W = int(os.environ.get("WALKS", "10"))
STEPS = int(os.environ.get("STEPS", "20"))
g = synthetic.rmat(20, 5_000_000, device="cuda")
start = rw.start_vertices(g)
parts = P.partition_graph(g, 8)
print("bytes of the largest part:", max(pt.nbytes() for pt in parts), "of", sum(pt.nbytes() for pt in parts), flush=True)
for fw in (True, False):
    P.walk_partitioned_local(parts, start[::50].contiguous(), 1, 3, 0.5, 2.0, 1, forwarding=fw)  # warm-up
PQS = [tuple(float(x) for x in pq.split(",")) for pq in os.environ["PQ"].split(";")] if os.environ.get("PQ") else \
    [(1.0, 1.0), (0.5, 2.0), (0.5, 1.0), (4.0, 0.25), (0.7, 1.3)]
# "ranks" = walk_partitioned's ranks with capacity-bounded mailboxes after the calibration steps (round 5),
# "ranks-exact" = every step with exact sizes and a host read (round 4's form)
FORMS = {"1": (True,), "0": (False,), "ranks": ("ranks", "ranks-exact")}.get(
    os.environ.get("FORWARD", ""), (True, "ranks", "ranks-exact", False))
NAMES = {True: "forwarding", "ranks": "ranks, bounded mailboxes", "ranks-exact": "ranks, exact sizes",
         False: "launch per stage"}
for p, q in PQS:
    want, wv = rw.walk(g, start, W, STEPS, p, q, 42)
    for fw in FORMS:
        best, t = None, {}
        P.BOUNDED = fw != "ranks-exact"
        for rep in range(2):
            t = {"sections": {}} if os.environ.get("SECTIONS") and rep == 1 else {}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            walks, valid = P.walk_partitioned_local(parts, start, W, STEPS, p, q, 42,
                                                    forwarding="ranks" if fw == "ranks-exact" else fw, timings=t)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        ok = torch.equal(valid, wv) and torch.equal(walks, want)
        print(f"  p={p} q={q} {NAMES[fw]}: {int(valid.sum())} walkers x {STEPS} steps in "
              f"{best * 1e3:.1f} ms = {int(valid.sum()) * STEPS / best / 1e9:.3f} G steps/s, bit-identical to n2v_walk: {ok}"
              + (f"; attempts {len(t['bounded_caps'])}, slots / words of all boxes "
                 f"{[(sum(map(sum, h)), sum(map(sum, w))) for h, w in t['bounded_caps']]}" if t.get("bounded_caps") else "")
              + (f"; ms per section {({k: round(v * 1e3, 1) for k, v in t['sections'].items()})}" if t.get("sections") else ""),
              flush=True)
