"""fast mode on a BASELINE graph for several (p, q): G steps/s and trials per step, with the wedge
table (layered sampler) and without it (rejection sampler).  GRAPH=cfg4|cfg3|cfg5|cfg2, 1 M start
vertices x 10 walks x 80 steps per launch."""
import os
import sys
import time

import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from node2vec_amd import randomwalk as rw  # noqa: E402
from node2vec_amd import synthetic  # noqa: E402

cfg = os.environ.get("GRAPH", "cfg4")
if cfg == "cfg4":
    g = synthetic.chung_lu(100_000_000, 500_000_000, device="cuda").trimmed(10_000, 42)
elif cfg == "cfg3":
    g = synthetic.chung_lu(10_000_000, 100_000_000, device="cuda").trimmed(10_000, 42)
elif cfg == "cfg5":
    g = synthetic.hub_bipartite(50_000_000, 5000, 10_000, device="cuda")
else:
    g = synthetic.rmat(20, 5_000_000, device="cuda")
start = rw.start_vertices(g)
b = min(1 << 20, start.numel())
walks = torch.empty((b * 10, 81), dtype=torch.int32, device="cuda")
valid = torch.empty(b * 10, dtype=torch.uint8, device="cuda")
PQ = [(0.5, 2.0), (0.25, 4.0), (1.0, 2.0), (2.0, 2.0), (4.0, 2.0), (2.0, 1.0), (4.0, 0.25), (0.5, 0.25), (3.0, 0.7), (0.7, 1.3)]
if os.environ.get("PQ"):
    PQ = [tuple(float(x) for x in pq.split(",")) for pq in os.environ["PQ"].split(";")]
# layered sampler reading the wedge slots, reading wedge_off + lists, rejection sampler
for wedges, slots in ((True, True), (True, False), (False, False)):
    for p, q in PQ:
        st = {}

        def run(k):
            lo = (k * b) % max(1, start.numel() - b + 1)
            rw.walk(g, start[lo:lo + b], 10, 80, p, q, 42, mode="fast", out=(walks, valid), check=False,
                    use_wedges=wedges, use_wedge_slots=slots, stats=st)

        run(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(1, 6):
            run(k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        steps = int(valid.sum()) * 80
        print(f"{cfg} fast p={p} q={q} {('layered, slots' if slots else 'layered, wedge_off') if wedges and g.wedge_off is not None else 'rejection'}: "
              f"{b * 800 / dt / 1e9:6.2f} G steps/s ({dt * 1e3:6.2f} ms per launch), "
              f"{int(st['trials'].item()) / max(steps, 1):.3f} trials per step", flush=True)
