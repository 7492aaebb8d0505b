"""Targeted fuzz of the margin kernels of weighted exact walks (csrc/n2v_walk_wlanes.hip): random graphs with hubs on
both sides of the cut between the lane and the wave kernel (768 slots), multi-edges, sinks, five kinds of weights,
p and q dyadic or not -- the step-synchronous walk with margins against the one-launch wave-per-walker kernel of
n2v_walk (itself checked against the oracle by tests/ and scripts/fuzz_walk.py), bit for bit.
  python scripts/r5/fuzz_weighted_margins.py [cases] [seed]"""
import os, sys, time
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from node2vec_amd import randomwalk as rw
from node2vec_amd.graph import DeviceGraph

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
PQ = [0.25, 0.5, 0.7, 1.0, 1.3, 2.0, 3.0, 4.0]
t0 = time.time()
walks_total = undecided_total = second_total = 0
for case in range(cases):
    nv = int(rng.integers(200, 4000))
    m = int(rng.integers(nv, 12 * nv))
    n_hubs = int(rng.integers(1, 6))
    hubs = rng.integers(0, nv, n_hubs)
    hub_deg = rng.integers(300, 6000, n_hubs)
    src = np.concatenate([rng.integers(0, nv, m), np.repeat(hubs, hub_deg)])
    dst = np.concatenate([rng.integers(0, nv, m), rng.integers(0, nv, int(hub_deg.sum()))])
    if rng.random() < 0.5:  # sinks
        keep = src % 13 != 5
        src, dst = src[keep], dst[keep]
    kind = rng.choice(["fp32", "fp64", "few", "wide", "zeros"])
    k = src.size
    if kind == "fp32":
        w = (rng.random(k) * 1.9 + 0.1).astype(np.float32)
    elif kind == "fp64":
        w = rng.random(k) * 1.9 + 0.1
    elif kind == "few":
        w = rng.integers(1, 5, k).astype(np.float32)
    elif kind == "wide":
        w = (10.0 ** rng.uniform(-8, 8, k)).astype(np.float32)
    else:
        w = ((rng.random(k) + 0.05) * (rng.random(k) > 0.2)).astype(np.float32)
    if rng.random() < 0.5:  # symmetric
        src, dst, w = np.concatenate([src, dst]), np.concatenate([dst, src]), np.concatenate([w, w])
    g = DeviceGraph.from_edges(src, dst, w, n_vertices=nv, device="cuda")
    p, q = float(rng.choice(PQ)), float(rng.choice(PQ))
    if p == 1.0 and q == 1.0:
        q = 2.0
    start = rw.start_vertices(g)
    if kind == "zeros":  # rows that sum to 0 raise in both paths alike: walk from vertices whose rows cannot
        sums = rw.weighted_row_sums(g)
        if sums is not None and bool((sums[:nv][g.degrees() > 0] <= 0).any()):
            continue
    nw, wl, seed = int(rng.integers(1, 5)), int(rng.integers(2, 30)), int(rng.integers(0, 2 ** 62))
    st = {}
    try:
        a, av = rw.walk(g, start, nw, wl, p, q, seed, use_weighted_lanes=True, stats=st)
        b, bv = rw.walk(g, start, nw, wl, p, q, seed, use_weighted_lanes=False)
    except ZeroDivisionError:
        continue
    if not (torch.equal(av, bv) and torch.equal(a, b)):
        bad = torch.nonzero((a != b).any(1)).reshape(-1)
        print(f"MISMATCH case {case}: {kind} weights, p={p} q={q}, nv={nv}, max degree {int(g.degrees().max())}, "
              f"rows differing {bad.numel()}: {bad[:5].tolist()}")
        sys.exit(1)
    walks_total += int(av.sum())
    undecided_total += int(st["undecided"])
    second_total += int(st["second_chance"])
print(f"weighted margins fuzz ok: {cases} cases, {walks_total} walks bit-identical to the wave-per-walker kernel; "
      f"{second_total} walker-steps had the second chance, {undecided_total} went to the exact kernel; {time.time() - t0:.0f} s")
