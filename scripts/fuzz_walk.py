"""Randomised differential test of the exact walk kernels against the CPU oracle
(test infrastructure; run on the GPU box):  python scripts/fuzz_walk.py [seconds] [seed]
FUZZ_PQ=extreme | two | rational: very small / large p, q; the shared-stack arrangements; class values
in small rational ratios on clique-heavy graphs (near-ties of the pairing loop).
FUZZ_PARTITIONED=1: the same cases also through graph-partitioned walking (1-6 parts, cut by edges or
by vertices, n2v_partition_step on every part, walkers migrating) against the same oracle walks."""
import os, sys, time
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import n2v_oracle
from node2vec_amd import randomwalk as rw
from node2vec_amd.graph import DeviceGraph
from node2vec_amd import partitioned as P
PARTITIONED = os.environ.get("FUZZ_PARTITIONED") == "1"

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
THREADS = min(64, len(os.sched_getaffinity(0)))

def graph(kind):
    nv = int(rng.choice([50, 400, 3000, 20000]))
    if kind == "er":
        ne = int(nv * rng.choice([2, 8, 30]))
        src, dst = rng.integers(0, nv, ne), rng.integers(0, nv, ne)
    elif kind == "powerlaw":
        ne = int(nv * rng.choice([4, 16]))
        a = 1.0 + rng.uniform(0.3, 1.5)
        src = np.minimum((rng.pareto(a, ne)).astype(np.int64), nv - 1)
        dst = rng.integers(0, nv, ne)
        if rng.random() < 0.7:
            src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
    elif kind == "hubs":
        nv = max(nv, 3000)
        src, dst = list(rng.integers(0, nv, nv * 3)), list(rng.integers(0, nv, nv * 3))
        for h in rng.integers(0, nv, int(rng.integers(1, 5))):
            k = int(rng.choice([70, 200, 700, 2500, 9000, 20000]))
            nb = rng.integers(0, nv, k)
            src += [h] * k + list(nb); dst += list(nb) + [h] * k
        src, dst = np.array(src), np.array(dst)
    elif kind == "cliques":  # overlapping cliques + bridges: rows whose class counts stand in small ratios
        src, dst, nv = [], [], 0
        for size in rng.integers(3, 70, int(rng.integers(8, 40))):
            ids = np.arange(nv, nv + int(size)); a, b = np.meshgrid(ids, ids); keep = a != b
            src.append(a[keep]); dst.append(b[keep]); nv += int(size)
        extra = rng.integers(0, nv, (int(nv * rng.choice([0.2, 1, 4])), 2)); extra = extra[extra[:, 0] != extra[:, 1]]
        src = np.concatenate(src + [extra[:, 0], extra[:, 1]]); dst = np.concatenate(dst + [extra[:, 1], extra[:, 0]])
    else:  # bipartite
        nh = int(rng.integers(2, 12)); nv = max(nv, 2000)
        leaves = rng.integers(nh, nv, nh * int(rng.choice([100, 3000, 12000])))
        hubs = rng.integers(0, nh, len(leaves))
        src, dst = np.concatenate([hubs, leaves]), np.concatenate([leaves, hubs])
    if rng.random() < 0.3:  # sinks
        keep = src % 7 != 3; src, dst = src[keep], dst[keep]
    if rng.random() < 0.3:  # de-duplicate (otherwise multi-edges stay)
        key = np.unique(src.astype(np.int64) * nv + dst); src, dst = key // nv, key % nv
    wk = rng.choice(["unit", "unit", "unit", "dyadic", "arbitrary", "arbitrary64"])
    if wk == "unit":
        w = np.ones(len(src), np.float32)
    elif wk == "dyadic":
        w = rng.choice([0.25, 0.5, 1.0, 2.0, 1.5], len(src)).astype(np.float32)
    elif wk == "arbitrary64":  # Python-float weights (the reference's own type): fp64 storage
        w = rng.uniform(0.01, 5.0, len(src))
    else:
        w = rng.uniform(0.01, 5.0, len(src)).astype(np.float32)
    return nv, src, dst, w, wk

t0 = time.time(); n_cases = 0; n_walks = 0
while time.time() - t0 < budget:
    kind = rng.choice(["er", "powerlaw", "hubs", "bipartite", "cliques"] if os.environ.get("FUZZ_PQ") != "rational"
                      else ["cliques", "cliques", "powerlaw", "hubs"])
    nv, src, dst, w, wk = graph(kind)
    if len(src) == 0:
        continue
    g = DeviceGraph.from_edges(src, dst, w, n_vertices=nv, device="cuda")
    # mixed wedge table: the lists of the edges into rows of this many entries or more are 32-bit and
    # have no slot (production: 65536) -- lowered here so that small graphs have such rows
    g.WEDGE_WIDE_FROM = int(rng.choice([65536, 65536, 2, 8, 64, 700]))
    # FUZZ_PQ=extreme: very small / very large and non-dyadic parameters
    if os.environ.get("FUZZ_PQ") == "extreme":
        vals = [0.001, 0.01, 0.03125, 0.03, 1.0 / 3.0, 0.999, 1.001, 16.0, 37.5, 100.0, 1000.0, 1024.0]
        p, q = float(rng.choice(vals)), float(rng.choice(vals))
    elif os.environ.get("FUZZ_PQ") == "rational":
        # class values in small rational ratios: the cumulative sums of the pairing loop meet exactly in
        # real arithmetic and within a few ulp in fp64 -- what the closed forms with margins must decline
        vals = [1.0 / 3.0, 2.0 / 3.0, 1.5, 3.0, 6.0, 0.75, 1.25, 5.0, 0.2, 0.6, 1.2, 2.5, 7.0, 1.0 / 7.0, 9.0, 12.0]
        p, q = float(rng.choice(vals)), float(rng.choice(vals))
    elif os.environ.get("FUZZ_PQ") == "two":
        # the return slot shares a stack with "other": q > 1 with p > q, q < 1 with p < q
        pairs = [(4.0, 2.0), (8.0, 2.0), (8.0, 4.0), (16.0, 2.0), (0.25, 0.5), (0.125, 0.5),
                 (0.125, 0.25), (0.0625, 0.5)]
        p, q = pairs[int(rng.integers(len(pairs)))]
    else:
        # (3, 1.5, 6, 0.75: class values in small rational ratios -- cumulative sums that meet within a few ulp)
        p = float(rng.choice([0.25, 0.5, 1.0, 2.0, 4.0, 3.0, 0.7, 1.5, 6.0]))
        q = float(rng.choice([0.25, 0.5, 1.0, 2.0, 4.0, 1.3, 0.1, 3.0, 0.75]))
    if rng.random() < 0.12:
        p = q = 1.0  # the reference's defaults: the hop-table kernels and the degree-ranked form
    nw, wl = int(rng.integers(1, 5)), int(rng.choice([1, 5, 20, 60, 130]))
    seed = int(rng.integers(0, 2 ** 62))
    deg = g.degrees()
    starts = torch.unique(torch.cat([torch.topk(deg, min(20, nv)).indices,
                                     torch.from_numpy(rng.integers(0, nv, 300)).cuda()])).to(torch.int32)
    if wk.startswith("arbitrary") and int(deg.max()) > 3000:
        starts = starts[:60]; wl = min(wl, 20)  # the oracle is O(degree) per step
    # unit weights, dyadic p, q: the lanes kernel (per-edge class counts) by default, the
    # wave-per-walker kernel without them -- both must match
    uec = bool(rng.random() < 0.7)
    # the passes of n2v_walk_ws (a `make WEDGE2=1` build; the default library ignores the workspace): any
    # number of main / resolve rounds before the finishing launch must give the same walks
    os.environ["N2V_WEDGE2_ROUNDS"] = str(int(rng.choice([0, 1, 2, 4, 9])))
    # weighted graphs: the lane-per-walker step kernel / a wave per walker with table classes / by search
    wlanes = bool(rng.random() < 0.5)
    # (with or without the wave kernel that decides long rows by margins)
    rw.WEIGHTED_LANES_MARGINS = bool(rng.random() < 0.7)
    try:
        got, gv = rw.walk(g, starts, nw, wl, p, q, seed, use_edge_classes=uec,
                          use_workspace=bool(rng.random() < 0.3), use_wedge_slots=bool(rng.random() < 0.7),
                          use_weighted_lanes=wlanes)
    except Exception:
        print("RAISED", dict(kind=kind, nv=nv, ne=len(src), weights=wk, p=p, q=q, nw=nw, wl=wl, seed=seed,
                             maxdeg=int(deg.max()), unit=g.unit_weights, edge_classes=uec, wide_from=g.WEDGE_WIDE_FROM,
                             wedge_mode=g.wedge_mode, slots=g.wedge_slots is not None, folded=g.slots_folded,
                             inline=g.hops_inline_rpos, weighted_lanes=wlanes), flush=True)
        raise
    want, wv = n2v_oracle.random_walk(g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.w.cpu().numpy(),
                                      starts.cpu().numpy(), nw, wl, p, q, seed, n_threads=THREADS)
    ok = np.array_equal(gv.cpu().numpy(), wv) and np.array_equal(got.cpu().numpy(), want)
    if ok and p == 1.0 and q == 1.0 and g.unit_weights:
        # the degree-ranked form (few classes: most ranks go through the head table), walks out in ranks
        g.RANK_MAX_CLASSES = int(rng.choice([8191, 64, 4, 1]))
        rk, rv = rw.walk(g, starts, nw, wl, p, q, seed, rank_ids=True)
        back = torch.where(rk >= 0, g.rank_vertex[rk.clamp(min=0).long()], rk)
        got2, gv2 = rw.walk(g, starts, nw, wl, p, q, seed, use_ranked=True)
        ok = (np.array_equal(rv.cpu().numpy(), wv) and np.array_equal(back.cpu().numpy(), want)
              and np.array_equal(gv2.cpu().numpy(), wv) and np.array_equal(got2.cpu().numpy(), want))
        if not ok:
            print("RANKED", g.RANK_MAX_CLASSES, flush=True)
    if ok and PARTITIONED:
        n_parts = int(rng.integers(1, 7))
        parts = P.partition_graph(g, n_parts, balance=str(rng.choice(["edges", "vertices"])),
                                  wedges=bool(rng.random() < 0.7))  # wedge lists or whole rows travel
        # forwarding (n2v_partition_forward, where it applies) or the launch-per-stage routing; a word
        # pool that starts too small makes steps repeat; parts on separate streams or one
        fw = [None, False, "ranks"][int(rng.integers(0, 3))]
        P.FORWARD_WORDS_PER_WALKER = int(rng.choice([8, 1, 0]))
        P.FORWARD_MIN_WORDS = int(rng.choice([1 << 16, 64, 0]))
        P.FORWARD_STREAMS = bool(rng.random() < 0.5)
        got, gv = P.walk_partitioned_local(parts, starts, nw, wl, p, q, seed, forwarding=fw)
        ok = np.array_equal(gv.cpu().numpy(), wv) and np.array_equal(got.cpu().numpy()[wv], want[wv])
        if not ok:
            print("PARTITIONED", n_parts, fw, P.FORWARD_WORDS_PER_WALKER, P.FORWARD_MIN_WORDS, P.FORWARD_STREAMS, flush=True)
    n_cases += 1; n_walks += int(wv.sum())
    if not ok:
        bad = np.nonzero((got.cpu().numpy() != want).any(1) | (gv.cpu().numpy() != wv))[0][:5]
        print("MISMATCH", dict(kind=kind, nv=nv, ne=len(src), weights=wk, p=p, q=q, nw=nw, wl=wl, seed=seed,
                               maxdeg=int(deg.max()), unit=g.unit_weights, edge_classes=uec,
                               wide_from=g.WEDGE_WIDE_FROM, wedge_mode=g.wedge_mode, weighted_lanes=wlanes), "rows", bad.tolist(), flush=True)
        for r in bad[:2]:
            print(" got ", got[r].tolist()[:12], "\n want", want[r].tolist()[:12])
        sys.exit(1)
print(f"fuzz ok: {n_cases} cases, {n_walks} walks bit-identical to the oracle in {time.time()-t0:.0f} s")
