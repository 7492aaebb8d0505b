#!/usr/bin/env python3
"""bench.py -- node2vec hot path on MI355X: walk-steps/s (+ embedding-updates/s).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], "cfg 2"): R-MAT scale 20, 5 M draws symmetrised
(~9.7 M directed edges, ~472 k non-isolated vertices), p=0.5 q=2, 10 walks per
vertex, walk_length 80.  One "step" = one launch of the walk kernel over one batch
of start vertices (inputs resident in HBM).  With N GPUs the graph is replicated and
start vertices are sharded by range: rank r walks batch (k*N + r); no collective is
on the data path ("weak" scaling: per-GPU work per step is fixed).

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline      algorithmic HBM bytes of the walk kernel / its HIP-event duration
  cpu_baseline  the CPU oracle (OpenMP, all host cores) on a bounded sample
  sgns          the same measurement for the SGNS kernel (embedding-updates/s)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md "HBM3E peak BW" (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", default="exact", choices=["exact", "fast"])
    ap.add_argument("--scale", type=int, default=20)
    ap.add_argument("--draws", type=int, default=5_000_000)
    ap.add_argument("--batch", type=int, default=47_104, help="start vertices per step per GPU")
    ap.add_argument("--num-walks", type=int, default=10)
    ap.add_argument("--walk-length", type=int, default=80)
    ap.add_argument("--p", type=float, default=0.5)
    ap.add_argument("--q", type=float, default=2.0)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sgns", action="store_true")
    ap.add_argument("--no-fast", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def algorithmic_bytes_exact(torch, g, walks, valid):
    """SURVEY.md 8(d), exact mode, summed over the emitted walks:
    B = 16 (rowptr v) + 8*deg(v) (col+w) + [s>=0: 16 (rowptr s) + 4*deg(s)] + 4 (path write)."""
    deg = g.degrees()
    w = walks[valid].long()
    d = deg[w[:, :-1]]                      # degree of the current vertex at every step
    total = (16 + 8 * d + 4).sum()
    total = total + (16 + 4 * d[:, :-1]).sum()  # previous vertex, steps >= 1
    return int(total)


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # under torch.distributed.run (RANK set) always go through RCCL, also at N=1: the same
    # code path at every N
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", device_id=dev)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    from node2vec_amd import randomwalk as rw
    from node2vec_amd import synthetic

    g = synthetic.rmat(args.scale, args.draws, seed=42, device=dev)
    if args.mode == "fast":
        g.build_alias()
    start_all = rw.start_vertices(g)
    n_batches = max(1, start_all.numel() // args.batch)
    W, L = args.num_walks, args.walk_length

    def batch(i):
        i = i % n_batches
        return start_all[i * args.batch:(i + 1) * args.batch].contiguous()

    walks = torch.empty((args.batch * W, L + 1), dtype=torch.int32, device=dev)
    valid = torch.empty(args.batch * W, dtype=torch.uint8, device=dev)

    def step(k):
        rw.walk(g, batch(k * world + rank), W, L, args.p, args.q, 42, mode=args.mode,
                out=(walks, valid), check=False)

    for k in range(args.warmup):
        step(k)
    # ---- timed region: EXACTLY K steps, barrier + synchronize on both sides ----
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    steps_done = 0
    abytes = 0
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        step(args.warmup + k)
        ev[k][1].record()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kernel_ms = [a.elapsed_time(b) for a, b in ev]  # events on the launch stream
    # unit counts (outside the timed region): re-derive from the last launch and
    # from a recount of every timed batch's valid walks
    for k in range(args.steps):
        step(args.warmup + k)
        torch.cuda.synchronize()
        v = valid.bool()
        steps_done += int(v.sum()) * L
        if args.mode == "exact":
            abytes += algorithmic_bytes_exact(torch, g, walks, v)
    t = torch.tensor([elapsed, float(steps_done), float(abytes), sum(kernel_ms)],
                     dtype=torch.float64, device=dev)
    if use_dist:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, steps_total = float(tmax[0]), float(tsum[1])
    else:
        steps_total = float(steps_done)
    value = steps_total / elapsed

    out = {
        "metric": "walk-steps/sec + embedding-updates/sec on 100M-node synthetic; 1/2/4/8 GPU",
        "value": value, "unit": "walk-steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"cfg2 RMAT scale {args.scale} / {g.n_edges} directed edges, "
                               f"p={args.p} q={args.q}, {W} walks x length {L}, "
                               f"{args.batch} start vertices per step per GPU",
                   "walk_mode": args.mode, "n_vertices": g.n_vertices, "n_edges": g.n_edges,
                   "start_vertices": int(start_all.numel()), "parallelism": f"range-shard x{world}"},
    }
    if rank == 0:
        avg_kernel_s = 1e-3 * sum(kernel_ms) / args.steps
        per_launch_bytes = abytes / args.steps if abytes else None
        if per_launch_bytes:
            ach = per_launch_bytes / avg_kernel_s
            out["roofline"] = {"bound": "hbm", "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9,
                               "unit": "GB/s", "frac": ach / HBM_PEAK,
                               "traffic": _pmc_traffic(),
                               "kernel": "walk_exact_unit_kernel" if g.unit_weights else "walk_exact_kernel",
                               "kernel_ms": 1e3 * avg_kernel_s,
                               "algorithmic_bytes_per_launch": per_launch_bytes,
                               "algorithmic_bytes_per_walk_step": abytes / max(steps_done, 1)}
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(args, g, start_all, W, L)
    if args.mode == "exact" and not args.no_fast:
        fm = bench_fast(args, torch, dist, g, rw, batch, walks, valid, rank, world, barrier, use_dist)
        if rank == 0:
            out["fast_mode"] = fm
    if not args.no_sgns:
        sg = bench_sgns(args, torch, dist, g, walks, valid, rank, world, barrier, use_dist)
        if rank == 0:
            out["sgns"] = sg
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def bench_fast(args, torch, dist, g, rw, batch, walks, valid, rank, world, barrier, use_dist=False):
    """Secondary figure: the same K steps with the rejection sampler (N2V_WALK_FAST,
    same transition distribution, not the same draws).  Algorithmic bytes per accepted
    step: 16 + T*(16 + [s>=0: 16 + 4*ceil(log2(deg(s)+1))]) + 4."""
    import math
    import time as _t

    W, L = args.num_walks, args.walk_length
    if g.slots is None:
        g.build_alias()
    stats = {}

    def step(k):
        rw.walk(g, batch(k * world + rank), W, L, args.p, args.q, 42, mode="fast",
                out=(walks, valid), check=False, stats=stats)

    for k in range(args.warmup):
        step(k)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    trials = 0
    barrier()
    t0 = _t.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        step(args.warmup + k)
        ev[k][1].record()
    barrier()
    elapsed = _t.perf_counter() - t0
    steps_done, abytes = 0, 0.0
    deg = g.degrees()
    for k in range(args.steps):  # recount outside the timed region
        step(args.warmup + k)
        torch.cuda.synchronize()
        v = valid.bool()
        n_steps = int(v.sum()) * L
        steps_done += n_steps
        tr = int(stats["trials"].item())
        trials += tr
        ds = deg[walks[v][:, :-2].long()].double()
        per_trial = 16.0 + (16.0 + 4.0 * torch.ceil(torch.log2(ds + 1.0))).mean().item() * (L - 1) / L
        abytes += 20.0 * n_steps + tr * per_trial
    t = torch.tensor([elapsed, float(steps_done)], dtype=torch.float64, device=walks.device)
    if use_dist:
        tm, ts = t.clone(), t.clone()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        elapsed, total = float(tm[0]), float(ts[1])
    else:
        total = float(steps_done)
    kernel_s = 1e-3 * sum(a.elapsed_time(b) for a, b in ev) / args.steps
    ach = abytes / args.steps / kernel_s
    del math
    return {"value": total / elapsed, "unit": "walk-steps/s", "walk_mode": "fast",
            "parity": "same transition distribution (chi-square tested), not the same draws",
            "trials_per_step": trials / max(steps_done, 1),
            "roofline": {"bound": "hbm", "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": ach / HBM_PEAK, "traffic": _pmc_traffic("fast_pmc_traffic.json"),
                         "kernel": "walk_fast_kernel",
                         "kernel_ms": 1e3 * kernel_s,
                         "algorithmic_bytes_per_walk_step": abytes / max(steps_done, 1)}}


def bench_sgns(args, torch, dist, g, walks, valid, rank, world, barrier, use_dist=False):
    """Second timed loop: K steps of the SGNS kernel (embedding-updates/s).  One step =
    one launch over the block of walks of one walk step (batch x W rows of L+1 tokens),
    vocabulary = every vertex (min_count=0, sample=0: deterministic unit counts), dim
    args.dim, window 5, k=5.  Unit: one positive (centre, context) pair with its k
    negative targets.  With N GPUs every rank trains its own walks on a full replica;
    the delta all-reduce is reported separately (it is per sync, not per step)."""
    import time as _t

    from node2vec_amd import sgns

    dev = walks.device
    deg = g.degrees().clamp(min=1)
    order = torch.sort(deg, descending=True, stable=True).indices
    index_of = torch.empty(g.n_vertices, dtype=torch.int32, device=dev)
    index_of[order] = torch.arange(g.n_vertices, dtype=torch.int32, device=dev)
    vocab = sgns.Vocab(order, deg[order], index_of)
    model = sgns.SgnsModel(vocab, args.dim, 5, 5, seed=1, sample=0.0, device=dev)
    idx = index_of[walks[valid.bool()].long()].contiguous()
    rows = idx.shape[0]
    for k in range(args.warmup):
        model.train_block(idx, 0.025, k * rows)
    model.pairs.zero_()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    barrier()
    t0 = _t.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        model.train_block(idx, 0.025, (args.warmup + k + rank * 1000) * rows)
        ev[k][1].record()
    barrier()
    elapsed = _t.perf_counter() - t0
    pairs = float(model.pairs.item())
    t = torch.tensor([elapsed, pairs], dtype=torch.float64, device=dev)
    if use_dist:
        tm = t.clone()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        ts = t.clone()
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        elapsed, pairs_total = float(tm[0]), float(ts[1])
    else:
        pairs_total = pairs
    kernel_s = 1e-3 * sum(a.elapsed_time(b) for a, b in ev) / args.steps
    bytes_per_pair = 8 * args.dim * (2 + 5)            # 2*4*D*(2+k)
    flops_per_pair = (1 + 5) * 6 * args.dim + args.dim  # (1+k)*6D + D
    ach = pairs / args.steps * bytes_per_pair / kernel_s
    res = {"value": pairs_total / elapsed, "unit": "embedding-updates/s (pairs incl. k=5 negatives)",
           "row_updates_per_s": pairs_total / elapsed * 6, "ms_per_step": 1e3 * elapsed / args.steps,
           "dtype": "f32", "config": {"dim": args.dim, "window": 5, "negative": 5,
                                      "rows_per_step": rows, "n_vocab": g.n_vertices,
                                      "sample": 0, "min_count": 0},
           "roofline": {"bound": "hbm", "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9,
                        "unit": "GB/s", "frac": ach / HBM_PEAK,
                        "traffic": _pmc_traffic("sgns_pmc_traffic.json") if args.dim == 128 else None,
                        "kernel": "sgns_kernel", "kernel_ms": 1e3 * kernel_s,
                        "algorithmic_bytes_per_pair": bytes_per_pair,
                        "fma_utilisation": pairs / args.steps * flops_per_pair / kernel_s / 157.3e12}}
    if use_dist:  # the exchange step: one delta all-reduce of both matrices
        sync = sgns.DeltaAllReduce([model.syn0, model.syn1neg], block_rows=1 << 18)
        barrier()
        t0 = _t.perf_counter()
        sync()
        barrier()
        res["delta_allreduce_s"] = _t.perf_counter() - t0
        res["delta_allreduce_bytes"] = 2 * model.syn0.numel() * 4
    return res


def _pmc_traffic(name="walk_pmc_traffic.json"):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/),
    corrected as MI355X_MICROARCH.md prescribes; None until such a profile exists."""
    path = os.path.join(ROOT, "profiles", name)
    if os.path.exists(path):
        with open(path) as f:
            return json.load(f).get("hbm_bytes_per_launch")
    return None


def cpu_baseline(args, g, start_all, W, L):
    """The CPU oracle (a port of the reference's algorithm: per-step biased alias
    rebuild + two-uniform draw) on the host cores, bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import n2v_oracle

    cores = os.cpu_count() or 1
    rowptr, col, w = g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.w.cpu().numpy()
    starts = start_all.cpu().numpy()
    # calibrate on a small slice, then size the sample for ~cpu_seconds of work
    n0 = 64
    t0 = time.perf_counter()
    n2v_oracle.random_walk(rowptr, col, w, starts[:n0], W, L, args.p, args.q, 42, n_threads=cores)
    dt0 = max(time.perf_counter() - t0, 1e-3)
    n = int(min(len(starts), max(n0, n0 * args.cpu_seconds / dt0)))
    stride = max(1, len(starts) // n)
    sample = starts[::stride][:n]
    t0 = time.perf_counter()
    _, valid = n2v_oracle.random_walk(rowptr, col, w, sample, W, L, args.p, args.q, 42,
                                      n_threads=cores)
    dt = time.perf_counter() - t0
    return {"value": float(valid.sum()) * L / dt, "unit": "walk-steps/s", "cores": cores,
            "kind": "port",
            "sample": f"{len(sample)} start vertices (every {stride}th) x {W} walks x {L} steps, "
                      f"{dt:.1f} s, oracle/n2v_oracle.c with OpenMP"}


if __name__ == "__main__":
    main()
