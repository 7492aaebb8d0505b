#!/usr/bin/env python3
"""bench.py -- node2vec hot path on MI355X: walk-steps/s + embedding-updates/s.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg4|cfg3|cfg2|cfg5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default workload = the configuration BASELINE.json's metric is quoted on ("100M-node
synthetic", BASELINE.md section 4 row 4, "cfg 4"): Chung-Lu power-law graph, gamma = 2.1,
10^8 vertices, 5 x 10^8 undirected draws symmetrised and de-duplicated (~0.9 x 10^9
directed edges), out-degree trimmed at 10 000 (trim_hotspot_vertices), 10 walks per
vertex, walk_length 80, p = q = 1, seed 42.  It fits one GPU (graph 4 GB + model 102 GB).

One "step" = one launch of the walk kernel over one batch of start vertices, inputs
resident in HBM.  With N GPUs the graph is replicated and start vertices are sharded by
range: rank r walks batch k * N + r; no collective is on the walk path ("weak" scaling:
per-GPU work per step is fixed).

ONE JSON line on rank 0.  Beside the contract's keys:
  roofline      the headline walk kernel: bytes / HIP-event duration against 8 TB/s
  biased        the same graph walked exactly at p = 0.5, q = 2 (at p = q = 1 a step is two
                gathers; the second-order bias is where the sampler works)
  fast_mode     fast mode at p = 0.5, q = 2 (same distribution, not the same draws)
  sgns          K launches of the SGNS kernel on a 10^8 x 128 model (embedding-updates/s)
  cpu_baseline  the CPU oracle (a port of the reference's algorithm) on this box's cores:
                walks single-thread and all-core, SGNS single-thread and all-core
  setup         one-off costs (graph build, trim, tables)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md "HBM3E peak BW" (spec)
FP32_PEAK = 157.3e12
# What the memory system of THIS box sustains for the access shapes of the two kernels is
# measured in this process, untimed, by n2v_mem_probe (measure_ceilings below): random 16-byte
# gathers over the hop table (K2) and random 512-byte row reads / read-modify-writes over the
# model (K3).  MEASURED holds the results; nothing is hard-coded.
MEASURED = {}

CONFIGS = {
    "cfg4": dict(gen="chung_lu", n=100_000_000, draws=500_000_000, trim=10_000, p=1.0, q=1.0,
                 batch=1 << 20, biased_batch=1 << 20, sgns_vertices=1 << 16, dim=128,
                 label="cfg4 Chung-Lu power-law gamma=2.1, 100M vertices / 5e8 undirected draws "
                       "symmetrised, out-degree trimmed at 10000"),
    "cfg3": dict(gen="chung_lu", n=10_000_000, draws=100_000_000, trim=10_000, p=1.0, q=1.0,
                 batch=1 << 20, biased_batch=1 << 20, sgns_vertices=1 << 16, dim=128,
                 label="cfg3 Chung-Lu power-law gamma=2.1, 10M vertices / 1e8 undirected draws "
                       "symmetrised, out-degree trimmed at 10000"),
    # BASELINE.json configs[4]: 5 000 hubs x 10 000 distinct leaves + one hub
    # per leaf (hub degree ~20 000 before the trim), p = 4, q = 0.25, dim 256 (model 2 x 51.2 GB)
    "cfg5": dict(gen="hub_bipartite", n=50_000_000, hubs=5_000, hub_degree=10_000, trim=10_000, p=4.0, q=0.25,
                 batch=1 << 20, biased_batch=1 << 20, sgns_vertices=1 << 16, dim=256,
                 label="cfg5 skewed bipartite, 50M vertices, 5000 hubs x 10000 leaves + one hub per leaf, "
                       "out-degree trimmed at 10000"),
    "cfg2": dict(gen="rmat", scale=20, draws=5_000_000, trim=0, p=0.5, q=2.0,
                 batch=47_104, biased_batch=47_104, sgns_vertices=47_104, dim=128,
                 label="cfg2 R-MAT scale 20, 5e6 draws symmetrised"),
}
BIASED_PQ = (0.5, 2.0)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg4", choices=sorted(CONFIGS))
    ap.add_argument("--no-ranked", action="store_true",
                    help="p = q = 1: keep vertex-id output as the headline (no degree-ranked form)")
    ap.add_argument("--num-walks", type=int, default=10)
    ap.add_argument("--walk-length", type=int, default=80)
    ap.add_argument("--batch", type=int, default=0, help="start vertices per step per GPU (0 = config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sgns", action="store_true")
    ap.add_argument("--no-fast", action="store_true")
    ap.add_argument("--no-batched", action="store_true", help="skip the opt-in batched SGNS leg")
    ap.add_argument("--no-hub", action="store_true", help="skip the hub_rows = 4096 SGNS leg (profiling "
                    "runs: every sgns_kernel launch then is a headline launch)")
    ap.add_argument("--no-biased", action="store_true")
    ap.add_argument("--no-regimes", action="store_true", help="skip the other (p, q) regimes of the exact sampler")
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="CPU time budget per baseline leg")
    ap.add_argument("--trim", type=int, default=-1,
                    help="out-degree cap of trim_hotspot_vertices (-1 = the config's, 10000; 100000 = the "
                         "reference's default cap, constants.py:6; 0 = no trim)")
    ap.add_argument("--no-ref-cap", action="store_true",
                    help="skip the legs on the graph trimmed at the reference's default cap (100 000)")
    ap.add_argument("--no-weighted", action="store_true",
                    help="skip the leg on the WEIGHTED cfg 2 graph (exact walks at (0.5, 2), every start vertex)")
    ap.add_argument("--no-api", action="store_true",
                    help="skip the API leg (random_walk() -> DataFrame -> Node2VecGensim.fit() -> embedding() on cfg 2)")
    ap.add_argument("--no-audition", action="store_true",
                    help="take the first output buffer the allocator hands out (no placement audition)")
    ap.add_argument("--spawn", action="store_true",
                    help="start the ranks through torch.distributed.run also at --gpus 1 (at --gpus N > 1 "
                         "this is what happens anyway when RANK is not in the environment)")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the spawned ranks (0 = a free one)")
    return ap.parse_args(argv)


def spawn_command(argv, n, port):
    """The command `bench.py --gpus N` re-launches itself as when it was started as ONE process
    (no RANK in the environment): N ranks under torch.distributed.run, the driver's own form."""
    rest = [a for a in argv if a != "--spawn"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + rest


def spawn_ranks(args, argv):
    """Parent of a self-launched run: NO GPU call is made here (a process that has touched the
    GPU must not start the ranks by exec, and this one does not need the GPU at all).  The
    ranks run as a CHILD process; its one JSON line is relayed, its exit code is ours."""
    import socket
    import subprocess

    port = args.master_port
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    child = subprocess.Popen(spawn_command(argv, args.gpus, port), env=env, stdout=subprocess.PIPE, text=True)
    lines = []
    for ln in child.stdout:  # stderr goes straight through; stdout carries the JSON line
        if ln.startswith("{"):
            lines.append(ln)
        else:
            sys.stderr.write(ln)
    code = child.wait()
    for ln in lines:
        sys.stdout.write(ln)
    sys.stdout.flush()
    if code == 0 and len(lines) != 1:
        sys.stderr.write(f"bench.py: the spawned ranks printed {len(lines)} JSON lines, expected 1\n")
        code = 1
    return code


def build_graph(cfg, torch, dev, setup, trim=-1):
    from node2vec_amd import synthetic

    t0 = time.perf_counter()
    if cfg["gen"] == "rmat":
        g = synthetic.rmat(cfg["scale"], cfg["draws"], seed=42, device=dev)
    elif cfg["gen"] == "hub_bipartite":
        g = synthetic.hub_bipartite(cfg["n"], cfg["hubs"], cfg["hub_degree"], seed=42, device=dev)
    else:
        g = synthetic.chung_lu(cfg["n"], cfg["draws"], seed=42, device=dev)
    torch.cuda.synchronize()
    setup["graph_generate_s"] = time.perf_counter() - t0
    setup["edges_before_trim"] = g.n_edges
    cap = cfg["trim"] if trim < 0 else trim
    setup["trim_cap"] = cap
    if cap:
        t0 = time.perf_counter()
        g = g.trimmed(cap, 42)  # trim_hotspot_vertices, randomwalk.py:238-262
        torch.cuda.synchronize()
        setup["trim_s"] = time.perf_counter() - t0
    torch.cuda.empty_cache()
    return g


def reference_algorithmic_bytes(torch, g, walks, valid, rows=65536):
    """SURVEY.md 8(d), exact mode, per walk-step (estimated on the first `rows` walks of the
    launch): B = 16 (rowptr v) + 8*deg(v) (col + w) + [s >= 0: 16 (rowptr s) + 4*deg(s)] + 4
    (path write) -- the bytes the REFERENCE's algorithm touches (it rebuilds the table of the
    whole row at every step); the kernels here move far fewer, see roofline.traffic."""
    deg = g.degrees()
    w = walks[:rows][valid[:rows].bool()].long()
    if w.numel() == 0:
        return None
    d = deg[w[:, :-1]]
    total = (16 + 8 * d + 4).sum() + (16 + 4 * d[:, :-1]).sum()
    return float(total) / float(d.numel())


class WalkLeg:
    """K timed launches of n2v_walk over batches of start vertices."""

    def __init__(self, torch, rw, g, start_all, W, L, p, q, mode, batch, rank, world, rank_ids=False,
                 audition=True):
        self.rank_ids = rank_ids  # the walks come out in degree ranks (fit_streaming's form at p = q = 1)
        self.torch, self.rw, self.g, self.start_all = torch, rw, g, start_all
        self.W, self.L, self.p, self.q, self.mode = W, L, p, q, mode
        self.batch = int(min(batch, start_all.numel()))
        self.rank, self.world = rank, world
        self.n_batches = max(1, start_all.numel() // self.batch)
        dev = g.device
        self.stats = {}
        self.audition = {}
        if audition:
            # where the output buffer lies decides up to 7 % of the launch time (DESIGN.md 5 "Placement"):
            # a few candidate buffers take a short launch each, the fastest is kept -- what
            # fit_streaming does for the buffers it walks its batches into (randomwalk.audition_buffers)
            self.walks, self.valid = rw.audition_buffers(g, self.starts(0), W, L, p, q, 42, mode,
                                                         report=self.audition, rank_ids=rank_ids)
        else:
            self.walks = torch.empty((self.batch * W, L + 1), dtype=torch.int32, device=dev)
            self.valid = torch.empty(self.batch * W, dtype=torch.uint8, device=dev)

    def starts(self, k):
        i = (k * self.world + self.rank) % self.n_batches
        return self.start_all[i * self.batch:(i + 1) * self.batch]

    def step(self, k):
        self.rw.walk(self.g, self.starts(k), self.W, self.L, self.p, self.q, 42, mode=self.mode,
                     out=(self.walks, self.valid), check=False, stats=self.stats, rank_ids=self.rank_ids)

    def run(self, steps, warmup, barrier):
        torch = self.torch
        for k in range(warmup):
            self.step(k)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(steps)]
        # ---- timed region: EXACTLY K steps, barrier + synchronize on both sides ----
        barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            ev[k][0].record()  # torch's current stream IS the launch stream (randomwalk.walk)
            self.step(warmup + k)
            ev[k][1].record()
        barrier()
        elapsed = time.perf_counter() - t0
        kernel_ms = [a.elapsed_time(b) for a, b in ev]
        # unit counts, outside the timed region: the same K batches walked again
        steps_done, trials = 0, 0
        for k in range(steps):
            self.step(warmup + k)
            torch.cuda.synchronize()
            steps_done += int(self.valid.sum()) * self.L
            if self.mode == "fast":
                trials += int(self.stats["trials"].item())
        return {"elapsed": elapsed, "steps_done": steps_done, "trials": trials, "launches": steps,
                "kernel_s": 1e-3 * sum(kernel_ms) / steps}


def reduce_job(torch, dist, use_dist, dev, elapsed, units):
    """max-over-ranks time, sum-over-ranks units"""
    if not use_dist:
        return elapsed, float(units)
    t = torch.tensor([elapsed, float(units)], dtype=torch.float64, device=dev)
    tm, ts = t.clone(), t.clone()
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    dist.all_reduce(ts, op=dist.ReduceOp.SUM)
    return float(tm[0]), float(ts[1])


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "RANK" not in os.environ and (args.gpus > 1 or args.spawn):
        # started as one process: --gpus N means N ranks, so start them (before any GPU call)
        raise SystemExit(spawn_ranks(args, argv))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU "
                         f"(python bench.py --gpus {args.gpus} does it itself)")
    import torch
    import torch.distributed as dist

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
        # RCCL prints a version banner on stdout when its communicator comes up; stdout carries
        # the ONE JSON line, so the banner is sent to stderr (fd level: it is printed by C code)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=dev)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    from node2vec_amd import randomwalk as rw

    cfg = CONFIGS[args.config]
    W, L = args.num_walks, args.walk_length
    setup = {}
    g = build_graph(cfg, torch, dev, setup, args.trim)
    setup["max_out_degree"] = int(g.degrees().max())
    start_all = rw.start_vertices(g)
    batch = args.batch or cfg["batch"]

    # ---- headline: exact walks at the config's p, q --------------------------------------
    p, q = cfg["p"], cfg["q"]
    # one-off tables, built explicitly (randomwalk.walk would build them on first use) so that
    # every launch of a walk kernel in this process has the bench's size: the rocprofv3
    # --kernel-trace average of the kernel is then the HIP-event average reported below
    prepare_tables(torch, g, p, q, "exact", setup, "headline")
    if rank == 0 and (g.hops8 is not None or g.hops is not None):
        # the ceiling the walk kernel is compared with, observed on THIS box: random gathers of
        # the kernel's entry width over this graph's own hop table (untimed; before the K steps)
        table, width = (g.hops8, 8) if (g.hops8 is not None and p == 1.0 and q == 1.0) else (g.hops, 16)
        if table is not None:
            c = measure_ceilings(torch, table, gather_width=width)
            setup["measured_gather_ceiling"] = dict(c, table_GB=table.numel() * table.element_size() / 1e9,
                                                    gather_bytes=width)
    # p = q = 1 on unit weights: fit_streaming walks the degree-ranked form (4-byte entries) and
    # takes the walks in RANKS -- the same walks, vertex for vertex, under the graph's other
    # numbering; rank -> vertex id is folded into the per-token vocabulary lookup that follows.
    # That launch is the headline; the launch that writes vertex ids (what random_walk() puts in
    # its DataFrame: the hop-table kernel) is timed first and reported beside it.
    in_ranks = False
    if p == 1.0 and q == 1.0 and g.unit_weights and not args.no_ranked:
        t0 = time.perf_counter()
        g.build_ranked()
        torch.cuda.synchronize()
        setup["headline_rank_table_build_s"] = time.perf_counter() - t0
        in_ranks = g.rank_hops is not None
    vertex_leg = None
    if in_ranks:
        leg = WalkLeg(torch, rw, g, start_all, W, L, p, q, "exact", batch, rank, world, audition=not args.no_audition)
        rv = leg.run(args.steps, args.warmup, barrier)
        ev, sv = reduce_job(torch, dist, use_dist, dev, rv["elapsed"], rv["steps_done"])
        if rank == 0:
            vertex_leg = {"value": sv / ev, "unit": "walk-steps/s", "ms_per_step": 1e3 * ev / args.steps,
                          "what": "the same walks written as vertex ids (randomwalk.walk's default, the "
                                  "random_walk() DataFrame): hop-table kernel, 16- or 8-byte entries",
                          "roofline": roofline(kernel_name(g, p, q), rv, leg, args.config, p, q, "exact", None)}
            c4 = measure_ceilings(torch, g.rank_hops, gather_width=4, classes=g.rank_class_first.numel())
            setup["measured_gather_ceiling_ranked"] = dict(c4, table_GB=g.rank_hops.numel() * 4 / 1e9,
                                                           gather_bytes=4)
        del leg
        torch.cuda.empty_cache()
    leg = WalkLeg(torch, rw, g, start_all, W, L, p, q, "exact", batch, rank, world, rank_ids=in_ranks, audition=not args.no_audition)
    res = leg.run(args.steps, args.warmup, barrier)
    elapsed, steps_total = reduce_job(torch, dist, use_dist, dev, res["elapsed"], res["steps_done"])
    value = steps_total / elapsed
    head_kernel = kernel_name(g, p, q)
    # the SGNS leg's vocabulary, built as fit_streaming builds it (pipeline.corpus_vocabulary: pass 1
    # over EVERY batch of the virtual corpus, descending token counts) -- untimed setup; the walks
    # are the headline's (ranks at p = q = 1), regenerated batch by batch into the leg's buffers
    sg_vocab = None
    if not args.no_sgns:
        from node2vec_amd.pipeline import corpus_vocabulary

        def count_walk(k):
            st = start_all[k * leg.batch:(k + 1) * leg.batch]
            full = st.numel() == leg.batch
            return rw.walk(g, st, W, L, p, q, 42, mode="exact", out=(leg.walks, leg.valid) if full else None,
                           check=False, rank_ids=in_ranks)

        t0 = time.perf_counter()
        sg_vocab, _ = corpus_vocabulary(g, count_walk, -(-start_all.numel() // leg.batch), 0, in_ranks)
        torch.cuda.synchronize()
        setup["sgns_vocabulary_pass_s"] = time.perf_counter() - t0
        leg.step(args.warmup + args.steps - 1)  # the buffers hold the last timed batch again
    if in_ranks:  # everything below reads vertex ids (outside the timed region)
        leg.walks = torch.where(leg.walks >= 0, g.rank_vertex[leg.walks.clamp(min=0).long()], leg.walks)
    # the SGNS legs train on walks of this leg (rows of the last batch walked)
    nv = min(cfg["sgns_vertices"], leg.batch)
    # one block of walks per SGNS launch, so that no launch trains rows the model has already seen
    # (a model that has seen its rows before skips more targets at |f| >= 6: measured 5 - 8 % faster)
    sg_blocks = max(1, min(args.steps + args.warmup, leg.batch // nv))
    sg_walks = leg.walks[: nv * W * sg_blocks][leg.valid[: nv * W * sg_blocks].bool()].clone()
    ref_bytes = reference_algorithmic_bytes(torch, g, leg.walks, leg.valid)
    # (short: the driver's record keeps 120 characters of a string; the details are separate keys)
    workload = (f"{args.config} {cfg['gen']} {g.n_vertices} vertices / {g.n_edges} edges, p={p:g} q={q:g}, "
                f"{W} walks x {L} steps, batch {leg.batch}")
    assert len(workload) <= 120, workload
    out = {
        "metric": "walk-steps/sec + embedding-updates/sec on 100M-node synthetic; 1/2/4/8 GPU",
        "value": value, "unit": "walk-steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload, "graph": cfg["label"].replace("trimmed at 10000", f"trimmed at {setup['trim_cap']}"),
                   "seed": 42, "num_walks": W,
                   "walk_length": L, "start_vertices_per_step_per_gpu": leg.batch,
                   "walk_mode": "exact", "n_vertices": g.n_vertices,
                   "walk_ids": ("vertex ids (what random_walk() returns); the launch fit_streaming makes, degree "
                                "ranks out: value_pipeline" if in_ranks else "vertex ids"),
                   "n_edges": g.n_edges, "start_vertices": int(start_all.numel()),
                   "trim_cap": setup["trim_cap"], "max_out_degree": setup["max_out_degree"],
                   "parallelism": f"graph replicated, start vertices range-sharded x{world}"},
    }
    if rank == 0:
        out["roofline"] = roofline(head_kernel, res, leg, args.config, p, q, "exact", ref_bytes)
    if in_ranks:
        # `value` is what the reference's contract returns -- walks as VERTEX IDS (fugue.py:153-155,
        # randomwalk.py:343-349): the hop-table launch timed first.  The launch fit_streaming makes (walks out in
        # degree ranks, consumed by n2v_corpus_index, which folds rank -> vertex id into its vocabulary lookup) is
        # carried beside it as value_pipeline / ms_per_step_pipeline / roofline_pipeline.
        out["value_pipeline"], out["ms_per_step_pipeline"] = out["value"], out["ms_per_step"]
        out["pipeline_consumer"] = ("pipeline.fit_streaming: walks in degree ranks -> n2v_corpus_count / "
                                    "n2v_corpus_index (rank -> vocabulary index in one lookup) -> n2v_sgns_train")
        out["value"], out["ms_per_step"] = sv / ev, 1e3 * ev / args.steps
        if rank == 0:
            out["roofline_pipeline"] = out["roofline"]
            out["roofline"] = vertex_leg["roofline"]
    del leg
    torch.cuda.empty_cache()

    # ---- the second-order bias on the same graph: exact mode and fast mode (layered or rejection sampler) -------------
    bp, bq = BIASED_PQ
    if not args.no_biased and (bp, bq) != (p, q):
        prepare_tables(torch, g, bp, bq, "exact", setup, "biased")
        leg = WalkLeg(torch, rw, g, start_all, W, L, bp, bq, "exact", cfg["biased_batch"], rank, world, audition=not args.no_audition)
        r2 = leg.run(args.steps, args.warmup, barrier)
        e2, s2 = reduce_job(torch, dist, use_dist, dev, r2["elapsed"], r2["steps_done"])
        if rank == 0:
            rb = reference_algorithmic_bytes(torch, g, leg.walks, leg.valid)
            out["biased"] = {"value": s2 / e2, "unit": "walk-steps/s", "walk_mode": "exact",
                             "p": bp, "q": bq, "ms_per_step": 1e3 * e2 / args.steps,
                             "start_vertices_per_step": leg.batch,
                             "parity": "bit-identical to the per-step alias rebuild of the reference",
                             "roofline": roofline(kernel_name(g, bp, bq), r2, leg, args.config, bp,
                                                  bq, "exact", rb)}
            # the one-off tables of the biased kernel amortised over ONE pass over the whole graph
            # (every start vertex x W walks x L steps at the measured rate): what a single
            # random_walk() call on a fresh graph gets
            tables_s = sum(v for k, v in setup.items() if k.startswith("biased_") and k.endswith("_s"))
            tables_s += setup.get("headline_hop_table_build_s", 0.0)
            pass_steps = float(start_all.numel()) * W * L  # symmetrised graph: no sinks
            pass_s = pass_steps / (s2 / e2)
            out["biased"]["value_incl_setup"] = pass_steps / (pass_s + tables_s)
            out["biased"]["setup_amortisation"] = {
                "tables_s": tables_s, "one_pass_walk_s": pass_s, "one_pass_walk_steps": pass_steps,
                "note": "edge classes + wedge table + hop table, built once per graph"}
        del leg
        torch.cuda.empty_cache()
    # the other arrangements of the three classes on the two stacks of the alias pairing, each
    # with its own closed form: "other" overfull (cfg 5's p, q), the return slot
    # sharing a stack with "other", the return slot alone overfull.  Same tables, same launches.
    if not args.no_biased and not args.no_regimes:
        regimes = []
        for rp, rq, what in ((4.0, 0.25, "other alone overfull"), (4.0, 2.0, "return + other underfull"),
                             (0.25, 0.5, "return + other overfull / return alone overfull"),
                             (3.0, 0.7, "1/p, 1/q not dyadic: closed forms with margins, the rest replayed")):
            leg = WalkLeg(torch, rw, g, start_all, W, L, rp, rq, "exact", cfg["biased_batch"], rank, world, audition=not args.no_audition)
            rr = leg.run(args.steps, args.warmup, barrier)
            er, sr = reduce_job(torch, dist, use_dist, dev, rr["elapsed"], rr["steps_done"])
            regimes.append({"p": rp, "q": rq, "arrangement": what, "value": sr / er,
                            "unit": "walk-steps/s", "ms_per_step": 1e3 * er / args.steps,
                            "kernel": kernel_name(g, rp, rq)})
            del leg
            torch.cuda.empty_cache()
        if rank == 0:
            out["biased_other_regimes"] = regimes
    if not args.no_fast:
        prepare_tables(torch, g, bp, bq, "fast", setup, "fast")
        leg = WalkLeg(torch, rw, g, start_all, W, L, bp, bq, "fast", batch, rank, world, audition=not args.no_audition)
        r3 = leg.run(args.steps, args.warmup, barrier)
        e3, s3 = reduce_job(torch, dist, use_dist, dev, r3["elapsed"], r3["steps_done"])
        if rank == 0:
            out["fast_mode"] = {"value": s3 / e3, "unit": "walk-steps/s", "walk_mode": "fast",
                                "p": bp, "q": bq, "ms_per_step": 1e3 * e3 / args.steps,
                                "start_vertices_per_step": leg.batch,
                                "sampler": ("layered (layer masses from the per-edge counts, listed slots by "
                                            "index from the wedge table)" if g.wedge_off is not None and g.unit_weights
                                            else "rejection (return edge folded out of the envelope)"),
                                "parity": "same transition distribution as generate_edge_alias_tables, not the "
                                          "same draws (chi-square against the oracle's exact probabilities: "
                                          "weighted tests/test_alias_trim_fast_gpu.py, unit-weight + every "
                                          "table combination tests/test_fast_unit_gpu.py)",
                                "trials_per_step": r3["trials"] / max(r3["steps_done"], 1),
                                "roofline": roofline("walk_fast_kernel", r3, leg, args.config, bp,
                                                     bq, "fast", None)}
        del leg
        torch.cuda.empty_cache()
    # the walk tables are not needed any more (the CPU baseline reads rowptr / col only)
    g.slots = g.pivots = g.hops = g.hops8 = g.edge_classes = g.wedge_off = g.wedge_pos = None
    g.wedge_slots = g.rank_hops = g.rank_of = g.rank_vertex = g.row_sums = None
    torch.cuda.empty_cache()

    # ---- the same graph trimmed at the REFERENCE's default cap (constants.py:6: 100 000; randomwalk.py:252-253)
    # instead of the examples' 10 000: rows of 65 536 slots and more (mixed wedge table), hub lists of thousands
    # of entries.  Two legs: p = q = 1 as fit_streaming launches it, and the exact biased kernel at (0.5, 2).
    if not args.no_ref_cap and cfg["gen"] == "chung_lu" and setup["trim_cap"] not in (0, 100_000):
        out_cap = bench_reference_cap(args, cfg, torch, dist, rw, dev, W, L, rank, world, barrier, use_dist, setup)
        if rank == 0 and out_cap:
            out["reference_trim_cap"] = out_cap

    # ---- exact biased walks on a WEIGHTED graph (no closed form: every table of a step is different): the cfg 2
    # graph with fp32 weights U[0.1, 2], every start vertex x W walks in ONE call, step-synchronous kernels
    if not args.no_weighted and world == 1:
        out_w = bench_weighted(args, torch, rw, dev, W, L)
        if rank == 0 and out_w:
            out["weighted"] = out_w

    # ---- the API surface, end to end: random_walk() -> DataFrame -> Node2VecGensim.fit() -> embedding() on cfg 2
    if not args.no_api and world == 1 and rank == 0:
        out["api"] = bench_api(args, torch, rw, dev)

    # ---- SGNS on the config's model ------------------------------------------------------------
    model = None
    if not args.no_sgns:
        sg, model = bench_sgns(args, cfg, torch, dist, g, sg_walks, rank, world, barrier, use_dist, sg_blocks,
                               sg_vocab)
        if rank == 0:
            out["sgns"] = sg
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args, cfg, torch, g, start_all, W, L, p, q,
                                           sg_walks[: sg_walks.shape[0] // sg_blocks],
                                           model)
    if rank == 0:
        setup["hbm_peak_allocated_GB"] = torch.cuda.max_memory_allocated() / 1e9
        out["setup"] = setup
        print(json.dumps(ordered_line(out)), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def bench_reference_cap(args, cfg, torch, dist, rw, dev, W, L, rank, world, barrier, use_dist, setup):
    """the workload's graph trimmed at 100 000 (trim_index(max_out_deg=0), the reference's default): exact
    p = q = 1 (ranks out) and exact (0.5, 2), K launches each"""
    sub = {}
    g2 = build_graph(cfg, torch, dev, sub, trim=100_000)
    if int(g2.degrees().max()) < 65536:
        return None
    start2 = rw.start_vertices(g2)
    res = {"trim_cap": 100_000, "n_edges": g2.n_edges, "max_out_degree": int(g2.degrees().max()),
           "share_of_steps_on_rows_of_65536_or_more": float(g2.degrees()[g2.degrees() >= 65536].sum()) / g2.n_edges}
    g2.build_ranked()
    legs = [("exact_pq1_ranks_out", (1.0, 1.0), g2.rank_hops is not None), ("exact_biased_0.5_2", BIASED_PQ, False)]
    if not args.no_regimes:
        legs += [("exact_biased_4_0.25", (4.0, 0.25), False), ("exact_biased_3_0.7", (3.0, 0.7), False)]
    for name, (p2, q2), in_ranks in legs:
        if (p2, q2) != (1.0, 1.0):
            # this leg owns the card (its tables go when it ends): lists (101 GB at cfg 4) + folded copies (17 GB) +
            # slots (26 GB) may take what is free less 24 GB for the walks and the other tables
            torch.cuda.empty_cache()
            free = torch.cuda.mem_get_info(dev)[0]
            res.setdefault("hbm_free_GB_before_tables", free / 1e9)
            prepare_tables(torch, g2, p2, q2, "exact", sub, "cap", wedge_bytes=max(free - (24 << 30), free // 2))
        leg = WalkLeg(torch, rw, g2, start2, W, L, p2, q2, "exact", cfg["biased_batch"], rank, world,
                      rank_ids=in_ranks, audition=not args.no_audition)
        r = leg.run(args.steps, args.warmup, barrier)
        e, s_ = reduce_job(torch, dist, use_dist, dev, r["elapsed"], r["steps_done"])
        res[name] = {"value": s_ / e, "unit": "walk-steps/s", "ms_per_step": 1e3 * e / args.steps,
                     "kernel": kernel_name(g2, p2, q2)}
        if rank == 0 and name == "exact_biased_0.5_2":
            # (its own key in profiles/pmc_traffic.json: another graph than the config's)
            res[name]["roofline"] = roofline(kernel_name(g2, p2, q2), r, leg, args.config + "@cap100000", p2, q2,
                                             "exact", None)
        del leg
        torch.cuda.empty_cache()
    res["wedge_table"] = {"mode": int(g2.wedge_mode), "slots": g2.wedge_slots is not None,
                          "folded_slots": bool(g2.slots_folded),
                          "row_sums_for": None if g2.row_sums is None else list(g2.row_sums[1:3]),
                          "GB": {"lists": 0.0 if g2.wedge_pos is None else g2.wedge_pos.numel() * 2 / 1e9,
                                 "slots": 0.0 if g2.wedge_slots is None else g2.wedge_slots.numel() * 2 / 1e9,
                                 "row_sums": 0.0 if g2.row_sums is None else g2.row_sums[0].numel() * 8 / 1e9},
                          "note": "65536 = mixed: the lists of the edges into rows of >= 65536 slots are 32-bit for "
                                  "every kernel but the exact slots kernel, which reads FOLDED 16-bit copies and slots "
                                  "(n2v_wedge_slots_fold); (3, 0.7): the row sums of the steps into rows of >= 1024 "
                                  "slots stored once (n2v_edge_row_sums_build)"}
    setup["reference_cap_tables_s"] = {k: v for k, v in sub.items() if k.endswith("_s")}
    del g2
    torch.cuda.empty_cache()
    return res


def bench_weighted(args, torch, rw, dev, W, L):
    """exact walks at (0.5, 2) on the weighted cfg 2 graph (R-MAT 2^20 vertices, 10^7 edges, fp32 weights
    U[0.1, 2]): every start vertex x W walks x L steps in one call of randomwalk.walk -- the per-edge tables, row
    sums and the block summaries of the long rows built before the clock starts, the clock around the whole call (per step: one sort, a wave with
    margins per walker on long rows, a lane per walker on short ones; csrc/n2v_walk_wlanes.hip)"""
    from node2vec_amd import synthetic

    g = synthetic.rmat(20, 5_000_000, device=dev, weights="uniform")
    start = rw.start_vertices(g)
    t0 = time.perf_counter()
    rw.weighted_lanes_tables(g)
    rw.weighted_row_sums(g)
    rw.weighted_hub_summaries(g)
    torch.cuda.synchronize()
    tables_s = time.perf_counter() - t0
    best, st, valid = None, {}, None
    for _ in range(2):
        st = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        walks, valid = rw.walk(g, start, W, L, BIASED_PQ[0], BIASED_PQ[1], 42, stats=st)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    steps = int(valid.sum()) * L
    # small batches (a partition of a Fugue job, a serving call): the same call on 4 710 and 47 104 start vertices
    small = {}
    for nb in (4_710, 47_104):
        bt = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, v2 = rw.walk(g, start[:nb].contiguous(), W, L, BIASED_PQ[0], BIASED_PQ[1], 42)
            torch.cuda.synchronize()
            bt = time.perf_counter() - t0 if bt is None else min(bt, time.perf_counter() - t0)
        small[f"{nb * W}_walkers"] = {"value": int(v2.sum()) * L / bt, "unit": "walk-steps/s", "s_per_call": bt}
    # the rows the steps stood on: what the reference's table of a step is built from (every weight of the row once)
    deg = g.degrees()
    stood = walks[:, :L][valid.bool()] if valid.dtype != torch.bool else walks[:, :L][valid]
    stood = stood[stood >= 0].long()
    mean_row = float(deg[stood].double().mean()) if stood.numel() else 0.0
    del stood
    res = {"graph": "cfg 2 (R-MAT scale 20, 1e7 directed edges), fp32 weights U[0.1, 2]", "p": BIASED_PQ[0],
           "q": BIASED_PQ[1], "walkers": int(valid.numel()), "value": steps / best, "unit": "walk-steps/s",
           "s_per_call": best, "tables_s": tables_s,
           "small_batches": small,
           "walker_steps_left_to_the_exact_kernel": int(st["undecided"]) if "undecided" in st else None,
           "mean_row_slots_per_step": mean_row,
           "roofline": {"bound": "hbm", "unit": "GB/s", "peak": 8000.0,
                        "achieved": steps / best * (4.0 * mean_row + 4.0) / 1e9,
                        "frac": steps / best * (4.0 * mean_row + 4.0) / 8e12, "traffic": None,
                        "algorithmic_formula": "4 B x the slots of the row stood on (every weight of the step's table "
                                               "once) + 4 (path write) per step; the rows of a step are hot in L2 "
                                               "(40 MB of weights in all): the kernels are bound by vector "
                                               "instructions, not by these bytes"},
           "kernel": "walk_weighted_lane_margin_kernel + walk_weighted_margin_kernel (n2v_walk_weighted_step)"}
    # What bounds this leg is vector-instruction issue, not bytes: the committed counter passes
    # of exactly this call (profiles/r9w_wm_pmc.txt: SQ_INSTS_VALU per dispatch, one dispatch of each kernel per
    # step) against what the chip's 1024 SIMDs issue -- one wave-instruction per 4 cycles each -- in the step
    # time measured HERE.  (The kernels have not changed since those passes; `frac` of the HBM object above counts
    # bytes that are hot in L2.)
    valu_per_step = 1.016e9 + 7.264e8 + 1.532e6 + 5.202e5 + 4.803e5  # margin (wave) + lane margin + keys + second chance + exact
    step_s = best / L
    res["valu_issue"] = {"wave_instructions_per_step_committed": valu_per_step,
                         "source": "profiles/r9w_wm_pmc.txt (rocprofv3 --pmc SQ_INSTS_VALU of this call, 160 steps)",
                         "issue_ms_per_step": 1e3 * valu_per_step * 4.0 / (1024 * 2.4e9),
                         "step_ms": 1e3 * step_s,
                         "frac": valu_per_step * 4.0 / (1024 * 2.4e9) / step_s,
                         "note": "256 CUs x 4 SIMDs, a 64-lane vector instruction every 4 cycles per SIMD at 2.4 GHz; "
                                 "the wave kernel of the long rows also waits on three dependent levels of loads "
                                 "(DESIGN.md 5)"}
    del g, walks, valid
    torch.cuda.empty_cache()
    return res


def bench_api(args, torch, rw, dev):
    """What a user of the reference's API gets (reference fugue.py:81-155, embedding.py:120-143), on BASELINE cfg 2
    (R-MAT scale 20, p = 0.5, q = 2, 10 walks x 80 steps from every vertex, dim 128, one epoch), wall clock of each call
    with the device part beside it:
        df = random_walk("hip", graph, params, random_seed=42)      kernel + D2H + the DataFrame of walks
        model = Node2VecGensim(df, w2v_params, random_seed=7).fit() vocabulary + training (the walks come from the
                                                                    device corpus behind the frame, not from its rows)
        vectors = n2v.embedding()                                   D2H + the DataFrame of vectors
    The per-edge tables of the graph are built before the clock starts (one-off, reported)."""
    from node2vec_amd import fugue, synthetic
    from node2vec_amd.embedding import Node2VecGensim

    W, L, p, q = 10, 80, 0.5, 2.0
    g = synthetic.rmat(20, 5_000_000, seed=42, device=dev)
    params = {"num_walks": W, "walk_length": L, "return_param": p, "inout_param": q}
    t0 = time.perf_counter()
    rw.walk(g, rw.start_vertices(g)[:4096].contiguous(), W, L, p, q, 1)  # builds the tables
    torch.cuda.synchronize()
    tables_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    walks, valid = fugue.random_walk_tensors(g, dict(params), None, 42)
    torch.cuda.synchronize()
    walk_device_s = time.perf_counter() - t0
    n_rows = int(valid.sum())
    del walks, valid
    res = {"workload": f"cfg2 R-MAT scale 20, {g.n_vertices} vertices / {g.n_edges} edges, p={p:g} q={q:g}, "
                       f"{W} walks x {L} steps from every vertex, dim 128, 1 epoch",
           "rows": n_rows, "tables_s": tables_s, "random_walk_device_s": walk_device_s}
    kinds = {}
    for kind in ("auto", "arrow"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        df = fugue.random_walk("hip", g, dict(params, walk_column=kind), random_seed=42)
        t1 = time.perf_counter()
        w2v = {"size": 128, "iter": 1, "min_count": 0, "sample": 0.0, "negative": 5, "window": 5}
        n2v = Node2VecGensim(df, w2v, random_seed=7)
        model = n2v.fit()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        emb = n2v.embedding()
        t3 = time.perf_counter()
        kinds[kind] = {"random_walk_s": t1 - t0, "fit_s": t2 - t1, "embedding_s": t3 - t2, "total_s": t3 - t0,
                       "walk_column": str(df["walk"].dtype), "vector_rows": len(emb), "pairs": int(model.pairs_trained)}
        steps, pairs = n_rows * L, int(model.pairs_trained)
        del df, n2v, model, emb
    a = kinds["auto"]
    res.update({"random_walk_s": a["random_walk_s"], "host_conversion_s": a["random_walk_s"] - walk_device_s,
                "fit_s": a["fit_s"], "embedding_s": a["embedding_s"], "total_s": a["total_s"],
                "walk_steps_per_s": steps / a["random_walk_s"], "pairs_per_s": pairs / a["fit_s"],
                "walk_column_default": "one read-only ndarray view per row beyond 2^22 vertices (corpus.list_column); "
                                       "Python lists of Python ints as in the reference: n2v_params['walk_column'] = "
                                       "'list' (3.8e8 int objects here: ~14 GB, not timed)",
                "with_arrow_backed_columns": kinds["arrow"],
                "note": "the walk kernel is %.0f %% of random_walk(): the rest is the D2H copy and the Python objects "
                        "of the DataFrame the reference's API returns" % (100.0 * walk_device_s / a["random_walk_s"])})
    del g
    torch.cuda.empty_cache()
    return res


def ordered_line(out):
    """The ONE line, ordered for its readers: the contract's keys first; the bulky sub-objects
    (setup, sgns, fast_mode, the regimes) in the middle; at the END -- the part of a long line
    that a record keeping only a tail of stdout retains -- the API leg, the exact biased leg, the rooflines
    of the headline (`roofline`: vertex ids out, what random_walk() returns, reference fugue.py:153-155;
    `roofline_pipeline`: ranks out, what fit_streaming launches), the CPU baseline and a compact summary of
    every leg."""
    head = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config"]
    tail = ["api", "biased", "cpu_baseline", "roofline_pipeline", "value_pipeline", "ms_per_step_pipeline",
            "pipeline_consumer", "roofline", "summary"]

    def frac(o):
        return None if not o or "roofline" not in o else round(o["roofline"]["frac"], 4)

    sg = out.get("sgns") or {}
    cpu = out.get("cpu_baseline") or {}
    out["summary"] = {
        "walk_steps_per_s": {
            "headline_vertex_ids_out": out["value"], "pq1_ranks_out_pipeline": out.get("value_pipeline"),
            "exact_biased_0.5_2": (out.get("biased") or {}).get("value"),
            **{"exact_%g_%g" % (r["p"], r["q"]): r["value"] for r in out.get("biased_other_regimes", [])},
            "fast_0.5_2": (out.get("fast_mode") or {}).get("value"),
            "weighted_cfg2_exact_0.5_2": (out.get("weighted") or {}).get("value"),
            "weighted_cfg2_exact_0.5_2_47k_walkers": (((out.get("weighted") or {}).get("small_batches") or {})
                                                      .get("47100_walkers") or {}).get("value"),
            "trim_cap_100000_exact_pq1_ranks_out": ((out.get("reference_trim_cap") or {}).get("exact_pq1_ranks_out") or {}).get("value"),
            "trim_cap_100000_exact_biased_0.5_2": ((out.get("reference_trim_cap") or {}).get("exact_biased_0.5_2") or {}).get("value"),
            "trim_cap_100000_exact_biased_4_0.25": ((out.get("reference_trim_cap") or {}).get("exact_biased_4_0.25") or {}).get("value"),
            "trim_cap_100000_exact_biased_3_0.7": ((out.get("reference_trim_cap") or {}).get("exact_biased_3_0.7") or {}).get("value")},
        "walk_roofline_frac": {"headline_vertex_ids_out": frac(out),
                               "pq1_ranks_out_pipeline": None if "roofline_pipeline" not in out
                               else round(out["roofline_pipeline"]["frac"], 4),
                               "exact_biased_0.5_2": frac(out.get("biased")), "fast_0.5_2": frac(out.get("fast_mode"))},
        "sgns_pairs_per_s": {"per_pair_default": sg.get("value"),
                             "per_pair_plain_stores": (sg.get("plain_stores") or {}).get("value"),
                             "batched_opt_in": (sg.get("batched") or {}).get("value")},
        "sgns_roofline_frac": {"per_pair_default": frac(sg), "batched_opt_in": frac(sg.get("batched"))},
        "sgns_exchange_world": (sg.get("exchange") or {}).get("world"),
        "sgns_exchange_plan_world8": None if "exchange_plan_world8" not in sg else {
            k: sg["exchange_plan_world8"][k] for k in ("hbm_bytes_per_rank", "fits", "bytes_per_link_per_direction_per_sync",
                                                       "link_seconds_per_sync_at_peak")},
        "cpu_port": {"walk_steps_per_s": cpu.get("value"), "sgns_pairs_per_s": (cpu.get("sgns") or {}).get("value"),
                     "cores": cpu.get("cores")},
        "api_cfg2": None if "api" not in out else {k: out["api"].get(k) for k in (
            "random_walk_s", "random_walk_device_s", "fit_s", "embedding_s", "total_s", "walk_steps_per_s", "pairs_per_s")},
        "hbm_peak_allocated_GB": (out.get("setup") or {}).get("hbm_peak_allocated_GB")}
    keys = head + [k for k in out if k not in head and k not in tail] + [k for k in tail if k in out]
    return {k: out[k] for k in keys if k in out}


def prepare_tables(torch, g, p, q, mode, setup, tag, wedge_bytes=None):
    """the one-off tables randomwalk.walk builds on first use for (p, q, mode), timed"""
    from node2vec_amd.randomwalk import _dyadic, tables_regime

    biased = not (p == 1.0 and q == 1.0)

    def timed(key, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        setup[f"{tag}_{key}_s"] = time.perf_counter() - t0

    if g.unit_weights:
        if mode == "fast" and g.pivots is None:
            timed("pivots_build", g.build_pivots)
        if biased and g.edge_classes is None and (mode == "fast" or tables_regime(p, q)):
            timed("edge_classes_build", g.build_edge_classes)
        if biased and (mode == "fast" or tables_regime(p, q)) and g.wedge_off is None and not g.wedge_tried:
            g.wedge_tried = True
            # (wedge_bytes: the budget of a leg that owns the card; None = the library's own rule, half of what is free)
            timed("wedge_table_build", lambda: g.build_wedges(max_bytes=wedge_bytes))
            setup[f"{tag}_wedge_table_GB"] = 0.0 if g.wedge_off is None else (
                g.wedge_off.numel() * 8 + g.wedge_pos.numel() * g.wedge_pos.element_size()) / 1e9
            setup[f"{tag}_wedge_slots_GB"] = 0.0 if g.wedge_slots is None else g.wedge_slots.numel() * 2 / 1e9

        if (biased and mode == "exact" and tables_regime(p, q) and g.wedge_slots is not None
                and (g.wedge_mode == 0 or g.slots_folded) and not (_dyadic(p) and _dyadic(q))):
            timed("row_sums_build", lambda: g.build_row_sums(p, q))  # (walk() would build them on first use)
        if not biased and mode == "exact" and g.hops8 is None and not g.hops8_tried:
            timed("hop8_table_build", g.build_hops8)  # 8 bytes per edge; p = q = 1 only
        if (biased or mode != "exact" or g.hops8 is None) and (
                g.hops is None or (g.edge_classes is not None and not g.hops_have_classes)):
            # (exact biased walks: class words with inline return positions, the slots kernel's form)
            timed("hop_table_build", lambda: g.build_hops(inline_rpos=(mode == "exact" and biased)))
    elif g.slots is None and (mode == "fast" or not biased):
        timed("alias_tables_build", g.build_alias)


def kernel_name(g, p, q):
    """the kernel n2v_walk dispatches exact mode to (node2vec_amd/csrc/n2v_capi.hip)"""
    from node2vec_amd.randomwalk import lanes_regime, tables_regime

    if g.unit_weights:
        if p == 1.0 and q == 1.0:
            return "walk_uniform_kernel"
        if tables_regime(p, q) and g.hops is not None and g.wedge_off is not None:
            if g.wedge_slots is not None and (g.wedge_mode == 0 or g.slots_folded):  # (all four instances)
                return "walk_exact_wedge_slots_kernel"
            return "walk_exact_wedge_kernel"
        if lanes_regime(p, q) and g.edge_classes is not None:
            return "walk_exact_unit_lanes_kernel"
        return "walk_exact_unit_kernel"
    return "walk_fast_kernel" if (p == 1.0 and q == 1.0) else "walk_exact_kernel"


def roofline(kernel, res, leg, config, p, q, mode, ref_bytes):
    """HBM roofline of one walk kernel.

    `traffic` = bytes per launch past L2 from the committed rocprofv3 --pmc passes of THIS
    workload (profiles/pmc_traffic.json, FETCH_SIZE + WRITE_SIZE corrected as
    MI355X_MICROARCH.md "HBM" prescribes), or null.  `achieved` = those bytes / the kernel's
    HIP-event duration when they exist -- the bytes the kernel really moves, so frac <= 1 by
    construction; without a profile, the kernel's own algorithmic bytes (what its algorithm
    must touch at byte granularity, formula in `algorithmic_formula`).  The reference
    algorithm's bytes are reported beside it, never as the roofline."""
    per_launch_steps = leg.batch * leg.W * leg.L
    hops = leg.g.hops is not None
    ranked = bool(getattr(leg, "rank_ids", False))
    hop8 = leg.g.hops8 is not None and mode == "exact" and p == 1.0 and q == 1.0 and not ranked
    if ranked:
        head = 0 if leg.g.rank_head is None else leg.g.rank_head.numel()
        alg = 4 + 4
        formula = ("4 (table entry: the neighbour's degree rank; its row follows from its degree class, "
                   f"{leg.g.rank_class_first.numel()}-entry table in LDS"
                   f"{f', the first {head} ranks from a cached table' if head else ''}) + 4 (path write) per step")
    elif mode == "fast":
        t = res["trials"] / max(res["steps_done"], 1)
        alg = (t * 16 + 4) if hops else (16 + t * 16 + 4)
        formula = ("trials * 16 (hop entry: neighbour, its row and degree, edge classes) + 4 (path)"
                   if hops else "16 (rowptr pair) + trials * 16 (slot) + 4 (path)") + \
            "; membership searches extra"
    elif hop8:
        cb, rb = leg.g.hops8_bits
        esc = (1 << (64 - cb - rb)) - 1
        share = escape_share(leg)
        alg = 8 + 4 + 16 * share
        formula = (f"8 (hop entry: neighbour id {cb} bits | its row start {rb} bits"
                   f"{' (rows padded to 8 entries)' if leg.g.hops8_shift else ''} | degree code "
                   f"{64 - cb - rb} bits) + 4 (path write) + 16 x {share:.3f} (share of the steps whose "
                   f"new vertex has degree >= {esc}: the degree is read from rowptr, cached)")
    elif p == 1.0 and q == 1.0:
        alg = (16 + 4) if hops else (16 + 4 + 4)
        formula = ("16 (hop entry: col[pick] with the row pointer and degree of that neighbour) + "
                   "4 (path write) per step" if hops else
                   "16 (rowptr pair of v) + 4 (col[pick]) + 4 (path write) per step")
    else:
        alg = (16 + 4) if hops else (16 + 4 + 4 + 4)
        wedges = leg.g.wedge_off is not None
        slots = wedges and getattr(leg.g, "wedge_slots", None) is not None and "slots" in kernel
        if slots and leg.g.edge_classes is not None:
            # a walk visits directed edges ~uniformly, and a step reads the 32-byte slot of the edge
            # it came along when that edge has shared neighbours (the steps that pair without a list
            # and read the slot for the return position alone are not counted)
            share = float(((leg.g.edge_classes & 0xffffff) != 0).float().mean())
            alg = 16 + 4 + 32 * share
        formula = (("16 (hop entry) + 4 (path write)" if hops else
                    "16 (rowptr pair) + 4 (edge class word) + 4 (col[pick]) + 4 (path write)") +
                   " per step" +
                   (", + 32 (the edge's wedge slot: return position + the list itself up to 14 entries, "
                    "else its offset and eight pivots) x the share of edges with shared neighbours (counted "
                    "in the figure above; probes of longer lists and slots read for the return position "
                    "alone are not)" if slots else
                    ", + 8 (wedge offset) + 2 per probe of the edge's shared-position list on steps "
                    "whose edge has shared neighbours; the steps that run the pairing read 2 bytes per "
                    "shared neighbour at most" if wedges else
                    ", + 4 per probe of the membership search; steps that run the pairing read both rows"))
    kernel_key = kernel + (":ranked" if ranked else ":hop8" if hop8 else (":hops" if hops else "")) + (
        ":wedges" if (leg.g.wedge_off is not None and mode == "exact" and not (p == 1.0 and q == 1.0)) else "")
    # bytes past L2 per launch from the COMMITTED rocprofv3 --pmc passes of this command: not
    # observed in this run, so it is reported as `traffic_committed`; `traffic` (the contract's
    # live PMC figure) stays null -- bench.py does not run under the profiler
    traffic = pmc_traffic(config, kernel_key, p, q, leg.batch)
    alg_launch = alg * per_launch_steps
    ach = alg_launch / res["kernel_s"]
    r = {"bound": "hbm", "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
         "frac": ach / HBM_PEAK, "traffic": None, "traffic_committed": traffic, "kernel": kernel,
         "hop_table": "4-byte degree-ranked" if ranked else "8-byte" if hop8 else hops,
         "wedge_table": ":wedges" in kernel_key,
         "kernel_ms": 1e3 * res["kernel_s"],
         "output_buffer_audition": getattr(leg, "audition", None) or None,
         "achieved_from": "algorithmic bytes (formula below) x walk-steps per launch / HIP-event duration",
         "frac_algorithmic": ach / HBM_PEAK,
         "algorithmic_bytes_per_launch": alg_launch, "algorithmic_bytes_per_walk_step": alg,
         "algorithmic_formula": formula,
         "traffic_committed_source": None if not traffic else
         "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command: committed, "
         "not observed in this run)"}
    if traffic:
        # counter bytes: what crossed the memory side of L2.  The walk kernels' reads are random
        # gathers and each costs one 64-byte sector whatever it uses of it
        # (profiles/r02_gather_fetch_calibration.txt), so counter bytes / algorithmic bytes is the
        # sector overhead of a gather, not re-reads.
        sectors = traffic / 64.0
        r["frac_counter"] = traffic / res["kernel_s"] / HBM_PEAK
        r["counter_over_algorithmic"] = traffic / alg_launch
        r["sectors_per_walk_step"] = sectors / per_launch_steps
    g_meas = MEASURED.get("gather_independent")
    if ranked and MEASURED.get("ranked_chain"):
        gathers = res["steps_done"] / max(res["launches"], 1) / res["kernel_s"]
        r["gather_ceiling"] = {
            "gathers_per_walk_step_min": 1.0, "gather_bytes": 4, "kernel_Ggathers_per_s": gathers / 1e9,
            "measured_Ggathers_per_s_dependent_chain": MEASURED["ranked_chain"] / 1e9,
            "measured_Ggathers_per_s_chain_with_class_search": MEASURED.get("ranked_chain_class_search", 0.0) / 1e9,
            "frac": gathers / MEASURED["ranked_chain"],
            "source": "n2v_mem_probe modes 1 (4-byte) / 4 over the ranked table of this graph, in this "
                      "process (untimed): what one dependent chain of random 4-byte reads per lane sustains"}
    elif g_meas:
        # the binding resource, observed on this box: random 16-byte gathers per second
        gathers = (res["trials"] if mode == "fast" and res.get("trials") else res["steps_done"]) / \
            max(res["launches"], 1) / res["kernel_s"]
        r["gather_ceiling"] = {
            "gathers_per_walk_step_min": 1.0 if (hops or hop8) else 2.0,
            "gather_bytes": MEASURED.get("gather_bytes", 16),
            "kernel_Ggathers_per_s": gathers / 1e9,
            "measured_Ggathers_per_s_independent": g_meas / 1e9,
            "measured_Ggathers_per_s_dependent_chain": MEASURED.get("gather_chain", 0.0) / 1e9,
            "frac": gathers / g_meas,
            "source": "n2v_mem_probe modes 0 / 1 over the hop table of this graph, in this process "
                      "(untimed); counts ONE table gather per step (per trial in fast mode) -- list and "
                      "membership probes of the biased kernels are extra"}
    if ref_bytes:
        r["reference_algorithmic_bytes_per_walk_step"] = ref_bytes
        r["reference_algorithmic_GBps_equivalent"] = ref_bytes * per_launch_steps / res["kernel_s"] / 1e9
    return r


def escape_share(leg):
    """share of the walk steps of the last launch whose new vertex has a degree the 8-byte hop
    entry cannot hold (the degree is then read from rowptr)"""
    cb, rb = leg.g.hops8_bits
    esc = (1 << (64 - cb - rb)) - 1
    deg = leg.g.degrees()
    w = leg.walks[:65536][leg.valid[:65536].bool()][:, 1:].long()
    return float((deg[w] >= esc).float().mean()) if w.numel() else 0.0


def measure_ceilings(torch, buffer, tag_rows=None, gather_width=16, classes=None):
    """n2v_mem_probe on `buffer` (a device tensor >> Infinity Cache): random 16-byte gathers,
    and -- when tag_rows names a row size -- random row reads / read-modify-writes.  Untimed
    with respect to the bench's K steps; HIP events around each probe launch."""
    import ctypes as C

    from node2vec_amd import _lib

    L = _lib.load()
    nbytes = buffer.numel() * buffer.element_size()
    sink = torch.zeros(4, dtype=torch.int32, device=buffer.device)
    out = {}

    def run(mode, iters, row_bytes):
        n = C.c_int64(0)
        best = None
        for rep in range(3):  # first = warm-up
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _lib.check(L.n2v_mem_probe(buffer.data_ptr(), nbytes, mode, iters, row_bytes, C.byref(n),
                                       sink.data_ptr(), _lib.current_stream_ptr()), "n2v_mem_probe")
            b.record()
            torch.cuda.synchronize()
            dt = 1e-3 * a.elapsed_time(b)
            if rep:
                best = dt if best is None else min(best, dt)
        return n.value / best

    if classes is not None:  # the ranked walk's shape; kept apart from the hop table's figures
        out["ranked_chain"] = run(1, 256, gather_width)
        out["ranked_chain_class_search"] = run(4, 256, int(min(max(classes, 64), 8192)))
        MEASURED.update(out)
        return out
    if tag_rows is None:
        out["gather_independent"] = run(0, 256, gather_width)
        out["gather_chain"] = run(1, 256, gather_width)
    else:
        out[f"rows{tag_rows}_read"] = run(2, 512, tag_rows)
        out[f"rows{tag_rows}_read_modify_write"] = run(3, 512, tag_rows)
    if tag_rows is None:
        out["gather_bytes"] = gather_width
    MEASURED.update(out)
    return out


def pmc_traffic(config, kernel, p, q, batch):
    """bytes per launch from profiles/pmc_traffic.json for exactly this workload, else None"""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        table = json.load(f)
    key = f"{config}:{kernel}:p{p}:q{q}:batch{batch}"
    ent = table.get(key)
    return float(ent["hbm_bytes_per_launch"]) if ent else None


def bench_sgns(args, cfg, torch, dist, g, walks, rank, world, barrier, use_dist, n_blocks=1, vocab=None):
    """K launches of the SGNS kernel (embedding-updates/s).  One step = one launch over the
    walks of `sgns_vertices` start vertices (x W rows of L+1 tokens), vocabulary = the one
    fit_streaming builds for this corpus (pipeline.corpus_vocabulary: every vertex the walks
    visit, min_count=0, sample=0, index order = descending token count over the WHOLE virtual
    corpus), dim from the config, window 5, k=5.  Unit: one positive (centre, context) pair
    with its k negative targets.  With N GPUs every rank trains its own walks on a full
    replica; the exchange step (bf16 delta all-reduce, sgns.DeltaSync) is timed beside it."""
    from node2vec_amd import sgns

    dev = walks.device
    dim = cfg["dim"]
    assert vocab is not None
    index_of = vocab.index_of
    model = sgns.SgnsModel(vocab, dim, 5, 5, seed=1, sample=0.0, device=dev)
    idx_all = index_of[walks.long()].contiguous()
    rows = idx_all.shape[0] // n_blocks
    blocks = [idx_all[i * rows:(i + 1) * rows] for i in range(n_blocks)]  # launch j trains blocks[j % n_blocks]
    idx = blocks[0]
    launches = [0]

    def next_block():
        launches[0] += 1
        return blocks[(launches[0] - 1) % n_blocks]

    row_ceiling = None
    if rank == 0:
        # K3's ceiling on this box: random `dim * 4`-byte rows read / read-modified-written by one
        # wave each over a scratch buffer far beyond the Infinity Cache (untimed)
        free = torch.cuda.mem_get_info()[0]
        nb = int(min(16 << 30, free // 4)) // 4096 * 4096
        if nb >= (1 << 30) and dim * 4 in (512, 1024, 2048):
            scratch = torch.empty(nb // 4, dtype=torch.float32, device=dev)
            scratch.zero_()
            row_ceiling = measure_ceilings(torch, scratch, dim * 4)
            # GB/s of row bytes; a read-modify-write moves the row twice (as 8*D*(2+k) counts it)
            row_ceiling = {k: v * dim * 4 * (2 if k.endswith("write") else 1) / 1e9
                           for k, v in row_ceiling.items()}
            row_ceiling["buffer_GB"] = nb / 1e9
            del scratch
            torch.cuda.empty_cache()
    for k in range(args.warmup):
        model.train_block(next_block(), 0.025, k * rows)
    model.pairs.zero_()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        model.train_block(next_block(), 0.025, (args.warmup + k + rank * 1000) * rows)
        ev[k][1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    pairs = float(model.pairs.item())
    elapsed, pairs_total = reduce_job(torch, dist, use_dist, dev, elapsed, pairs)
    kernel_s = 1e-3 * sum(a.elapsed_time(b) for a, b in ev) / args.steps
    bytes_per_pair = 8 * dim * (2 + 5)            # 2*4*D*(2+k)
    flops_per_pair = (1 + 5) * 6 * dim + dim      # (1+k)*6D + D
    ach = pairs / args.steps * bytes_per_pair / kernel_s
    traffic = pmc_traffic(args.config, "sgns_kernel", 0, 0, rows)
    res = {"value": pairs_total / elapsed, "unit": "embedding-updates/s (pairs incl. k=5 negatives)",
           "row_updates_per_s": pairs_total / elapsed * 6, "ms_per_step": 1e3 * elapsed / args.steps,
           "dtype": "f32", "config": {"dim": dim, "window": 5, "negative": 5, "rows_per_step": rows,
                                      "distinct_blocks_of_rows": n_blocks,
                                      "n_vocab": len(vocab), "sample": 0, "min_count": 0,
                                      "vocabulary": "pipeline.corpus_vocabulary: token counts of the whole "
                                                    "virtual corpus, descending (fit_streaming's pass 1)",
                                      "model_bytes": 2 * len(vocab) * dim * 4},
           "roofline": {"bound": "hbm", "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9,
                        "unit": "GB/s", "frac": ach / HBM_PEAK, "traffic": None,
                        "traffic_committed": traffic, "kernel": "sgns_kernel", "kernel_ms": 1e3 * kernel_s,
                        "achieved_from": "algorithmic bytes (SURVEY 8d: 8*D*(2+k) per pair).  `traffic` = "
                                         "2 * FETCH_SIZE + WRITE_SIZE of a committed rocprofv3 pass (the "
                                         "counter tallies half of this kernel's row reads: calibrated in "
                                         "profiles/r3e_fetch_write_calibration_rows.txt); it is bytes past "
                                         "L2, Infinity-Cache hits included, so it can exceed what HBM "
                                         "itself delivers",
                        "traffic_GBps": None if not traffic else traffic / kernel_s / 1e9,
                        "frac_algorithmic": ach / HBM_PEAK,
                        "frac_counter": None if not traffic else traffic / kernel_s / HBM_PEAK,
                        "measured_row_ceiling_GBps": row_ceiling,
                        "algorithmic_bytes_per_pair": bytes_per_pair,
                        "fma_utilisation": pairs / args.steps * flops_per_pair / kernel_s / FP32_PEAK}}
    # The launches above ran the DEFAULT: atomic adds on the rows that more than one wave holds at
    # a time on average (SgnsModel.auto_hub_rows), the reference's concurrency regime (gensim: <= 16
    # threads) at 8192 waves.  The same launches with plain stores everywhere (hub_rows = 0, gensim's
    # code as written; cfg 2 link AUC 0.897 instead of 0.909) for comparison:
    res["hub_rows_auto"] = {"rows": int(model.hub_rows or 0), "n_vocab": len(model.vocab),
                            "waves_in_flight": model.hub_waves, "rows_the_lambda_rule_selects": model.hub_candidates,
                            "their_share_of_all_row_holds": model.hub_share,
                            "rule": "rows with waves x (token share + k x negative-draw share) >= 1.5, when together "
                                    "they carry >= 10 % of all row-holds (cfg 2: 18 %, + 0.009 link AUC; cfg 3: 5 %, "
                                    "no difference in AUC for 12 % of the rate: profiles/r10r_auc_cfg3.log)"}
    if not args.no_hub:
        auto_rows = model.hub_rows
        model.hub_rows = 0
        try:
            model.train_block(next_block(), 0.025, 50 * rows)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            p0 = float(model.pairs.item())
            for k in range(args.steps):
                model.train_block(next_block(), 0.025, (60 + k + rank * 1000) * rows)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res["plain_stores"] = {"value": (float(model.pairs.item()) - p0) / dt,
                                   "unit": "embedding-updates/s on this GPU",
                                   "ms_per_step": 1e3 * dt / args.steps,
                                   "what": "hub_rows = 0: every row updated by read-modify-write stores "
                                           "(w2v_params['hub_rows'] = 0).  This leg runs AFTER the default "
                                           "one and goes through the same blocks of rows a second time: the "
                                           "model has seen them once, more targets reach |f| >= 6 and are "
                                           "skipped, so its launches are a few per cent shorter whatever "
                                           "hub_rows is"}
            if auto_rows:
                res["hub_rows_auto"]["throughput_vs_plain_stores"] = res["value"] / max(res["plain_stores"]["value"], 1.0)
            else:
                res["hub_rows_auto"]["note"] = ("no row qualifies at this vocabulary size: the two legs run the same "
                                                "kernel configuration, their difference is run-to-run")
        finally:
            model.hub_rows = auto_rows
    if not args.no_batched and dim in (64, 128, 256):
        res["batched"] = bench_sgns_batched(args, torch, dist, model, next_block, rows, rank, barrier,
                                            use_dist, dev, dim)
    # what ONE rank of an 8-GPU job holds and moves per exchange, from the size rules the exchange allocates by
    # (sgns.exchange_plan; a two-rank gloo test compares them with live buffers) -- no 8-GPU box is needed to know it
    resident = sum(t.numel() * t.element_size() for t in (g.rowptr, g.col) if t is not None)
    res["exchange_plan_world8"] = sgns.exchange_plan([tuple(model.syn0.shape), tuple(model.syn1neg.shape)], 8, "bf16",
                                                     resident_bytes=resident + int(30e9))
    res["exchange_plan_world8"]["resident_note"] = ("CSR of this graph + 30 GB for the walk tables and one corpus batch "
                                                    "(cfg 4: hop table 12.1 GB, ranked form 3.8 GB, batch + index 6.8 GB)")
    if use_dist:  # the exchange step of the multi-GPU path, timed once (blocking, bf16 deltas)
        # (at N = 1 under torch.distributed.run the same calls run on a group of one rank: the
        # RCCL path is rehearsed, the mean is of one replica)
        torch.cuda.reset_peak_memory_stats()
        sync = sgns.DeltaSync(model, wire="bf16", rehearse=True)
        barrier()
        t0 = time.perf_counter()
        sync.sync(blocking=True)
        barrier()
        dt = time.perf_counter() - t0
        peak_live = torch.tensor([float(torch.cuda.max_memory_allocated())], dtype=torch.float64, device=dev)
        dist.all_reduce(peak_live, op=dist.ReduceOp.MAX)
        step_s = 1e-3 * res["ms_per_step"]
        every = max(1, -(-int(1e6 * dt * 0.9) // max(1, int(1e6 * 0.1 * step_s))))
        res["exchange"] = {"world": dist.get_world_size(), "backend": "nccl (RCCL)",
                           "path": "n2v_delta_pack -> all_to_all (bytes) -> n2v_delta_reduce (fp32, rank "
                                   "order) -> all_gather (bytes) -> n2v_delta_apply",
                           "blocks_exchanged": sync.exchanged_blocks,
                           "tensors_on_device": bool(sync.on_gpu),
                           "hbm_peak_allocated_GB_with_exchange_live": float(peak_live.item()) / 1e9,
                           "delta_allreduce_s": dt, "wire_dtype": sync.wire_dtype_name,
                           "wire_bytes_per_rank": sync.wire_bytes,
                           "share_of_step_time_if_every_launch": dt / (dt + step_s),
                           "sync_every_for_10pct": every,
                           "share_at_that_period": dt / (dt + every * step_s)}
        del sync
    return res, model


def bench_sgns_batched(args, torch, dist, model, next_block, rows, rank, barrier, use_dist, dev, dim):
    """The opt-in batched trainer (n2v_sgns_params.batched: the k negatives are drawn once per
    centre position and shared by its pairs -- NOT gensim's sampling) on the same model and
    corpus: K launches.  A position is three small dense products on v_mfma_f32_16x16x4_f32 and
    moves 8*D*(2+k) bytes of HBM per POSITION instead of per pair."""
    model.batched = True
    keep_hub, model.hub_rows = model.hub_rows, 0  # the opt-in kernel runs with plain stores
    try:
        for k in range(args.warmup):
            model.train_block(next_block(), 0.025, (100 + k) * rows)
        model.pairs.zero_()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(args.steps)]
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            ev[k][0].record()
            idx = next_block()
            model.train_block(idx, 0.025, (200 + k + rank * 1000) * rows)
            ev[k][1].record()
        barrier()
        elapsed = time.perf_counter() - t0
        pairs = float(model.pairs.item())
    finally:
        model.batched = False
        model.hub_rows = keep_hub
    elapsed, pairs_total = reduce_job(torch, dist, use_dist, dev, elapsed, pairs)
    kernel_s = 1e-3 * sum(a.elapsed_time(b) for a, b in ev) / args.steps
    positions = float((idx >= 0).sum().item())  # per launch: every token is a centre position
    bytes_per_position = 8 * dim * (2 + 5)
    ach = positions * bytes_per_position / kernel_s
    mfma_per_position = dim // 4 + (dim // 16) * 3 + (dim // 16) * 2  # F + Tgt update + Ctx update
    sclk = 2.4e9
    return {"value": pairs_total / elapsed, "unit": "embedding-updates/s (pairs; k=5 negatives shared "
                                                    "by the pairs of a centre position)",
            "ms_per_step": 1e3 * elapsed / args.steps, "dtype": "f32",
            "sampling": "opt-in, NOT gensim's per-pair negatives: one draw of k negatives per centre "
                        "position (Ji et al. 2016); bit-identical to its own oracle in deterministic mode",
            "pairs_per_position": pairs / args.steps / max(positions, 1.0),
            "roofline": {"bound": "hbm", "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": ach / HBM_PEAK,
                         "traffic": None,
                         "traffic_committed": pmc_traffic(args.config, "sgns_batched_kernel", 0, 0, rows),
                         "kernel": "sgns_batched_kernel",
                         "kernel_ms": 1e3 * kernel_s,
                         "achieved_from": "algorithmic bytes 8*D*(2+k) per centre POSITION (centre + k "
                                          "negative rows and one context row, read and written once)",
                         "algorithmic_bytes_per_position": bytes_per_position,
                         "mfma": {"instruction": "v_mfma_f32_16x16x4_f32 (exact f32, 32 cycles)",
                                  "instructions_per_position": mfma_per_position,
                                  "busy_fraction_analytic": positions * mfma_per_position * 32 /
                                  (kernel_s * sclk * 1024),
                                  "note": "instructions x 32 cycles / (1024 SIMDs x 2.4 GHz x kernel time); "
                                          "the measured SQ_VALU_MFMA_BUSY_CYCLES are in profiles/"}}}


def host_cores():
    """threads this process may really use: affinity mask, capped by a cgroup CPU quota"""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def cpu_baseline(args, cfg, torch, g, start_all, W, L, p, q, sg_walks, model):
    """The CPU oracle -- a C port of the reference's algorithm (per-step biased alias rebuild
    + two-uniform draw; per-pair SGNS) -- timed on this box's host cores on a bounded sample
    of the same workload: single-thread and all cores (OpenMP over walkers; hogwild threads
    over sentences, as gensim's workers)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np

    import n2v_oracle

    cores = host_cores()
    budget = args.cpu_seconds
    rowptr, col = g.rowptr.cpu().numpy(), g.col.cpu().numpy()
    w = None if g.unit_weights else g.w.cpu().numpy()
    starts = start_all.cpu().numpy()
    rng = np.random.default_rng(42)

    def sample(n):
        n = int(max(1, min(len(starts), n)))
        return np.sort(rng.choice(starts, n, replace=False)).astype(np.int32)

    def run(s, threads):
        t0 = time.perf_counter()
        _, valid = n2v_oracle.random_walk(rowptr, col, w, s, W, L, p, q, 42, n_threads=threads)
        dt = time.perf_counter() - t0
        return float(valid.sum()) * L / dt, dt

    rate0, _ = run(sample(16), 1)                       # calibration
    per_vertex = W * L / rate0
    s1 = sample(0.4 * budget / per_vertex)
    r1, dt1 = run(s1, 1)
    per_vertex = W * L / r1
    # all cores: at least 16 dynamic chunks (4 walkers each) per thread
    sN = sample(max(cores * 8, 0.6 * budget * cores / per_vertex))
    rN, dtN = run(sN, cores)
    out = {"value": rN, "unit": "walk-steps/s", "cores": cores, "kind": "port",
           "sample": f"{len(sN)} start vertices (uniform sample, seed 42) x {W} walks x {L} steps "
                     f"at p={p} q={q} on the bench graph, {dtN:.1f} s, oracle/n2v_oracle.c, "
                     f"OpenMP {cores} threads (affinity mask / cgroup quota; os.cpu_count()="
                     f"{os.cpu_count()})",
           "single_thread": {"value": r1, "unit": "walk-steps/s", "cores": 1,
                             "sample": f"{len(s1)} start vertices x {W} x {L}, {dt1:.1f} s"},
           "reference_python": {"value": 2.8e4, "unit": "walk-steps/s", "cores": 1,
                                "note": "the reference's own next_step_random_walk, CPython, measured "
                                        "in the survey container (BASELINE.md section 2); cannot run here"}}
    if sg_walks is not None:
        out["sgns"] = cpu_baseline_sgns(args, cfg, torch, g, sg_walks, cores, np, n2v_oracle)
    return out


def cpu_baseline_sgns(args, cfg, torch, g, walks, cores, np, n2v_oracle):
    """oracle/n2v_oracle_sgns.c (restated gensim-3.8 SGNS, parity unpinned) on host cores.
    The CPU model is capped at 16 M rows x dim (2 x 8 GB at dim 128: far beyond every CPU
    cache, so row gathers are DRAM-bound like the full 100 M-row model) and tokens are
    folded into it; initialising two 51 GB host matrices would dominate the bench."""
    from concurrent.futures import ThreadPoolExecutor

    from node2vec_amd import sgns

    dim = cfg["dim"]
    n_cpu = int(min(g.n_vertices, 16_000_000))
    deg = g.degrees().clamp(min=1)
    order = torch.sort(deg, descending=True, stable=True).indices[:n_cpu]
    cum = sgns.make_cum_table(deg[order]).cpu().numpy()
    index_of = torch.empty(g.n_vertices, dtype=torch.int32, device=walks.device)
    full = torch.sort(deg, descending=True, stable=True).indices
    index_of[full] = (torch.arange(g.n_vertices, device=walks.device) % n_cpu).to(torch.int32)
    idx = index_of[walks.long()].cpu().numpy()
    del index_of, full
    rng = np.random.default_rng(1)
    block = ((rng.random((1 << 20, dim), dtype=np.float32) - 0.5) / dim).astype(np.float32)
    syn0 = np.empty((n_cpu, dim), np.float32)
    for lo in range(0, n_cpu, 1 << 20):
        syn0[lo:lo + (1 << 20)] = block[:min(1 << 20, n_cpu - lo)]
    syn1 = np.zeros((n_cpu, dim), np.float32)
    syn1[:] = 0.0  # touch the pages: no first-touch faults inside the timed region
    et = sgns.exp_table()

    def train(rows, base):
        return n2v_oracle.sgns_train(rows, syn0, syn1, cum, None, et, n_cpu, base, 1, dim, 5, 5, 0.025)

    t0 = time.perf_counter()
    pairs0 = train(idx[:8], 0)
    rate0 = pairs0 / max(time.perf_counter() - t0, 1e-6)
    pairs_per_row = pairs0 / 8
    budget = args.cpu_seconds
    n1 = int(max(8, min(len(idx), 0.4 * budget * rate0 / pairs_per_row)))
    t0 = time.perf_counter()
    p1 = train(idx[:n1], 1 << 20)
    dt1 = time.perf_counter() - t0
    r1 = p1 / dt1
    nN = int(max(cores * 4, min(len(idx), 0.6 * budget * r1 * cores / pairs_per_row)))
    per = max(1, nN // (cores * 4))
    jobs = [(idx[lo:lo + per], (2 << 20) + lo) for lo in range(0, nN, per)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:  # ctypes releases the GIL: real threads, hogwild
        pN = sum(ex.map(lambda a: train(*a), jobs))
    dtN = time.perf_counter() - t0
    return {"value": pN / dtN, "unit": "embedding-updates/s (pairs incl. k=5 negatives)",
            "cores": cores, "kind": "port",
            "sample": f"{nN} walks of {idx.shape[1]} tokens in {len(jobs)} hogwild jobs on {cores} "
                      f"threads, {dtN:.1f} s; model {n_cpu} x {dim} fp32 (tokens folded modulo "
                      f"{n_cpu}), oracle/n2v_oracle_sgns.c",
            "single_thread": {"value": r1, "cores": 1, "sample": f"{n1} walks, {dt1:.1f} s"}}


if __name__ == "__main__":
    main()
