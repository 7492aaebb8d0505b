"""Device walk sampler: host-side mirror of the reference's node2vec/randomwalk.py.

The reference's per-partition transformers (initiate_random_walk :279-296,
next_step_random_walk :300-339, to_path :343-349) and the loop that drives them
(fugue.py:137-153) run as ONE kernel launch here (n2v_walk, include/n2v_hip.h);
this module only marshals tensors.  Semantics kept: W walks per start vertex,
start set = vertices with out-edges, walkers that hit a sink vanish, every
emitted walk has walk_length + 1 vertices and starts at its source.
"""
import os
from typing import Optional, Tuple

import torch

from node2vec_amd import _lib
from node2vec_amd.graph import DeviceGraph

MODES = {"exact": _lib.WALK_EXACT, "fast": _lib.WALK_FAST}


def _dyadic(x: float) -> bool:
    """1/x * 2^20 is an exact integer below 2^31 (the lanes kernel's integer row sum)"""
    t = (1.0 / x) * 1048576.0
    return 0.0 < t < 2147483648.0 and t == float(int(t))


def lanes_regime(p: float, q: float) -> bool:
    """The (p, q) for which exact walks on a unit-weight graph use the per-edge class counts
    (same predicate as n2v_walk_exact_unit_try, csrc/n2v_walk_unit.hip): dyadic values with
    1/q <= 1 and 1/p >= 1/q, i.e. the bulk of every row ("other" neighbours) is underfull."""
    return _dyadic(p) and _dyadic(q) and 1.0 / q <= 1.0 and 1.0 / p >= 1.0 / q


def tables_regime(p: float, q: float) -> bool:
    """The (p, q) for which exact walks on a unit-weight graph run from the per-edge tables alone
    (class counts + wedge lists + hop table, csrc/n2v_walk_wedge.hip): every pair but p = q = 1
    (which needs no table but the hop table).  Dyadic 1/p, 1/q: the pairing of a row has a closed
    form whether the bulk class "other" is alone on its stack (q >= 1 with p <= q, q <= 1 with
    p >= q) or shares it with the return slot; other values of ordinary magnitude (2^-20 .. 2^20,
    the bound of n2v_walk_exact_unit_try): the row is summed in the reference's order and the
    pairing replayed run by run, one lane per walker all the same."""
    if p == 1.0 and q == 1.0:
        return False
    if _dyadic(p) and _dyadic(q):
        return True
    return all(2.0 ** -20 <= 1.0 / x <= 2.0 ** 20 for x in (p, q))


def fresh_seed() -> int:
    """random_seed=None in the reference means an unseeded `random` (randomwalk.py:314)."""
    return int.from_bytes(os.urandom(8), "little")


def start_vertices(graph: DeviceGraph, walk_seed_ids=None) -> torch.Tensor:
    """fugue.py:132-134: ids of df_adj (vertices with >= 1 out-edge), inner-joined
    with walk_seed.id when given.  Sorted ascending, int32, on the graph's device."""
    has_out = graph.degrees() > 0
    if walk_seed_ids is not None:
        ids = torch.as_tensor(walk_seed_ids).to(device=graph.device, dtype=torch.int64).reshape(-1)
        ids = torch.unique(ids)
        ids = ids[(ids >= 0) & (ids < graph.n_vertices)]
        ids = ids[has_out[ids]]
    else:
        ids = torch.nonzero(has_out).reshape(-1)
    return ids.to(torch.int32)


def walk(graph: DeviceGraph, start_ids: torch.Tensor, num_walks: int, walk_length: int,
         return_param: float, inout_param: float, seed: int, mode: str = "exact",
         out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, check: bool = True,
         stats: Optional[dict] = None, use_edge_classes: bool = True, use_hops: bool = True,
         use_wedges: bool = True, use_wedge_kernel: bool = True, use_hops8: bool = True,
         use_workspace: bool = False, use_wedge_slots: bool = True, use_ranked: Optional[bool] = None,
         rank_ids: bool = False, use_weighted_lanes: Optional[bool] = None, use_row_sums: bool = True):
    """Launch K2.  Returns (walks int32 [n_start*num_walks, walk_length+1], valid bool).

    mode "fast", and on weighted graphs mode "exact" with return_param == inout_param == 1
    (the reference's defaults), read the first-order alias tables of the graph: they are
    built on first use (graph.build_alias(): 16 bytes per edge plus the search index, kept
    on the graph).  Exact walks on a unit-weight graph with other p, q use the per-edge
    class counts (graph.build_edge_classes(): 4 bytes per edge, built on first use;
    use_edge_classes=False walks without them, one wave per walker: same bits, slower).
    Unit-weight graphs also get the hop table (graph.build_hops(): 16 bytes per edge, built on
    first use; use_hops=False walks the CSR arrays instead: same bits, more gathers per step)
    and, for biased exact walks, the wedge table (graph.build_wedges(): the shared neighbours of
    every edge by position, 8 bytes per edge + 2 per entry, skipped when it would not fit;
    use_wedges=False walks without it: same bits, searches at the steps that need the pairing).
    With all three tables the walk runs in the kernel where no step needs the wave
    (n2v_walk_wedge.hip); use_wedge_kernel=False keeps the class-count kernel: same bits.  For
    dyadic p, q that walk then runs in passes over a workspace lent to the library (n2v_walk_ws:
    closed forms in the main launches, the ~1 % of steps they decline replayed out of line;
    64 bytes per walker, allocated here); use_workspace=False keeps the one-launch kernel: same
    bits.
    Exact p = q = 1 walks on unit weights can run on the degree-ranked form (graph.build_ranked():
    4-byte entries, the neighbour's RANK by descending degree).  rank_ids=True returns the walks in
    ranks (map with graph.rank_vertex, or compose it into the per-token lookup that follows, as
    fit_streaming does): the form is built on first use and one step is one 4-byte gather.  With
    rank_ids=False the kernel translates every token back (one more gather): use_ranked=True asks
    for that, the default (None) keeps the hop tables for vertex-id output.
    Exact biased walks on a WEIGHTED graph run step-synchronously, one lane per walker, the walkers
    of every step ordered by the degree of the vertex they stand on (n2v_walk_weighted_step; the
    per-edge class counts and wedge lists are built on first use -- they depend on the ids alone):
    the same walks as the wave-per-walker kernel of n2v_walk, which use_weighted_lanes=False keeps.
    Values of p, q that are not dyadic: the row sums of the steps into rows of 1024 entries and more are computed
    once per (p, q) (graph.build_row_sums(): 8 bytes per edge, skipped when they do not fit) and read by the steps
    that need them; use_row_sums=False has those steps add the row up themselves: same bits."""
    L = _lib.load()
    _lib.require_gpu()
    if mode not in MODES:
        raise ValueError(f"unknown walk mode {mode!r}")
    if not graph.rowptr.is_cuda:
        raise RuntimeError("walk: graph is not on the GPU")
    if return_param == 0 or inout_param == 0:
        # generate_edge_alias_tables, randomwalk.py:214-217
        raise ValueError(f"Zero return ({return_param}) or inout ({inout_param}) parameter!")
    biased = not (return_param == 1.0 and inout_param == 1.0)
    if mode == "fast":
        if not graph.unit_weights:
            if graph.slots is None:
                graph.build_alias()  # candidates come from the first-order tables
        else:
            # unit weights: candidates are col[row + int(r1 * n)]; the per-edge class counts take
            # the return edge out of the rejection envelope and skip hopeless membership searches
            if graph.pivots is None:
                graph.build_pivots()
            if use_edge_classes and biased and graph.edge_classes is None:
                graph.build_edge_classes()
            # with the wedge table as well a step is one draw from the layers of its table
            # (n2v_walk_fast.hip, kClassFirst): 1 trial per step instead of ~2, and at q >= 1 with
            # p <= q a single gather
            if (use_edge_classes and use_wedges and biased and graph.wedge_off is None
                    and not graph.wedge_tried):
                graph.wedge_tried = True
                graph.build_wedges()
    if mode == "exact" and graph.unit_weights:
        # unit weights: the per-step table follows from two counts per edge, computed once
        # (n2v_edge_classes_build, 4 bytes per edge); p == q == 1 needs nothing at all
        if use_edge_classes and biased and tables_regime(return_param, inout_param):
            if graph.edge_classes is None:
                graph.build_edge_classes()
            # which slots are which, per edge (wedge table): built once if it fits in half of the
            # free device memory; the steps that run the pairing then search and stream nothing
            if use_wedges and graph.wedge_off is None and not graph.wedge_tried:
                graph.wedge_tried = True
                graph.build_wedges()
            # values that are not dyadic: the reference's rounded row sums of the steps into long rows, once
            # (the slots kernel reads them: a mixed table without folded slots walks through the other kernel)
            if (use_wedges and use_row_sums and graph.wedge_slots is not None
                    and (graph.wedge_mode == 0 or graph.slots_folded)
                    and not (_dyadic(return_param) and _dyadic(inout_param))):
                graph.build_row_sums(return_param, inout_param)
    elif mode == "exact" and biased and use_edge_classes and use_wedges:
        # weighted graph, biased exact walk: the per-edge class counts and wedge lists (they depend on
        # the ids alone) tell both weighted kernels which slots of a step's table are return / shared /
        # other -- no filter over N(s), no membership search, no pass over col
        weighted_lanes_tables(graph)
    elif mode == "exact" and not biased and graph.slots is None:
        # the reference's default p = q = 1: every per-step table is the first-order table of
        # the current vertex, i.e. the K1 slots (bit-identical); build them once (milliseconds)
        try:
            graph.build_alias()
        except ZeroDivisionError:
            pass  # some row sums to 0: keep the per-step path, which raises only if it is visited
    ranked = False
    if graph.unit_weights and not biased and mode == "exact" and (rank_ids or use_ranked):
        if graph.rank_hops is None and not graph.rank_tried:
            graph.build_ranked()
        ranked = graph.rank_hops is not None
    if rank_ids and not ranked:
        raise ValueError("rank_ids: exact p = q = 1 walks on a unit-weight graph that has a degree-ranked "
                         "form (graph.build_ranked()) only")
    uniform8 = False
    if ranked:
        pass
    elif graph.unit_weights and use_hops and use_hops8 and not biased and mode == "exact":
        # p == q == 1: the 8-byte hop table (built once, when the graph's field widths allow it and
        # its high-degree rows are few enough to stay cached) -- one 8-byte gather per step
        if graph.hops8 is None and not graph.hops8_tried:
            graph.build_hops8()
        uniform8 = graph.hops8 is not None
    if graph.unit_weights and use_hops and not uniform8 and not ranked:
        # hop table (16 bytes per edge): one gather per step instead of two or three.  It embeds
        # the class counts, so it is (re)built after them when a biased walk first needs them.
        want_classes = graph.edge_classes is not None
        # the exact slots kernel reads return positions out of the hop table's class words; every
        # other kernel needs the plain form (the table is rebuilt in milliseconds when it changes hands)
        # (fast mode reads either form)
        want_inline = (biased and use_edge_classes and use_wedges and use_wedge_kernel
                       and use_wedge_slots and not use_workspace and graph.wedge_slots is not None
                       and (mode == "fast" or tables_regime(return_param, inout_param))
                       and graph.can_inline_rpos())
        if (graph.hops is None or (want_classes and not graph.hops_have_classes)
                or (biased and mode == "exact" and graph.hops_inline_rpos != want_inline)):
            graph.build_hops(inline_rpos=want_inline)
    start_ids = start_ids.to(device=graph.device, dtype=torch.int32).contiguous()
    n_start = start_ids.numel()
    total = n_start * num_walks
    if (mode == "exact" and biased and not graph.unit_weights and use_weighted_lanes is not False
            and total > 0 and walk_length > 0
            and use_edge_classes and use_wedges  # (the flags that force the table-free wave kernel)
            and (use_weighted_lanes or total >= WEIGHTED_LANES_MIN_WALKERS)
            and weighted_lanes_tables(graph, bool(use_weighted_lanes))):
        return _walk_weighted_lanes(graph, start_ids, num_walks, walk_length, return_param, inout_param,
                                    seed, out, check, stats)
    if out is None:
        walks = torch.empty((total, walk_length + 1), dtype=torch.int32, device=graph.device)
        valid = torch.empty(total, dtype=torch.uint8, device=graph.device)
    else:
        walks, valid = out
    # include/n2v_hip.h: four words (diagnostic builds of the library count in more: N2V_DIAG_STATUS_WORDS)
    status = torch.zeros(max(4, int(os.environ.get("N2V_DIAG_STATUS_WORDS", 4))), dtype=torch.int32,
                         device=graph.device)
    g = graph.c_struct()
    if not use_edge_classes:  # the wave-per-walker kernel (what a C caller without the counts gets)
        g.edge_classes = 0
    if not use_hops or (not use_edge_classes and biased):
        g.hops = 0
    if not uniform8:
        g.hops8 = 0
    if not ranked:
        g.rank_hops = 0
    g.rank_emit = 1 if rank_ids else 0
    if not use_wedges or not use_edge_classes:
        g.wedge_off = 0
        g.wedge_pos = 0
        g.wedge_slots = 0
    if not use_wedge_kernel:  # keep the tables but walk with the lanes kernel (tests: same bits)
        g.reserved = 1
    if not use_wedge_slots:  # the all-tables kernel through wedge_off (tests: same bits)
        g.wedge_slots = 0
    if not use_row_sums:  # every row added up by the lane that needs its sum (tests: same bits)
        g.row_sums = 0
    with torch.cuda.device(graph.device):
        ws_bytes = 0
        if use_workspace and n_start > 0:
            ws_bytes = int(L.n2v_walk_workspace_bytes(g, n_start, num_walks, walk_length,
                                                      float(return_param), float(inout_param),
                                                      MODES[mode]))
        # (torch's caching allocator: the block is reused by the next launch on this stream)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=graph.device) if ws_bytes else None
        rc = L.n2v_walk_ws(g, start_ids.data_ptr(), n_start, num_walks, walk_length,
                           float(return_param), float(inout_param), seed & (2 ** 64 - 1),
                           MODES[mode], walks.data_ptr(), valid.data_ptr(), status.data_ptr(),
                           ws.data_ptr() if ws is not None else None, ws_bytes,
                           _lib.current_stream_ptr())
    _lib.check(rc, "n2v_walk")
    if check:
        _lib.check_status_word(int(status[0].item()), "n2v_walk")
    if stats is not None:  # device tensors; read after synchronising
        stats["trials"] = status[2:4].view(torch.int64)
        stats["status"] = status
    return walks, valid.bool() if out is None else valid


# The step-synchronous form -- per step the sort keys, one sort and five launches: the slot a draw asks for decided
# from sums over the row with margins (WEIGHTED_LANES_MARGINS; csrc/n2v_walk_wlanes.hip) by a wave per walker on
# the long rows and by a lane per walker on the rows below the cut (768 slots for a full batch, down to 48 for a small
# one: the library chooses it from the number of walkers), a second chance on the reference-order row sum and the
# exact wave kernel for what the margins leave undecided -- wins over the one-launch wave-per-walker kernel from a
# thousand walkers on (weighted cfg 2 at (0.5, 2), round 6: 1 000 walkers 9.6 M steps/s against 4.2 M, 10 k: 69.5 M
# against 25.6 M, 47 k: 202 M against 38 M, 471 k: 490 M against 44 M, 4.7 M: 826 M; profiles/r10j_wm_small_batches.log).
# Without the margins the exact lane kernel (the pairing replayed) has the wave on the longest row as the tail of
# every step (78 ms whatever the batch) and only wins from 2 M walkers on.
WEIGHTED_LANES_MARGINS = True
WEIGHTED_LANES_MIN_WALKERS = 1 << 9
# capture steps 1 .. L - 1 of a call into one hipGraph (below).  OFF: measured in round 6 and slower at every batch
# size (profiles/r10g_time_weighted_graph.log: 10 k walkers 37.8 -> 41.2 ms, 471 k 83.0 -> 89.3 ms, same walks) -- the
# ~0.47 ms a step takes whatever the batch is spent ON THE GPU, in a dozen dependent stream operations (the sort's
# passes, three memsets, four persistent launches), not in issuing them; capture + instantiation add 4 - 6 ms.
WEIGHTED_LANES_GRAPH = False
WEIGHTED_LANES_GRAPH_MAX_WALKERS = 1 << 22


def weighted_lanes_tables(graph: DeviceGraph, insist: bool = False) -> bool:
    """the per-edge class counts and wedge lists of a WEIGHTED graph, built on first use (they depend
    on the ids alone): what n2v_walk_weighted_step reads.  False when they do not fit (or were
    declined before)."""
    if graph.edge_classes is None:
        graph.build_edge_classes()
    if graph.wedge_off is None and (insist or not graph.wedge_tried):
        graph.wedge_tried = True
        graph.build_wedges(slots=False)
    return graph.wedge_off is not None


def weighted_row_sums(graph: DeviceGraph) -> Optional[torch.Tensor]:
    """fp64 sum of the stored weights of every row (any order), kept on the graph: what the wave kernel for
    long rows takes the row sum of a step from (n2v_walk_weighted_step).  None when some weight is negative
    or not finite -- the margins of that kernel assume neither -- and the lane kernel then walks alone."""
    got = graph._row_weight_sums
    if got is None:
        w = graph.w
        if not bool((torch.isfinite(w) & (w >= 0)).all()):
            got = False
        else:
            # (per-row sums as differences of one running sum would carry the error of the whole graph;
            # index_add_ serialises its atomics on the hub rows: 1.6 s on cfg 2)
            got = torch.empty(graph.n_vertices + 2, dtype=torch.float64, device=graph.device)
            got[:graph.n_vertices] = torch.segment_reduce(w.double(), "sum", offsets=graph.rowptr)
            # behind the sums: a power of two that divides every stored weight (fp32: the place of the last
            # mantissa bit of the smallest one; fp64 weights: none claimed) and the largest weight -- with
            # them the kernel knows when the reference's own row sum rounds nowhere (include/n2v_hip.h)
            pos = w[w > 0]
            grid = 0.0
            if w.dtype == torch.float32 and pos.numel():
                grid = 2.0 ** (int(torch.frexp(pos)[1].min()) - 24)
            got[graph.n_vertices] = grid
            got[graph.n_vertices + 1] = float(w.max()) if w.numel() else 0.0
        graph._row_weight_sums = got
    return None if got is False else got


WEIGHTED_HUB_SLOTS = 768  # rows of at least this many slots get block summaries (weighted_hub_summaries)
WEIGHTED_LANE_CUT = 0     # > 0: n2v_weighted_hubs.lane_cut (rows from this many slots on: a wave per walker)


def weighted_hub_summaries(graph: DeviceGraph):
    """struct n2v_weighted_hubs for the graph (kept on it), or None: for every row of WEIGHTED_HUB_SLOTS slots or
    more, cut into blocks of 256 slots in row order, the weights of each block sorted ascending (the last block
    padded with +inf) and their fp64 prefix sums -- what lets the wave kernel of n2v_walk_weighted_step take the
    sums of a step over a hub row from one binary search per block instead of a pass over the row (two thirds of
    the slots the steps of a walk on cfg 2 stand on belong to 211 such rows).  None also when the summaries (12
    bytes per hub slot that stay, ~60 of temporaries while they are built) do not fit in half of the free
    memory: the wave kernel then makes its pass over the row -- same bits, slower."""
    got = graph._weighted_hubs
    if got is None:
        got = False
        deg = graph.degrees()
        rows = torch.nonzero(deg >= WEIGHTED_HUB_SLOTS).reshape(-1)
        if rows.numel() and WEIGHTED_HUB_SLOTS > 0:
            nblk = (deg[rows] + 255) // 256
            fits = True
            if graph.device.type == "cuda":
                fits = int(nblk.sum()) * 256 * 72 <= torch.cuda.mem_get_info(graph.device)[0] // 2
            if fits:
                try:
                    got = _hub_summaries(graph, deg, rows, nblk)
                except torch.cuda.OutOfMemoryError:
                    got = False
        graph._weighted_hubs = got
    return None if got is False else got[0]


def _hub_summaries(graph: DeviceGraph, deg, rows, nblk):
    dev, w = graph.device, graph.w
    first = torch.cumsum(nblk, 0) - nblk  # first block of every hub row
    n_blocks = int(nblk.sum())
    block0 = torch.full((graph.n_vertices,), -1, dtype=torch.int32, device=dev)
    block0[rows] = first.to(torch.int32)
    # slot s of block b of hub row r = weight rowptr[r] + 256 (b - first[r]) + s, +inf beyond the row
    blk_row = torch.repeat_interleave(torch.arange(rows.numel(), device=dev), nblk)
    blk_in_row = torch.arange(n_blocks, device=dev) - first[blk_row]
    slot = blk_in_row[:, None] * 256 + torch.arange(256, device=dev)[None, :]
    inside = slot < deg[rows][blk_row][:, None]
    src = (graph.rowptr[rows][blk_row][:, None] + slot).clamp_(max=max(graph.n_edges - 1, 0))
    vals = torch.where(inside, w[src], torch.full((), float("inf"), dtype=w.dtype, device=dev))
    srt = torch.sort(vals, dim=1).values.contiguous()
    prefix = torch.zeros((n_blocks, 257), dtype=torch.float64, device=dev)
    torch.cumsum(torch.where(torch.isfinite(srt), srt, torch.zeros((), dtype=w.dtype, device=dev)).double(), 1,
                 out=prefix[:, 1:])
    st = _lib.WeightedHubs(block0.data_ptr(), srt.data_ptr(), prefix.data_ptr(), WEIGHTED_HUB_SLOTS, 0)
    return (st, block0, srt, prefix)  # (the tensors are kept alive beside the struct that points at them)


def _walk_weighted_lanes(graph: DeviceGraph, start_ids: torch.Tensor, num_walks: int, walk_length: int,
                         p: float, q: float, seed: int, out, check: bool, stats: Optional[dict]):
    """Exact biased walks on a weighted graph, step by step (n2v_walk_weighted_step): per step one
    sort of the walkers by the degree of the vertex they stand on, one launch with a lane per
    walker.  Initialisation = initiate_random_walk (randomwalk.py:279-296), the loop = fugue.py:137-153."""
    L = _lib.load()
    dev = graph.device
    n_start, W, Lw = start_ids.numel(), int(num_walks), int(walk_length)
    total = n_start * W
    if out is None:
        walks = torch.empty((total, Lw + 1), dtype=torch.int32, device=dev)
        valid = torch.empty(total, dtype=torch.uint8, device=dev)
    else:
        walks, valid = out
    status = torch.zeros(4, dtype=torch.int32, device=dev)
    deg = graph.degrees().to(torch.int32)
    # walkers are ordered by the RANK of the vertex they stand on (descending degree, ties by id): rows
    # of one length come together, and so do the walkers on one and the same row -- a walk stands on a
    # vertex in proportion to its degree, so most waves have all 64 lanes on one or two rows and their
    # loads of the row coalesce instead of fetching 64 different lines
    if graph._degree_rank is None:
        order0 = torch.sort(deg, descending=True, stable=True).indices
        rk = torch.empty(graph.n_vertices, dtype=torch.int32, device=dev)
        rk[order0] = torch.arange(graph.n_vertices, dtype=torch.int32, device=dev)
        graph._degree_rank = rk
        del order0
    rank_of = graph._degree_rank
    s64 = start_ids.long()
    in_range = (s64 >= 0) & (s64 < graph.n_vertices)
    alive = in_range & (deg[s64.clamp(0, max(graph.n_vertices - 1, 0))] > 0)  # fugue.py:132
    status[0] = torch.where(in_range.all(), 0, int(_lib.ST_RANGE)).to(torch.int32)
    walks.fill_(-1)
    walks[:, 0] = torch.where(alive, start_ids, torch.full_like(start_ids, -1)).repeat_interleave(W)
    valid.copy_(alive.repeat_interleave(W).to(torch.uint8))
    edge_state = torch.full((total,), -1, dtype=torch.int64, device=dev)
    # the list of the walkers whose step the wave kernel for long rows leaves to the exact kernel
    row_sums = weighted_row_sums(graph) if WEIGHTED_LANES_MARGINS else None
    scratch = torch.empty(2 * (total + 2), dtype=torch.int64, device=dev) if row_sums is not None else None
    hubs = weighted_hub_summaries(graph) if row_sums is not None and WEIGHTED_HUB_SLOTS > 0 else None
    import ctypes as C

    if WEIGHTED_LANE_CUT > 0:  # (tuning / tests: the library chooses the cut from the batch otherwise)
        hubs = (_lib.WeightedHubs(0, 0, 0, 0, int(WEIGHTED_LANE_CUT)) if hubs is None else
                _lib.WeightedHubs(hubs.block0, hubs.sorted, hubs.prefix, hubs.min_slots, int(WEIGHTED_LANE_CUT)))
    hubs_ref = C.byref(hubs) if hubs is not None else None
    undecided = torch.zeros(2, dtype=torch.int64, device=dev)  # second chances; left to the exact kernel
    key = torch.empty(total, dtype=torch.int32, device=dev)
    g = graph.c_struct()
    def steps(first: int, last: int):
        stream = _lib.current_stream_ptr()
        for step in range(first, last):
            # (vanished walkers last; they are skipped)
            _lib.check(L.n2v_walk_weighted_keys(walks.data_ptr(), valid.data_ptr(), rank_of.data_ptr(),
                                                graph.n_vertices, total, step, Lw, key.data_ptr(), stream),
                       "n2v_walk_weighted_keys")
            order = torch.sort(key).indices
            _lib.check(L.n2v_walk_weighted_step(g, start_ids.data_ptr(), W, order.data_ptr(), total, step, Lw,
                                                float(p), float(q), seed & (2 ** 64 - 1),
                                                edge_state.data_ptr(), walks.data_ptr(), valid.data_ptr(),
                                                status.data_ptr(), 0 if scratch is None else scratch.data_ptr(),
                                                0 if row_sums is None else row_sums.data_ptr(), hubs_ref, stream),
                       "n2v_walk_weighted_step")
            if scratch is not None and stats is not None:
                undecided.add_(scratch[0::total + 2])

    with torch.cuda.device(dev):
        # A step is ~12 stream operations (keys, the sort, three memsets, four kernels, the counters); nothing in it
        # reads anything on the host, so steps 1 .. L - 1 CAN be captured into one hipGraph and replayed
        # (WEIGHTED_LANES_GRAPH; same walks, tested) -- it does not pay: see the flag.
        graphed = False
        if WEIGHTED_LANES_GRAPH and Lw > 2 and total <= WEIGHTED_LANES_GRAPH_MAX_WALKERS:
            steps(0, 1)  # eager: also the warm-up of everything a capture may not do for the first time
            try:
                cg = torch.cuda.CUDAGraph()
                with torch.cuda.graph(cg):
                    steps(1, Lw)
                cg.replay()
                graphed = True
            except RuntimeError:  # (a capture that fails has executed nothing: the steps run eagerly below)
                graphed = False
            if not graphed:
                steps(1, Lw)
            else:
                del cg
        else:
            steps(0, Lw)
        if stats is not None:
            stats["graph"] = graphed
    if check:
        _lib.check_status_word(int(status[0].item()), "n2v_walk")
    if stats is not None:
        stats["trials"] = status[2:4].view(torch.int64)
        stats["status"] = status
        # walker-steps the margins did not decide (stepped by the exact kernel); those that needed the
        # reference-order row sum before they were decided
        stats["undecided"] = undecided[1]
        stats["second_chance"] = undecided[0]
    return walks, valid.bool() if out is None else valid


def audition_buffers(graph: DeviceGraph, start_ids: torch.Tensor, num_walks: int, walk_length: int,
                     return_param: float, inout_param: float, seed: int, mode: str = "exact",
                     candidates: int = 4, probe_vertices: int = 1 << 17, report: Optional[dict] = None,
                     **walk_kw) -> Tuple[torch.Tensor, torch.Tensor]:
    """Output buffers (walks int32 [n_start * num_walks, walk_length + 1], valid uint8) for REPEATED
    launches of `walk(..., out=...)` over batches of start_ids.numel() start vertices, chosen by
    audition: the same walk kernel on the same tables runs up to 7 % faster or slower depending on
    WHICH allocation it writes to (DESIGN.md 5 "Placement": the reads alone and the stores alone do not
    care where the buffers are, their mix does, for every pair of table and output differently; no
    counter up to the L2 shows why), and a launch over 2^17 start vertices ranks the candidates the way
    the full launches do (profiles/r7d_placement_parts.log).  So `candidates` buffers are allocated,
    each takes a short launch (untimed warm-up + one timed by HIP events on the current stream), the
    fastest is kept and the others are freed.  Costs candidates x the buffer transiently and
    ~5 ms per candidate; callers that launch once (random_walk()) do not bother."""
    n_start = int(start_ids.numel())
    total = n_start * int(num_walks)
    dev = graph.device

    def alloc():
        return (torch.empty((total, walk_length + 1), dtype=torch.int32, device=dev),
                torch.empty(total, dtype=torch.uint8, device=dev))

    free = torch.cuda.mem_get_info(dev)[0]
    need = total * (walk_length + 2) * 4
    k = int(max(1, min(candidates, (free // 2) // max(need, 1))))
    probe = start_ids[: min(n_start, int(probe_vertices))]
    if k <= 1 or probe.numel() == 0 or walk_length == 0:
        return alloc()
    bufs = [alloc() for _ in range(k)]
    rows = probe.numel() * int(num_walks)
    times = []
    for w, v in bufs:
        for rep in range(2):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            walk(graph, probe, num_walks, walk_length, return_param, inout_param, seed, mode,
                 out=(w[:rows], v[:rows]), check=False, **walk_kw)
            b.record()
        torch.cuda.synchronize(dev)
        times.append(a.elapsed_time(b))
    best = min(range(k), key=times.__getitem__)
    if report is not None:
        report.update(candidates=k, probe_ms=[round(t, 3) for t in times], chosen=best,
                      probe_start_vertices=int(probe.numel()))
    return bufs[best]


# the reference's row-level surface of this module (Neighbors, AliasProb, RandomPath,
# generate_alias_tables, generate_edge_alias_tables, trim_hotspot_vertices, get_vertex_neighbors,
# initiate_random_walk, next_step_random_walk, to_path -- randomwalk.py:17-349) lives in
# node2vec_amd/transformers.py and is importable from here under the reference's names
from node2vec_amd.transformers import (AliasProb, Neighbors, RandomPath,  # noqa: E402,F401
                                       generate_alias_tables, generate_edge_alias_tables,
                                       get_vertex_neighbors, initiate_random_walk,
                                       next_step_random_walk, to_path, trim_hotspot_vertices)
