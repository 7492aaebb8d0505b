"""Graph-partitioned walking: walks over a graph that does not fit one GPU.

The reference never holds the graph in one place: every step of `fugue.random_walk`
(fugue.py:144-149) joins the walker rows with the adjacency rows of their previous and their
current vertex -- two shuffles per step, the neighbour list of the previous vertex travelling
with the walker -- and calls `next_step_random_walk` on the joined rows.  This module is that
design point on a GPU node:

* the CSR is partitioned by contiguous VERTEX RANGE; rank r stores the rows of its range only
  (`partition_graph`), so per-GPU memory is E / N instead of E;
* a walker lives on the rank that stores the row of its CURRENT vertex.  One step there is the
  reference's transformer on a batch -- bias N(v) against what travelled with the walker, build
  the table, draw with the walker's own uniforms -- as ONE launch (n2v_partition_step, the table
  never materialised; `tables_step` keeps the launch-per-stage form on materialised tables as a
  cross-check), bit-identical to n2v_walk's exact mode, whose RNG is keyed by (seed, start
  vertex, ordinal, step) and not by where the walker happens to be; the routing that follows is
  one more launch (n2v_partition_route), a sort by destination and a gather;
* then the walker MIGRATES to the owner of the vertex it drew, carrying a 40-byte header
  (output row, RNG key, previous/current vertex, step, classes) and -- only when q != 1, the one
  case in which N(s) decides anything (randomwalk.py:226-229) -- either the row it just left (the
  `src_neighbors` of its next step) or, when the parts hold the per-edge tables of a unit-weight
  graph, just the WEDGE LIST of the edge it leaves along (the positions, in the row it is going
  to, of the neighbours that row shares with this one: usually a handful of words instead of a
  row, and the receiving rank then needs neither a search over N(s) nor a pass over N(v)): one
  all-to-all per step (`torch.distributed.all_to_all_single`, RCCL over xGMI on the GPUs, gloo in
  the CPU tests) -- of variable size, read on the host, for the first three steps; from then on of
  mailboxes with a fixed capacity per (source, destination) set from the flows seen so far, sent
  whole, empty slots included, so that a step reads nothing on the host (`Outboxes`);
* every appended vertex is logged as (row, position, vertex) and sent once, at the end, to the
  rank that emits the walk (the owner of its start vertex); walkers that reach a vertex without
  out-edges vanish (inner join, fugue.py:147) and their row is marked invalid -- the step finds
  them itself (nothing to draw from), the arrivals are not inspected.

`walk_partitioned` runs one rank of it; `walk_partitioned_local` runs all ranks of a
partition in one process (the exchange is a list transpose) and is what the single-GPU tests
use to prove that partitioned == unpartitioned, bit for bit.  The step function is injectable
(`step_fn`); the default is the HIP one and raises without a GPU.
"""
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import torch

from node2vec_amd.graph import DeviceGraph
from node2vec_amd.shard import shard_range

WEDGE_OFF_MASK = (1 << 40) - 1  # include/n2v_hip.h: N2V_WEDGE_OFF_MASK
HEAD_COLS = 5  # (output row, RNG key, s << 32 | v, step, classes of the step's table)


@dataclass
class GraphPart:
    """rows [lo, hi) of the CSR: rowptr rebased to 0, the rows' neighbour ids (global) and
    weights (None = unit); `bounds` = the lo of every part, ascending (owner lookup)"""
    rank: int
    lo: int
    hi: int
    rowptr: torch.Tensor
    col: torch.Tensor
    w: Optional[torch.Tensor]
    bounds: torch.Tensor
    # the part's slice of the per-edge tables of a unit-weight graph (n2v_edge_classes_build,
    # n2v_wedge_build; list offsets rebased to the slice) or None: with them a walker that
    # leaves along edge e carries the wedge list of e instead of the whole row of its vertex
    edge_classes: Optional[torch.Tensor] = None
    wedge_off: Optional[torch.Tensor] = None
    wedge_pos: Optional[torch.Tensor] = None

    @property
    def device(self):
        return self.rowptr.device

    def owner(self, v: torch.Tensor) -> torch.Tensor:
        return torch.searchsorted(self.bounds, v.to(torch.int64), right=True) - 1

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in (self.rowptr, self.col, self.w, self.edge_classes,
                                                         self.wedge_off, self.wedge_pos) if t is not None)


def partition_graph(g: DeviceGraph, n_parts: int, balance: str = "edges",
                    wedges: bool = True) -> List[GraphPart]:
    """Contiguous vertex ranges with about equal numbers of EDGES (default) or of vertices.
    A rank of a real run would load only its own part; here they are cut from a whole graph.
    `wedges`: on a unit-weight graph on the GPU the per-edge tables (class counts and wedge
    lists) are built once on the whole graph and every part keeps the slice of ITS edges -- part
    of the partitioned graph's stored form, like the rows themselves."""
    tables = None
    if wedges and g.unit_weights and g.rowptr.is_cuda and g.n_edges > 0:
        if g.wedge_off is None and not g.wedge_tried:
            g.wedge_tried = True
            g.build_wedges()
        if g.wedge_off is not None and g.wedge_mode >= 2:
            # a mixed table (16-bit lists, then the 32-bit lists of the edges into wide rows) cannot be
            # cut by edge ranges: the parts of a graph with such rows keep 32-bit lists throughout
            gw = DeviceGraph(g.rowptr, g.col, None)
            gw.edge_classes = g.edge_classes
            gw.build_wedges(wide=True, slots=False)
            if gw.wedge_off is not None:
                tables = (gw.edge_classes, gw.wedge_off, gw.wedge_pos)
        elif g.wedge_off is not None:
            tables = (g.edge_classes, g.wedge_off, g.wedge_pos)
    V = g.n_vertices
    if balance == "edges" and g.n_edges > 0:
        targets = torch.arange(1, n_parts, device=g.device, dtype=torch.int64) * g.n_edges // n_parts
        cuts = torch.searchsorted(g.rowptr, targets, right=False).clamp(0, V).tolist()
    else:
        cuts = [shard_range(V, r, n_parts)[0] for r in range(1, n_parts)]
    los = [0] + [int(c) for c in cuts]
    his = los[1:] + [V]
    bounds = torch.tensor(los + [V], dtype=torch.int64, device=g.device)
    # owner lookup uses searchsorted(right=True) - 1 over the lows; empty ranges own nothing
    parts = []
    for r, (lo, hi) in enumerate(zip(los, his)):
        hi = max(hi, lo)
        e0, e1 = int(g.rowptr[lo]), int(g.rowptr[hi])
        part = GraphPart(r, lo, hi, (g.rowptr[lo:hi + 1] - e0).contiguous(),
                         g.col[e0:e1].contiguous(),
                         None if g.unit_weights else g.w[e0:e1].contiguous(), bounds[:-1].contiguous())
        if tables is not None and e1 > e0:
            ec, off, pos = tables
            cnt = (ec[e0:e1] & 0xffffff).to(torch.int64)
            o0 = int(off[e0] & WEDGE_OFF_MASK)
            o1 = int(off[e1 - 1] & WEDGE_OFF_MASK) + int(cnt[-1])
            part.edge_classes = ec[e0:e1].contiguous()
            part.wedge_off = (off[e0:e1] - o0).contiguous()  # the return position (bits 40+) stays
            part.wedge_pos = pos[o0:o1].contiguous() if o1 > o0 else pos.new_zeros(1)
        elif tables is not None:  # a part without edges keeps (empty) tables: same mode on every rank
            part.edge_classes = tables[0][:0].contiguous()
            part.wedge_off = tables[1][:0].contiguous()
            part.wedge_pos = tables[2].new_zeros(1)
        parts.append(part)
    return parts


# ---- messages ---------------------------------------------------------------------------------
@dataclass
class Walkers:
    """a batch of walker states: header int64 [k, 5] = (output row, RNG key, s << 32 | v, step,
    classes) and what travels with each walker, packed (lens int64 [k], ids int32 [nnz]): the
    row N(s), or -- when the parts carry wedge tables -- the wedge list of the edge
    (s -> v), with classes = edge_classes | return position << 32 (else 0)"""
    head: torch.Tensor
    lens: torch.Tensor
    ids: torch.Tensor
    _ptr: Optional[torch.Tensor] = None

    @property
    def ptr(self) -> torch.Tensor:
        """offsets of every walker's words in `ids` (int64 [k + 1]): built once, on the rank
        that steps the batch -- on the wire the lengths travel"""
        if self._ptr is None:
            self._ptr = torch.zeros(self.lens.numel() + 1, dtype=torch.int64, device=self.lens.device)
            torch.cumsum(self.lens, 0, out=self._ptr[1:])
        return self._ptr

    @classmethod
    def empty(cls, device):
        return cls(torch.zeros((0, HEAD_COLS), dtype=torch.int64, device=device),
                   torch.zeros(0, dtype=torch.int64, device=device),
                   torch.zeros(0, dtype=torch.int32, device=device))

    def __len__(self):
        return self.head.shape[0]

    def select(self, idx: torch.Tensor) -> "Walkers":
        lens = self.lens[idx]
        ptr = torch.zeros(idx.numel() + 1, dtype=torch.int64, device=self.head.device)
        torch.cumsum(lens, 0, out=ptr[1:])
        return Walkers(self.head[idx], lens, _gather_rows(self.ptr, self.ids, idx, ptr), ptr)

    @classmethod
    def cat(cls, parts: Sequence["Walkers"], device) -> "Walkers":
        parts = [p for p in parts if len(p)]
        if not parts:
            return cls.empty(device)
        if len(parts) == 1:
            return parts[0]
        return cls(torch.cat([p.head for p in parts]), torch.cat([p.lens for p in parts]),
                   torch.cat([p.ids for p in parts]))


@dataclass
class Mail:
    """walkers as n2v_partition_forward leaves them in a mailbox: headers int64 [k, 5], where every
    walker's wedge list starts in `words` (int64 [k]: no prefix array -- the order in a mailbox is
    whatever the atomics gave), the lists int32 [w]"""
    head: torch.Tensor
    off: torch.Tensor
    words: torch.Tensor

    def __len__(self):
        return self.head.shape[0]

    @classmethod
    def empty(cls, device):
        return cls(torch.zeros((0, HEAD_COLS), dtype=torch.int64, device=device),
                   torch.zeros(0, dtype=torch.int64, device=device),
                   torch.zeros(0, dtype=torch.int32, device=device))

    @classmethod
    def cat(cls, parts: Sequence["Mail"], device) -> "Mail":
        """one batch out of the mail of several sources: list starts rebased to the joined pool"""
        parts = [p for p in parts if len(p)]
        if not parts:
            return cls.empty(device)
        if len(parts) == 1:
            return parts[0]
        base, offs = 0, []
        for p in parts:
            offs.append(p.off + base)
            base += int(p.words.numel())
        return cls(torch.cat([p.head for p in parts]), torch.cat(offs), torch.cat([p.words for p in parts]))


def _gather_rows(ptr: torch.Tensor, ids: torch.Tensor, rows: torch.Tensor, out_ptr: torch.Tensor):
    """concatenation of ids[ptr[r] : ptr[r + 1]] for r in rows (out_ptr = the new offsets)"""
    total = int(out_ptr[-1])
    if total == 0:
        return ids[:0]
    lens = out_ptr[1:] - out_ptr[:-1]
    which = torch.repeat_interleave(torch.arange(rows.numel(), device=ids.device), lens)
    within = torch.arange(total, device=ids.device) - out_ptr[:-1][which]
    return ids[ptr[rows][which] + within]


# ---- the step on one rank ---------------------------------------------------------------------
def tables_step(dst_ptr, dst_ids, dst_w, src_id, src_ptr, src_ids, keys, steps, p, q, seed):
    """One step of a batch of walkers on the GPU that stores their current rows, the way the
    reference does it: next_step_random_walk (randomwalk.py:300-339) on MATERIALISED tables
    (n2v_edge_bias, n2v_alias_build, n2v_alias_draw) with n2v_walk's uniforms.  ~10 launches and
    24 bytes of table per neighbour and step; kept as the cross-check of the fused step."""
    from node2vec_amd import _lib
    from node2vec_amd import transformers as T

    L = _lib.load()
    _lib.require_gpu()
    dev = dst_ptr.device
    n = keys.numel()
    r1 = torch.empty(n, dtype=torch.float64, device=dev)
    r2 = torch.empty(n, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.n2v_walk_uniforms(int(seed) & (2 ** 64 - 1), keys.data_ptr(), steps.data_ptr(), n,
                                       r1.data_ptr(), r2.data_ptr(), _lib.current_stream_ptr()),
                   "n2v_walk_uniforms")
    nbs = src_ids if src_ids.numel() else torch.zeros(1, dtype=torch.int32, device=dev)
    biased = T._bias_rows(dst_ptr, dst_ids, dst_w, src_id, src_ptr, nbs, p, q)
    slots = T._build_tables(dst_ptr, dst_ids, biased)
    return T._draw_device(dst_ptr, slots, r1, r2)


def hip_step(dst_ptr, dst_ids, dst_w, src_id, src_ptr, src_ids, keys, steps, p, q, seed):
    """The default step.  RankState recognises it and runs the whole step of a rank as ONE launch
    of n2v_partition_step straight from the part's CSR (`_advance_fused`); called directly, with
    the joined rows already packed, it is `tables_step`."""
    return tables_step(dst_ptr, dst_ids, dst_w, src_id, src_ptr, src_ids, keys, steps, p, q, seed)


class Outboxes:
    """The capacity-bounded mailboxes of one rank.  caps_h[s][d] / caps_w[s][d] = how many walkers /
    list words rank s can send to rank d in one step (the same matrices on every rank).  To SEND:
    the boxes of this rank's row back to back (send_head / send_off / send_words; `box_starts` is
    their layout as n2v_partition_forward_boxes takes it); to RECEIVE: the boxes of its column back
    to back, which is exactly what an all-to-all with those fixed split sizes delivers.  Slots
    nobody wrote are EMPTY (output row -1: the kernels skip them).  Allocated once per walk -- with
    the path records of all its steps and the step's scratch -- and reused by every step; nothing
    about the exchange is read on the host.  `count` = walkers, then words, per destination of the
    last step; `need` = their maxima over the steps so far (what a second attempt is sized by when
    a box overflowed)."""

    def __init__(self, world: int, rank: int, caps_h, caps_w, dev, send=None, recv=None):
        self.world, self.rank = world, rank
        self.send_h, self.send_w = [int(c) for c in caps_h[rank]], [int(c) for c in caps_w[rank]]
        self.recv_h = [int(caps_h[s][rank]) for s in range(world)]
        self.recv_w = [int(caps_w[s][rank]) for s in range(world)]

        def make(nh, nw):
            return (torch.empty((nh, HEAD_COLS), dtype=torch.int64, device=dev),
                    torch.empty(nh, dtype=torch.int64, device=dev),
                    torch.empty(max(nw, 1), dtype=torch.int32, device=dev))

        self.send_head, self.send_off, self.send_words = \
            send if send is not None else make(sum(self.send_h), sum(self.send_w))
        self.recv_head, self.recv_off, self.recv_words = \
            recv if recv is not None else make(sum(self.recv_h), sum(self.recv_w))
        if self.send_words.numel() == 0:  # (no lists travel: the kernels still want an address)
            self.send_words = torch.zeros(1, dtype=torch.int32, device=dev)
        if self.recv_words.numel() == 0:
            self.recv_words = torch.zeros(1, dtype=torch.int32, device=dev)
        starts, run = [], 0
        for caps in (self.send_h, self.send_w):
            run = 0
            for c in caps:
                starts.append(run)
                run += c
            starts.append(run)
        self.box_starts = torch.tensor(starts, dtype=torch.int64, device=dev)
        # a list start is written relative to the pool of its destination d; there that pool lies behind
        # the pools of the ranks before this one
        base_there = [sum(int(caps_w[s][d]) for s in range(rank)) for d in range(world)]
        self.off_add = torch.repeat_interleave(torch.tensor(base_there, dtype=torch.int64, device=dev),
                                               torch.tensor(self.send_h, dtype=torch.int64, device=dev))
        self.count = torch.zeros(2 * world, dtype=torch.int64, device=dev)
        self.need = torch.zeros(2 * world, dtype=torch.int64, device=dev)
        self.inbox_live = False  # False: the next step still reads the exact mail of the calibration
        self.logs: List[torch.Tensor] = []
        self.nxt = self.edge = None

    def begin(self, first_k: int, steps: int, wedge: bool):
        """the path records of the `steps` bounded steps (rows -1 until written: empty slots leave none) and
        the step's scratch, allocated before the first of them"""
        dev, k_in = self.count.device, int(self.recv_head.shape[0])
        self.logs = [torch.full((first_k, 3), -1, dtype=torch.int64, device=dev)]
        if steps > 1:
            self.logs += list(torch.full((steps - 1, k_in, 3), -1, dtype=torch.int64, device=dev).unbind(0))
        k = max(first_k, k_in, 1)
        self.nxt = torch.empty(k, dtype=torch.int32, device=dev)
        self.edge = torch.empty(k, dtype=torch.int64, device=dev) if wedge else None
        self.step_no = 0

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in (self.send_head, self.send_off, self.send_words,
                                                         self.recv_head, self.recv_off, self.recv_words))


class RankState:
    """what one rank holds between steps"""

    def __init__(self, part: GraphPart, num_walks: int, walk_length: int, p: float, q: float, seed: int,
                 step_fn: Callable = hip_step):
        if p == 0 or q == 0:
            raise ValueError(f"Zero return ({p}) or inout ({q}) parameter!")
        self.part, self.W, self.L, self.p, self.q, self.seed = part, num_walks, walk_length, p, q, seed
        self.step_fn = step_fn
        self.walkers = Walkers.empty(part.device)
        self.log: List[torch.Tensor] = []  # int64 [k, 3]: (output row, position, vertex or -1 = dropped)
        self.n_rows = 0
        self.row_base = 0
        self.status = None  # uint32 [4] of n2v_partition_step, on the part's device
        self._arange = None
        self.use_tables = True  # walk_partitioned clears it unless every rank holds the tables
        self.last_status = 0       # status bits of n2v_partition_step seen so far
        self.defer_status = False  # True: collected in last_status instead of raised at once
        # route with n2v_partition_forward (one launch; walkers travel as Mail) where the step is
        # per-lane work, instead of route + group + prefix sum + gather (walkers travel as Walkers)
        self.forward = False
        self.mail = Mail.empty(part.device)
        self.last_counts: Optional[List[int]] = None  # what _advance_forward sent: walkers, then words, per rank
        self.flow_counts: Optional[List[int]] = None
        self.boxes: Optional[Outboxes] = None          # capacity-bounded mailboxes (advance_bounded)

    # -- initiate_random_walk (randomwalk.py:279-296) for the start vertices this rank owns ----
    def initiate(self, start_ids_global: torch.Tensor):
        """start_ids_global: the WHOLE sorted start list (every rank passes the same); this rank
        creates the walkers of the start vertices in its range.  Output row of (start index i,
        ordinal o) is i * W + o - 1, as in n2v_walk."""
        part, dev = self.part, self.part.device
        s = start_ids_global.to(device=dev, dtype=torch.int64)
        if s.numel() > 1 and not bool((s[1:] >= s[:-1]).all()):
            raise ValueError("walk_partitioned: the start list must be sorted (the rows of a rank are a range of it)")
        idx = torch.nonzero((s >= part.lo) & (s < part.hi)).reshape(-1)
        mine = s[idx]
        deg = (part.rowptr[1:] - part.rowptr[:-1])[mine - part.lo]
        ok = deg > 0  # fugue.py:132: vertices without out-edges start no walk
        self.n_rows = int(s.numel()) * self.W
        ords = torch.arange(self.W, device=dev, dtype=torch.int64)
        rows = (idx[:, None] * self.W + ords[None, :]).reshape(-1)
        keys = (mine[:, None] * self.W + ords[None, :]).reshape(-1)
        v = mine.repeat_interleave(self.W)
        live = ok.repeat_interleave(self.W)
        # position 0 of every row this rank emits (invalid rows are logged as dropped)
        self.log.append(torch.stack([rows, torch.zeros_like(rows), torch.where(live, v, torch.full_like(v, -1))], 1))
        rows, keys, v = rows[live], keys[live], v[live]
        if self.L == 0:
            return
        head = torch.stack([rows, keys, (torch.full_like(v, -1) << 32) | (v & 0xffffffff),
                            torch.zeros_like(rows), torch.zeros_like(rows)], 1)
        self.walkers = Walkers(head, torch.zeros(rows.numel(), dtype=torch.int64, device=dev),
                               torch.zeros(0, dtype=torch.int32, device=dev))
        self.mail = Mail(head, torch.zeros(rows.numel(), dtype=torch.int64, device=dev),
                         torch.zeros(0, dtype=torch.int32, device=dev))

    # -- the step as ONE launch: n2v_partition_step reads N(v) from the part's CSR ---------------
    def _lane_mode(self) -> int:
        """0: one wave per walker (rows travel when q != 1); else the step is per-lane work
        (partition_step_wedge_kernel) and the value says what travels: 1 nothing (p == q == 1 on
        unit weights: the draw is `pick`), 2 the wedge list of the edge, 3 only its counts and
        return position (q == 1).  2 and 3 need the per-edge tables on the parts and (p, q) in
        the range of the unit-weight kernels."""
        part = self.part
        if part.w is not None:
            return 0
        if self.p == 1.0 and self.q == 1.0:
            return 1
        ordinary = all(2.0 ** -20 <= 1.0 / x <= 2.0 ** 20 for x in (self.p, self.q))
        if part.wedge_off is None or not self.use_tables or not ordinary:
            return 0
        return 2 if self.q != 1.0 else 3

    def _advance_fused(self, n_parts: int) -> List[Walkers]:
        """next_step_random_walk (randomwalk.py:300-339) for every resident walker in one launch
        (the table never materialised), then the routing: one elementwise launch
        (n2v_partition_route: path record, next header, destination, words to carry), a stable
        counting sort by destination (n2v_partition_group), one gather of what travels.  N(s) only decides shared / other
        (randomwalk.py:226-229): with q == 1 -- the reference's defaults p == q == 1 included,
        where every table is probs == [1.0] * n and the draw is pick = int(r1 * n) -- the walker
        travels as its 40-byte header alone.  Otherwise it takes the WEDGE LIST of the edge it
        leaves along (n2v_gather_wedges: the positions in the next row of the neighbours shared
        with this one, from this part's slice of the wedge table) when the parts carry the
        per-edge tables, else the whole row it leaves (n2v_gather_rows).  Walkers that stand on
        a vertex without out-edges draw nothing and are logged as vanished (fugue.py:147)."""
        from node2vec_amd import _lib

        L = _lib.load()
        part, dev, wk = self.part, self.part.device, self.walkers
        k = len(wk)
        head_in = wk.head.contiguous()
        lanes = self._lane_mode()
        wedge = lanes >= 2  # the edge drawn is needed: something of it travels
        carry = self.q != 1.0
        nxt32 = torch.empty(k, dtype=torch.int32, device=dev)
        edge = torch.empty(k, dtype=torch.int64, device=dev) if wedge else None
        if self.status is None:
            self.status = torch.zeros(4, dtype=torch.int32, device=dev)
        w = part.w
        w32 = w.data_ptr() if (w is not None and w.dtype == torch.float32) else 0
        w64 = w.data_ptr() if (w is not None and w.dtype == torch.float64) else 0
        src_ids = wk.ids if wk.ids.numel() else torch.zeros(1, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.n2v_partition_step(part.rowptr.data_ptr(), part.col.data_ptr(), w32, w64,
                                            part.lo, part.hi - part.lo, head_in.data_ptr(), HEAD_COLS,
                                            wk.ptr.data_ptr() if (carry or wedge) else 0,
                                            src_ids.data_ptr(), int(lanes > 0), k,
                                            float(self.p), float(self.q),
                                            int(self.seed) & (2 ** 64 - 1), nxt32.data_ptr(),
                                            edge.data_ptr() if wedge else 0, self.status.data_ptr(),
                                            _lib.current_stream_ptr()), "n2v_partition_step")
        # routing: one elementwise kernel (path record, next header, destination, words to carry),
        # a stable sort by destination, one gather of what travels
        carry_kind = lanes if wedge else (1 if (carry and not lanes) else 0)
        log = torch.empty((k, 3), dtype=torch.int64, device=dev)
        head = torch.empty((k, HEAD_COLS), dtype=torch.int64, device=dev)
        dest = torch.empty(k, dtype=torch.int32, device=dev)
        lens = torch.empty(k, dtype=torch.int64, device=dev)
        src = torch.empty(k, dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.n2v_partition_route(head_in.data_ptr(), HEAD_COLS, nxt32.data_ptr(),
                                             edge.data_ptr() if wedge else 0, k, self.L,
                                             part.bounds.data_ptr(), n_parts, carry_kind,
                                             part.rowptr.data_ptr(), part.lo,
                                             part.edge_classes.data_ptr() if wedge else 0,
                                             log.data_ptr(), head.data_ptr(), dest.data_ptr(),
                                             lens.data_ptr(), src.data_ptr(),
                                             _lib.current_stream_ptr()), "n2v_partition_route")
        self.log.append(log)
        if n_parts <= 64:  # stable counting sort by destination (three small launches)
            work = torch.empty(((k + 255) // 256 + 1) * (n_parts + 1), dtype=torch.int64, device=dev)
            head_s, lens_s, src_s = torch.empty_like(head), torch.empty_like(lens), torch.empty_like(src)
            cuts = torch.empty(n_parts + 1, dtype=torch.int64, device=dev)
            with torch.cuda.device(dev):
                _lib.check(L.n2v_partition_group(dest.data_ptr(), head.data_ptr(), HEAD_COLS, lens.data_ptr(),
                                                 src.data_ptr(), k, n_parts, work.data_ptr(),
                                                 head_s.data_ptr(), lens_s.data_ptr(), src_s.data_ptr(),
                                                 cuts.data_ptr(), _lib.current_stream_ptr()),
                           "n2v_partition_group")
            head, lens, src = head_s, lens_s, src_s
        else:
            dest, order = torch.sort(dest, stable=True)
            cuts = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev),
                              torch.searchsorted(dest, self._parts_arange(n_parts))])
            head, lens, src = head[order], lens[order], src[order]
        ptr = torch.zeros(k + 1, dtype=torch.int64, device=dev)
        torch.cumsum(lens, 0, out=ptr[1:])
        # one transfer: the status word, the walkers per destination, the words per destination
        host = torch.cat([self.status[:1].to(torch.int64), cuts, ptr[cuts]]).tolist()
        # one rank of several must not raise alone (the others would wait for it in the exchange):
        # walk_partitioned sets defer_status and raises on every rank after the size exchange
        self.last_status |= int(host[0])
        if not self.defer_status:
            _lib.check_status_word(host[0], "n2v_partition_step")
        cuts_h, at = host[1:n_parts + 2], host[n_parts + 2:]
        k2 = cuts_h[-1]  # forwarded walkers: the first k2 of the sorted batch
        ids = torch.zeros(0, dtype=torch.int32, device=dev)
        if carry_kind and k2 > 0:
            ids = torch.empty(max(at[-1], 1), dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                if wedge:
                    _lib.check(L.n2v_gather_wedges(part.edge_classes.data_ptr(), part.wedge_off.data_ptr(),
                                                   part.wedge_pos.data_ptr(),
                                                   int(part.wedge_pos.dtype == torch.int32), src.data_ptr(),
                                                   ptr.data_ptr(), k2, ids.data_ptr(), head.data_ptr(),
                                                   HEAD_COLS, _lib.current_stream_ptr()), "n2v_gather_wedges")
                else:
                    _lib.check(L.n2v_gather_rows(part.rowptr.data_ptr(), part.col.data_ptr(),
                                                 src.data_ptr(), ptr.data_ptr(), k2, ids.data_ptr(),
                                                 _lib.current_stream_ptr()), "n2v_gather_rows")
        return [Walkers(head[cuts_h[r]:cuts_h[r + 1]], lens[cuts_h[r]:cuts_h[r + 1]], ids[at[r]:at[r + 1]])
                for r in range(n_parts)]

    def _note_counts(self, counts: List[int]):
        """what this step sent (walkers, then words, per rank); `flow_counts` = the larger of this and the
        previous step's, the figure the capacity-bounded mailboxes are sized by"""
        prev = self.last_counts
        self.last_counts = counts
        self.flow_counts = counts if prev is None else [max(a, b) for a, b in zip(prev, counts)]

    def _parts_arange(self, n_parts: int) -> torch.Tensor:
        if self._arange is None or self._arange.numel() != n_parts:
            self._arange = torch.arange(1, n_parts + 1, dtype=torch.int32, device=self.part.device)
        return self._arange

    # -- one step of every resident walker; returns the migrating walkers per destination -----
    def forwarding(self) -> bool:
        """this rank's walkers travel as Mail (n2v_partition_forward: at most FORWARD_MAX_PARTS destinations --
        its per-block counters live in LDS; a larger world routes launch by stage)"""
        return (self.forward and self.step_fn is hip_step and self.part.rowptr.is_cuda
                and self._lane_mode() >= 1 and int(self.part.bounds.numel()) <= FORWARD_MAX_PARTS)

    def _advance_forward(self, n_parts: int) -> List[Mail]:
        """_advance_fused with the routing in ONE launch: n2v_partition_step on the resident mail,
        then n2v_partition_forward into one outbox per destination (headers, list starts, a word
        pool per destination: what goes to a rank is contiguous).  One host read: the counts (the
        sizes of the exchange that follows) and the status word; a pool that was too small is
        enlarged to what the launch reported and the launch repeated."""
        from node2vec_amd import _lib

        L = _lib.load()
        part, dev, ml = self.part, self.part.device, self.mail
        k = len(ml)
        lanes = self._lane_mode()
        wedge = lanes >= 2
        carry = lanes if wedge else 0
        head_in = ml.head.contiguous()
        off_in = ml.off.contiguous()
        words_in = ml.words if ml.words.numel() else torch.zeros(1, dtype=torch.int32, device=dev)
        nxt32 = torch.empty(k, dtype=torch.int32, device=dev)
        edge = torch.empty(k, dtype=torch.int64, device=dev) if wedge else None
        if self.status is None:
            self.status = torch.zeros(4, dtype=torch.int32, device=dev)
        stream = _lib.current_stream_ptr()
        with torch.cuda.device(dev):
            _lib.check(L.n2v_partition_step(part.rowptr.data_ptr(), part.col.data_ptr(), 0, 0, part.lo,
                                            part.hi - part.lo, head_in.data_ptr(), HEAD_COLS, off_in.data_ptr(),
                                            words_in.data_ptr(), 2,  # N2V_SRC_WEDGES_AT
                                            k, float(self.p), float(self.q), int(self.seed) & (2 ** 64 - 1),
                                            nxt32.data_ptr(), edge.data_ptr() if wedge else 0,
                                            self.status.data_ptr(), stream), "n2v_partition_step")
        log = torch.empty((k, 3), dtype=torch.int64, device=dev)
        box_head = torch.empty((n_parts, k, HEAD_COLS), dtype=torch.int64, device=dev)
        box_off = torch.zeros((n_parts, k), dtype=torch.int64, device=dev)
        count = torch.zeros(2 * n_parts, dtype=torch.int64, device=dev)
        wcap = max(FORWARD_WORDS_PER_WALKER * k // n_parts, FORWARD_MIN_WORDS, 1) if lanes == 2 else 1
        while True:
            box_words = torch.empty((n_parts, wcap), dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(L.n2v_partition_forward(
                    head_in.data_ptr(), HEAD_COLS, nxt32.data_ptr(), edge.data_ptr() if wedge else 0, k, self.L,
                    part.bounds.data_ptr(), n_parts, carry,
                    part.edge_classes.data_ptr() if wedge else 0, part.wedge_off.data_ptr() if wedge else 0,
                    part.wedge_pos.data_ptr() if wedge else 0, int(wedge and part.wedge_pos.dtype == torch.int32),
                    box_head.data_ptr(), box_off.data_ptr(), box_words.data_ptr(), count.data_ptr(), k, wcap,
                    log.data_ptr(), 0, 0, self.status.data_ptr(), stream), "n2v_partition_forward")
            host = torch.cat([count, self.status[:1].to(torch.int64)]).tolist()  # the step's one host read
            word = int(host[-1])
            if not (word & _lib.ST_OVERFLOW):
                break
            wcap = int(max(host[n_parts:2 * n_parts])) + max(FORWARD_MIN_WORDS, 1)
            count.zero_()
            self.status[0] = word & ~_lib.ST_OVERFLOW
        self.log.append(log)
        self.last_status |= word
        self._note_counts([int(c) for c in host[:2 * n_parts]])
        if not self.defer_status:
            _lib.check_status_word(word, "n2v_partition_step")
        return [Mail(box_head[d, :host[d]], box_off[d, :host[d]], box_words[d, :host[n_parts + d]])
                for d in range(n_parts)]

    def advance_bounded(self, n_parts: int):
        """_advance_forward WITHOUT a host read: the outboxes have a fixed capacity per destination
        (`self.boxes`, kept across steps) and travel whole, so neither the sizes of the exchange nor
        the next launch depend on a count.  A walker or list that does not fit sets N2V_ST_OVERFLOW
        in the status word, which the driver reads once, after the last step, and then repeats the
        walk with larger boxes.  Input: the resident mail of the calibration steps the first time,
        from then on the inbox the exchange filled (`boxes.recv_*`).  Allocates nothing (the parts of
        a one-process run are stepped on separate streams)."""
        from node2vec_amd import _lib

        L = _lib.load()
        part, dev, bx = self.part, self.part.device, self.boxes
        lanes = self._lane_mode()
        wedge = lanes >= 2
        carry = lanes if wedge else 0
        if bx.inbox_live:
            head_in, off_in, words_in = bx.recv_head, bx.recv_off, bx.recv_words
        else:
            ml = self.mail
            head_in, off_in = ml.head, ml.off
            words_in = ml.words if ml.words.numel() else bx.recv_words
        k = int(head_in.shape[0])
        bx.send_head[:, 0].fill_(-1)  # every slot empty until the forwarding writes it
        bx.count.zero_()
        if k:
            log = bx.logs[bx.step_no]
            stream = _lib.current_stream_ptr()
            with torch.cuda.device(dev):
                _lib.check(L.n2v_partition_step(part.rowptr.data_ptr(), part.col.data_ptr(), 0, 0, part.lo,
                                                part.hi - part.lo, head_in.data_ptr(), HEAD_COLS, off_in.data_ptr(),
                                                words_in.data_ptr(), 2,  # N2V_SRC_WEDGES_AT
                                                k, float(self.p), float(self.q), int(self.seed) & (2 ** 64 - 1),
                                                bx.nxt.data_ptr(), bx.edge.data_ptr() if wedge else 0,
                                                self.status.data_ptr(), stream), "n2v_partition_step")
                _lib.check(L.n2v_partition_forward_boxes(
                    head_in.data_ptr(), HEAD_COLS, bx.nxt.data_ptr(), bx.edge.data_ptr() if wedge else 0, k, self.L,
                    part.bounds.data_ptr(), n_parts, carry,
                    part.edge_classes.data_ptr() if wedge else 0, part.wedge_off.data_ptr() if wedge else 0,
                    part.wedge_pos.data_ptr() if wedge else 0, int(wedge and part.wedge_pos.dtype == torch.int32),
                    bx.send_head.data_ptr(), bx.send_off.data_ptr(), bx.send_words.data_ptr(), bx.count.data_ptr(),
                    bx.box_starts.data_ptr(), log.data_ptr(), self.status.data_ptr(), stream),
                    "n2v_partition_forward_boxes")
            if lanes == 2:
                bx.send_off.add_(bx.off_add)
            torch.maximum(bx.need, bx.count, out=bx.need)
        bx.step_no += 1
        bx.inbox_live = True

    def begin_bounded(self, boxes: Outboxes, steps: int):
        """before the first bounded step (may allocate; on the stream the walk was started on)"""
        dev = self.part.device
        self.boxes = boxes
        self.mail = Mail(self.mail.head.contiguous(), self.mail.off.contiguous(), self.mail.words)
        if self.status is None:
            self.status = torch.zeros(4, dtype=torch.int32, device=dev)
        boxes.begin(len(self.mail), steps, self._lane_mode() >= 2)

    def end_bounded(self):
        self.log += [lg for lg in self.boxes.logs[:self.boxes.step_no] if lg.numel()]
        self.mail = Mail.empty(self.part.device)

    def advance(self, n_parts: int) -> List[Walkers]:
        part, dev, wk = self.part, self.part.device, self.walkers
        if self.forwarding():
            if len(self.mail) == 0:
                self._note_counts([0] * (2 * n_parts))
                return [Mail.empty(dev) for _ in range(n_parts)]
            return self._advance_forward(n_parts)
        if len(wk) == 0:
            return [Walkers.empty(dev) for _ in range(n_parts)]
        if self.step_fn is hip_step and part.rowptr.is_cuda:
            return self._advance_fused(n_parts)
        # an injected step function (the tests pass the oracle's; `tables_step` is the reference's
        # transformer on materialised tables): it takes the joined rows packed
        rows, keys, sv, step = wk.head[:, 0], wk.head[:, 1], wk.head[:, 2], wk.head[:, 3]
        s = (sv >> 32).to(torch.int32)
        v = (sv & 0xffffffff).to(torch.int64)
        local = v - part.lo
        # pack the rows of the current vertices (the `dst_neighbors` of the joined row)
        lens = (part.rowptr[1:] - part.rowptr[:-1])[local]
        dptr = torch.zeros(len(wk) + 1, dtype=torch.int64, device=dev)
        torch.cumsum(lens, 0, out=dptr[1:])
        dids = _gather_rows(part.rowptr, part.col, local, dptr)
        dw = None if part.w is None else _gather_rows(part.rowptr, part.w, local, dptr)
        nxt = self.step_fn(dptr, dids, dw, s, wk.ptr, wk.ids, keys.contiguous(),
                           step.to(torch.int32).contiguous(), self.p, self.q, self.seed)
        nxt = torch.as_tensor(nxt, device=dev).to(torch.int64)
        self.log.append(torch.stack([rows, step + 1, nxt], 1))
        done = step + 1 >= self.L
        keep = ~done
        head = torch.stack([rows, keys, (v << 32) | nxt, step + 1, torch.zeros_like(rows)], 1)[keep]
        sub_ptr, sub_ids = self._subrows(dptr, dids, keep)
        moving = Walkers(head, sub_ptr[1:] - sub_ptr[:-1], sub_ids, sub_ptr)
        dest = part.owner(nxt[keep])
        return [moving.select(torch.nonzero(dest == r).reshape(-1)) for r in range(n_parts)]

    @staticmethod
    def _subrows(ptr, ids, keep):
        idx = torch.nonzero(keep).reshape(-1)
        lens = (ptr[1:] - ptr[:-1])[idx]
        out = torch.zeros(idx.numel() + 1, dtype=torch.int64, device=ptr.device)
        torch.cumsum(lens, 0, out=out[1:])
        return out, _gather_rows(ptr, ids, idx, out)

    # -- arrivals: walkers whose new current vertex has no out-edges vanish (fugue.py:147) -----
    def receive(self, inbox: Sequence[Walkers]):
        part, dev = self.part, self.part.device
        if self.forwarding():
            self.mail = Mail.cat(inbox, dev)
            return
        wk = Walkers.cat(inbox, dev)
        # (the fused step finds the sinks itself: it draws nothing from an empty row and the
        # routing logs the walker as vanished -- no look at the arrivals, no host sync)
        if len(wk) and not (self.step_fn is hip_step and part.rowptr.is_cuda):
            v = (wk.head[:, 2] & 0xffffffff) - part.lo
            deg = (part.rowptr[1:] - part.rowptr[:-1])[v]
            dead = deg == 0
            if bool(dead.any()):
                self.log.append(torch.stack([wk.head[dead, 0], torch.full_like(wk.head[dead, 0], -1),
                                             torch.full_like(wk.head[dead, 0], -1)], 1))
                wk = wk.select(torch.nonzero(~dead).reshape(-1))
        self.walkers = wk

    def log_by_home(self, start_ids_global: torch.Tensor, n_parts: int) -> List[torch.Tensor]:
        """the path records, split by the rank that emits each row (owner of its start vertex).  The
        start list is sorted and the parts are vertex ranges, so the rows of a home rank are a RANGE of
        output rows: the home of a record is a search over n_parts cuts; the records are then grouped by a
        sort of that key -- one byte while the sentinel `n_parts` fits in it, int16 / int32 for larger worlds
        (empty slots of the bounded inboxes -- row -1 -- last, and dropped)."""
        dev = self.part.device
        rec = torch.cat(self.log) if self.log else torch.zeros((0, 3), dtype=torch.int64, device=dev)
        if rec.shape[0] == 0:
            return [rec for _ in range(n_parts)]
        s = start_ids_global.to(device=dev, dtype=torch.int64)
        # first output row of every rank's start vertices (bounds = the first vertex of every part)
        cuts = torch.searchsorted(s, self.part.bounds.to(torch.int64)) * self.W
        row = rec[:, 0].contiguous()
        home = torch.searchsorted(cuts, row, right=True) - 1  # (-1 for an empty slot)
        key_dtype = torch.uint8 if n_parts < 256 else (torch.int16 if n_parts < 32768 else torch.int32)
        key = torch.where(row >= 0, home, torch.full_like(home, n_parts)).to(key_dtype)
        key, order = torch.sort(key)
        counts = torch.bincount(key.to(torch.int64), minlength=n_parts + 1).tolist()  # one host read
        rec = rec[order]
        return list(torch.split(rec, counts)[:n_parts])

    def assemble(self, records: Sequence[torch.Tensor], start_ids_global: torch.Tensor):
        """rows of this rank (start vertices in its range, in start-list order):
        (walks int32 [k * W, L + 1], valid bool [k * W], global row ids int64)"""
        part, dev = self.part, self.part.device
        s = start_ids_global.to(device=dev, dtype=torch.int64)
        # (sorted start list, contiguous vertex range: this rank's start vertices are a range of the list)
        i0, i1 = torch.searchsorted(s, torch.tensor([part.lo, part.hi], dtype=torch.int64, device=dev)).tolist()
        n_rows, width = (i1 - i0) * self.W, self.L + 1
        rows = torch.arange(i0 * self.W, i1 * self.W, device=dev, dtype=torch.int64)
        # one spare cell behind the rows: where the writes of records that carry no vertex go
        flat = torch.full((n_rows * width + 1,), -1, dtype=torch.int32, device=dev)
        ok = torch.ones(n_rows + 1, dtype=torch.bool, device=dev)
        rec = torch.cat([r.to(dev) for r in records]) if records else torch.zeros((0, 3), dtype=torch.int64, device=dev)
        if rec.numel():
            r = rec[:, 0] - i0 * self.W
            pos, vertex = rec[:, 1], rec[:, 2]
            dropped = pos < 0  # vanished on arrival (fugue.py:147)
            flat[torch.where(dropped, n_rows * width, r * width + pos)] = vertex.to(torch.int32)
            # (vertex < 0 at position 0: a start vertex without out-edges)
            ok[torch.where(dropped | (vertex < 0), r, n_rows)] = False
        ok[n_rows:] = True
        return flat[:n_rows * width].view(n_rows, width), ok[:n_rows], rows


# ---- drivers ----------------------------------------------------------------------------------
# first size of the pool the wedge lists of a step are appended to (_walk_local_forwarding): a step
# that needs more reports how much, the pool is enlarged and the step repeated
FORWARD_WORDS_PER_WALKER = 8
FORWARD_STREAMS = True  # step the parts of a step on separate streams
FORWARD_MIN_WORDS = 1 << 16
FORWARD_MAX_PARTS = 256  # kFwdMaxParts of csrc/n2v_partition.hip
# walk_partitioned's ranks: after this many steps with exact sizes (one host read each: their counts
# are what the capacities are set from -- the walkers start spread over the VERTICES and are spread
# like the EDGES one step later) the mailboxes get a fixed capacity per destination,
# BOUNDED_SLACK times the largest flow between two ranks seen, and the remaining steps read nothing
# on the host.  BOUNDED = False: every step with exact sizes, as in round 4.
BOUNDED = True
BOUNDED_CALIBRATION_STEPS = 3  # capacities from the larger flow of the last TWO (p < 1 sends walkers back: the
#                                flows of even and odd steps differ until the walkers have mixed)
BOUNDED_SLACK = 1.5
BOUNDED_MAX_RETRIES = 2  # walk_partitioned: overflowing attempts with enlarged boxes before exact sizes take over
BOUNDED_MIN_SLOTS = 256


def _forward_mode(parts: Sequence[GraphPart], p: float, q: float, step_fn: Callable) -> int:
    """the lane mode (RankState._lane_mode) when every part can run the step as per-lane work and
    forward its walkers with n2v_partition_forward -- unit weights, all parts on one GPU, the
    per-edge tables on every part unless p == q == 1 --, else 0"""
    if step_fn is not hip_step or not parts or not all(pt.rowptr.is_cuda for pt in parts):
        return 0
    if len({pt.device for pt in parts}) != 1 or any(pt.w is not None for pt in parts):
        return 0
    if len(parts) > FORWARD_MAX_PARTS:
        return 0
    tables = all(pt.wedge_off is not None for pt in parts)
    modes = set()
    for pt in parts:
        st = RankState(pt, 1, 1, p, q, 0)
        st.use_tables = tables
        modes.add(st._lane_mode())
    return modes.pop() if len(modes) == 1 else 0


def _walk_local_forwarding(parts: Sequence[GraphPart], start_ids: torch.Tensor, num_walks: int,
                           walk_length: int, p: float, q: float, seed: int, lanes: int,
                           timings: Optional[dict] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """walk_partitioned_local with TWO launches per part and step and one host read per step:
    n2v_partition_step on the part's mailbox, then n2v_partition_forward, which writes the path
    straight into the output rows and appends every walker (and the wedge list of the edge it
    leaves along) to the mailbox of the part it goes to -- no sort, no prefix sum, no size on the
    host.  The host reads the mailbox counts once per step (they are the next step's launch
    sizes) together with the status word; a word pool that turns out too small is enlarged to
    the size the step reported and the step repeated (its writes are idempotent)."""
    from node2vec_amd import _lib

    L = _lib.load()
    _lib.require_gpu()
    n, dev, W, Lw = len(parts), parts[0].device, int(num_walks), int(walk_length)
    if p == 0 or q == 0:
        raise ValueError(f"Zero return ({p}) or inout ({q}) parameter!")
    s = start_ids.to(device=dev, dtype=torch.int64)
    total = int(s.numel()) * W
    walks = torch.full((total, Lw + 1), -1, dtype=torch.int32, device=dev)
    valid = torch.ones(total, dtype=torch.uint8, device=dev)
    # initiate_random_walk (randomwalk.py:279-296): the walkers of every part's start vertices
    ords = torch.arange(W, device=dev, dtype=torch.int64)
    heads = []
    for pt in parts:
        idx = torch.nonzero((s >= pt.lo) & (s < pt.hi)).reshape(-1)
        mine = s[idx]
        live = ((pt.rowptr[1:] - pt.rowptr[:-1])[mine - pt.lo] > 0).repeat_interleave(W)  # fugue.py:132
        rows = (idx[:, None] * W + ords[None, :]).reshape(-1)
        keys = (mine[:, None] * W + ords[None, :]).reshape(-1)
        v = mine.repeat_interleave(W)
        walks[rows, 0] = torch.where(live, v, torch.full_like(v, -1)).to(torch.int32)
        valid[rows[~live]] = 0
        rows, keys, v = rows[live], keys[live], v[live]
        heads.append(torch.stack([rows, keys, (torch.full_like(v, -1) << 32) | (v & 0xffffffff),
                                  torch.zeros_like(rows), torch.zeros_like(rows)], 1))
    counts = [int(h.shape[0]) for h in heads]
    cap = max(sum(counts), 1)
    if Lw == 0 or sum(counts) == 0:
        return walks, valid.bool()
    wedge = lanes >= 2          # the edge drawn is needed: something of it travels
    carry = lanes if wedge else 0  # N2V_SRC_WEDGES + 1 (the list) / + 2 (counts only) / nothing
    wcap = max(FORWARD_WORDS_PER_WALKER * cap // n, FORWARD_MIN_WORDS, 1) if lanes == 2 else 1

    class Boxes:
        def __init__(self):
            self.head = torch.empty((n, cap, HEAD_COLS), dtype=torch.int64, device=dev)
            self.off = torch.zeros((n, cap), dtype=torch.int64, device=dev)
            self.words = torch.zeros((n, wcap), dtype=torch.int32, device=dev)  # one pool per destination
            self.count = torch.zeros(2 * n, dtype=torch.int64, device=dev)     # walkers, then words

    cur, nxt = Boxes(), Boxes()
    for r, h in enumerate(heads):
        cur.head[r, :counts[r]] = h
    del heads
    nxt32 = torch.empty(cap, dtype=torch.int32, device=dev)
    edge = torch.empty(cap, dtype=torch.int64, device=dev) if wedge else None
    status = torch.zeros((n, 4), dtype=torch.int32, device=dev)  # one word set per part
    seed64 = int(seed) & (2 ** 64 - 1)
    # The parts of a step are independent (in a real run they are different GPUs): each is stepped
    # on its own stream, so that the launch of one part does not wait for the slowest lane of the
    # previous one (a launch is one or two generations of waves: its time is its longest walker).
    main = torch.cuda.current_stream(dev)
    streams = [torch.cuda.Stream(device=dev) for _ in parts] if FORWARD_STREAMS else [main] * n
    for _ in range(Lw):
        while True:
            nxt.count.zero_()
            at = 0
            if FORWARD_STREAMS:
                for st in streams:
                    st.wait_stream(main)
            for r, pt in enumerate(parts):
                k = counts[r]
                if k == 0:
                    continue
                nx_r = nxt32[at:at + k]
                ed_r = edge[at:at + k] if wedge else None
                at += k
                with torch.cuda.device(dev), torch.cuda.stream(streams[r]):
                    stream = _lib.current_stream_ptr()
                    _lib.check(L.n2v_partition_step(pt.rowptr.data_ptr(), pt.col.data_ptr(), 0, 0, pt.lo,
                                                    pt.hi - pt.lo, cur.head[r].data_ptr(), HEAD_COLS,
                                                    cur.off[r].data_ptr(), cur.words[r].data_ptr(),
                                                    2,  # N2V_SRC_WEDGES_AT
                                                    k, float(p), float(q), seed64, nx_r.data_ptr(),
                                                    ed_r.data_ptr() if wedge else 0, status[r].data_ptr(), stream),
                               "n2v_partition_step")
                    _lib.check(L.n2v_partition_forward(
                        cur.head[r].data_ptr(), HEAD_COLS, nx_r.data_ptr(), ed_r.data_ptr() if wedge else 0, k,
                        Lw, pt.bounds.data_ptr(), n, carry,
                        pt.edge_classes.data_ptr() if wedge else 0, pt.wedge_off.data_ptr() if wedge else 0,
                        pt.wedge_pos.data_ptr() if wedge else 0,
                        int(wedge and pt.wedge_pos.dtype == torch.int32), nxt.head.data_ptr(),
                        nxt.off.data_ptr(), nxt.words.data_ptr(), nxt.count.data_ptr(), cap,
                        nxt.words.shape[1], 0, walks.data_ptr(), valid.data_ptr(), status[r].data_ptr(), stream),
                        "n2v_partition_forward")
            if FORWARD_STREAMS:
                for st in streams:
                    main.wait_stream(st)
            word = status[0, 0]
            for r in range(1, n):
                word = word | status[r, 0]
            host = torch.cat([nxt.count, word.reshape(1).to(torch.int64)]).tolist()  # the step's one host read
            word = int(host[-1])
            if word & ~_lib.ST_OVERFLOW:
                _lib.check_status_word(word & ~_lib.ST_OVERFLOW, "n2v_partition_step")
            if not (word & _lib.ST_OVERFLOW):
                break
            # a pool was too small: the step says how many words every destination needs
            nxt.words = torch.zeros((n, int(max(host[n:2 * n])) + max(FORWARD_MIN_WORDS, 1)), dtype=torch.int32,
                                    device=dev)
            if timings is not None:
                timings["pool_enlarged"] = timings.get("pool_enlarged", 0) + 1
            status.zero_()
        if timings is not None:
            timings.setdefault("walkers_per_step", []).append(host[:n])
        counts = [int(c) for c in host[:n]]
        cur, nxt = nxt, cur
        if sum(counts) == 0:
            break
    return walks, valid.bool()


def _ST_OVERFLOW() -> int:
    from node2vec_amd import _lib

    return _lib.ST_OVERFLOW


def _check_word(word: int):
    from node2vec_amd import _lib

    _lib.check_status_word(int(word), "n2v_partition_step")


def _bounded_caps(flows_h, flows_w, lanes: int):
    """capacities per (source, destination) -- header slots, list words -- from the flows [s][d] of the
    last step with exact sizes"""
    caps_h = [[max(int(BOUNDED_SLACK * f) + 1, BOUNDED_MIN_SLOTS) for f in row] for row in flows_h]
    caps_w = [[max(int(BOUNDED_SLACK * f) + 1, 8 * BOUNDED_MIN_SLOTS) if lanes == 2 else 0 for f in row]
              for row in flows_w]
    return caps_h, caps_w


def _bounded_caps_retry(caps, need_h, need_w, lanes: int):
    """after an overflow: every box at least what the failed attempt asked of it (a lower bound: a
    walker that did not fit was not walked further) with the slack on top; a box that overflowed at
    least doubles"""
    def grow(c, need):
        new = max(c, int(BOUNDED_SLACK * need) + 1)
        return max(new, 2 * c) if need > c else new

    caps_h = [[grow(c, f) for c, f in zip(cr, fr)] for cr, fr in zip(caps[0], need_h)]
    caps_w = [[grow(c, f) if lanes == 2 else 0 for c, f in zip(cr, fr)] for cr, fr in zip(caps[1], need_w)]
    return caps_h, caps_w


def _transposed_blocks(t: torch.Tensor, caps) -> List[torch.Tensor]:
    """`t` holds boxes laid out by (source, destination): its blocks in (destination, source) order --
    their concatenation is what the all-to-all delivers (a one-process run does it as ONE cat)"""
    n = len(caps)
    at, run = [[0] * n for _ in range(n)], 0
    for s_ in range(n):
        for d in range(n):
            at[s_][d] = run
            run += int(caps[s_][d])
    return [t[at[s_][d]:at[s_][d] + int(caps[s_][d])] for d in range(n) for s_ in range(n)]


def walk_partitioned_local(parts: Sequence[GraphPart], start_ids: torch.Tensor, num_walks: int,
                           walk_length: int, p: float, q: float, seed: int,
                           step_fn: Callable = hip_step, forwarding: Optional[bool] = None,
                           timings: Optional[dict] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Every rank of the partition in ONE process (the all-to-all is a list transpose): returns
    (walks, valid) in the row order of n2v_walk over the same start list.  `forwarding`: None =
    the two-launch form (_walk_local_forwarding) wherever it applies, False = always the
    launch-per-stage routing (route, group, prefix sum, gather), True = insist, "ranks" = every
    part as a rank of walk_partitioned steps it (n2v_partition_forward into per-destination
    outboxes, walkers travelling as Mail), the exchange being a list transpose."""
    lanes = _forward_mode(parts, p, q, step_fn) if forwarding is not False else 0
    if forwarding is True and not lanes:
        raise ValueError("forwarding: unit-weight parts on one GPU with the per-edge tables (or p == q == 1) only")
    if not lanes:
        forwarding = False  # ("ranks": the ranks then route launch by stage, as walk_partitioned's would)
    if lanes and forwarding != "ranks":
        return _walk_local_forwarding(parts, start_ids, num_walks, walk_length, p, q, seed, lanes, timings)
    n = len(parts)
    ranks = [RankState(pt, num_walks, walk_length, p, q, seed, step_fn) for pt in parts]
    # what travels with a walker (wedge lists or rows) must be the same on every part: lists only
    # if EVERY part stores them (walk_partitioned agrees on this with an all-reduce)
    tables = all(pt.wedge_off is not None for pt in parts)
    for r in ranks:
        r.use_tables = tables
        r.forward = forwarding == "ranks"  # what walk_partitioned's ranks do, the exchange a list transpose
        r.initiate(start_ids)
    dev = parts[0].device

    def mark(name):  # timings["sections"] = {}: wall time per phase (synchronising: a diagnostic)
        if timings is not None and "sections" in timings:
            import time

            torch.cuda.synchronize(dev)
            now = time.perf_counter()
            timings["sections"][name] = timings["sections"].get(name, 0.0) + now - timings.get("_t", now)
            timings["_t"] = now

    mark("start")
    bounded = BOUNDED and forwarding == "ranks" and all(r.forwarding() for r in ranks)
    exact_steps = min(BOUNDED_CALIBRATION_STEPS, walk_length) if bounded else walk_length
    caps = None
    while True:
        for _ in range(exact_steps):
            out = [r.advance(n) for r in ranks]
            for d, r in enumerate(ranks):
                r.receive([out[src][d] for src in range(n)])
        mark("exact steps")
        if exact_steps == walk_length:
            break
        # the remaining steps with capacity-bounded mailboxes: here the boxes of all ranks are slices of
        # one tensor per kind and the all-to-all is one gather per kind
        lanes = ranks[0]._lane_mode()
        if caps is None:
            caps = _bounded_caps([r.flow_counts[:n] for r in ranks], [r.flow_counts[n:] for r in ranks], lanes)
        caps_h, caps_w = caps
        if timings is not None:  # (one entry per attempt: an overflow repeats the walk with larger boxes)
            timings.setdefault("bounded_caps", []).append(caps)
            timings["exact_steps"] = exact_steps
        tot_h, tot_w = sum(map(sum, caps_h)), sum(map(sum, caps_w))

        def make():  # (nothing is read that was not written: a slot is empty unless its header says otherwise)
            return (torch.empty((tot_h, HEAD_COLS), dtype=torch.int64, device=dev),
                    torch.empty(tot_h, dtype=torch.int64, device=dev),
                    torch.empty(max(tot_w, 1), dtype=torch.int32, device=dev))

        send, recv = make(), make()
        blocks = [_transposed_blocks(send[0], caps_h), _transposed_blocks(send[1], caps_h)]
        if lanes == 2:
            blocks.append(_transposed_blocks(send[2], caps_w))
        sh = sw = rh = rw_ = 0
        for i, r in enumerate(ranks):
            nsh, nsw = sum(caps_h[i]), sum(caps_w[i])
            nrh, nrw = sum(caps_h[s_][i] for s_ in range(n)), sum(caps_w[s_][i] for s_ in range(n))
            r.begin_bounded(Outboxes(n, i, caps_h, caps_w, dev,
                                     (send[0][sh:sh + nsh], send[1][sh:sh + nsh], send[2][sw:sw + nsw]),
                                     (recv[0][rh:rh + nrh], recv[1][rh:rh + nrh], recv[2][rw_:rw_ + nrw])),
                            walk_length - exact_steps)
            sh, sw, rh, rw_ = sh + nsh, sw + nsw, rh + nrh, rw_ + nrw
        # (the parts are different GPUs in a real run: each is stepped on its own stream, as in
        # _walk_local_forwarding)
        main = torch.cuda.current_stream(dev)
        streams = [torch.cuda.Stream(device=dev) for _ in ranks] if FORWARD_STREAMS else [main] * n
        mark("boxes allocated")
        for step in range(exact_steps, walk_length):
            if FORWARD_STREAMS:
                for st_ in streams:
                    st_.wait_stream(main)
            for r, st_ in zip(ranks, streams):
                with torch.cuda.stream(st_):
                    r.advance_bounded(n)
            if FORWARD_STREAMS:
                for st_ in streams:
                    main.wait_stream(st_)
            if step + 1 < walk_length:  # (nobody is forwarded by the last step)
                for kind, blk in enumerate(blocks):
                    torch.cat(blk, out=recv[kind])
        host = torch.cat([r.status[:1].to(torch.int64) for r in ranks]
                         + [r.boxes.need for r in ranks]).tolist()  # the one host read of these steps
        mark("bounded steps")
        word = 0
        for x in host[:n]:
            word |= int(x) & 0xffffffff
        if not (word & _ST_OVERFLOW()):
            _check_word(word)
            for r in ranks:
                r.end_bounded()
            break
        _check_word(word & ~_ST_OVERFLOW())
        need = [host[n + 2 * n * i:n + 2 * n * (i + 1)] for i in range(n)]
        caps = _bounded_caps_retry(caps, [x[:n] for x in need], [x[n:] for x in need], lanes)
        del send, recv, blocks
        ranks = [RankState(pt, num_walks, walk_length, p, q, seed, step_fn) for pt in parts]
        for r in ranks:
            r.use_tables, r.forward = tables, True
            r.initiate(start_ids)
    logs = [r.log_by_home(start_ids, n) for r in ranks]
    mark("records by home rank")
    total = int(start_ids.numel()) * num_walks
    walks = torch.full((total, walk_length + 1), -1, dtype=torch.int32, device=dev)
    valid = torch.zeros(total, dtype=torch.bool, device=dev)
    for d, r in enumerate(ranks):
        w, v, rows = r.assemble([logs[src][d] for src in range(n)], start_ids)
        walks[rows], valid[rows] = w, v
    mark("rows assembled")
    return walks, valid


def _all_to_all_var(tensors: List[torch.Tensor], group, dist) -> List[torch.Tensor]:
    """variable-size all-to-all of one tensor per destination rank (first dimension varies)"""
    world = dist.get_world_size(group)
    dev = tensors[0].device
    cpu = dist.get_backend(group) == "gloo"  # gloo moves host tensors
    send = torch.cat(tensors)
    if cpu:
        send = send.cpu()
    n_send = torch.tensor([t.shape[0] for t in tensors], dtype=torch.int64, device=send.device)
    n_recv = torch.empty(world, dtype=torch.int64, device=send.device)
    dist.all_to_all_single(n_recv, n_send, group=group)
    recv = torch.empty((int(n_recv.sum()),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
    dist.all_to_all_single(recv, send.contiguous(), n_recv.tolist(), n_send.tolist(), group=group)
    return [t.to(dev) for t in torch.split(recv, n_recv.tolist())]


def _exchange_walkers(out: List[Walkers], group, dist, dev, status: int = 0) -> List[Walkers]:
    """The migration of one step: three collectives -- the sizes (walkers and row words per
    destination, one [world, 2] exchange), the headers with the length of what travels with each
    walker as an extra column, and those words.  Segments arrive in source order in both
    payloads, so the receiver rebuilds ONE batch."""
    cpu = dist.get_backend(group) == "gloo"  # gloo moves host tensors
    wire = torch.device("cpu") if cpu else dev
    head5 = torch.cat([torch.cat([w.head, w.lens[:, None]], 1) for w in out]).to(wire)
    ids = torch.cat([w.ids for w in out]).to(wire)
    sizes = [[len(w), int(w.ids.numel())] for w in out]  # shapes: known on the host, no sync
    # (the third column: this rank's status word, so that an error of one rank is every rank's)
    n_send = torch.tensor([z + [int(status)] for z in sizes], dtype=torch.int64, device=wire)
    n_recv = torch.empty_like(n_send)
    dist.all_to_all_single(n_recv, n_send, group=group)
    got = n_recv.tolist()
    bad = 0
    for g3 in got:
        bad |= int(g3[2])
    if bad:
        from node2vec_amd import _lib

        _lib.check_status_word(bad, "n2v_partition_step (on some rank)")
    recv_h = torch.empty((sum(g[0] for g in got), HEAD_COLS + 1), dtype=torch.int64, device=wire)
    dist.all_to_all_single(recv_h, head5.contiguous(), [g[0] for g in got], [z[0] for z in sizes], group=group)
    recv_i = torch.empty(sum(g[1] for g in got), dtype=torch.int32, device=wire)
    dist.all_to_all_single(recv_i, ids.contiguous(), [g[1] for g in got], [z[1] for z in sizes], group=group)
    recv_h, recv_i = recv_h.to(dev), recv_i.to(dev)
    return [Walkers(recv_h[:, :HEAD_COLS].contiguous(), recv_h[:, HEAD_COLS].contiguous(), recv_i)]


def _exchange_mail(out: List[Mail], group, dist, dev, status: int = 0) -> List[Mail]:
    """_exchange_walkers for walkers that travel as Mail: the sizes, the headers with the list
    start as a sixth column, the word pools.  Pool d of this rank is what rank d gets, so the
    payloads are the outboxes as they lie; the receiver rebases the list starts of every source
    to where that source's words landed."""
    cpu = dist.get_backend(group) == "gloo"  # gloo moves host tensors
    wire = torch.device("cpu") if cpu else dev
    head6 = torch.cat([torch.cat([m.head, m.off[:, None]], 1) for m in out]).to(wire)
    words = torch.cat([m.words for m in out]).to(wire)
    sizes = [[len(m), int(m.words.numel())] for m in out]  # shapes: known on the host, no sync
    n_send = torch.tensor([z + [int(status)] for z in sizes], dtype=torch.int64, device=wire)
    n_recv = torch.empty_like(n_send)
    dist.all_to_all_single(n_recv, n_send, group=group)
    got = n_recv.tolist()
    bad = 0
    for g3 in got:
        bad |= int(g3[2])
    if bad:
        from node2vec_amd import _lib

        _lib.check_status_word(bad, "n2v_partition_step (on some rank)")
    recv_h = torch.empty((sum(g[0] for g in got), HEAD_COLS + 1), dtype=torch.int64, device=wire)
    dist.all_to_all_single(recv_h, head6.contiguous(), [g[0] for g in got], [z[0] for z in sizes], group=group)
    recv_w = torch.empty(sum(g[1] for g in got), dtype=torch.int32, device=wire)
    dist.all_to_all_single(recv_w, words.contiguous(), [g[1] for g in got], [z[1] for z in sizes], group=group)
    recv_h, recv_w = recv_h.to(dev), recv_w.to(dev)
    bases, run = [], 0
    for g3 in got:  # where the words of every source start in recv_w
        bases.append(run)
        run += int(g3[1])
    add = torch.repeat_interleave(torch.tensor(bases, dtype=torch.int64, device=dev),
                                  torch.tensor([int(g3[0]) for g3 in got], dtype=torch.int64, device=dev))
    return [Mail(recv_h[:, :HEAD_COLS].contiguous(), recv_h[:, HEAD_COLS] + add, recv_w)]


def walk_partitioned(part: GraphPart, start_ids: torch.Tensor, num_walks: int, walk_length: int,
                     p: float, q: float, seed: int, group=None, step_fn: Callable = hip_step,
                     timings: Optional[dict] = None):
    """One rank of the partitioned walk under torch.distributed (one process per GPU; backend
    "nccl" = RCCL over xGMI, or gloo on CPU tensors).  Every rank passes the same sorted
    `start_ids` and the same seed.  Returns this rank's rows: (walks, valid, global row ids) --
    the rows of the start vertices in its range, bit-identical to n2v_walk on the whole graph."""
    import torch.distributed as dist

    from node2vec_amd.shard import all_reduce

    world = dist.get_world_size(group)
    st = RankState(part, num_walks, walk_length, p, q, seed, step_fn)
    # what travels with a walker must mean the same thing on both ends: wedge lists only when
    # EVERY rank holds the per-edge tables of its part
    have = torch.tensor([0 if part.wedge_off is None else 1], dtype=torch.int32, device=part.device)
    all_reduce(have, dist.ReduceOp.MIN, group)
    use_tables = bool(have.item())
    dev = part.device
    cpu = dist.get_backend(group) == "gloo"  # gloo moves host tensors
    caps = None
    overflows = 0  # attempts that ended in N2V_ST_OVERFLOW (the status word is all-reduced: the same on every rank)
    while True:
        st.use_tables = use_tables
        st.defer_status = True
        # per-lane steps (unit weights; the tables on every rank, or p == q == 1): route with
        # n2v_partition_forward, one launch; every rank takes the same branch (use_tables is agreed)
        st.forward = True
        st.initiate(start_ids)
        # a flow that keeps outgrowing its boxes (BOUNDED_MAX_RETRIES enlargements did not hold it): the
        # walk is then stepped with exact sizes throughout, a host read per step -- it always terminates
        bounded = BOUNDED and st.forwarding() and overflows <= BOUNDED_MAX_RETRIES
        exact_steps = min(BOUNDED_CALIBRATION_STEPS, walk_length) if bounded else walk_length
        for _ in range(exact_steps):
            out = st.advance(world)
            exchange = _exchange_mail if st.forwarding() else _exchange_walkers
            st.receive(exchange(out, group, dist, dev, st.last_status))
        if exact_steps == walk_length:
            break
        # the remaining steps: capacity-bounded mailboxes, exchanged whole -- no size, count or status
        # is read on the host until the last step is done (fugue.py:146-149: one shuffle per step)
        lanes = st._lane_mode()
        rank = part.rank
        if caps is None:
            m = torch.zeros((world, 2 * world), dtype=torch.int64, device=dev)
            m[rank] = torch.tensor(st.flow_counts or [0] * (2 * world), dtype=torch.int64, device=dev)
            all_reduce(m, dist.ReduceOp.SUM, group)
            m = m.tolist()  # (the last host read of the calibration: the flows between every two ranks)
            caps = _bounded_caps([row[:world] for row in m], [row[world:] for row in m], lanes)
        if timings is not None:  # (one entry per attempt: an overflow repeats the walk with larger boxes)
            timings.setdefault("bounded_caps", []).append(caps)
            timings["exact_steps"] = exact_steps
        bx = Outboxes(world, rank, caps[0], caps[1], dev)
        st.begin_bounded(bx, walk_length - exact_steps)
        for step in range(exact_steps, walk_length):
            st.advance_bounded(world)
            if step + 1 < walk_length:  # (nobody is forwarded by the last step)
                _exchange_bounded(bx, group, dist, cpu, lanes == 2)
        fin = torch.zeros((world, 1 + 2 * world), dtype=torch.int64, device=dev)
        fin[rank, 0] = st.status[0].to(torch.int64) & 0xffffffff
        fin[rank, 1:] = bx.need
        all_reduce(fin, dist.ReduceOp.SUM, group)
        fin = fin.tolist()  # the one host read of these steps
        word = 0
        for row in fin:
            word |= int(row[0])
        if not (word & _ST_OVERFLOW()):
            _check_word(word)
            st.end_bounded()
            break
        _check_word(word & ~_ST_OVERFLOW())
        overflows += 1
        if timings is not None:
            timings["bounded_overflows"] = overflows
        caps = _bounded_caps_retry(caps, [row[1:1 + world] for row in fin], [row[1 + world:] for row in fin], lanes)
        del bx
        st = RankState(part, num_walks, walk_length, p, q, seed, step_fn)
    records = _all_to_all_var(st.log_by_home(start_ids, world), group, dist)
    return st.assemble(records, start_ids)


def _exchange_bounded(bx: Outboxes, group, dist, cpu: bool, words: bool):
    """the migration of one step with capacity-bounded mailboxes: box d of every rank goes to rank d
    whole -- headers, list starts and, when lists travel, the word pools; the split sizes are the
    capacities, fixed for the walk"""
    todo = [(bx.recv_head, bx.send_head, bx.recv_h, bx.send_h), (bx.recv_off, bx.send_off, bx.recv_h, bx.send_h)]
    if words:
        todo.append((bx.recv_words, bx.send_words, bx.recv_w, bx.send_w))
    for recv, send, n_recv, n_send in todo:
        if cpu:
            got = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_to_all_single(got, send.cpu(), n_recv, n_send, group=group)
            recv.copy_(got)
        else:
            dist.all_to_all_single(recv, send, n_recv, n_send, group=group)
