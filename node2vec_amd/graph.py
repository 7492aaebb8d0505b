"""CSR adjacency in HBM: the device form of the reference's df_adj.

The reference aggregates the edge DataFrame into one serialized neighbour list
per source vertex (`partition(by=src, presort=dst).transform(get_vertex_neighbors)`,
fugue.py:130, randomwalk.py:266-275).  Here the same data is a CSR triple kept
resident on the GPU: rowptr int64[V+1], col int32[E] (sorted by dst within a
row, multi-edges kept in input order), w fp32[E]; plus, for the fast sampler,
CSR-aligned first-order alias tables packed as 16-byte slots {col, alias vertex, prob}.
"""
import ctypes as C
from typing import Optional

import torch

from node2vec_amd import _lib


def rank_tables(deg: torch.Tensor, max_classes: int, max_head: int):
    """The host half of the degree-ranked form (DeviceGraph.build_ranked; plain torch, any device):
    from the out-degrees int64 [V] -- rank_vertex / rank_of int32 [V] (stable sort by descending
    degree), rank_rowptr int64 [V + 1] (row starts in rank order), the head table int64 [H] or None
    (row offset | degree << 40 of the ranks that are looked up one by one) and the class table
    first / off int32 [P] as n2v_graph wants them (P a power of two > the classes kept; the entry
    after the last class and the padding hold V / E; uint32 bits).  None when more than `max_head`
    top vertices would have to be listed."""
    n, dev = int(deg.numel()), deg.device
    n_edges = int(deg.sum())
    order = torch.sort(deg, descending=True, stable=True).indices
    deg_r = deg[order]
    rank_vertex = order.to(torch.int32)
    rank_of = torch.empty(n, dtype=torch.int32, device=dev)
    rank_of[order] = torch.arange(n, dtype=torch.int32, device=dev)
    del order
    rank_rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(deg_r, 0, out=rank_rowptr[1:])
    cls_deg, cls_count = torch.unique_consecutive(deg_r, return_counts=True)
    cls_first = torch.cumsum(cls_count, 0) - cls_count
    n_cls = int(cls_deg.numel())
    keep = min(n_cls, int(max_classes))
    head_n = int(cls_first[n_cls - keep].item())  # ranks below it are looked up one by one
    if head_n > max_head:
        return None
    P = 2
    while P < keep + 1:  # one entry after the last class closes it
        P *= 2
    first = torch.full((P,), n, dtype=torch.int64, device=dev)
    off = torch.full((P,), n_edges, dtype=torch.int64, device=dev)
    kept_first = cls_first[n_cls - keep:]
    first[:keep] = kept_first
    off[:keep] = rank_rowptr[kept_first]
    head = None
    if head_n:
        head = (rank_rowptr[:head_n] | (deg_r[:head_n] << 40)).contiguous()
    as_u32 = lambda t: torch.where(t >= (1 << 31), t - (1 << 32), t).to(torch.int32)  # noqa: E731  (uint32 bits)
    return rank_vertex, rank_of, rank_rowptr, head, as_u32(first), as_u32(off)


class DeviceGraph:
    def __init__(self, rowptr: torch.Tensor, col: torch.Tensor, w: Optional[torch.Tensor]):
        """`w` None = every weight is 1.0 (what index_graph_* produces, indexer.py:20-21):
        nothing is stored for the weights and the walk kernels never read them."""
        if rowptr.dtype != torch.int64 or col.dtype != torch.int32:
            raise TypeError("DeviceGraph wants rowptr int64, col int32")
        if w is not None and w.dtype not in (torch.float32, torch.float64):
            raise TypeError("DeviceGraph wants w float32 / float64 (or None for unit weights)")
        if (w is not None and col.numel() != w.numel()) or rowptr.numel() < 1:
            raise ValueError("DeviceGraph: ragged CSR arrays")
        self.rowptr = rowptr.contiguous()
        self.col = col.contiguous()
        # unweighted graph: the kernels are told through w == NULL
        if w is not None and (w.numel() == 0 or bool((w == 1.0).all())):
            w = None
        self._w = None if w is None else w.contiguous()
        self.unit_weights = self._w is None
        self.slots: Optional[torch.Tensor] = None  # int32 [E, 4] view of n2v_slot[E]
        self.pivots: Optional[torch.Tensor] = None  # int32 [(E + 31) / 32] search index (fast mode)
        self.edge_classes: Optional[torch.Tensor] = None  # uint32-in-int32 [E] (exact mode, unit weights)
        self.hops: Optional[torch.Tensor] = None  # int32 [E, 4] view of n2v_hop[E] (unit weights)
        self.hops_have_classes = False
        self.hops8: Optional[torch.Tensor] = None  # int64 [E]: 8-byte hop entries (p = q = 1 walks)
        self.hops8_bits = (0, 0)  # (col_bits, row_bits) of a hops8 entry
        self.hops8_rowptr: Optional[torch.Tensor] = None  # int64 [V + 1]: padded rows of the hops8 table
        self.hops8_shift = 0
        self.hops8_tried = False
        self.wedge_off: Optional[torch.Tensor] = None  # int64 [E]: list offset | return position << 40
        self.wedge_pos: Optional[torch.Tensor] = None  # int16 / int32 [sum of shared counts]
        self.wedge_mode = 0  # n2v_graph.wedge_wide: 0 16-bit lists, 1 32-bit, T >= 2 mixed (build_wedges)
        self.wedge_tried = False  # randomwalk.walk tries to build the table once
        self.wedge_slots: Optional[torch.Tensor] = None  # int16 [E, 16]: n2v_wedge_slots_build
        self.slots_folded = False  # the slots of the edges into wide rows are folded slots (n2v_wedge_slots_fold)
        # row sums of the steps into long rows for ONE (p, q) that is not dyadic (build_row_sums):
        # (fp64 [E] tensor, p, q, the wedge_off tensor they were computed from) | None
        self.row_sums = None
        self.hops_inline_rpos = False  # the hop table's class words carry return positions (slots kernel)
        self._inline_ok = None  # (edge_classes tensor, every return count < 128): can_inline_rpos()
        # the degree-ranked form (build_ranked): 4-byte entries for p = q = 1 walks
        self.rank_hops: Optional[torch.Tensor] = None  # int32 [E] (uint32 ranks)
        self.rank_of: Optional[torch.Tensor] = None  # int32 [V] vertex id -> rank
        self.rank_vertex: Optional[torch.Tensor] = None  # int32 [V] rank -> vertex id
        self.rank_head: Optional[torch.Tensor] = None  # int64 [H]
        self.rank_class_first: Optional[torch.Tensor] = None  # int32 [P] (uint32)
        self.rank_class_off: Optional[torch.Tensor] = None  # int32 [P] (uint32)
        self.rank_tried = False
        # caches of the weighted step kernels (randomwalk.weighted_row_sums, weighted_hub_summaries,
        # _walk_weighted_lanes): functions of rowptr / the stored weights alone.  `w` has no setter and
        # the arrays of a DeviceGraph are never replaced in place, so they cannot go stale.
        self._row_weight_sums = None  # fp64 [V + 2] | False (a weight negative or not finite)
        self._weighted_hubs = None    # (struct n2v_weighted_hubs, its tensors) | False (none / did not fit)
        self._degree_rank = None      # int32 [V]: place of every vertex by descending out-degree, ties by id

    @property
    def w(self) -> torch.Tensor:
        """Edge weights as a tensor (ones are materialised on demand for a unit-weight graph)."""
        if self._w is None:
            return torch.ones(self.n_edges, dtype=torch.float32, device=self.device)
        return self._w

    # -- construction ---------------------------------------------------------
    @classmethod
    def from_edges(cls, src, dst, weight=None, n_vertices: Optional[int] = None, device=None):
        """Stable sort by (src, dst) == partition(by=src, presort=dst) (fugue.py:130).

        `weight` None = unit weights.  The reference carries weights as Python floats
        (randomwalk.py:20, indexer.py:24): float64 input stays float64 in HBM unless every
        value is exactly representable in float32, in which case the 4-byte form is stored
        (both are widened to fp64 before any arithmetic, so the results are the same bits)."""
        src = torch.as_tensor(src).to(device=device, dtype=torch.int64).reshape(-1)
        dst = torch.as_tensor(dst).to(device=device, dtype=torch.int64).reshape(-1)
        w = None
        if weight is not None:
            w = torch.as_tensor(weight).to(device=device).reshape(-1)
            if w.dtype in (torch.float16, torch.bfloat16, torch.float32):
                w = w.to(torch.float32)
            else:
                w = w.to(torch.float64)
                w32 = w.to(torch.float32)
                if bool((w32.to(torch.float64) == w).all()):
                    w = w32
        if src.numel() != dst.numel() or (w is not None and w.numel() != src.numel()):
            raise ValueError("src, dst and weight differ in length")
        if src.numel() and (int(src.min()) < 0 or int(dst.min()) < 0):
            raise ValueError("vertex ids must be non-negative (negative ids mark first steps, "
                             "randomwalk.py:295)")
        hi = int(max(src.max(), dst.max())) + 1 if src.numel() else 0
        if n_vertices is None:
            n_vertices = hi
        if hi > n_vertices or n_vertices >= 2 ** 31:
            raise ValueError("vertex id out of range for int32 CSR")
        key = src * max(n_vertices, 1) + dst
        del dst
        key, order = torch.sort(key, stable=True)
        counts = torch.bincount(src, minlength=n_vertices)
        del src
        rowptr = torch.zeros(n_vertices + 1, dtype=torch.int64, device=key.device)
        torch.cumsum(counts, 0, out=rowptr[1:])
        col = (key % max(n_vertices, 1)).to(torch.int32)
        return cls(rowptr, col, None if w is None else w[order])

    @classmethod
    def from_sorted_keys(cls, key: torch.Tensor, n_vertices: int) -> "DeviceGraph":
        """Unit-weight CSR from edge keys src * n_vertices + dst that are already sorted
        ascending (torch.unique output): no second sort, no weight array."""
        src = key // max(n_vertices, 1)
        counts = torch.bincount(src, minlength=n_vertices)
        del src
        rowptr = torch.zeros(n_vertices + 1, dtype=torch.int64, device=key.device)
        torch.cumsum(counts, 0, out=rowptr[1:])
        return cls(rowptr, (key % max(n_vertices, 1)).to(torch.int32), None)

    @classmethod
    def from_pandas(cls, df, n_vertices: Optional[int] = None, device=None):
        # Neighbors(df) reads obj["dst"] and obj["weight"] (randomwalk.py:21-22)
        for name in ("src", "dst", "weight"):
            if name not in df.columns:
                raise KeyError(name)
        return cls.from_edges(df["src"].to_numpy(), df["dst"].to_numpy(),
                              df["weight"].to_numpy(), n_vertices, device)

    # -- views ------------------------------------------------------------------
    @property
    def n_vertices(self) -> int:
        return self.rowptr.numel() - 1

    @property
    def n_edges(self) -> int:
        return self.col.numel()

    @property
    def device(self):
        return self.rowptr.device

    def degrees(self) -> torch.Tensor:
        return self.rowptr[1:] - self.rowptr[:-1]

    def to(self, device) -> "DeviceGraph":
        g = DeviceGraph(self.rowptr.to(device), self.col.to(device),
                        None if self._w is None else self._w.to(device))
        for name in ("slots", "pivots", "edge_classes", "hops", "hops8", "hops8_rowptr", "wedge_off",
                     "wedge_pos", "wedge_slots", "rank_hops", "rank_of", "rank_vertex", "rank_head",
                     "rank_class_first", "rank_class_off"):
            t = getattr(self, name)
            if t is not None:
                setattr(g, name, t.to(device))
        g.hops_have_classes = self.hops_have_classes
        g.hops8_bits, g.hops8_shift = self.hops8_bits, self.hops8_shift
        # a declined build (escape share, memory budget) stays declined on the copy
        g.hops8_tried, g.wedge_tried, g.rank_tried = self.hops8_tried, self.wedge_tried, self.rank_tried
        g.hops_inline_rpos = self.hops_inline_rpos
        g.wedge_mode = self.wedge_mode
        g.slots_folded = self.slots_folded
        return g

    def c_struct(self) -> _lib.Graph:
        w32 = self._w is not None and self._w.dtype == torch.float32
        w64 = self._w is not None and self._w.dtype == torch.float64
        return _lib.Graph(self.n_vertices, self.n_edges, self.rowptr.data_ptr(),
                          self.col.data_ptr(), self._w.data_ptr() if w32 else 0,
                          self._w.data_ptr() if w64 else 0,
                          0 if self.slots is None else self.slots.data_ptr(),
                          0 if self.pivots is None else self.pivots.data_ptr(),
                          0 if self.edge_classes is None else self.edge_classes.data_ptr(),
                          0 if self.hops is None else self.hops.data_ptr(),
                          0 if self.wedge_off is None else self.wedge_off.data_ptr(),
                          0 if self.wedge_pos is None else self.wedge_pos.data_ptr(),
                          0 if self.wedge_pos is None else self.wedge_mode, 0,
                          0 if self.hops8 is None else self.hops8.data_ptr(),
                          self.hops8_bits[0], self.hops8_bits[1],
                          0 if self.hops8_rowptr is None else self.hops8_rowptr.data_ptr(),
                          self.hops8_shift,
                          int(self.hops_inline_rpos and self.hops is not None)
                          | (2 if self.slots_folded and self.wedge_slots is not None else 0),  # N2V_SLOTS_FOLDED
                          0 if self.wedge_slots is None else self.wedge_slots.data_ptr(),
                          *self._rank_fields(), *self._row_sum_fields())

    ROW_SUMS_FROM = 1024  # rows of this many entries and more have the sums of their steps' tables stored

    def _row_sum_fields(self):
        rs = self.row_sums
        if rs is None or self.wedge_off is None or rs[3] is not self.wedge_off:
            return (0, 0.0, 0.0, 0, 0)
        return (rs[0].data_ptr(), rs[1], rs[2], self.ROW_SUMS_FROM, 0)

    def build_row_sums(self, p: float, q: float, max_bytes: Optional[int] = None) -> "DeviceGraph":
        """sum(node_weights) (randomwalk.py:172) of the table of every step into a row of ROW_SUMS_FROM entries or
        more, for one (p, q) whose 1/p or 1/q is not dyadic (n2v_edge_row_sums_build): 8 bytes per edge, kept for
        the last (p, q) asked.  A step that has to know the reference's rounded sum -- 1 % of the steps on long
        rows -- then reads it instead of adding the row up by one lane.  Same bits either way."""
        L = _lib.load()
        if self.row_sums is not None and self.row_sums[1:3] == (float(p), float(q)) and \
                self.row_sums[3] is self.wedge_off:
            return self
        self.row_sums = None
        if (not self.unit_weights or self.edge_classes is None or self.wedge_off is None or self.wedge_pos is None
                or self.n_edges == 0):
            return self
        deg = self.degrees()
        if int(deg.max()) < self.ROW_SUMS_FROM:
            return self
        if max_bytes is None:
            max_bytes = torch.cuda.mem_get_info(self.device)[0] // 4
        if 8 * self.n_edges > max_bytes:
            return self
        long_row = deg >= self.ROW_SUMS_FROM
        edges = torch.nonzero(long_row[self.col.long()]).reshape(-1)
        # longest lists first: the lanes of a wave then have lists of one length
        order = torch.argsort((self.edge_classes[edges] & 0xffffff), descending=True)
        edges = edges[order].contiguous()
        del order, long_row
        sums = torch.empty(self.n_edges, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            rc = L.n2v_edge_row_sums_build(self.c_struct(), float(p), float(q), edges.data_ptr(), edges.numel(),
                                           sums.data_ptr(), _lib.current_stream_ptr())
        _lib.check(rc, "n2v_edge_row_sums_build")
        self.row_sums = (sums, float(p), float(q), self.wedge_off)
        return self

    def _rank_fields(self):
        if self.rank_hops is None:
            return (0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
        return (self.rank_hops.data_ptr(), self.rank_of.data_ptr(), self.rank_vertex.data_ptr(),
                0 if self.rank_head is None else self.rank_head.data_ptr(),
                self.rank_class_first.data_ptr(), self.rank_class_off.data_ptr(),
                0 if self.rank_head is None else self.rank_head.numel(), self.rank_class_first.numel(), 0, 0)

    # -- a9 -----------------------------------------------------------------------
    def trimmed(self, max_out_degree: int, seed: int) -> "DeviceGraph":
        """trim_hotspot_vertices (randomwalk.py:238-262) on the CSR itself: rows above the cap
        keep a uniform sample without replacement of exactly `cap` edges (n2v_trim_mark), in
        their original order; other rows and all weights are untouched.  Returns a new graph
        (or self when no row exceeds the cap)."""
        from node2vec_amd.constants import MAX_OUT_DEGREES

        cap = int(max_out_degree) if max_out_degree > 0 else MAX_OUT_DEGREES  # randomwalk.py:252-253
        deg = self.degrees()
        if self.n_edges == 0 or int(deg.max()) <= cap:
            return self
        L = _lib.load()
        _lib.require_gpu()
        keep = torch.ones(self.n_edges, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            rc = L.n2v_trim_mark(self.rowptr.data_ptr(), self.n_vertices, cap,
                                 int(seed) & (2 ** 64 - 1), keep.data_ptr(),
                                 _lib.current_stream_ptr())
        _lib.check(rc, "n2v_trim_mark")
        keep = keep.bool()
        rowptr = torch.zeros_like(self.rowptr)
        torch.cumsum(deg.clamp(max=cap), 0, out=rowptr[1:])
        return DeviceGraph(rowptr, self.col[keep], None if self._w is None else self._w[keep])

    # -- K1 -----------------------------------------------------------------------
    def build_alias(self) -> "DeviceGraph":
        """First-order Walker tables for every row == generate_alias_tables(row weights)
        (randomwalk.py:157-190), written CSR-aligned by the K1 kernel."""
        L = _lib.load()
        _lib.require_gpu()
        if not self.rowptr.is_cuda:
            raise RuntimeError("build_alias: graph is not on the GPU")
        slots = torch.zeros((self.n_edges, 4), dtype=torch.int32, device=self.device)
        if self.n_edges == 0:  # nothing to build (and no buffer to hand over)
            self.slots = slots
            return self
        status = torch.zeros(4, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            rc = L.n2v_alias_build(self.c_struct(), slots.data_ptr(), status.data_ptr(),
                                   _lib.current_stream_ptr())
        _lib.check(rc, "n2v_alias_build")
        _lib.check_status_word(int(status[0].item()), "n2v_alias_build")
        self.slots = slots
        return self.build_pivots()

    def build_edge_classes(self) -> "DeviceGraph":
        """Per-edge class counts (n2v_edge_classes_build) for exact walks: 4 bytes per edge,
        computed once, kept on the graph.  They depend on the ids alone (which slots of the table of
        step (s -> v) are return / shared / other), so weighted graphs have them too: the
        lane-per-walker kernel of weighted exact walks reads them (n2v_walk_weighted_step)."""
        L = _lib.load()
        _lib.require_gpu()
        if not self.rowptr.is_cuda:
            raise RuntimeError("build_edge_classes: graph is not on the GPU")
        ec = torch.zeros(self.n_edges, dtype=torch.int32, device=self.device)
        if self.n_edges:
            status = torch.zeros(4, dtype=torch.int32, device=self.device)
            with torch.cuda.device(self.device):
                rc = L.n2v_edge_classes_build(self.c_struct(), ec.data_ptr(), status.data_ptr(),
                                              _lib.current_stream_ptr())
            _lib.check(rc, "n2v_edge_classes_build")
        self.edge_classes = ec
        return self

    WEDGE_WIDE_FROM = 65536  # rows of this many entries or more need 32-bit positions

    def build_wedges(self, max_bytes: Optional[int] = None, wide: Optional[bool] = None,
                     slots: bool = True, wide_from: Optional[int] = None, fold: bool = True) -> "DeviceGraph":
        """Shared-position lists (n2v_wedge_build): for every edge (s -> v) the positions in
        N(v) of the neighbours v shares with s -- what generate_edge_alias_tables recomputes by a
        set intersection at every step (randomwalk.py:226), stored once.  8 bytes per edge + 2
        per (edge, common neighbour) pair; `max_bytes` (default: half of the free device memory)
        bounds it -- a graph with more triangles than that walks without the lists (same bits,
        slower on the steps that need the pairing).  With room for 32 more bytes per edge the
        wedge slots are built as well (build_wedge_slots).

        Rows of 65 536 entries or more (the reference's own trim cap is 100 000, constants.py:6) need
        32-bit positions.  Only the lists of the edges INTO such a row are widened ("mixed" table,
        n2v_graph.wedge_wide = 65536): they lie, 4 bytes per entry, behind the 16-bit lists of all
        other edges, and a walker standing on a wide row reads them through wedge_off -- every
        other step keeps the 16-bit lists and the wedge slots.  `wide=True` forces 32-bit positions
        for every list (no slots: what partitioned walking slices); `wide_from` (tests) lowers
        the row length from which lists are widened."""
        L = _lib.load()
        _lib.require_gpu()
        if self.edge_classes is None:
            self.build_edge_classes()
        self.wedge_off = self.wedge_pos = self.wedge_slots = None
        self.wedge_mode = 0
        self.slots_folded = False
        if self.n_edges == 0 or int(self.degrees().max()) >= self.HOP_MAX_DEGREE:
            return self
        counts = (self.edge_classes & 0xffffff).to(torch.int64)
        if bool((counts == 0xffffff).any()) or bool((((self.edge_classes >> 24) & 0xff) == 0xff).any()):
            return self  # a saturated count (16 M shared or 255 parallel return edges): no lists
        total = int(counts.sum())
        deg = self.degrees()
        t_from = int(wide_from) if wide_from else self.WEDGE_WIDE_FROM
        if not 2 <= t_from <= 65536:
            raise ValueError("wide_from: 2 .. 65536")
        mode = 1 if wide else (t_from if int(deg.max()) >= t_from else 0)
        total_wide = 0
        wide_e = None
        if mode >= 2:
            wide_e = (deg >= t_from)[self.col.long()]  # edges into a wide row
            total_wide = int(counts[wide_e].sum())
        need = 8 * self.n_edges + (total * 4 if mode == 1 else (total - total_wide) * 2 + total_wide * 4 + 4)
        own_budget = max_bytes is None
        if own_budget:
            max_bytes = torch.cuda.mem_get_info(self.device)[0] // 2
        if need > max_bytes or total >= (1 << 40):
            return self
        # folded copies (n2v_wedge_slots_fold) of the lists of more than 14 entries of the edges into rows of
        # t_from .. t_from + 65536 entries: 2 bytes per entry behind the 32-bit lists, when they fit as well
        fold_e, fold_total = None, 0
        # (ALL wide rows must fold -- a row of more than t_from + 65536 entries leaves the table as it is, and the
        # walk to the kernel that reads wedge_off: the slots kernel has no 32-bit instance of the step any more)
        if mode >= 2 and slots and fold and int(deg.max()) - t_from <= 65536:
            fold_e = wide_e & (counts > 14)
            fold_total = int(counts[fold_e].sum())
            if need + 2 * fold_total > max_bytes:
                fold_e, fold_total = None, 0
            need += 2 * fold_total
        fold_off = None
        if mode >= 2:
            # the 16-bit lists first (offsets in 2-byte units), then the 32-bit lists of the edges into
            # wide rows (offsets in 4-byte units from the same base)
            c16 = torch.where(wide_e, torch.zeros_like(counts), counts)
            off = torch.cumsum(c16, 0) - c16
            base32 = (total - total_wide + 1) // 2
            c32 = counts - c16
            off32 = torch.cumsum(c32, 0) - c32 + base32
            off = torch.where(wide_e, off32, off)
            del c16, c32, off32
            pos = torch.empty(2 * (base32 + max(total_wide, 1)) + fold_total, dtype=torch.int16, device=self.device)
            if fold_e is not None:
                fc = torch.where(fold_e, counts, torch.zeros_like(counts))
                fold_off = torch.cumsum(fc, 0) - fc + 2 * (base32 + max(total_wide, 1))
                del fc
        else:
            off = torch.cumsum(counts, 0) - counts  # exclusive prefix sum; becomes wedge_off in place
            pos = torch.empty(max(total, 1), dtype=torch.int32 if mode == 1 else torch.int16, device=self.device)
        del counts, wide_e, fold_e
        status = torch.zeros(4, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            rc = L.n2v_wedge_build(self.c_struct(), off.data_ptr(), off.data_ptr(), pos.data_ptr(),
                                   mode, status.data_ptr(), _lib.current_stream_ptr())
        _lib.check(rc, "n2v_wedge_build")
        if int(status[0].item()) & _lib.ST_RANGE:
            raise RuntimeError("n2v_wedge_build: list lengths disagree with edge_classes")
        self.wedge_off, self.wedge_pos, self.wedge_mode = off, pos, mode
        if slots and mode != 1:
            if own_budget:
                # (the lists are in place: the slots may take half of what is free NOW, counting the blocks torch
                # holds cached -- the 100 GB of lists of cfg 4 trimmed at the reference's cap of 100 000 left the
                # old rule, lists + slots within half of what was free before, 1 % short of the slots, and the walk
                # without them is 30 % slower: 12.8 against 16.7 G steps/s at (0.5, 2), profiles/r11b_*)
                free = (torch.cuda.mem_get_info(self.device)[0] + torch.cuda.memory_reserved(self.device)
                        - torch.cuda.memory_allocated(self.device))
                fits = 32 * self.n_edges <= free // 2
            else:
                fits = need + 32 * self.n_edges <= max_bytes
            if fits:
                self.build_wedge_slots()
            if self.wedge_slots is not None and mode >= 2 and fold_off is not None:
                # the edges into wide rows: folded lists and slots, so that the slots kernel steps those rows with
                # the instructions of every other row (the 32-bit lists stay for the other kernels)
                with torch.cuda.device(self.device):
                    rc = L.n2v_wedge_slots_fold(self.c_struct(), fold_off.data_ptr(), pos.data_ptr(),
                                                self.wedge_slots.data_ptr(), _lib.current_stream_ptr())
                _lib.check(rc, "n2v_wedge_slots_fold")
                self.slots_folded = True
        return self

    def build_wedge_slots(self) -> "DeviceGraph":
        """Wedge slots (n2v_wedge_slots_build): 32 bytes per edge at a place the walker knows a
        step ahead -- the return position and, for the four fifths of the lists that have at most
        14 entries, the list itself; offset + eight pivots for longer ones.  The exact biased
        kernel then fetches hop entry and list in two independent gathers instead of hop, offset
        and list (DESIGN.md "K2 exact, biased").  Needs the wedge table with 16-bit positions."""
        L = _lib.load()
        self.wedge_slots = None
        if self.wedge_off is None or self.wedge_pos is None or self.wedge_mode == 1:
            return self
        slots = torch.empty((self.n_edges, 16), dtype=torch.int16, device=self.device)
        with torch.cuda.device(self.device):
            rc = L.n2v_wedge_slots_build(self.c_struct(), slots.data_ptr(), _lib.current_stream_ptr())
        _lib.check(rc, "n2v_wedge_slots_build")
        self.wedge_slots = slots
        return self

    HOP_MAX_DEGREE = 1 << 24  # n2v_hop packs the degree into 24 bits

    def can_inline_rpos(self) -> bool:
        """the hop table may carry return positions in its class words (N2V_EC_INLINE): wedge table
        with 16-bit positions and wedge slots at hand, every return count below 128"""
        if self.edge_classes is None or self.wedge_off is None or self.wedge_slots is None:
            return False
        if self.wedge_pos is None or self.wedge_mode == 1:
            return False
        if self.wedge_mode >= 2 and not self.slots_folded:
            return False  # a mixed table without folded slots walks through wedge_off: the plain hop table
        if self._inline_ok is None or self._inline_ok[0] is not self.edge_classes:
            # one pass over the class words and a host sync: once per table, not per walk() call
            ok = self.n_edges == 0 or int((self.edge_classes >> 24 & 0xff).max()) < 128
            self._inline_ok = (self.edge_classes, ok)
        return self._inline_ok[1]

    def build_hops(self, with_classes: bool = False, inline_rpos: bool = False) -> "DeviceGraph":
        """Hop table (n2v_hops_build): per edge {neighbour id, class counts of the edge, row
        pointer and degree of the neighbour} in 16 bytes, so that a walk step is one gather.
        `with_classes` builds the per-edge class counts first (biased walks need them inside
        the table).  Unit-weight graphs with rows below 2^24 entries only; otherwise the graph
        is left without the table and the kernels walk the CSR arrays as before."""
        L = _lib.load()
        _lib.require_gpu()
        if not self.unit_weights:
            raise ValueError("the hop table exists for unit-weight graphs only")
        if not self.rowptr.is_cuda:
            raise RuntimeError("build_hops: graph is not on the GPU")
        if with_classes and self.edge_classes is None:
            self.build_edge_classes()
        if self.n_edges == 0 or self.n_edges >= (1 << 40) or int(self.degrees().max()) >= self.HOP_MAX_DEGREE:
            self.hops = None
            return self
        self.hops = None  # the kernel must not read a half-written table through c_struct()
        inline_rpos = bool(inline_rpos) and self.can_inline_rpos()
        hops = torch.empty((self.n_edges, 4), dtype=torch.int32, device=self.device)
        status = torch.zeros(4, dtype=torch.int32, device=self.device)
        cs = self.c_struct()
        cs.reserved2 = 1 if inline_rpos else 0  # N2V_HOPS_INLINE_RPOS
        with torch.cuda.device(self.device):
            rc = L.n2v_hops_build(cs, hops.data_ptr(), status.data_ptr(), _lib.current_stream_ptr())
        _lib.check(rc, "n2v_hops_build")
        if int(status[0].item()) & _lib.ST_RANGE:
            self.hops_inline_rpos = False
            return self
        self.hops = hops
        self.hops_inline_rpos = inline_rpos
        self.hops_have_classes = self.edge_classes is not None
        return self

    # Measured (profiles/r3x_time_hop8_*.log): with 0 - 11 % of the steps taking the escape the 8-byte
    # table walks cfg 3 at 51.5 - 51.8 G steps/s against 45.7 G for the 16-byte table; with 26 - 39 %
    # (cfg 4: 27-bit ids leave 7 - 9 bits for the degree) it LOSES, 37.4 - 38.3 G against 42.7 G --
    # the second, dependent lookup costs more than the narrower gather saves.
    HOP8_MAX_ESCAPE_SHARE = 0.12

    def build_hops8(self, force: bool = False, col_bits: Optional[int] = None,
                    row_bits: Optional[int] = None, align_shift: Optional[int] = None) -> "DeviceGraph":
        """The 8-byte hop table (n2v_hops8_build) for exact walks with p == q == 1: neighbour id,
        its row start and (a code for) its degree in ONE 8-byte gather -- the chip serves those a
        quarter faster than the 16-byte entries of build_hops.  Field widths follow the graph; a
        degree that does not fit its field is read from rowptr (the "escape"), which pays only
        while few steps take it (HOP8_MAX_ESCAPE_SHARE).  First choice: rows as in the CSR (cfg 3:
        24 + 28 + 12 bits); second: rows padded to multiples of 8 entries, which frees 3 bits of
        the row field for the degree (cfg 3: 24 + 25 + 15, no escape at all).  cfg 4's 27-bit ids
        leave 7 - 9 bits: a quarter to a third of the steps would escape, and the graph keeps the
        16-byte table."""
        L = _lib.load()
        _lib.require_gpu()
        self.hops8 = self.hops8_rowptr = None
        self.hops8_tried = True
        if not self.unit_weights or not self.rowptr.is_cuda or self.n_edges == 0:
            return self
        deg = self.degrees()
        cb = int(col_bits or max(1, int(self.n_vertices - 1).bit_length()))  # (tests pass wider fields:
        if cb > 31:                                                           #  more escapes)
            return self

        def escape_share(esc):
            # share of the walk steps that would take the escape (a walk on an undirected graph
            # visits a vertex in proportion to its degree)
            return float(deg[deg >= esc].sum()) / max(self.n_edges, 1)

        for shift in ((0, 3) if align_shift is None else (int(align_shift),)):
            if shift == 0:
                trow, entries = None, self.n_edges
            else:
                pad = (1 << shift) - 1
                trow = torch.zeros(self.n_vertices + 1, dtype=torch.int64, device=self.device)
                torch.cumsum((deg + pad) & ~pad, 0, out=trow[1:])
                entries = int(trow[-1])
            rb = int(row_bits or max(1, int(entries >> shift).bit_length()))
            if cb + rb > 62 or (entries >> shift) >= (1 << rb):
                continue
            esc = (1 << (64 - cb - rb)) - 1
            if not force and escape_share(esc) > self.HOP8_MAX_ESCAPE_SHARE:
                continue
            hops8 = torch.zeros(entries, dtype=torch.int64, device=self.device)
            with torch.cuda.device(self.device):
                rc = L.n2v_hops8_build(self.c_struct(), cb, rb, shift, 0 if trow is None else trow.data_ptr(),
                                       hops8.data_ptr(), _lib.current_stream_ptr())
            _lib.check(rc, "n2v_hops8_build")
            self.hops8, self.hops8_bits, self.hops8_rowptr, self.hops8_shift = hops8, (cb, rb), trow, shift
            return self
        return self

    # The degree-ranked form keeps its class table in LDS (8 bytes per class, 64 KB for each of the
    # two blocks of 1024 threads of a CU); should a graph have more distinct degrees, its top ranks
    # go to a small table in HBM that stays cached.
    RANK_MAX_CLASSES = 8191
    RANK_MAX_HEAD = 1 << 22

    def build_ranked(self) -> "DeviceGraph":
        """The degree-ranked form for p = q = 1 walks on unit weights (n2v_graph.rank_*,
        n2v_rank_hops_build): vertices numbered by descending degree (stable: ties by ascending
        id), rows laid out in rank order, an entry = the 4-byte rank of the neighbour.  The row of a
        rank follows from its degree class: offset(class) + (rank - first(class)) * degree(class).
        4 bytes per edge + 8 per vertex; declined (rank_hops stays None) when the graph is
        weighted, has 2^32 edges or more, a degree needs more than 24 bits, or more than
        RANK_MAX_HEAD top vertices would have to be listed one by one."""
        L = _lib.load()
        _lib.require_gpu()
        self.rank_tried = True
        self.rank_hops = None
        if not self.unit_weights or not self.rowptr.is_cuda:
            return self
        n, dev = self.n_vertices, self.device
        deg = self.degrees()
        if (self.n_edges == 0 or self.n_edges >= (1 << 32) or n >= (1 << 31)
                or int(deg.max()) >= self.HOP_MAX_DEGREE):
            return self
        tables = rank_tables(deg, self.RANK_MAX_CLASSES, self.RANK_MAX_HEAD)
        if tables is None:
            return self
        rank_vertex, rank_of, rank_rowptr, head, first, off = tables
        hops = torch.empty(self.n_edges, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.n2v_rank_hops_build(self.c_struct(), rank_of.data_ptr(), rank_vertex.data_ptr(),
                                             rank_rowptr.data_ptr(), hops.data_ptr(),
                                             _lib.current_stream_ptr()), "n2v_rank_hops_build")
        self.rank_of, self.rank_vertex, self.rank_head = rank_of, rank_vertex, head
        self.rank_class_first, self.rank_class_off = first, off
        self.rank_hops = hops
        return self

    def build_pivots(self) -> "DeviceGraph":
        """Block-end search index over `col` (n2v_pivots_build): fast-mode membership tests
        then touch 2-3 cache lines instead of log2(degree)."""
        L = _lib.load()
        _lib.require_gpu()
        pivots = torch.zeros(((self.n_edges + 31) // 32,), dtype=torch.int32, device=self.device)
        if self.n_edges:
            with torch.cuda.device(self.device):
                rc = L.n2v_pivots_build(self.col.data_ptr(), self.n_edges, pivots.data_ptr(),
                                        _lib.current_stream_ptr())
            _lib.check(rc, "n2v_pivots_build")
        self.pivots = pivots
        return self

    # -- views of the packed tables (tests, debugging) -------------------------------
    @property
    def alias(self) -> Optional[torch.Tensor]:
        """neighbour id behind each slot's alias index (row[alias[i]])"""
        return None if self.slots is None else self.slots[:, 1]

    @property
    def prob(self) -> Optional[torch.Tensor]:
        return None if self.slots is None else self.slots.view(torch.float64)[:, 1]
