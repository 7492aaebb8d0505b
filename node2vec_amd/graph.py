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


class DeviceGraph:
    def __init__(self, rowptr: torch.Tensor, col: torch.Tensor, w: torch.Tensor):
        if rowptr.dtype != torch.int64 or col.dtype != torch.int32 or w.dtype != torch.float32:
            raise TypeError("DeviceGraph wants rowptr int64, col int32, w float32")
        if col.numel() != w.numel() or rowptr.numel() < 1:
            raise ValueError("DeviceGraph: ragged CSR arrays")
        self.rowptr = rowptr.contiguous()
        self.col = col.contiguous()
        self.w = w.contiguous()
        self.slots: Optional[torch.Tensor] = None  # int32 [E, 4] view of n2v_slot[E]
        self.pivots: Optional[torch.Tensor] = None  # int32 [(E + 31) / 32] search index (fast mode)
        # unweighted graph (index_graph_* gives weight 1.0, indexer.py:20-21): the walk
        # kernels are told through w == NULL and never read the weights
        self.unit_weights = bool((self.w == 1.0).all()) if self.w.numel() else True

    # -- construction ---------------------------------------------------------
    @classmethod
    def from_edges(cls, src, dst, weight, n_vertices: Optional[int] = None, device=None):
        """Stable sort by (src, dst) == partition(by=src, presort=dst) (fugue.py:130)."""
        src = torch.as_tensor(src).to(device=device, dtype=torch.int64).reshape(-1)
        dst = torch.as_tensor(dst).to(device=device, dtype=torch.int64).reshape(-1)
        w = torch.as_tensor(weight).to(device=device, dtype=torch.float32).reshape(-1)
        if not (src.numel() == dst.numel() == w.numel()):
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
        order = torch.sort(key, stable=True).indices
        src, dst, w = src[order], dst[order], w[order]
        counts = torch.bincount(src, minlength=n_vertices)
        rowptr = torch.zeros(n_vertices + 1, dtype=torch.int64, device=src.device)
        torch.cumsum(counts, 0, out=rowptr[1:])
        return cls(rowptr, dst.to(torch.int32), w)

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
        g = DeviceGraph(self.rowptr.to(device), self.col.to(device), self.w.to(device))
        if self.slots is not None:
            g.slots = self.slots.to(device)
        if self.pivots is not None:
            g.pivots = self.pivots.to(device)
        return g

    def c_struct(self) -> _lib.Graph:
        return _lib.Graph(self.n_vertices, self.n_edges, self.rowptr.data_ptr(),
                          self.col.data_ptr(), 0 if self.unit_weights else self.w.data_ptr(),
                          0 if self.slots is None else self.slots.data_ptr(),
                          0 if self.pivots is None else self.pivots.data_ptr())

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
            rc = L.n2v_alias_build(self.rowptr.data_ptr(), self.col.data_ptr(),
                                   self.w.data_ptr(), self.n_vertices, slots.data_ptr(),
                                   status.data_ptr(), _lib.current_stream_ptr())
        _lib.check(rc, "n2v_alias_build")
        _lib.check_status_word(int(status[0].item()), "n2v_alias_build")
        self.slots = slots
        return self.build_pivots()

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
