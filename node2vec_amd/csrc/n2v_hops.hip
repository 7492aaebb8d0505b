// n2v_hops.hip -- the hop table of a unit-weight graph (struct n2v_hop, include/n2v_hip.h).
//
// The reference advances a walker by two joins per step (fugue.py:146-148): the walker row is
// joined with the adjacency row of its current vertex and with that of its previous vertex.
// On the GPU each of these is a dependent random gather, and the walk kernels are bound by
// the chip's rate of random 64-byte sector reads (~50 G/s, profiles/r02_gather_ceiling.log),
// not by bytes: a step of the p == q == 1 kernel was {col[row + pick]} then {rowptr[x],
// rowptr[x + 1]} -- two sectors.  The hop table stores, beside every neighbour id, the row
// pointer and degree of that neighbour and the class counts of the edge (edge_classes[e]), so
// the entry that names the next vertex already says where its row starts, how long it is and
// what the per-step table of the NEXT step looks like: one sector per step.  16 bytes per
// edge; built in one streaming pass (one random 16-byte gather of rowptr per edge).
#include "n2v_common.h"

namespace n2v {

__global__ __launch_bounds__(256) void hops_build_kernel(n2v_graph g, n2v_hop *__restrict__ out,
                                                         uint32_t *__restrict__ overflow) {
  bool bad = false;
  const bool inline_rpos = g.edge_classes && g.wedge_off && (g.reserved2 & N2V_HOPS_INLINE_RPOS);
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < g.n_edges;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int32_t x = g.col[e];
    const int64_t b = g.rowptr[x];
    const int64_t d = g.rowptr[x + 1] - b;
    bad = bad || d >= (1ll << (64 - N2V_HOP_DEG_SHIFT));
    const uint64_t row = (uint64_t)b | ((uint64_t)d << N2V_HOP_DEG_SHIFT);
    uint32_t cls = g.edge_classes ? g.edge_classes[e] : 0xffffffffu;
    if (inline_rpos && cls != 0xffffffffu) {
      // an edge without shared neighbours carries its return position instead of the zero count
      const uint32_t fR = cls >> N2V_EC_RETURN_SHIFT;
      bad = bad || fR >= 0x80u;  // (the caller checked: every return count is below 128)
      // (an edge into a wide row of a mixed wedge table keeps its plain class word: its return
      // position needs more than the 16 bits the slots kernel reads back)
      if ((cls & N2V_EC_SHARED_MASK) == 0u && !(g.wedge_wide >= 2 && d >= (int64_t)g.wedge_wide)) {
        const uint64_t rpos = g.wedge_off[e] >> N2V_WEDGE_RPOS_SHIFT;
        cls = N2V_EC_INLINE | ((fR & 0x7fu) << N2V_EC_RETURN_SHIFT) | (uint32_t)(rpos & N2V_EC_SHARED_MASK);
      }
    }
    int4 v;
    v.x = x;
    v.y = (int)cls;
    v.z = (int)(uint32_t)row;
    v.w = (int)(uint32_t)(row >> 32);
    *reinterpret_cast<int4 *>(out + e) = v;
  }
  if (bad) atomicOr(overflow, N2V_ST_RANGE);
}

// the 8-byte form for the p == q == 1 kernel (include/n2v_hip.h, n2v_hops8_build)
__global__ __launch_bounds__(256) void hops8_build_kernel(n2v_graph g, int col_bits, int row_bits,
                                                          int align_shift,
                                                          const int64_t *__restrict__ trow,
                                                          uint64_t *__restrict__ out) {
  const uint64_t esc = (1ull << (64 - col_bits - row_bits)) - 1ull;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < g.n_edges;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int32_t x = g.col[e];
    const int64_t b = g.rowptr[x];
    const uint64_t d = (uint64_t)(g.rowptr[x + 1] - b);
    int64_t at = e;
    uint64_t start = (uint64_t)b;
    if (trow) {  // padded rows: the source vertex of edge e, then its slot in the padded table
      int64_t lo = 0, hi = g.n_vertices;  // last v with rowptr[v] <= e
      while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (g.rowptr[mid] <= e)
          lo = mid;
        else
          hi = mid;
      }
      at = trow[lo] + (e - g.rowptr[lo]);
      start = (uint64_t)trow[x];
    }
    out[at] = (uint64_t)(uint32_t)x | ((start >> align_shift) << col_bits) |
              ((d < esc ? d : esc) << (col_bits + row_bits));
  }
}

}  // namespace n2v

extern "C" int n2v_hops8_build(const n2v_graph *g, int32_t col_bits, int32_t row_bits,
                               int32_t align_shift, const int64_t *hop8_rowptr,
                               uint64_t *hops8_out, void *stream) {
  if (!g || !g->rowptr || g->n_vertices < 0 || g->n_edges < 0) return N2V_EINVAL;
  if (g->w || g->w64) return N2V_EINVAL;  // unit-weight graphs only
  if (col_bits < 1 || row_bits < 1 || col_bits > 31 || col_bits + row_bits > 62) return N2V_EINVAL;
  if (align_shift < 0 || align_shift > 6 || (align_shift > 0 && !hop8_rowptr)) return N2V_EINVAL;
  if (g->n_vertices > (1ll << col_bits)) return N2V_EINVAL;
  if (!hop8_rowptr && g->n_edges >= (1ll << row_bits)) return N2V_EINVAL;  // (padded: the caller
  if (g->n_edges == 0) return N2V_OK;                                       //  checked its own size)
  if (!g->col || !hops8_out) return N2V_EINVAL;
  int64_t blocks = (g->n_edges + 255) / 256;
  const int64_t cap = n2v::resident_blocks((const void *)n2v::hops8_build_kernel, 256, 0) * 2;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(n2v::hops8_build_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, *g, col_bits, row_bits, align_shift, hop8_rowptr, hops8_out);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}

extern "C" int n2v_hops_build(const n2v_graph *g, n2v_hop *hops_out, uint32_t *status,
                              void *stream) {
  if (!g || !g->rowptr || g->n_vertices < 0 || g->n_edges < 0) return N2V_EINVAL;
  if (g->w || g->w64) return N2V_EINVAL;  // unit-weight graphs only
  if (g->n_edges >= (1ll << N2V_HOP_DEG_SHIFT)) return N2V_EINVAL;
  if (g->n_edges == 0) return N2V_OK;
  if (!g->col || !hops_out || !status) return N2V_EINVAL;
  // a degree of 2^24 or more cannot be packed: the kernel then sets N2V_ST_RANGE in status[0]
  // and the caller must not use the table
  int64_t blocks = (g->n_edges + 255) / 256;
  const int64_t cap = n2v::resident_blocks((const void *)n2v::hops_build_kernel, 256, 0) * 2;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(n2v::hops_build_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, *g, hops_out, status);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}

namespace n2v {
// One wave per 64 consecutive ranks (neighbouring ranks have neighbouring degrees): rows of 64
// entries or more are copied by the whole wave one after the other, shorter ones lane per row.
__global__ __launch_bounds__(256) void rank_hops_build_kernel(n2v_graph g, const int32_t *__restrict__ rank_of,
                                                              const int32_t *__restrict__ rank_vertex,
                                                              const int64_t *__restrict__ rank_rowptr,
                                                              uint32_t *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int64_t groups = (g.n_vertices + 63) >> 6;
  for (int64_t grp = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); grp < groups; grp += n_waves) {
    const int64_t r = grp * 64 + lane;
    int64_t src = 0, dst = 0;
    int d = 0;
    if (r < g.n_vertices) {
      const int32_t v = rank_vertex[r];
      src = g.rowptr[v];
      d = (int)(g.rowptr[v + 1] - src);
      dst = rank_rowptr[r];
    }
    uint64_t big = ballot64(d >= 64);
    while (big) {
      const int l = __builtin_ctzll(big);
      big &= big - 1;
      const int64_t s_src = __shfl(src, l, 64), s_dst = __shfl(dst, l, 64);
      const int s_d = __shfl(d, l, 64);
      for (int k = lane; k < s_d; k += 64) out[s_dst + k] = (uint32_t)rank_of[g.col[s_src + k]];
    }
    if (d < 64)
      for (int k = 0; k < d; ++k) out[dst + k] = (uint32_t)rank_of[g.col[src + k]];
  }
}
}  // namespace n2v

extern "C" int n2v_rank_hops_build(const n2v_graph *g, const int32_t *rank_of, const int32_t *rank_vertex,
                                   const int64_t *rank_rowptr, uint32_t *out, void *stream) {
  if (!g || !g->rowptr || g->n_vertices < 0 || g->n_edges < 0) return N2V_EINVAL;
  if (g->w || g->w64) return N2V_EINVAL;  // unit-weight graphs only
  if (g->n_edges >= (1ll << N2V_HOP_DEG_SHIFT) || g->n_vertices >= (1ll << 31)) return N2V_EINVAL;
  if (g->n_edges == 0) return N2V_OK;
  if (!g->col || !rank_of || !rank_vertex || !rank_rowptr || !out) return N2V_EINVAL;
  int64_t blocks = ((g->n_vertices + 63) / 64 + 3) / 4;
  const int64_t cap = n2v::resident_blocks((const void *)n2v::rank_hops_build_kernel, 256, 0) * 4;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(n2v::rank_hops_build_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     *g, rank_of, rank_vertex, rank_rowptr, out);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}
