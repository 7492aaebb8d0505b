// n2v_edge_classes.hip -- per-edge class counts for exact walks on unit-weight graphs.
//
// generate_edge_alias_tables(s, N_out(s), N(v), p, q) (reference randomwalk.py:193-232)
// gives neighbour x of v the weight 1/p if x == s (:223-224), 1 if x is an out-neighbour
// of s (:226-227) and 1/q otherwise (:229-230).  That table is a function of the edge
// (s -> v) alone, and with unit weights its row sum (:172) and every probs[i] before the
// pairing (:173) follow from two counts:
//     n_return = #{ j : N(v)[j] == s }
//     n_shared = #{ j : N(v)[j] != s and N(v)[j] in N_out(s) }       (multi-edges of v counted)
// The reference recomputes the set intersection behind them at every step.  A walk of
// W * L = 800 steps per vertex crosses each edge many times, so this pass computes the two
// counts once per edge and the exact walk kernel (n2v_walk_unit.hip, lanes kernel) reads
// them with one 4-byte gather.
//
// Work item = 64 consecutive edges, one per lane (the source row of an edge is found by
// binary search over rowptr).  An edge whose shorter list has at most kLaneMax ids is
// handled by its lane alone (iterate the shorter list, binary search in the longer one);
// the others are taken one at a time by the whole wave (the shorter list strided over the
// lanes, four searches in flight per lane).  Batches of edges are handed out through a
// counter (status[1]), because rows differ by four orders of magnitude in cost.
#include "n2v_alias_core.h"

namespace n2v {

constexpr int kLaneMax = 24;       // shorter list handled by one lane up to this length
constexpr int kEdgeBatch = 256;    // edges per counter grab (4 wave passes)

// number of entries equal to x in the sorted row a[0, m)
__device__ __forceinline__ int count_sorted(const int32_t *a, int m, int32_t x) {
  int lo = 0, hi = m;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < x)
      lo = mid + 1;
    else
      hi = mid;
  }
  int c = 0;
  while (lo + c < m && a[lo + c] == x) ++c;
  return c;
}

// the two counts of edge (s -> v) by one lane
__device__ __forceinline__ void classes_lane(const int32_t *scol, int ds, const int32_t *vcol,
                                             int dv, int32_t s, int &nR, int &nM) {
  nR = 0;
  nM = 0;
  if (dv <= ds) {  // walk N(v), look each id up in N(s)
    for (int j = 0; j < dv; ++j) {
      const int32_t x = vcol[j];
      if (x == s)
        ++nR;
      else
        nM += count_sorted(scol, ds, x) > 0 ? 1 : 0;
    }
  } else {  // walk the distinct ids of N(s), count their occurrences in N(v)
    nR = count_sorted(vcol, dv, s);
    int32_t prev = -1;
    for (int k = 0; k < ds; ++k) {
      const int32_t y = scol[k];
      if (y != s && y != prev) nM += count_sorted(vcol, dv, y);
      prev = y;
    }
  }
}

// the same by the whole wave (all arguments wave-uniform); result in every lane
__device__ __forceinline__ void classes_wave(const int32_t *scol, int ds, const int32_t *vcol,
                                             int dv, int32_t s, int lane, int &nR, int &nM) {
  int r = 0, m = 0;
  if (dv <= ds) {
    const int iters = 32 - __clz(ds);
    for (int base = 0; base < dv; base += 256) {
      int32_t x[4];
      bool found[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = base + u * 64 + lane;
        x[u] = j < dv ? vcol[j] : -1;
      }
      member_sorted_x4(scol, ds, x, iters, found);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool act = base + u * 64 + lane < dv;
        r += (act && x[u] == s) ? 1 : 0;
        m += (act && x[u] != s && found[u]) ? 1 : 0;
      }
    }
  } else {
    const int iters = 32 - __clz(dv);
    if (lane == 0) r = count_sorted(vcol, dv, s);
    for (int base = 0; base < ds; base += 256) {
      int32_t y[4];
      int lo[4];
      bool found[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = base + u * 64 + lane;
        const int32_t yy = k < ds ? scol[k] : -1;
        const int32_t prev = (k >= 1 && k < ds) ? scol[k - 1] : -1;
        y[u] = (yy == s || yy == prev) ? -1 : yy;  // the return slot / a repeated id: skip
      }
      lower_bound_x4(vcol, dv, y, iters, lo, found);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (found[u] && y[u] >= 0) {
          int c = 1;
          while (lo[u] + c < dv && vcol[lo[u] + c] == y[u]) ++c;  // multi-edges of v
          m += c;
        }
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    r += __shfl_xor(r, off, 64);
    m += __shfl_xor(m, off, 64);
  }
  nR = __builtin_amdgcn_readfirstlane(r);
  nM = __builtin_amdgcn_readfirstlane(m);
}

__global__ __launch_bounds__(256) void edge_classes_kernel(n2v_graph g,
                                                          uint32_t *__restrict__ classes,
                                                          uint32_t *__restrict__ counter) {
  const int lane = threadIdx.x & 63;
  const int64_t n_edges = g.n_edges;
  const int64_t n_batches = (n_edges + kEdgeBatch - 1) / kEdgeBatch;
  for (;;) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(counter, 1u);
    const int64_t batch = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    if (batch >= n_batches) break;
    for (int sub = 0; sub < kEdgeBatch / 64; ++sub) {
      const int64_t e = batch * kEdgeBatch + sub * 64 + lane;
      const bool act = e < n_edges;
      int32_t s = 0, v = 0;
      int64_t sb = 0, vb = 0;
      int ds = 0, dv = 0;
      if (act) {
        // source row of edge e: the last row with rowptr[row] <= e
        int64_t lo = 0, hi = g.n_vertices;  // answer in [lo, hi)
        while (hi - lo > 1) {
          const int64_t mid = (lo + hi) >> 1;
          if (g.rowptr[mid] <= e)
            lo = mid;
          else
            hi = mid;
        }
        s = (int32_t)lo;
        sb = g.rowptr[s];
        ds = (int)(g.rowptr[s + 1] - sb);
        v = g.col[e];
        vb = g.rowptr[v];
        dv = (int)(g.rowptr[v + 1] - vb);
      }
      int nR = 0, nM = 0;
      const bool small = act && min(ds, dv) <= kLaneMax;
      if (small) classes_lane(g.col + sb, ds, g.col + vb, dv, s, nR, nM);
      uint64_t big = ballot64(act && !small);
      while (big != 0ull) {
        const int l = (int)__builtin_ctzll(big);
        big &= big - 1ull;
        const int64_t sb_l = readfirstlane_i64(__shfl(sb, l, 64));
        const int64_t vb_l = readfirstlane_i64(__shfl(vb, l, 64));
        const int ds_l = __builtin_amdgcn_readlane(ds, l);
        const int dv_l = __builtin_amdgcn_readlane(dv, l);
        const int32_t s_l = __builtin_amdgcn_readlane(s, l);
        int r = 0, m = 0;
        classes_wave(g.col + sb_l, ds_l, g.col + vb_l, dv_l, s_l, lane, r, m);
        if (lane == l) {
          nR = r;
          nM = m;
        }
      }
      if (act) {
        const uint32_t fR = nR >= (int)N2V_EC_RETURN_SAT ? N2V_EC_RETURN_SAT : (uint32_t)nR;
        const uint32_t fM = nM >= (int)N2V_EC_SHARED_MASK ? N2V_EC_SHARED_MASK : (uint32_t)nM;
        classes[e] = (fR << N2V_EC_RETURN_SHIFT) | fM;
      }
    }
  }
}

}  // namespace n2v

extern "C" int n2v_edge_classes_build(const n2v_graph *g, uint32_t *classes_out,
                                      uint32_t *status, void *stream) {
  if (!g || !g->rowptr || g->n_vertices < 0 || g->n_edges < 0) return N2V_EINVAL;
  // (weights play no part: which slots are return / shared / other depends on the ids alone)
  if (g->n_edges == 0) return N2V_OK;
  if (!g->col || !classes_out || !status) return N2V_EINVAL;
  const int64_t n_batches = (g->n_edges + n2v::kEdgeBatch - 1) / n2v::kEdgeBatch;
  if (n_batches >= 0xffff0000ll) return N2V_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  // status[1] is the kernel's batch counter: start it at zero on the same stream
  if (hipMemsetAsync(status + 1, 0, sizeof(uint32_t), st) != hipSuccess) return N2V_ELAUNCH;
  int64_t blocks = (n_batches + 3) / 4;
  const int64_t cap = n2v::resident_blocks((const void *)n2v::edge_classes_kernel, 256, 0);
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(n2v::edge_classes_kernel, dim3((unsigned)blocks), dim3(256), 0, st, *g,
                     classes_out, status + 1);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}
