// n2v_wedge.hip -- per-edge shared-position lists ("wedge table") for exact walks on
// unit-weight graphs.
//
// generate_edge_alias_tables(s, N_out(s), N(v), p, q) (reference randomwalk.py:193-232) gives
// slot j of the table of step (s -> v) one of three values according to N(v)[j]: == s
// (:223-224), in N_out(s) (:226-227), neither (:229-230).  n2v_edge_classes_build already
// stores HOW MANY slots of each kind the table of every edge has; that decides ~80 % of the
// steps (n2v_walk_unit.hip, lanes kernel).  The other steps run the pairing loop (:182-189)
// and need to know WHICH slots are which -- the reference recomputes the set intersection at
// every step, and on the GPU that recomputation (searches over N(s), streaming N(v)) was
// 58 % of those steps' time and 94 % of the kernel's.  This pass stores the answer once per
// edge:
//     wedge_pos[off(e) .. off(e) + n_shared(e))  the positions j, ascending, with
//                                                N(v)[j] in N_out(s) and N(v)[j] != s
//     wedge_off[e] = off(e) | rpos(e) << 40       rpos = position of the first N(v)[j] == s
// (n_shared(e) and the return count are edge_classes[e]; off = their exclusive prefix sum,
// computed by the caller).  The lists hold one entry per (edge, common neighbour) pair -- six
// per triangle of the graph; 1.2 x 10^10 entries at BASELINE cfg 4 (23 GB as uint16).
// Same traversal as n2v_edge_classes.hip (the shorter list searched in the longer one, one
// lane per short edge, the wave for long ones), with ordered emission.
#include "n2v_alias_core.h"

namespace n2v {

constexpr int kWLaneMax = 24;     // shorter list handled by one lane up to this length
constexpr int kWEdgeBatch = 256;  // edges per counter grab

template <typename P>
__device__ __forceinline__ int lower_bound_1(const int32_t *a, int m, int32_t x) {
  int lo = 0, hi = m;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < x)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

// one lane: positions into out[0, cap), returns how many were written (never beyond cap)
template <typename P>
__device__ __forceinline__ int wedge_lane(const int32_t *scol, int ds, const int32_t *vcol, int dv,
                                          int32_t s, P *out, int cap, int &rpos) {
  int k = 0;
  rpos = 0;
  if (dv <= ds) {  // walk N(v), look each id up in N(s)
    bool seen = false;
    for (int j = 0; j < dv; ++j) {
      const int32_t x = vcol[j];
      if (x == s) {
        if (!seen) rpos = j;
        seen = true;
      } else if (member_sorted_lane(scol, ds, x)) {
        if (k < cap) out[k] = (P)j;
        ++k;
      }
    }
  } else {  // walk the distinct ids of N(s), locate their occurrences in N(v)
    rpos = lower_bound_1<P>(vcol, dv, s);
    if (rpos >= dv) rpos = 0;
    int32_t prev = -1;
    for (int i = 0; i < ds; ++i) {
      const int32_t y = scol[i];
      if (y != s && y != prev) {
        int lo = lower_bound_1<P>(vcol, dv, y);
        while (lo < dv && vcol[lo] == y) {
          if (k < cap) out[k] = (P)lo;
          ++k;
          ++lo;
        }
      }
      prev = y;
    }
  }
  return k;
}

// exclusive prefix sum of a small per-lane count over the wave, and the total
__device__ __forceinline__ int wave_excl_scan(int v, int lane, int &total) {
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  total = __builtin_amdgcn_readlane(incl, 63);
  return incl - v;
}

// the whole wave (all arguments wave-uniform)
template <typename P>
__device__ __forceinline__ int wedge_wave(const int32_t *scol, int ds, const int32_t *vcol, int dv,
                                          int32_t s, int lane, P *out, int cap, int &rpos) {
  int cnt = 0;
  int rp = 0x7fffffff;
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
      for (int u = 0; u < 4; ++u) {  // sub-chunks in index order: ordered emission
        const int j = base + u * 64 + lane;
        const bool act = j < dv;
        if (act && x[u] == s) rp = min(rp, j);
        const bool hit = act && x[u] != s && found[u];
        const uint64_t mask = ballot64(hit);
        const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
        if (hit && pos < cap) out[pos] = (P)j;
        cnt += __popcll(mask);
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) rp = min(rp, __shfl_xor(rp, off, 64));
  } else {
    const int iters = 32 - __clz(dv);
    if (lane == 0) rp = lower_bound_1<P>(vcol, dv, s);
    rp = __builtin_amdgcn_readfirstlane(rp);
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
        int c = 0;
        if (found[u] && y[u] >= 0) {
          c = 1;
          while (lo[u] + c < dv && vcol[lo[u] + c] == y[u]) ++c;  // multi-edges of v
        }
        int total;
        const int at = cnt + wave_excl_scan(c, lane, total);
        for (int t = 0; t < c; ++t)
          if (at + t < cap) out[at + t] = (P)(lo[u] + t);
        cnt += total;
      }
    }
  }
  rpos = (rp == 0x7fffffff || rp >= dv) ? 0 : __builtin_amdgcn_readfirstlane(rp);
  return __builtin_amdgcn_readfirstlane(cnt);
}

// kWide 0: uint16 lists, 1: uint32 lists, 2: mixed -- uint32 for the edges into rows of `wide_from`
// entries or more (offsets in units of the list's own width from the one base)
template <int kWide>
__global__ __launch_bounds__(256) void wedge_fill_kernel(n2v_graph g,
                                                        const uint64_t *list_off,  // may alias wedge_off (in-place use is allowed)
                                                        uint64_t *wedge_off,
                                                        void *__restrict__ wedge_pos_v,
                                                        int wide_from,
                                                        uint32_t *__restrict__ status,
                                                        uint32_t *__restrict__ counter) {
  uint16_t *pos16 = reinterpret_cast<uint16_t *>(wedge_pos_v);
  uint32_t *pos32 = reinterpret_cast<uint32_t *>(wedge_pos_v);
  const int lane = threadIdx.x & 63;
  const int64_t n_edges = g.n_edges;
  const int64_t n_batches = (n_edges + kWEdgeBatch - 1) / kWEdgeBatch;
  for (;;) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(counter, 1u);
    const int64_t batch = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    if (batch >= n_batches) break;
    for (int sub = 0; sub < kWEdgeBatch / 64; ++sub) {
      const int64_t e = batch * kWEdgeBatch + sub * 64 + lane;
      const bool act = e < n_edges;
      int32_t s = 0, v = 0;
      int64_t sb = 0, vb = 0;
      int ds = 0, dv = 0, want = 0;
      uint64_t off = 0;
      if (act) {
        int64_t lo = 0, hi = g.n_vertices;  // source row of edge e: last row with rowptr[row] <= e
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
        off = list_off[e];
        want = (int)(g.edge_classes[e] & N2V_EC_SHARED_MASK);
      }
      int got = 0, rpos = 0;
      const bool small = act && min(ds, dv) <= kWLaneMax;
      const bool wide_e = kWide == 1 || (kWide == 2 && dv >= wide_from);
      if (small) {
        if (wide_e)
          got = wedge_lane<uint32_t>(g.col + sb, ds, g.col + vb, dv, s, pos32 + off, want, rpos);
        else
          got = wedge_lane<uint16_t>(g.col + sb, ds, g.col + vb, dv, s, pos16 + off, want, rpos);
      }
      uint64_t big = ballot64(act && !small);
      while (big != 0ull) {
        const int l = (int)__builtin_ctzll(big);
        big &= big - 1ull;
        const int64_t sb_l = readfirstlane_i64(__shfl(sb, l, 64));
        const int64_t vb_l = readfirstlane_i64(__shfl(vb, l, 64));
        const int64_t off_l = readfirstlane_i64(__shfl((int64_t)off, l, 64));
        const int ds_l = __builtin_amdgcn_readlane(ds, l);
        const int dv_l = __builtin_amdgcn_readlane(dv, l);
        const int want_l = __builtin_amdgcn_readlane(want, l);
        const int32_t s_l = __builtin_amdgcn_readlane(s, l);
        int rp = 0;
        int k;
        if (kWide == 1 || (kWide == 2 && dv_l >= wide_from))  // (wave-uniform)
          k = wedge_wave<uint32_t>(g.col + sb_l, ds_l, g.col + vb_l, dv_l, s_l, lane, pos32 + off_l, want_l, rp);
        else
          k = wedge_wave<uint16_t>(g.col + sb_l, ds_l, g.col + vb_l, dv_l, s_l, lane, pos16 + off_l, want_l, rp);
        if (lane == l) {
          got = k;
          rpos = rp;
        }
      }
      if (act) {
        // the list must have exactly the length edge_classes promised (the prefix sum was
        // taken over those counts); anything else means the two tables disagree
        if (got != want) atomicOr(status, N2V_ST_RANGE);
        wedge_off[e] = off | ((uint64_t)(uint32_t)rpos << N2V_WEDGE_RPOS_SHIFT);
      }
    }
  }
}

}  // namespace n2v

extern "C" int n2v_wedge_build(const n2v_graph *g, const uint64_t *list_off, uint64_t *wedge_off_out,
                               void *wedge_pos_out, int32_t wide, uint32_t *status, void *stream) {
  if (!g || !g->rowptr || g->n_vertices < 0 || g->n_edges < 0) return N2V_EINVAL;
  // (weights play no part: which slots are return / shared / other depends on the ids alone)
  if (g->n_edges == 0) return N2V_OK;
  if (!g->col || !g->edge_classes || !list_off || !wedge_off_out || !wedge_pos_out || !status)
    return N2V_EINVAL;
  const int64_t n_batches = (g->n_edges + n2v::kWEdgeBatch - 1) / n2v::kWEdgeBatch;
  if (n_batches >= 0xffff0000ll) return N2V_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(status + 1, 0, sizeof(uint32_t), st) != hipSuccess) return N2V_ELAUNCH;
  int64_t blocks = (n_batches + 3) / 4;
  if (wide < 0 || wide > 65536) return N2V_EINVAL;
  if (wide >= 2 && (reinterpret_cast<uintptr_t>(wedge_pos_out) & 3u) != 0) return N2V_EINVAL;
  auto fn = wide == 0 ? n2v::wedge_fill_kernel<0> : wide == 1 ? n2v::wedge_fill_kernel<1> : n2v::wedge_fill_kernel<2>;
  const int64_t cap = n2v::resident_blocks((const void *)fn, 256, 0);
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(256), 0, st, *g, list_off, wedge_off_out, wedge_pos_out,
                     (int)wide, status, status + 1);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}

namespace n2v {
// one thread per edge: copy a short list into its slot, or leave its offset and eight pivots
__global__ __launch_bounds__(256) void wedge_slots_kernel(n2v_graph g, uint16_t *__restrict__ slots) {
  const uint16_t *pos = reinterpret_cast<const uint16_t *>(g.wedge_pos);
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < g.n_edges;
       e += (int64_t)gridDim.x * blockDim.x) {
    if (g.wedge_wide >= 2) {  // mixed: an edge into a wide row has a uint32 list and no slot
      const int32_t x = g.col[e];
      if (g.rowptr[x + 1] - g.rowptr[x] >= (int64_t)g.wedge_wide) {
        reinterpret_cast<int4 *>(slots + e * 16)[0] = make_int4(0, 0, 0, 0);
        reinterpret_cast<int4 *>(slots + e * 16)[1] = make_int4(0, 0, 0, 0);
        continue;
      }
    }
    const uint32_t ec = g.edge_classes[e];
    const uint64_t wraw = g.wedge_off[e];
    const uint64_t off = wraw & N2V_WEDGE_OFF_MASK;
    const int rpos = (int)(wraw >> N2V_WEDGE_RPOS_SHIFT);
    int nM = (int)(ec & N2V_EC_SHARED_MASK);
    if ((uint32_t)nM == N2V_EC_SHARED_MASK) nM = 0;  // saturated: the walk kernels flag it
    const uint16_t *list = pos + off;
    uint32_t hw[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) hw[k] = 0;
    hw[0] = (uint32_t)rpos & 0xffffu;
    int below = 0;  // list entries below the return position
    if (nM <= 14) {
      for (int k = 0; k < nM; ++k) {
        const uint32_t v = list[k];
        below += (int)v < rpos ? 1 : 0;
#pragma unroll
        for (int u = 0; u < 14; ++u)
          if (u == k) hw[2 + u] = v;
      }
    } else {
      int lo = 0, hi = nM;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((int)list[mid] < rpos)
          lo = mid + 1;
        else
          hi = mid;
      }
      below = lo;
      hw[4] = (uint32_t)(off & 0xffffu);
      hw[5] = (uint32_t)((off >> 16) & 0xffffu);
      hw[6] = (uint32_t)((off >> 32) & 0xffffu);
      hw[7] = (uint32_t)((off >> 48) & 0xffffu);
#pragma unroll
      for (int k = 0; k < 8; ++k) hw[8 + k] = list[((int64_t)(k + 1) * nM) / 9];
    }
    hw[1] = (uint32_t)below & 0xffffu;
    int4 a, b;
    a.x = (int)(hw[0] | (hw[1] << 16));
    a.y = (int)(hw[2] | (hw[3] << 16));
    a.z = (int)(hw[4] | (hw[5] << 16));
    a.w = (int)(hw[6] | (hw[7] << 16));
    b.x = (int)(hw[8] | (hw[9] << 16));
    b.y = (int)(hw[10] | (hw[11] << 16));
    b.z = (int)(hw[12] | (hw[13] << 16));
    b.w = (int)(hw[14] | (hw[15] << 16));
    reinterpret_cast<int4 *>(slots + e * 16)[0] = a;
    reinterpret_cast<int4 *>(slots + e * 16)[1] = b;
  }
}
}  // namespace n2v

extern "C" int n2v_wedge_slots_build(const n2v_graph *g, uint16_t *slots_out, void *stream) {
  if (!g || g->n_edges < 0) return N2V_EINVAL;
  if (g->n_edges == 0) return N2V_OK;
  if (!g->edge_classes || !g->wedge_off || !g->wedge_pos || g->wedge_wide == 1 || !slots_out)
    return N2V_EINVAL;
  if (g->wedge_wide >= 2 && (!g->rowptr || !g->col || g->wedge_wide > 65536)) return N2V_EINVAL;
  if ((reinterpret_cast<uintptr_t>(slots_out) & 31u) != 0) return N2V_EINVAL;
  int64_t blocks = (g->n_edges + 255) / 256;
  const int64_t cap = 8 * n2v::resident_blocks((const void *)n2v::wedge_slots_kernel, 256, 0);
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(n2v::wedge_slots_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, *g, slots_out);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}

// ---- folded lists and slots for the edges into wide rows (include/n2v_hip.h, n2v_wedge_slots_fold) ----
namespace n2v {
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(256) void wedge_slots_fold_kernel(n2v_graph g, const uint64_t *__restrict__ fold_off,
                                                               uint16_t *__restrict__ pos16,
                                                               uint16_t *__restrict__ slots) {
  const int T = g.wedge_wide;
  const int lane = threadIdx.x & 63;
  const uint32_t *pos32 = reinterpret_cast<const uint32_t *>(g.wedge_pos);
  const int64_t n_chunks = (g.n_edges + 63) / 64;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
  auto fold = [&](int v) -> uint32_t { return (uint32_t)(v < T ? v : v - T) & 0xffffu; };
  for (int64_t c = wave; c < n_chunks; c += n_waves) {
    const int64_t e = c * 64 + lane;
    bool wide = false;
    int nM = 0, rpos = 0;
    int64_t off32 = 0;
    if (e < g.n_edges) {
      const int64_t x = g.col[e];
      const int64_t d = g.rowptr[x + 1] - g.rowptr[x];
      wide = d >= (int64_t)T && d - (int64_t)T <= 65536;
    }
    if (wide) {
      const uint32_t ec = g.edge_classes[e];
      const uint64_t wraw = g.wedge_off[e];
      off32 = (int64_t)(wraw & N2V_WEDGE_OFF_MASK);
      rpos = (int)(wraw >> N2V_WEDGE_RPOS_SHIFT);
      nM = (int)(ec & N2V_EC_SHARED_MASK);
      if ((uint32_t)nM == N2V_EC_SHARED_MASK) nM = 0;  // saturated: the walk kernels flag it
    }
    const uint32_t upper = rpos >= T ? 1u : 0u;
    if (wide && nM <= 14) {  // the list lives in the slot: one lane
      const uint32_t *list = pos32 + off32;
      uint32_t hw[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) hw[k] = 0;
      int below = 0, nlow = 0;
      for (int k = 0; k < nM; ++k) {
        const int v = (int)list[k];
        below += v < rpos ? 1 : 0;
        nlow += v < T ? 1 : 0;
#pragma unroll
        for (int u = 0; u < 14; ++u)
          if (u == k) hw[2 + u] = fold(v);
      }
      hw[0] = fold(rpos);
      hw[1] = (uint32_t)below | ((uint32_t)nlow << 4) | (upper << 8);
      int4 a, b;
      a.x = (int)(hw[0] | (hw[1] << 16));
      a.y = (int)(hw[2] | (hw[3] << 16));
      a.z = (int)(hw[4] | (hw[5] << 16));
      a.w = (int)(hw[6] | (hw[7] << 16));
      b.x = (int)(hw[8] | (hw[9] << 16));
      b.y = (int)(hw[10] | (hw[11] << 16));
      b.z = (int)(hw[12] | (hw[13] << 16));
      b.w = (int)(hw[14] | (hw[15] << 16));
      reinterpret_cast<int4 *>(slots + e * 16)[0] = a;
      reinterpret_cast<int4 *>(slots + e * 16)[1] = b;
    }
    // a longer list: the whole wave folds it into its place among the 16-bit lists
    uint64_t todo = __ballot(wide && nM > 14);
    while (todo != 0ull) {
      const int src = __ffsll((unsigned long long)todo) - 1;
      todo &= todo - 1ull;
      const int64_t e1 = c * 64 + src;
      const int n1 = __builtin_amdgcn_readlane(nM, src);
      const int r1 = __builtin_amdgcn_readlane(rpos, src);
      const int64_t o32 = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)off32 >> 32), src) << 32) |
                                    (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)off32, src));
      const uint64_t o16 = fold_off[e1];
      const uint32_t *list = pos32 + o32;
      uint16_t *out = pos16 + o16;
      int below = 0, nlow = 0;
      for (int k = lane; k < n1; k += 64) {
        const int v = (int)list[k];
        below += v < r1 ? 1 : 0;
        nlow += v < T ? 1 : 0;
        out[k] = (uint16_t)fold(v);
      }
      below = wave_sum_i32(below);
      nlow = wave_sum_i32(nlow);
      uint32_t piv = 0;
      if (lane < 8) piv = fold((int)list[((int64_t)(lane + 1) * n1) / 9]);
      uint32_t pv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) pv[k] = (uint32_t)__builtin_amdgcn_readlane((int)piv, k);
      if (lane == 0) {
        const uint32_t up1 = r1 >= T ? 1u : 0u;
        int4 a, b;
        a.x = (int)(fold(r1) | (((uint32_t)below & 0xffffu) << 16));
        a.y = (int)(((uint32_t)nlow & 0xffffu) |
                    ((up1 | ((((uint32_t)below >> 16) & 0xfu) << 4) | ((((uint32_t)nlow >> 16) & 0xfu) << 8)) << 16));
        a.z = (int)(uint32_t)(o16 & 0xffffffffull);
        a.w = (int)(uint32_t)(o16 >> 32);
        b.x = (int)(pv[0] | (pv[1] << 16));
        b.y = (int)(pv[2] | (pv[3] << 16));
        b.z = (int)(pv[4] | (pv[5] << 16));
        b.w = (int)(pv[6] | (pv[7] << 16));
        reinterpret_cast<int4 *>(slots + e1 * 16)[0] = a;
        reinterpret_cast<int4 *>(slots + e1 * 16)[1] = b;
      }
    }
  }
}
}  // namespace n2v

extern "C" int n2v_wedge_slots_fold(const n2v_graph *g, const uint64_t *fold_off, void *wedge_pos_rw, uint16_t *slots,
                                    void *stream) {
  if (!g || g->n_edges < 0) return N2V_EINVAL;
  if (g->n_edges == 0) return N2V_OK;
  if (!g->rowptr || !g->col || !g->edge_classes || !g->wedge_off || !g->wedge_pos || !fold_off || !slots ||
      g->wedge_wide < 2 || g->wedge_wide > 65536 || wedge_pos_rw != g->wedge_pos)
    return N2V_EINVAL;
  if ((reinterpret_cast<uintptr_t>(slots) & 31u) != 0) return N2V_EINVAL;
  int64_t blocks = ((g->n_edges + 63) / 64 + 3) / 4;
  const int64_t cap = 8 * n2v::resident_blocks((const void *)n2v::wedge_slots_fold_kernel, 256, 0);
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(n2v::wedge_slots_fold_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *g,
                     fold_off, reinterpret_cast<uint16_t *>(wedge_pos_rw), slots);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}
