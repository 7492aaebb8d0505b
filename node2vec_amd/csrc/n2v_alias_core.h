// n2v_alias_core.h -- device code shared by K1 (n2v_alias.hip) and K2 exact
// (n2v_walk.hip): the p/q bias of one 64-neighbour chunk (randomwalk.py:219-231)
// and the exact row sum of generate_alias_tables (randomwalk.py:172).
#pragma once
#include "n2v_common.h"

namespace n2v {

constexpr int kWavesPerBlock = 4;
constexpr int kLdsChunks = 128;  // class ballots kept in LDS for rows <= 8192

// diagnostic build only (-DN2V_STATS): per-wave counters / cycle stamps, flushed once
#ifdef N2V_STATS
static __device__ unsigned long long n2v_stats[40];  // one copy per translation unit
struct WaveStats { unsigned long long v[40]; };
#define N2V_STATS_ARG , WaveStats &WS
#define N2V_STATS_PASS , WS
#define N2V_STAT(i, v_) do { WS.v[i] += (unsigned long long)(v_); } while (0)
#define N2V_T0 unsigned long long n2v_tprev = __builtin_readcyclecounter();
#define N2V_T(i) do { unsigned long long tn_ = __builtin_readcyclecounter(); WS.v[i] += tn_ - n2v_tprev; n2v_tprev = tn_; } while (0)
#else
#define N2V_T0
#define N2V_T(i) do { } while (0)
#define N2V_STATS_ARG
#define N2V_STATS_PASS
#define N2V_STAT(i, v) do { } while (0)
#endif

constexpr int kBitWordsMax = 256;  // membership filter: up to 8192 bits
constexpr int kMaybeCap = 128;     // filter hits waiting for exact verification

// multiplicative hash of a vertex id onto the filter's bit range
__device__ __forceinline__ uint32_t hash_id(int32_t y, int shift) {
  return ((uint32_t)y * 2654435761u) >> shift;
}

struct StepCtx {
  const int32_t *vcol;  // N(v) ids
  const float *vw;      // N(v) weights, fp32 storage ...
  const double *vw64;   // ... or fp64 storage (at most one is set; both NULL = unit weights)
  const int32_t *scol;  // N(s) ids
  int n, nch, m, iters;
  int32_t s;
  bool need_cls, need_mem;
  double p, q;
  // Optional (round 5): the per-edge tables of the edge (s -> v) walked last -- the shared positions
  // of N(v) (ascending), the return run -- when the caller has them (n2v_edge_classes_build,
  // n2v_wedge_build: they depend on the ids alone, weighted graphs have them too).  The classes of the
  // slots then come from them: no filter over N(s), no membership search, no pass over col.  lst == NULL:
  // classes by search (rows of another kind: first steps, C callers without the tables).
  const void *lst = nullptr;
  int lst_n = 0, lst_wide = 0, rpos = 0, n_ret = 0;
};

// entry k of the shared-position list of the step (a plain branch on the width)
__device__ __forceinline__ int lst_at(const StepCtx &c, int k) {
  if (c.lst_wide) return (int)reinterpret_cast<const uint32_t *>(c.lst)[k];
  return (int)reinterpret_cast<const uint16_t *>(c.lst)[k];
}
// is position i one of the shared positions?  one lane, binary search
__device__ __forceinline__ bool lst_has(const StepCtx &c, int i) {
  int lo = 0, hi = c.lst_n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (lst_at(c, mid) < i)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo < c.lst_n && lst_at(c, lo) == i;
}
// The shared positions inside [64 chunk, 64 chunk + 64) as a ballot, by the whole wave, from a cursor
// `lm` (wave-uniform: entries below it lie in earlier chunks) that the call advances.  `flags`: 64 bytes
// of this wave's LDS, zero on entry and on return.
__device__ __forceinline__ uint64_t lst_chunk_mask(const StepCtx &c, int chunk, int lane, int &lm,
                                                   uint8_t *flags) {
  const int c0 = chunk * 64;
  bool any = false;
  for (;;) {
    const int k = lm + lane;
    const int pos = k < c.lst_n ? lst_at(c, k) : 0x7fffffff;
    const bool in = pos < c0 + 64;
    if (in) flags[pos - c0] = 1;
    const int cnt = __popcll(ballot64(in));
    lm += cnt;
    any = any || cnt > 0;
    if (cnt < 64) break;  // (64 in range: the next 64 entries may hold more)
  }
  if (!any) return 0ull;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const bool mine = flags[lane] != 0;
  __builtin_amdgcn_wave_barrier();
  flags[lane] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  return ballot64(mine);
}

// weight of neighbour i of the current row, widened to the fp64 the reference computes in
// (wave-uniform pointers: the selection is a scalar branch)
__device__ __forceinline__ double weight_at(const StepCtx &c, int i) {
  if (c.vw64) return c.vw64[i];
  return c.vw ? (double)c.vw[i] : 1.0;
}

__device__ __forceinline__ bool member_sorted(const int32_t *a, int m, int32_t x, int iters) {
  int lo = 0, hi = m;
  for (int it = 0; it < iters; ++it) {
    int mid = (lo + hi) >> 1;
    int32_t val = a[mid < m ? mid : m - 1];
    bool act = lo < hi;
    bool less = val < x;
    lo = (act && less) ? mid + 1 : lo;
    hi = (act && !less) ? mid : hi;
  }
  return a[lo < m ? lo : m - 1] == x && lo < m;
}

// four independent searches advanced in lockstep: 4 gathers in flight per round
__device__ __forceinline__ void member_sorted_x4(const int32_t *a, int m, const int32_t (&x)[4],
                                                 int iters, bool (&found)[4]) {
  int lo[4], hi[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    lo[u] = 0;
    hi[u] = m;
  }
  for (int it = 0; it < iters; ++it) {
    int32_t val[4];
    int mid[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      mid[u] = (lo[u] + hi[u]) >> 1;
      val[u] = a[mid[u] < m ? mid[u] : m - 1];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool act = lo[u] < hi[u];
      const bool less = val[u] < x[u];
      lo[u] = (act && less) ? mid[u] + 1 : lo[u];
      hi[u] = (act && !less) ? mid[u] : hi[u];
    }
  }
  int32_t fin[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) fin[u] = a[lo[u] < m ? lo[u] : m - 1];
#pragma unroll
  for (int u = 0; u < 4; ++u) found[u] = fin[u] == x[u] && lo[u] < m;
}

// four lower_bound searches in lockstep: lo[u] = first index with a[lo] >= x[u],
// found[u] = a[lo[u]] == x[u]
__device__ __forceinline__ void lower_bound_x4(const int32_t *a, int m, const int32_t (&x)[4],
                                               int iters, int (&lo)[4], bool (&found)[4]) {
  int hi[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    lo[u] = 0;
    hi[u] = m;
  }
  for (int it = 0; it < iters; ++it) {
    int32_t val[4];
    int mid[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      mid[u] = (lo[u] + hi[u]) >> 1;
      val[u] = a[mid[u] < m ? mid[u] : m - 1];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool act = lo[u] < hi[u];
      const bool less = val[u] < x[u];
      lo[u] = (act && less) ? mid[u] + 1 : lo[u];
      hi[u] = (act && !less) ? mid[u] : hi[u];
    }
  }
  int32_t fin[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) fin[u] = a[lo[u] < m ? lo[u] : m - 1];
#pragma unroll
  for (int u = 0; u < 4; ++u) found[u] = fin[u] == x[u] && lo[u] < m;
}

// lower_bound of x[u] in a[0, m), each search confined to its slice
// [slice*stride, min(m, (slice+1)*stride)) (the caller guarantees the answer lies in
// the slice or at its end); found[u] = a[lo[u]] == x[u]
__device__ __forceinline__ void lower_bound_slices_x4(const int32_t *a, int m, int stride,
                                                      const int (&slice)[4],
                                                      const int32_t (&x)[4], int iters,
                                                      int (&lo)[4], bool (&found)[4]) {
  int hi[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    lo[u] = slice[u] * stride;
    hi[u] = lo[u] + stride;
    lo[u] = lo[u] < m ? lo[u] : m;
    hi[u] = hi[u] < m ? hi[u] : m;
  }
  for (int it = 0; it < iters; ++it) {
    int32_t val[4];
    int mid[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      mid[u] = (lo[u] + hi[u]) >> 1;
      val[u] = a[mid[u] < m ? mid[u] : m - 1];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool act = lo[u] < hi[u];
      const bool less = val[u] < x[u];
      lo[u] = (act && less) ? mid[u] + 1 : lo[u];
      hi[u] = (act && !less) ? mid[u] : hi[u];
    }
  }
  int32_t fin[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) fin[u] = a[lo[u] < m ? lo[u] : m - 1];
#pragma unroll
  for (int u = 0; u < 4; ++u) found[u] = fin[u] == x[u] && lo[u] < m;
}

// biased weight of element chunk*64+lane (randomwalk.py:219-231); 0 past the row
template <bool kFromCache>
__device__ __forceinline__ double chunk_bias(const StepCtx &c, int chunk, int lane,
                                             uint64_t *cls, bool &valid) {
  const int i = chunk * 64 + lane;
  valid = i < c.n;
  double wt = valid ? weight_at(c, i) : 0.0;
  if (!c.need_cls) return wt;
  bool is_ret, is_mem;
  if (kFromCache && chunk < kLdsChunks) {
    is_ret = (cls[2 * chunk] >> lane) & 1ull;
    is_mem = (cls[2 * chunk + 1] >> lane) & 1ull;
  } else {
    if (c.lst) {  // the per-edge tables: the return run by position, the shared slots by a search of the list
      is_ret = valid && i >= c.rpos && i < c.rpos + c.n_ret;
      is_mem = c.need_mem && valid && !is_ret && lst_has(c, i);
    } else {
    int32_t x = valid ? c.vcol[i] : -1;
    is_ret = valid && x == c.s;
    is_mem = false;
#if defined(N2V_ABLATE) && (N2V_ABLATE & 2)  // timing-only build: no membership search
    if (c.need_mem) is_mem = false;
#else
    if (c.need_mem) is_mem = member_sorted(c.scol, c.m, x, c.iters) && valid && !is_ret;
#endif
    }
    if (!kFromCache && chunk < kLdsChunks) {
      uint64_t rm = ballot64(is_ret), mm = ballot64(is_mem);
      if (lane == 0) {
        cls[2 * chunk] = rm;
        cls[2 * chunk + 1] = mm;
      }
    }
  }
  if (is_ret) return wt / c.p;                  // :223-224
  if (is_mem || !c.need_mem) return wt;         // :226-227 (and q == 1: w / 1.0 == w)
  return wt / c.q;                              // :229-230
}


// avg-independent part of generate_alias_tables: sum of the biased weights in the
// reference's order.  When every b is a multiple of 2^-20 below 2^11 all partial
// sums are exactly representable, so the wave reduces in int64 (any order gives
// the same bits); otherwise: serial left-to-right fp64 adds.  Also returns the
// biased weight of element `pick` (pass pick < 0 to skip).
__device__ __forceinline__ double row_sum(const StepCtx &c, int lane, uint64_t *cls, int pick,
                                          double &b_pick) {
  int64_t isum = 0;
  bool exact = true;
  b_pick = 0.0;
  for (int chunk = 0; chunk < c.nch; ++chunk) {
    bool valid;
    double b = chunk_bias<false>(c, chunk, lane, cls, valid);
    double t = b * 1048576.0;
    bool ok = (t >= 0.0) && (t < 2147483648.0) && (t == trunc(t));
    exact = exact && ok;
    isum += ok ? (int64_t)t : 0;
    if (chunk == (pick >> 6)) b_pick = readlane_f64(b, pick & 63);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (ballot64(!exact) == 0ull && c.n <= (1 << 21))
    return (double)readfirstlane_i64(wave_sum_i64(isum)) * (1.0 / 1048576.0);
  double total = 0.0;  // reference order: left to right, one rounding per add
  for (int chunk = 0; chunk < c.nch; ++chunk) {
    bool valid;
    double b = chunk_bias<true>(c, chunk, lane, cls, valid);
    int cnt = min(64, c.n - chunk * 64);
    for (int j = 0; j < cnt; ++j) total = total + readlane_f64(b, j);
  }
  return total;
}

}  // namespace n2v
