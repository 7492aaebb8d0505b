// n2v_walk_unit.hip -- K2 exact mode, specialised for UNIT edge weights (gfx950).
//
// Same contract and same bits as walk_exact_kernel (n2v_walk.hip): per step the
// index sampling_from_alias(r1, r2) returns on the table generate_edge_alias_tables
// builds (reference randomwalk.py:86-99, :157-232).  With every weight 1.0 the
// biased weight of a neighbour takes one of three values (:223-230):
//     return  1.0 / p     shared  1.0     other  1.0 / q
// and the whole step is integer / scalar work:
//   * when those scale to exact integers (x * 2^20, true for the dyadic p, q of every
//     BASELINE config) the row sum of :172 is three popcounts times three constants
//     (every partial sum is exactly representable, so order does not matter); for other
//     p, q it is added in the reference's order, one closed-form step per run of equal
//     addends (rep_add below, kernel instance <false>);
//   * the LDS cache is the two class ballots per 64-neighbour chunk, w is never read;
//   * probs0 has three values (one fp64 division decides most steps, three at most), so
//     the underfull / overfull stacks of the pairing (:175-189) are sequences of RUNS of
//     equal values: an over absorbs a run in a tight loop, a demoted residual cascades
//     through a run in closed form, a single-class stack is consumed by rank alone
//     (see "pairing, run engine" in unit_draw);
//   * a uniform row (first step, or p == q == 1) has probs0 == 1.0 everywhere, no
//     underfull slot, and the draw is `pick` itself: O(1).
// Membership "x in N(s)" (:226) picks one of five strategies by the two row lengths
// (reverse / staged / filter / merge / direct, see unit_draw).  One wave64 per walker;
// waves take walkers from a shared counter (status[1]), because walks differ widely in cost.
#include "n2v_alias_core.h"
#include "n2v_unit_core.h"

namespace n2v {

constexpr int kUC = 128;     // class ballots cached for the TOP 128 chunks (8192 neighbours)
constexpr int kYsCap = 384;   // N(s) staged in LDS when it has at most this many ids
constexpr int kBitsA = 256;   // filter words beside a staged N(s)        (8192 bits)
constexpr int kBitsB = 256;   // filter words when N(s) is not staged (a power of two <= kYsCap)
constexpr int kMaybeU = 128;  // filter hits waiting for verification
constexpr int kSparseCap = 64;  // reverse mode: return/shared slots below the cached chunks, by position

// 5 KB per wave: 8 blocks of 4 waves per CU, the hardware maximum of 8 waves per SIMD.  `pool` is either {staged ids of N(s)} (m <= kYsCap, filter in
// `bits`) or one large filter (kYsCap < m <= 8192); reverse classification uses it for the positions of
// the return/shared slots that lie below the cached chunks (kSparseCap entries: index | class << 31).
static_assert(kBitsB <= kYsCap && (kBitsB & (kBitsB - 1)) == 0, "large filter must fit the pool");
struct UnitLds {
  uint64_t cls[2 * kUC];     // slot nch-1-chunk: ballot(return), ballot(shared)
  uint32_t bits[kBitsA];     // small filter
  int32_t mlist[kMaybeU];    // filter hits waiting for verification
  uint32_t pool[kYsCap];     // staged N(s)  |  large filter
};

// binary search over ids staged in LDS (no global gathers on the dependent chain)
__device__ __forceinline__ bool member_lds(const uint32_t *ys, int m, int32_t x, int iters) {
  int lo = 0, hi = m;
  for (int it = 0; it < iters; ++it) {
    const int mid = (lo + hi) >> 1;
    const int32_t val = (int32_t)ys[mid < m ? mid : m - 1];
    const bool act = lo < hi;
    const bool less = val < x;
    lo = (act && less) ? mid + 1 : lo;
    hi = (act && !less) ? mid : hi;
  }
  return (int32_t)ys[lo < m ? lo : m - 1] == x && lo < m;
}

// four LDS searches in lock-step (independent chains, so their LDS round trips overlap)
__device__ __forceinline__ void member_lds_x4(const uint32_t *ys, int m, const int32_t (&x)[4],
                                              int iters, bool (&found)[4]) {
  int lo[4] = {0, 0, 0, 0}, hi[4] = {m, m, m, m};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int mid = (lo[u] + hi[u]) >> 1;
      const int32_t val = (int32_t)ys[mid < m ? mid : m - 1];
      const bool act = lo[u] < hi[u];
      const bool less = val < x[u];
      lo[u] = (act && less) ? mid + 1 : lo[u];
      hi[u] = (act && !less) ? mid : hi[u];
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u)
    found[u] = found[u] | ((int32_t)ys[lo[u] < m ? lo[u] : m - 1] == x[u] && lo[u] < m);
}

// ---- the reference's row sum when 1/p or 1/q is not dyadic ------------------------------
// sum(node_weights) (:172) adds left to right in fp64, so with values that are not
// multiples of 2^-20 the result depends on the order of the classes.  A row is a few
// return/shared slots separated by long runs of "other" slots, and a run of k equal
// addends needs no loop: while s stays inside one binade [2^e, 2^(e+1)) it is a multiple
// of ulp = 2^(e-52), c = m*ulp + f, and fl(s + c) = s + (m or m+1)*ulp according to f
// alone -- the same step for every addition.  (If f is exactly ulp/2 the tie goes to the
// even neighbour: after one addition s is an even multiple of ulp and from then on the
// step is constant again.)  So: per binade, one real addition to measure the step, an
// exact multiply for all additions that stay below 2^(e+1), one real addition to cross.
__device__ __forceinline__ double rep_add(double s, double c, int k) {
  const uint64_t c_man =
      ((uint64_t)__double_as_longlong(c) & 0x000fffffffffffffull) | 0x0010000000000000ull;
  const int c_exp = biased_exp(c);
  while (k > 0) {
    const int es = biased_exp(s);
    if (es == 0) {  // s == 0.0: the first addend, exact
      s = s + c;
      --k;
      continue;
    }
    const int shift = es - c_exp;  // low bits of c below ulp(s)
    if (shift >= 1 && shift <= 53 && (c_man & ((1ull << shift) - 1ull)) == (1ull << (shift - 1))) {
      s = readfirstlane_f64(s + c);  // a tie: one real addition makes s an even multiple of ulp
      --k;
      if (k == 0 || biased_exp(s) != es) continue;
    }
    const double t = readfirstlane_f64(s + c);
    if (biased_exp(t) != es) {  // this addition leaves the binade: the real operation
      s = t;
      --k;
      continue;
    }
    const double step = t - s;  // exact
    if (step == 0.0) return s;  // c vanishes against s
    // the largest j with s + j*step < 2^(e+1), at least 1 here
    const double need = __longlong_as_double((long long)(es + 1) << 52) - s;  // exact
    if ((double)k * step < need) {  // the usual case: the whole run stays in the binade
      // (a product that compares below need is below 2^(e+1) and a multiple of ulp: exact)
      return readfirstlane_f64(s + (double)k * step);
    }
    double j = ceil(need / step) - 1.0;
    while ((j + 1.0) * step < need) j += 1.0;
    while (j * step >= need) j -= 1.0;
    j = fmin(j, (double)k);
    s = readfirstlane_f64(s + j * step);  // exact: a multiple of ulp below 2^(e+1)
    k -= __builtin_amdgcn_readfirstlane((int)j);
  }
  return s;
}

struct UnitStep {
  const int32_t *vcol, *scol;
  int n, nch, m, iters;
  int32_t s;
  bool need_mem;
  // the classes of this step's table are already known by position (wedge table of the graph,
  // n2v_wedge_build): return slots [w_rpos, w_rpos + w_nR), shared slots w_pos[0, w_nM)
  bool w_have = false, w_wide = false;
  const void *w_pos = nullptr;
  int w_nR = 0, w_nM = 0, w_rpos = 0;
};

__device__ __forceinline__ uint64_t valid_mask(const UnitStep &c, int chunk) {
  const int rem = c.n - chunk * 64;
  return rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
}

// number of pivots < x among the 64 pivots parked in LDS (0..63): the 1/64 slice of the
// sorted row that can contain x
__device__ __forceinline__ int pivot_slice(const int32_t *piv, int32_t x) {
  int a = 0, b = 64;
#pragma unroll
  for (int it = 0; it < 7; ++it) {
    const int mid = (a + b) >> 1;
    const bool act = a < b;
    const bool less = act && piv[mid < 64 ? mid : 63] < x;
    a = (act && less) ? mid + 1 : a;
    b = (act && !less) ? mid : b;
  }
  return a < 63 ? a : 63;
}

// park 64 pivots row[(l+1)*stride - 1] of a sorted row in LDS
__device__ __forceinline__ void park_pivots(int32_t *piv, const int32_t *row, int len, int stride,
                                            int lane) {
  const int pi = (lane + 1) * stride - 1;
  piv[lane] = row[pi < len ? pi : len - 1];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// class ballots of one chunk: from LDS when cached, else by searching again
__device__ __forceinline__ void chunk_classes(const UnitStep &c, UnitLds &L, int chunk, int lane,
                                              int sp_n, uint64_t &rm, uint64_t &mm) {
  const int ci = c.nch - 1 - chunk;
  if (ci < kUC) {
    // uniform address, but an LDS load is lane-varying to the compiler: say it is scalar,
    // so that everything derived from the class masks stays on the scalar unit
    rm = readfirstlane_u64(L.cls[2 * ci]);
    mm = readfirstlane_u64(L.cls[2 * ci + 1]);
    return;
  }
  if (sp_n >= 0) {  // reverse mode kept every return/shared slot of the uncached part by position
    const int32_t e = lane < sp_n ? (int32_t)L.pool[lane] : 0;
    uint64_t b = ballot64(lane < sp_n && ((e & 0x7fffffff) >> 6) == chunk);
    rm = 0ull;
    mm = 0ull;
    while (b) {
      const int32_t el = __builtin_amdgcn_readlane(e, (int)__builtin_ctzll(b));
      b &= b - 1ull;
      if (el < 0)
        rm |= 1ull << (el & 63);
      else
        mm |= 1ull << (el & 63);
    }
    return;
  }
  const int i = chunk * 64 + lane;
  const bool valid = i < c.n;
  const int32_t x = valid ? c.vcol[i] : -1;
  const bool is_ret = valid && x == c.s;
  bool is_mem = false;
  if (c.need_mem) is_mem = member_sorted(c.scol, c.m, x, c.iters) && valid && !is_ret;
  rm = ballot64(is_ret);
  mm = ballot64(is_mem);
}

// one lower_bound confined to a slice of a sorted row (see lower_bound_slices_x4)
__device__ __forceinline__ int lower_bound_slice(const int32_t *a, int m, int stride, int slice,
                                                 int32_t x, int iters, bool &found) {
  int lo = slice * stride, hi = lo + stride;
  lo = lo < m ? lo : m;
  hi = hi < m ? hi : m;
  for (int it = 0; it < iters; ++it) {
    const int mid = (lo + hi) >> 1;
    const int32_t val = a[mid < m ? mid : m - 1];
    const bool act = lo < hi;
    const bool less = val < x;
    lo = (act && less) ? mid + 1 : lo;
    hi = (act && !less) ? mid : hi;
  }
  found = a[lo < m ? lo : m - 1] == x && lo < m;
  return lo;
}

// member test through the 64 pivots of N(s) parked in `piv`: LDS search for the slice,
// then a binary search confined to it (log2(m/64) global rounds instead of log2(m))
__device__ __forceinline__ bool member_pivoted(const UnitStep &c, const int32_t *piv, int stride,
                                               int iters, int32_t x) {
  int lo = pivot_slice(piv, x) * stride;
  int hi = lo + stride;
  lo = lo < c.m ? lo : c.m;
  hi = hi < c.m ? hi : c.m;
  for (int it = 0; it < iters; ++it) {
    const int mid = (lo + hi) >> 1;
    const int32_t val = c.scol[mid < c.m ? mid : c.m - 1];
    const bool act = lo < hi;
    const bool less = val < x;
    lo = (act && less) ? mid + 1 : lo;
    hi = (act && !less) ? mid : hi;
  }
  return c.scol[lo < c.m ? lo : c.m - 1] == x && lo < c.m;
}

__device__ __forceinline__ void verify_unit(const UnitStep &c, UnitLds &L, int count, int lane,
                                            bool staged, int &nM) {
  const int stride_s = (c.m + 63) >> 6;
  const int iters_s = 32 - __clz(stride_s);
  for (int k = 0; k < count; k += 64) {
    const bool act = k + lane < count;
    const int i = act ? L.mlist[k + lane] : 0;
    const int32_t x = act ? c.vcol[i] : -1;
    // not staged = large-filter mode: the small filter `bits` is idle and holds the pivots
    const bool mem = (staged ? member_lds(L.pool, c.m, x, c.iters)
                             : member_pivoted(c, reinterpret_cast<const int32_t *>(L.bits),
                                              stride_s, iters_s, x)) && act;
    const int ci = c.nch - 1 - (i >> 6);
    if (mem && ci < kUC)
      atomicOr(reinterpret_cast<unsigned long long *>(&L.cls[2 * ci + 1]), 1ull << (i & 63));
    nM += __popcll(ballot64(mem));
  }
}

template <bool kDyadic>
__device__ __forceinline__ int unit_draw(const UnitStep &c, const UnitConsts &K, uint32_t u1,
                                         uint32_t u2, int lane, UnitLds &L N2V_STATS_ARG) {
  const int n = c.n;
  const int pick = pick_index(u1, n);  // int(r1 * n)
  const double r2 = (double)u2 * (1.0 / 4294967296.0);

  N2V_T0
  N2V_STAT(0, 1);
  int nR = 0, nM = 0;
  int sp_n = -1;  // >= 0: L.pool holds the position of every return/shared slot below the cached chunks
  // ---- reverse classification: search from the SHORTER list --------------------------
  // The classes of N(v) are needed as ballots and counts, not as a stream.  When N(s)
  // is much shorter than N(v) (or only the return slot matters, q == 1) it is cheaper
  // to locate s and every distinct id of N(s) inside the sorted row of v by binary
  // search and scatter the class bits: O(m log n) instead of O(n), and N(v) is never
  // read.  Multi-edges: every occurrence of an id in N(v) is marked; a repeated id of
  // N(s) is searched once.
  // (thresholds 128..512 and ratios 1..4 were timed on cfg 2: flat within 2 %)
  const bool reverse = n > 256 && (!c.need_mem || 2 * c.m < n);
  if (c.w_have) {
    // the graph's wedge table names every return / shared slot by position: scatter the class
    // bits, no search over N(s) and no pass over N(v)
    const int ncached = c.nch < kUC ? c.nch : kUC;
    for (int wv = lane; wv < 2 * ncached; wv += 64) L.cls[wv] = 0ull;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    sp_n = 0;
    nR = c.w_nR;
    nM = c.need_mem ? c.w_nM : 0;
    const int items = nR + nM;
    for (int base = 0; base < items; base += 64) {
      const int k = base + lane;
      const bool act = k < items;
      const bool isr = k < nR;
      int j = 0;
      if (act) {
        if (isr)
          j = c.w_rpos + k;
        else if (c.w_wide)
          j = wedge_at_t<uint32_t>(c.w_pos, k - nR);
        else
          j = wedge_at_t<uint16_t>(c.w_pos, k - nR);
      }
      N2V_CHECK_RANGE(5, j, 0, n);
      const int ci = c.nch - 1 - (j >> 6);
      if (act && ci < kUC)
        atomicOr(reinterpret_cast<unsigned long long *>(&L.cls[2 * ci + (isr ? 0 : 1)]),
                 1ull << (j & 63));
      const bool far = act && ci >= kUC;
      const uint64_t fb = ballot64(far);
      if (fb != 0ull) {
        const int slot = sp_n + __popcll(fb & ((1ull << lane) - 1ull));
        if (far && slot < kSparseCap) L.pool[slot] = (uint32_t)j | (isr ? 0x80000000u : 0u);
        sp_n += __popcll(fb);
      }
    }
    if (sp_n > kSparseCap) sp_n = -1;  // too many: the uncached chunks are searched on demand
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    N2V_T(24);
    N2V_STAT(13, 1);
  } else if (reverse) {
    const int ncached = c.nch < kUC ? c.nch : kUC;
    for (int wv = lane; wv < 2 * ncached; wv += 64) L.cls[wv] = 0ull;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int items = 1 + (c.need_mem ? c.m : 0);  // item 0 = s, item k = N(s)[k-1]
    // first level shared by every search: 64 pivots N(v)[(l+1) * stride - 1], one
    // gather, parked in LDS (mlist is idle here); each lane then finds its 1/64
    // slice by LDS binary search, leaving log2(n/64) global rounds instead of log2(n)
    const int stride = (n + 63) >> 6;
    sp_n = 0;
    park_pivots(L.mlist, c.vcol, n, stride, lane);
    const int iters_n = 32 - __clz(stride);
    auto mark = [&](bool f, int j, int32_t xv, bool isr) {  // every occurrence of xv in N(v)
      while (ballot64(f) != 0ull) {
        const int ci = c.nch - 1 - (j >> 6);
        if (f && ci < kUC)
          atomicOr(reinterpret_cast<unsigned long long *>(&L.cls[2 * ci + (isr ? 0 : 1)]),
                   1ull << (j & 63));
        const bool far = f && ci >= kUC;
        const uint64_t fb = ballot64(far);
        if (fb != 0ull) {
          const int slot = sp_n + __popcll(fb & ((1ull << lane) - 1ull));
          if (far && slot < kSparseCap) L.pool[slot] = (uint32_t)j | (isr ? 0x80000000u : 0u);
          sp_n += __popcll(fb);
        }
        nR += __popcll(ballot64(f && isr));
        nM += __popcll(ballot64(f && !isr));
        ++j;
        f = f && j < n && c.vcol[j < n ? j : n - 1] == xv;
      }
    };
    auto item = [&](int k, bool &isr) -> int32_t {  // item 0 = s, item k = N(s)[k-1]
      isr = k == 0;
      if (k == 0) return c.s;
      if (k >= items) return -1;
      const int32_t y = c.scol[k - 1];
      const int32_t prev = k >= 2 ? c.scol[k - 2] : -1;
      return (y == c.s || y == prev) ? -1 : y;  // return slot / repeated id: skip
    };
    if (items <= 64) {  // one search per lane is enough
      bool isr, found;
      const int32_t xv = item(lane, isr);
      const int lo1 = lower_bound_slice(c.vcol, n, stride, pivot_slice(L.mlist, xv), xv, iters_n,
                                        found);
      mark(found && xv >= 0, lo1, xv, isr);
    } else {
      for (int base = 0; base < items; base += 256) {
        int32_t x[4];
        bool isret[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = item(base + u * 64 + lane, isret[u]);
        int lo[4], slice[4];
        bool found[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) slice[u] = pivot_slice(L.mlist, x[u]);
        lower_bound_slices_x4(c.vcol, n, stride, slice, x, iters_n, lo, found);
#pragma unroll
        for (int u = 0; u < 4; ++u) mark(found[u] && x[u] >= 0, lo[u], x[u], isret[u]);
      }
    }
    if (sp_n > kSparseCap) sp_n = -1;  // too many: the uncached chunks are searched again on demand
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    N2V_T(24);
    N2V_STAT(13, 1);
  } else {
  // ---- pass 0: membership strategy for "x in N(s)" (:226) ---------------------------
  //  staged  m <= 1024: N(s) is copied into LDS; rows of v up to 2 chunks search it
  //          directly, longer rows test a hashed-id filter first and verify only the
  //          hits, by LDS binary search (no global gather on any dependent chain)
  //  filter  m <= 8192 and not much longer than N(v): large filter, hits verified by
  //          binary search over global memory, batched once per step
  //  merge   N(s) too long for a filter and N(v) of comparable length (hub to hub): both rows
  //          are sorted, so the ids N(s) can share with a group of 256 neighbours form one
  //          window that only moves forward; it is staged in LDS 384 ids at a time and
  //          searched there (one coalesced load per window instead of 8 dependent gathers)
  //  direct  otherwise (N(s) a hub, N(v) short): per-lane global binary search
  const bool staged = c.need_mem && c.m <= kYsCap;
  const bool big_filter = c.need_mem && !staged && c.m <= 4096 && c.m <= 8 * n + 64;
  const bool use_filter = big_filter || (staged && c.nch > 2);
  uint32_t *fbits = staged ? L.bits : L.pool;
  int shift = 32;
  const bool merge = c.need_mem && !staged && !big_filter && c.m <= 4 * n;
  const bool direct = c.need_mem && !staged && !big_filter && !merge;
  const int stride_s = (c.m + 63) >> 6;
  const int iters_s = 32 - __clz(stride_s);
  if (big_filter) park_pivots(reinterpret_cast<int32_t *>(L.bits), c.scol, c.m, stride_s, lane);
  if (direct) park_pivots(L.mlist, c.scol, c.m, stride_s, lane);  // no filter hits in this mode
  if (staged || big_filter) {
    if (use_filter) {
      const int cap = staged ? kBitsA : kBitsB;
      int words = 64;
      while (words < cap && words * 32 < 16 * c.m) words <<= 1;
      shift = 32 - (5 + (31 - __clz(words)));
      for (int wv = lane; wv < words; wv += 64) fbits[wv] = 0u;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    for (int yb = 0; yb < c.m; yb += 256) {  // 4 loads in flight per lane
      int32_t y[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = yb + u * 64 + lane;
        y[u] = j < c.m ? c.scol[j] : -1;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = yb + u * 64 + lane;
        if (j < c.m) {
          if (staged) L.pool[j] = (uint32_t)y[u];
          if (use_filter) {
            const uint32_t h = hash_id(y[u], shift);
            atomicOr(&fbits[h >> 5], 1u << (h & 31));
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  N2V_T(16);
  N2V_STAT(1, (staged && use_filter) ? 1 : 0); N2V_STAT(2, (c.need_mem && !staged && !big_filter) ? 1 : 0);
  N2V_STAT(11, (staged && !use_filter) ? 1 : 0); N2V_STAT(12, big_filter ? 1 : 0);
  // ---- pass 1: stream N(v) ids, classify, count ---------------------------------
  int mcount = 0;
  // classify one chunk given its ids and (outside filter mode) the membership flags
  auto finish_chunk = [&](int chunk, int32_t x, bool memflag) {
    const int i = chunk * 64 + lane;
    const bool valid = i < n;
    const bool is_ret = valid && x == c.s;
    bool is_mem = false, maybe = false;
    if (use_filter) {
      const uint32_t h = hash_id(x, shift);
      maybe = valid && !is_ret && ((fbits[h >> 5] >> (h & 31)) & 1u);
    } else {
      is_mem = memflag && valid && !is_ret;
    }
    const uint64_t rm = ballot64(is_ret), mm = ballot64(is_mem);
    nR += __popcll(rm);
    nM += __popcll(mm);
    const int ci = c.nch - 1 - chunk;
    if (ci < kUC && lane == 0) {
      L.cls[2 * ci] = rm;
      L.cls[2 * ci + 1] = mm;
    }
    const uint64_t ym = ballot64(maybe);
    if (ym) {
      const int cnt = __popcll(ym);
      if (mcount + cnt > kMaybeU) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        verify_unit(c, L, mcount, lane, staged, nM);
        mcount = 0;
      }
      if (maybe) L.mlist[mcount + __popcll(ym & ((1ull << lane) - 1ull))] = i;
      mcount += cnt;
    }
  };
  const bool search_here = c.need_mem && !use_filter;  // staged (short rows), merge or direct
  int chunk0 = 0;
  if (merge) {
    int sp = 0;  // every id of N(s)[0, sp) is below the ids of N(v) still to come
    for (; chunk0 < c.nch; chunk0 += 4) {
      int32_t xs[4];
      bool memv[4] = {false, false, false, false};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = (chunk0 + u) * 64 + lane;
        xs[u] = i < n ? c.vcol[i] : -1;
      }
      const int32_t x_hi = __builtin_amdgcn_readfirstlane(c.vcol[min(n, (chunk0 + 4) * 64) - 1]);
      for (;;) {
        const int wn = min(kYsCap, c.m - sp);
        if (wn <= 0) break;
        int32_t y[kYsCap / 64];
#pragma unroll
        for (int k = 0; k < kYsCap / 64; ++k) {
          const int j = lane + 64 * k;
          y[k] = j < wn ? c.scol[sp + j] : 0x7fffffff;
        }
        int below = 0;  // window ids smaller than the last id of the group
#pragma unroll
        for (int k = 0; k < kYsCap / 64; ++k) {
          const int j = lane + 64 * k;
          if (j < wn) L.pool[j] = (uint32_t)y[k];
          below += (int)(y[k] < x_hi);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        member_lds_x4(L.pool, wn, xs, 9, memv);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) below += __shfl_xor(below, off, 64);
        below = __builtin_amdgcn_readfirstlane(below);
        __builtin_amdgcn_wave_barrier();
        sp += below;
        if (below < wn) break;  // the window reaches the last id of the group: nothing further can match
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (chunk0 + u < c.nch) finish_chunk(chunk0 + u, xs[u], memv[u]);
    }
  }
  for (; chunk0 + 4 <= c.nch; chunk0 += 4) {  // full groups: 4 loads / 4 searches in flight
    int32_t xs[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = (chunk0 + u) * 64 + lane;
      xs[u] = i < n ? c.vcol[i] : -1;
    }
    bool memv[4] = {false, false, false, false};
    if (search_here) {
      if (staged) {
#pragma unroll
        for (int u = 0; u < 4; ++u) memv[u] = member_lds(L.pool, c.m, xs[u], c.iters);
      } else {  // direct: shared pivot level, then 4 interleaved confined searches
        int slice[4], lo4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) slice[u] = pivot_slice(L.mlist, xs[u]);
        lower_bound_slices_x4(c.scol, c.m, stride_s, slice, xs, iters_s, lo4, memv);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) finish_chunk(chunk0 + u, xs[u], memv[u]);
  }
  for (; chunk0 < c.nch; ++chunk0) {  // remainder (and every row of <= 3 chunks): one at a time
    const int i = chunk0 * 64 + lane;
    const int32_t x = i < n ? c.vcol[i] : -1;
    bool memflag = false;
    if (search_here) {
      if (staged)
        memflag = member_lds(L.pool, c.m, x, c.iters);
      else
        lower_bound_slice(c.scol, c.m, stride_s, pivot_slice(L.mlist, x), x, iters_s, memflag);
    }
    finish_chunk(chunk0, x, memflag);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#ifdef N2V_STATS
  if (merge) { N2V_T(22); N2V_STAT(5, 1); } else if (direct) { N2V_T(20); } else { N2V_T(17); }
#endif
  N2V_STAT(3, mcount);
  if (mcount) verify_unit(c, L, mcount, lane, staged, nM);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  }
  N2V_T(18);
  // ---- :172-173 on three values -----------------------------------------------------
  const int nO = n - nR - nM;
  double avg;
  if (kDyadic) {
    const int64_t isum = (int64_t)nR * K.TR + (int64_t)nM * K.TM + (int64_t)nO * K.TO;
    avg = ((double)isum * (1.0 / 1048576.0)) / (double)n;
  } else {
    // left to right over the classes of N(v); runs of "other" slots by rep_add
    double sum = 0.0;
    int pending = 0;
    for (int chunk = 0; chunk < c.nch; ++chunk) {
      uint64_t rm, mm;
      chunk_classes(c, L, chunk, lane, sp_n, rm, mm);
      const uint64_t vm = valid_mask(c, chunk);
      uint64_t spec = (rm | mm) & vm;
      int pos = 0;
      while (spec != 0ull) {
        const int l = (int)__builtin_ctzll(spec);
        spec &= spec - 1ull;
        sum = rep_add(sum, K.bO, pending + l - pos);
        sum = readfirstlane_f64(sum + (((rm >> l) & 1ull) ? K.bR : K.bM));
        pending = 0;
        pos = l + 1;
      }
      pending += __popcll(vm) - pos;
    }
    sum = rep_add(sum, K.bO, pending);
    avg = sum / (double)n;
  }
  // probs0 of slot `pick` first: one fp64 division decides most steps (an underfull slot that is
  // accepted never changes); the other two class values are only needed past this exit
  uint64_t prm, pmm;
  chunk_classes(c, L, pick >> 6, lane, sp_n, prm, pmm);
  const bool pR = (prm >> (pick & 63)) & 1ull, pM = (pmm >> (pick & 63)) & 1ull;
  const double p_pick = readfirstlane_f64(pick3(pR, pM, K.bR, K.bM, K.bO) / avg);
  N2V_T(19);
  if (p_pick < 1.0 && r2 < p_pick) return pick;  // untouched underfull slot
  // uniform values: scalar registers, so they cost no VGPRs across the pairing code
  const double vR = pR ? p_pick : readfirstlane_f64(K.bR / avg);
  const double vM = (pM && !pR) ? p_pick : readfirstlane_f64(K.bM / avg);
  const double vO = (!pR && !pM) ? p_pick : readfirstlane_f64(K.bO / avg);
  const bool uR = vR < 1.0, uM = vM < 1.0, uO = vO < 1.0;
  const bool any_under = (nR && uR) || (nM && uM) || (nO && uO);
  const bool any_over = (nR && !uR) || (nM && !uM) || (nO && !uO);
  if (!any_under || !any_over) return (r2 < p_pick) ? pick : 0;  // the loop of :182 never runs

  N2V_STAT(6, 1);
  // ---- pairing, count-based fast path ----------------------------------------------
  // When "other" is the ONLY underfull class every absorbed slot has the same value
  // vO, so the fp64 sequence of :186 depends on how MANY slots an overfull absorbs,
  // not on which ones; the identity of a slot matters only to know when `pick` is
  // next, i.e. its rank among the underfull slots above it -- a popcount over the
  // class ballots.  The stack discipline is unchanged: overfull slots are taken in
  // descending index order, a demoted one is absorbed first by its successor.
  const int top_cached_chunk = c.nch - kUC;  // chunks below this are not in LDS
#ifndef N2V_NO_CASE_A
  if (uO && !(nR && uR) && !(nM && uM) && (pick >> 6) >= top_cached_chunk) {
    const int total_u = nO;
    int above = total_u + 1;  // underfull slots consumed before `pick` is next (never, if overfull)
    if (!pR && !pM) {
      int nrm = 0;  // return/shared slots with index > pick
      const int pc = pick >> 6;
      for (int base = pc; base < c.nch; base += 64) {
        const int ch = base + lane;
        uint64_t w = 0;
        if (ch < c.nch) {
          const int ci = c.nch - 1 - ch;
          w = L.cls[2 * ci] | L.cls[2 * ci + 1];
          if (ch == pc) w &= ~((2ull << (pick & 63)) - 1ull);
        }
        nrm += __popcll(w);
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) nrm += __shfl_xor(nrm, off, 64);
      // a shuffle result is lane-varying to the compiler: make the count a scalar, or
      // the absorb loop below is compiled as a divergent (exec-masked) loop
      above = (n - 1 - pick) - __builtin_amdgcn_readfirstlane(nrm);
    }
    int co2 = c.nch, consumed = 0;
    uint64_t om2 = 0, orm2 = 0, omm2 = 0;
    bool carry2 = false;
    double carry2_r = 0.0;
    int carry2_idx = 0;
    double fprob = p_pick;
    int falias = 0;
    for (;;) {
      while (om2 == 0ull && co2 > 0) {  // next overfull candidates: return | shared
        --co2;
        chunk_classes(c, L, co2, lane, sp_n, orm2, omm2);
        om2 = orm2 | omm2;
      }
      if (om2 == 0ull) {  // `overfull` empty
        if (carry2 && carry2_idx == pick) fprob = carry2_r;
        break;
      }
      const int lo = 63 - __clzll((long long)om2);
      om2 ^= 1ull << lo;
      double r = ((orm2 >> lo) & 1ull) ? vR : vM;
      const int o_idx = co2 * 64 + lo;
      if (carry2) {
        if (carry2_idx == pick) {
          fprob = carry2_r;
          falias = o_idx;
          break;
        }
        r = readfirstlane_f64(r + carry2_r - 1.0);
        carry2 = false;
        if (r < 1.0) {
          carry2 = true;
          carry2_r = r;
          carry2_idx = o_idx;
          continue;
        }
      }
      const int limit = __builtin_amdgcn_readfirstlane(min(total_u, above) - consumed);
      int j = 0;
      bool demoted = false;
      // probs[over] = probs[over] + probs[under] - 1.0, one absorbed slot per step.  Long runs
      // are skipped in closed form (absorb_skip); then, with vO < 1 the residual never
      // increases, so four steps can be taken at once and only the last one tested; the exact
      // exit step is resolved when it dropped below 1.0.
      if (limit >= 16) {
        absorb_skip(r, vO, j, limit);
        r = readfirstlane_f64(r);
        j = __builtin_amdgcn_readfirstlane(j);
      }
      while (j + 4 <= limit) {
        N2V_STAT(8, 4);
        const double a1 = r + vO - 1.0;
        const double a2 = a1 + vO - 1.0;
        const double a3 = a2 + vO - 1.0;
        const double a4 = a3 + vO - 1.0;
        if (!(a4 < 1.0)) {
          r = a4;
          j += 4;
          continue;
        }
        demoted = true;
        if (a1 < 1.0) {
          r = a1;
          j += 1;
        } else if (a2 < 1.0) {
          r = a2;
          j += 2;
        } else if (a3 < 1.0) {
          r = a3;
          j += 3;
        } else {
          r = a4;
          j += 4;
        }
        break;
      }
      while (!demoted && j < limit) {
        N2V_STAT(8, 1);
        r = r + vO - 1.0;
        ++j;
        if (r < 1.0) demoted = true;
      }
      r = readfirstlane_f64(r);
      consumed += j;
      if (demoted) {
        carry2 = true;
        carry2_r = r;
        carry2_idx = o_idx;
        continue;
      }
      if (consumed == above) {  // the next `under` is pick: alias[pick] = over
        fprob = vO;
        falias = o_idx;
      } else if (o_idx == pick) {  // `underfull` empty
        fprob = r;
      }
      break;
    }
    N2V_T(21);
    return (r2 < fprob) ? pick : falias;
  }
#endif
  // ---- pairing, run engine (every other class arrangement) -----------------------------
  // Both Python stacks are consumed in descending index order and their interleaving does
  // not matter, only each stack's own order.  With three class values a stack is a
  // sequence of RUNS of same-class slots; slot `pick` is always a run of its own.  Two
  // things happen, over and over (:182-189):
  //   absorb   the current `over` (r >= 1) takes a run of equal under values v < 1:
  //            r = r + v - 1 per slot, non-increasing, four slots per loop iteration;
  //   cascade  a demoted residual (the next `under`) meets a run of equal over values
  //            V in [1, 2): each slot becomes fl(V + a) - 1 and is demoted in turn until
  //            one settles at >= 1.  a, V - 1 and V are multiples of 2^-52 below 2, so
  //            V + a is exactly representable while it stays below 2, i.e. while the slot
  //            is demoted: slot i holds a1 + (i-1)*(V-1) EXACTLY.  The number of demoted
  //            slots follows from integer-valued arithmetic (all products stay below 2^53
  //            ulps), and the slot that settles is computed with the real operations again
  //            because its sum reaches [2, 3) where it may round.  A run costs O(1).
  {
    const int pc = pick >> 6;
    const uint64_t pbit = 1ull << (pick & 63);
    // A stream holds the run handed out last, consumed by count: `used` of its `cnt`
    // slots, highest index first.  Two ways to produce runs:
    //   by chunk  cm = candidates of the chunk being scanned that were not handed out yet;
    //             a run is a bit mask inside one chunk;
    //   by rank   when the stack holds ONE class (and every class ballot of the row is in
    //             LDS) all its slots have the same value, so the stack is one run from the
    //             top down to `pick`, `pick`, and one run below it: nothing but counts, and
    //             a rank -> neighbour index conversion once, for the alias that is returned.
    struct RunStream {
      int c;
      uint64_t cm, rm, mm;
      uint64_t run;
      int rc, cnt, used;
      double val, inv;
      bool is_pick;
      bool homog;
      int total, handed, rho, base;  // by rank: size, slots handed out, rank of pick (-1: none)
      int cid;                       // by chunk: class of the run `inv` was computed for
    };
    RunStream U{c.nch, 0ull, 0ull, 0ull, 0ull, 0, 0, 0, 0.0, 0.0, false, false, 0, 0, -1, 0, -1};
    RunStream O{c.nch, 0ull, 0ull, 0ull, 0ull, 0, 0, 0, 0.0, 0.0, false, false, 0, 0, -1, 0, -1};
    {
      const bool cR = nR > 0, cM = nM > 0, cO = nO > 0;
      const int n_under = (int)(cR && uR) + (int)(cM && uM) + (int)(cO && uO);
      const int n_over = (int)(cR && !uR) + (int)(cM && !uM) + (int)(cO && !uO);
      // every class ballot of the row is at hand: in LDS, or (reverse mode) by position
      const bool known = c.nch <= kUC || sp_n >= 0;
      U.homog = known && n_under == 1;
      O.homog = known && n_over == 1;
      U.total = (uR ? nR : 0) + (uM ? nM : 0) + (uO ? nO : 0);
      O.total = n - U.total;
      if (U.homog || O.homog) {
        // return / shared slots with a higher index than pick: one popcount pass
        int packed = 0;
        for (int base = max(pc, c.nch - kUC); base < c.nch; base += 64) {
          const int ch = base + lane;
          if (ch < c.nch) {
            const int ci = c.nch - 1 - ch;
            uint64_t wr = L.cls[2 * ci], wm = L.cls[2 * ci + 1];
            if (ch == pc) {
              const uint64_t keep = ~((2ull << (pick & 63)) - 1ull);
              wr &= keep;
              wm &= keep;
            }
            packed += __popcll(wr) + (__popcll(wm) << 16);
          }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) packed += __shfl_xor(packed, off, 64);
        packed = __builtin_amdgcn_readfirstlane(packed);
        int aR = packed & 0xffff, aM = packed >> 16;
        if (sp_n > 0) {  // the slots kept by position (all below the cached chunks)
          const int32_t e = lane < sp_n ? (int32_t)L.pool[lane] : 0;
          const bool hi = lane < sp_n && (e & 0x7fffffff) > pick;
          aR += __popcll(ballot64(hi && e < 0));
          aM += __popcll(ballot64(hi && e >= 0));
        }
        const int aO = (n - 1 - pick) - aR - aM;
        const int above_u = (uR ? aR : 0) + (uM ? aM : 0) + (uO ? aO : 0);
        const bool pick_under = p_pick < 1.0;
        if (U.homog) {
          U.rho = pick_under ? above_u : -1;
          U.val = pick3(cR && uR, cM && uM, vR, vM, vO);
        }
        if (O.homog) {
          O.rho = pick_under ? -1 : (n - 1 - pick) - above_u;
          O.val = pick3(cR && !uR, cM && !uM, vR, vM, vO);
          O.inv = readfirstlane_f64(1.0 / (O.val - 1.0));  // seeds the search for the demoted count
        }
      }
    }
    // next run of a stream: same class as the top candidate, down to (excluding) the next
    // candidate of another class or slot `pick`; `pick` itself is a run of one
    auto fetch = [&](RunStream &S, bool under) __attribute__((always_inline)) -> bool {
      if (S.used < S.cnt) return true;
      if (S.homog) {
        const int remaining = S.total - S.handed;
        if (remaining <= 0) return false;
        S.base = S.handed;
        S.is_pick = S.handed == S.rho;
        S.cnt = S.is_pick ? 1 : (S.handed < S.rho ? S.rho - S.handed : remaining);
        S.handed += S.cnt;
        S.used = 0;
        return true;
      }
      // a stack without the "other" class has candidates in few chunks: after one empty
      // chunk, look at the ballots of the next 64 cached chunks at once and jump
      const bool sparse = under ? !(uO && nO > 0) : (uO || nO == 0);
      bool looked = false;
      while (S.cm == 0ull && S.c > 0) {
        const int low_cached = max(c.nch - kUC, 0);
        if (looked && sparse && S.c > low_cached) {
          const int ch = S.c - 1 - lane;
          bool has = false;
          if (ch >= low_cached) {
            const int ci = c.nch - 1 - ch;
            const uint64_t wr = L.cls[2 * ci], wm = L.cls[2 * ci + 1];
            const uint64_t um = (uR ? wr : 0ull) | (uM ? wm : 0ull);
            has = (under ? um : ((wr | wm) & ~um)) != 0ull;
          }
          const uint64_t hit = ballot64(has);
          const int span = min(64, S.c - low_cached);
          S.c -= hit ? (int)__builtin_ctzll(hit) : span;
          if (!hit) continue;
        }
        looked = true;
        --S.c;
        N2V_STAT(10, 1);
        if (c.nch - 1 - S.c >= kUC) N2V_STAT(7, 1);
        chunk_classes(c, L, S.c, lane, sp_n, S.rm, S.mm);
        const uint64_t vm = valid_mask(c, S.c);
        const uint64_t um = (uR ? S.rm : 0ull) | (uM ? S.mm : 0ull) | (uO ? (vm & ~(S.rm | S.mm)) : 0ull);
        S.cm = under ? um : (vm & ~um);
      }
      if (S.cm == 0ull) return false;
      const int l = 63 - __clzll((long long)S.cm);
      const uint64_t lbit = 1ull << l;
      const uint64_t pk = S.c == pc ? pbit : 0ull;
      const bool isR = S.rm & lbit, isM = S.mm & lbit;
      // (a conditional on captured lvalues is an lvalue: clang then selects ADDRESSES and the
      // values are forced into scratch memory; pick3 takes them by value)
      S.val = pick3(isR, isM, vR, vM, vO);
      if (!under) {  // 1 / (V - 1) seeds the search for the demoted count: once per class
        const int cid = isR ? 0 : (isM ? 1 : 2);
        if (cid != S.cid) {
          S.cid = cid;
          S.inv = readfirstlane_f64(1.0 / (S.val - 1.0));
        }
      }
      S.is_pick = (pk & lbit) != 0ull;
      uint64_t run = lbit;
      if (!S.is_pick) {
        const uint64_t cmask = pick3(isR, isM, S.rm, S.mm, ~(S.rm | S.mm));
        run = S.cm & cmask & (lbit | (lbit - 1ull));
        const uint64_t stop = ((S.cm & ~cmask) | (S.cm & pk)) & (lbit - 1ull);
        if (stop) run &= ~((2ull << (63 - __clzll((long long)stop))) - 1ull);
      }
      S.run = run;
      S.cm &= ~run;
      S.rc = S.c;
      S.cnt = __popcll(run);
      S.used = 0;
      return true;
    };
    // neighbour index of the k-th highest slot of a run of `overfull` (k >= 1)
    auto kth_index = [&](uint64_t run, int rc, int k) __attribute__((always_inline)) -> int {
      if (O.homog) {
        // by rank: `run` is unused, rc is the rank of the run's first slot.  Find the
        // chunk that holds rank K (popcounts of 64 chunks at a time + a wave scan).
        const int K = rc + k - 1;
        int running = 0;
        const int ncached = c.nch < kUC ? c.nch : kUC;
        for (int base = 0; base < ncached; base += 64) {
          const int ci = base + lane;
          uint64_t mem = 0ull;
          if (ci < ncached) {
            const int ch = c.nch - 1 - ci;
            const uint64_t wr = L.cls[2 * ci], wm = L.cls[2 * ci + 1];
            const uint64_t vm = valid_mask(c, ch);
            const uint64_t um = (uR ? wr : 0ull) | (uM ? wm : 0ull) | (uO ? (vm & ~(wr | wm)) : 0ull);
            mem = vm & ~um;
          }
          const int mine = __popcll(mem);
          int incl = mine;
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
          }
          const int block = __builtin_amdgcn_readlane(incl, 63);
          if (K < running + block) {
            const int sel = (int)__builtin_ctzll(ballot64(running + incl > K));
            const int excl = __builtin_amdgcn_readlane(incl - mine, sel);
            const uint64_t m = readfirstlane_u64(__shfl(mem, sel, 64));
            const int kk = K - running - excl + 1;
            const uint64_t sh = m >> lane;
            const uint64_t hit = ballot64((sh & 1ull) && __popcll(sh) == kk);
            return (c.nch - 1 - (base + sel)) * 64 + (int)__builtin_ctzll(hit);
          }
          running += block;
        }
        // below the cached chunks (reverse mode): indices [0, lim) with the return/shared
        // slots known by position
        const int lim = (c.nch - kUC) * 64, kk = K - running;
        const int32_t e = lane < sp_n ? (int32_t)L.pool[lane] : 0;
        const int pos = e & 0x7fffffff;
        if (nO > 0 && !uO) {
          // the stack is the "other" class: the (kk+1)-th highest index that is not in the
          // list.  idx = lim-1-kk-#(listed >= idx), iterated downwards to its largest fixed point
          int idx = lim - 1 - kk;
          for (;;) {
            const int nidx = lim - 1 - kk - __popcll(ballot64(lane < sp_n && pos >= idx));
            if (nidx == idx) break;
            idx = nidx;
          }
          return idx;
        }
        // the stack is the return or the shared class: the (kk+1)-th highest listed position
        const bool want_r = nR > 0 && !uR;
        const bool match = lane < sp_n && ((e < 0) == want_r);
        int greater = 0;
        for (uint64_t b = ballot64(match); b != 0ull; b &= b - 1ull)
          greater += (int)(__builtin_amdgcn_readlane(pos, (int)__builtin_ctzll(b)) > pos);
        const uint64_t hit = ballot64(match && greater == kk);
        return __builtin_amdgcn_readlane(pos, (int)__builtin_ctzll(hit));
      }
      const uint64_t sh = run >> lane;
      const uint64_t hit = ballot64((sh & 1ull) && __popcll(sh) == k);
      return rc * 64 + (int)__builtin_ctzll(hit);
    };
    // The loop of :182-189 as two alternating phases.  An under value pu < 1 (the first
    // slot of `underfull`, later always the residual of the `over` demoted last) cascades
    // through `overfull` until a slot survives it; that slot, r >= 1, then absorbs slots of
    // `underfull` until it drops below 1 and becomes the next under value.
    double fprob = p_pick;
    int falias = 0;
    [&]() __attribute__((always_inline)) {
      if (!fetch(U, true)) return;  // not reached: both stacks are non-empty here
      double pu = U.val, r = 0.0;
      bool u_is_pick = U.is_pick, cur_is_pick = false;
      ++U.used;
      for (;;) {
        // ---- cascade
        int cur_rc, cur_k;
        uint64_t cur_run;
        for (;;) {
          if (!fetch(O, false)) {  // `overfull` empty: the under keeps its value
            if (u_is_pick) fprob = pu;
            return;
          }
          if (u_is_pick) {  // alias[pick] = the top of `overfull`
            fprob = pu;
            falias = kth_index(O.run, O.homog ? O.base : O.rc, O.used + 1);
            return;
          }
          N2V_STAT(9, 1);
          const double val = O.val;
          const int avail = O.cnt - O.used;
          int k;
          bool settled = true;
          const double a1 = val + pu - 1.0;  // slot 1, the reference's two operations
          if (!(a1 < 1.0)) {
            r = a1;
            k = 1;
          } else {
            const double d = val - 1.0;    // a1 < 1 implies val < 2: exact by Sterbenz
            const double need = 1.0 - a1;  // exact, > 0
            const double m1 = (double)(avail - 1);
            if (m1 * d < need) {  // even the last slot of the run is demoted
              k = avail;
              pu = a1 + m1 * d;  // exact, < 1
              settled = false;
            } else {
              // j = the first i >= 1 with a1 + i*d >= 1.0: seeded by a multiplication, then
              // fixed up with exact products (j*d stays below 2)
              double j = fmin(fmax(ceil(need * O.inv), 1.0), m1);
              while (j * d < need) j += 1.0;
              while (j >= 2.0 && (j - 1.0) * d >= need) j -= 1.0;
              k = (int)j + 1;                            // slot j+1 settles
              const double a_prev = a1 + (j - 1.0) * d;  // slot j: exact, < 1
              r = val + a_prev - 1.0;
            }
          }
          O.used += __builtin_amdgcn_readfirstlane(k);
          if (settled) break;
          pu = readfirstlane_f64(pu);
          u_is_pick = O.is_pick;
        }
        r = readfirstlane_f64(r);
        cur_is_pick = O.is_pick;  // the slot consumed last is the current `over`
        cur_run = O.run;
        cur_rc = O.homog ? O.base : O.rc;
        cur_k = O.used;
        // ---- absorb
        for (;;) {
          if (!fetch(U, true)) {  // `underfull` empty
            if (cur_is_pick) fprob = r;
            return;
          }
          if (U.is_pick) {  // alias[pick] = over; probs[pick] is final
            fprob = U.val;
            falias = kth_index(cur_run, cur_rc, cur_k);
            return;
          }
          const double val = U.val;
          int j = U.used;
          const int count = U.cnt;
          bool demoted = false;
          if (count - j >= 16) {
            absorb_skip(r, val, j, count);
            r = readfirstlane_f64(r);
            j = __builtin_amdgcn_readfirstlane(j);
          }
          while (j + 4 <= count) {
            N2V_STAT(8, 4);
            const double a1 = r + val - 1.0;
            const double a2 = a1 + val - 1.0;
            const double a3 = a2 + val - 1.0;
            const double a4 = a3 + val - 1.0;
            if (!(a4 < 1.0)) {
              r = a4;
              j += 4;
              continue;
            }
            demoted = true;
            if (a1 < 1.0) {
              r = a1;
              j += 1;
            } else if (a2 < 1.0) {
              r = a2;
              j += 2;
            } else if (a3 < 1.0) {
              r = a3;
              j += 3;
            } else {
              r = a4;
              j += 4;
            }
            break;
          }
          while (!demoted && j < count) {
            N2V_STAT(8, 1);
            r = r + val - 1.0;
            ++j;
            if (r < 1.0) demoted = true;
          }
          U.used = __builtin_amdgcn_readfirstlane(j);
          r = readfirstlane_f64(r);
          if (demoted) break;
        }
        pu = r;  // the demoted `over` is the next under
        u_is_pick = cur_is_pick;
      }
    }();
    N2V_T(21);
    return (r2 < fprob) ? pick : falias;
  }
}

// 8 waves per SIMD (<= 64 VGPRs) matches the 8 resident blocks the 20 KB of LDS allow
#ifndef N2V_UNIT_WAVES
#define N2V_UNIT_WAVES 8
#endif
template <bool kDyadic>
__global__ __launch_bounds__(kWavesPerBlock * 64, N2V_UNIT_WAVES) void walk_exact_unit_kernel(
    n2v_graph g, const int32_t *__restrict__ start_ids, int64_t n_start, int32_t num_walks,
    int32_t walk_length, double p, double q, UnitConsts K, uint64_t seed,
    int32_t *__restrict__ walks_out, uint8_t *__restrict__ valid_out,
    uint32_t *__restrict__ status) {
  __shared__ UnitLds lds_all[kWavesPerBlock];
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  UnitLds &L = lds_all[wave_in_block];
  const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t total = n_start * (int64_t)num_walks;
  const int L1 = walk_length + 1;
  const bool biased = !(p == 1.0 && q == 1.0);
  UnitStep c;
  c.need_mem = q != 1.0;
#ifdef N2V_STATS
  WaveStats WS;
  for (int i = 0; i < 40; ++i) WS.v[i] = 0;
  const unsigned long long t_kernel0 = __builtin_readcyclecounter();
#endif

  // Walkers differ a lot in cost (a walk that lingers among hubs is many times dearer than one
  // in the periphery), so waves take them from a shared counter (status[1], zero at launch)
  // instead of a fixed stride: no wave is left with a long queue while others idle.  Results
  // are addressed by walker row, so they do not depend on who walks what.
  // (short walks are fetched eight at a time: fewer atomics on the one counter)
  const bool dynamic = total < 0xfffffff0ll;
  const uint32_t grab = walk_length >= 16 ? 1u : 8u;
  int64_t rr = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block;
  uint32_t left = 0;
  for (;;) {
    if (dynamic) {
      if (left == 0) {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(&status[1], grab);
        rr = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        left = grab;
      } else {
        ++rr;
      }
      --left;
    }
    if (rr >= total) break;
    const int64_t r = readfirstlane_i64(rr);
    if (!dynamic) rr += n_waves;
#ifdef N2V_STATS
    const unsigned long long t_walker0 = __builtin_readcyclecounter();
#endif
    int32_t *out = walks_out + r * L1;
    // the path lives in registers (lane t holds vertices t and 64 + t) and is stored as
    // whole rows at the end; walks longer than 128 vertices fall back to direct stores
    const bool buffered = L1 <= 128;
    int32_t path0 = -1, path1 = -1;
    if (!buffered)
      for (int t = lane; t < L1; t += 64) out[t] = -1;
    const int32_t start = __builtin_amdgcn_readfirstlane(start_ids[r / num_walks]);
    const int32_t ordinal = (int32_t)(r % num_walks) + 1;
    bool alive = true;
    if (start < 0 || (int64_t)start >= g.n_vertices) {
      if (lane == 0) atomicOr(status, N2V_ST_RANGE);
      alive = false;
    }
    int32_t s = -1, v = start;
    if (alive) {
      const int64_t vb = readfirstlane_i64(g.rowptr[v]);
      const int64_t ve = readfirstlane_i64(g.rowptr[v + 1]);
      alive = ve > vb;  // fugue.py:132
    }
    if (alive) {
      const uint64_t key = (uint64_t)start * (uint64_t)num_walks + (uint64_t)(ordinal - 1);
      const uint64_t h0 = walker_stream(seed, key);
      __builtin_amdgcn_wave_barrier();
      if (buffered) {
        if (lane == 0) path0 = start;
      } else if (lane == 0) {
        out[0] = start;
      }
      int64_t sb = 0;
      int m = 0;
      for (int step = 0; step < walk_length; ++step) {
        const int64_t vb = readfirstlane_i64(g.rowptr[v]);
        const int64_t ve = readfirstlane_i64(g.rowptr[v + 1]);
        const int n = (int)(ve - vb);
        if (n == 0) {  // fugue.py:147: the walker vanishes at a sink
          alive = false;
          break;
        }
        const uint64_t bits = step_bits(h0, (uint32_t)step);
        int idx;
        if (s < 0 || !biased) {
          // uniform row: probs0 == 1.0 everywhere, no underfull slot, alias unused
          idx = pick_index((uint32_t)(bits >> 32), n);
        } else {
          c.vcol = g.col + vb;
          c.n = n;
          c.nch = (n + 63) >> 6;
          c.s = s;
          c.scol = g.col + sb;
          c.m = m;
          c.iters = 32 - __clz(m);
#ifdef N2V_STATS
          const unsigned long long t_s0 = __builtin_readcyclecounter();
#endif
          idx = unit_draw<kDyadic>(c, K, (uint32_t)(bits >> 32), (uint32_t)bits, lane, L N2V_STATS_PASS);
#ifdef N2V_STATS
          {
            const unsigned long long dt = __builtin_readcyclecounter() - t_s0;
            const int bk = n <= 64 ? 0 : (n <= 1024 ? 1 : (n <= 4096 ? 2 : (n <= 8192 ? 3 : 4)));
            WS.v[25 + bk] += dt;
            if (bk >= 3) WS.v[14 + bk - 3] += 1;  // 14: 4096 < n <= 8192, 15: larger
          }
#endif
        }
        const int32_t next = __builtin_amdgcn_readfirstlane(g.col[vb + idx]);
        if (buffered) {
          const int t = step + 1;
          if (lane == (t & 63)) {
            if (t < 64)
              path0 = next;
            else
              path1 = next;
          }
        } else if (lane == 0) {
          out[step + 1] = next;
        }
        s = v;  // the row of the new previous vertex is the row just walked
        sb = vb;
        m = n;
        v = next;
      }
    }
    if (buffered) {
      if (lane < L1) out[lane] = path0;
      if (64 + lane < L1) out[64 + lane] = path1;
    }
    if (lane == 0) valid_out[r] = alive ? 1 : 0;
#ifdef N2V_STATS
    {  // slowest walker and when the wave took it (tail diagnostics)
      const unsigned long long tw = __builtin_readcyclecounter() - t_walker0;
      if (lane == 0) atomicMax(&n2v_stats[30], tw);
      WS.v[31] += tw;
    }
#endif
  }
#ifdef N2V_STATS
  WS.v[23] = __builtin_readcyclecounter() - t_kernel0;
  if (lane == 0)
    for (int i = 0; i < 40; ++i) atomicAdd(&n2v_stats[i], WS.v[i]);
#endif
}

// ---- one step of the walkers resident on one part of a partitioned graph -------------------
// (n2v_partition_step, n2v_walk.hip: the same contract; this is its unit-weight instance.)  N(v)
// comes from the part's CSR, N(s) from the rows that travelled with the walkers; the draw is
// unit_draw above, the routine n2v_walk runs on the whole graph, hence the same vertex.
// (Rows travel here.  When wedge lists travel instead the step is per-lane work:
// partition_step_wedge_kernel, n2v_walk_wedge.hip.)
template <bool kDyadic>
__global__ __launch_bounds__(kWavesPerBlock * 64, N2V_UNIT_WAVES) void partition_step_unit_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, int64_t lo,
    int64_t n_local, const int64_t *__restrict__ head, int head_cols,
    const int64_t *__restrict__ src_ptr, const int32_t *__restrict__ src_ids,
    int64_t k, double p, double q, UnitConsts K, uint64_t seed, int32_t *__restrict__ next_out,
    int64_t *__restrict__ edge_out, uint32_t *__restrict__ status) {
  __shared__ UnitLds lds_all[kWavesPerBlock];
  const int lane = threadIdx.x & 63;
  UnitLds &L = lds_all[threadIdx.x >> 6];
  const bool biased = !(p == 1.0 && q == 1.0);
  UnitStep c;
  c.need_mem = q != 1.0;
#ifdef N2V_STATS
  WaveStats WS;
  for (int i = 0; i < 40; ++i) WS.v[i] = 0;
#endif
  ItemQueue queue(k, kWavesPerBlock);
  for (;;) {
    const int64_t i = queue.next(&status[1], lane);
    if (i < 0) break;
    const int64_t *hd = head + i * head_cols;
    const uint64_t key = (uint64_t)readfirstlane_i64(hd[1]);
    const int64_t sv = readfirstlane_i64(hd[2]);
    const uint32_t step = (uint32_t)readfirstlane_i64(hd[3]);
    const int32_t s = (int32_t)(sv >> 32);
    const int64_t local = (int64_t)(uint32_t)sv - lo;
    int32_t next = -1;
    int64_t edge = -1;
    if (readfirstlane_i64(hd[0]) < 0) {
      // an empty slot of a capacity-bounded mailbox (negative output row): nothing to step
    } else if (local < 0 || local >= n_local) {  // a walker that is not resident here
      if (lane == 0) atomicOr(status, N2V_ST_RANGE);
    } else {
      const int64_t vb = readfirstlane_i64(rowptr[local]);
      const int n = (int)(readfirstlane_i64(rowptr[local + 1]) - vb);
      if (n > 0) {  // (arrivals at a sink were dropped by the caller, fugue.py:147)
        const uint64_t bits = step_bits(walker_stream(seed, key), step);
        int idx = -1;
        if (s < 0 || !biased) {
          // uniform row: probs0 == 1.0 everywhere, no underfull slot, alias unused
          idx = pick_index((uint32_t)(bits >> 32), n);
        } else {
          c.vcol = col + vb;
          c.n = n;
          c.nch = (n + 63) >> 6;
          c.s = s;
          c.scol = col;
          c.m = 1;
          c.iters = 1;
          bool ok = true;
          if (c.need_mem) {
            const int64_t sb = readfirstlane_i64(src_ptr[i]);
            c.m = (int)(readfirstlane_i64(src_ptr[i + 1]) - sb);
            c.scol = src_ids + sb;
            c.iters = 32 - __clz(c.m > 0 ? c.m : 1);
            ok = c.m > 0;  // the previous vertex had out-edges: its row must have travelled
            if (!ok && lane == 0) atomicOr(status, N2V_ST_RANGE);
          }
          if (ok)
            idx = unit_draw<kDyadic>(c, K, (uint32_t)(bits >> 32), (uint32_t)bits, lane, L N2V_STATS_PASS);
        }
        if (idx >= 0) {
          next = __builtin_amdgcn_readfirstlane(col[vb + idx]);
          edge = vb + idx;
        }
      }
    }
    if (lane == 0) {
      next_out[i] = next;
      if (edge_out) edge_out[i] = edge;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- lanes kernel: one LANE per walker, the wave only for the pairing -------------------
// With the class counts of every edge at hand (n2v_edge_classes_build) a step (s -> v) needs
// no classification of N(v) unless the pairing loop (:182-189) has to run for slot `pick`:
//   avg     = (nR / p + nM + nO / q) / n              from edge_classes[e], e = the edge walked last
//   probs0  = the class value of slot `pick` / avg     one membership test "N(v)[pick] in N(s)"
//   exit 1  probs0 < 1 and r2 < probs0 -> pick         (an underfull slot that is accepted never
//                                                       changes, same exit as unit_draw)
//   exit 2  one of the two stacks is empty -> pick or 0 (the loop of :182 never runs)
// Those exits decide ~80 % of the steps of the BASELINE graphs, and they are per-lane work:
// four to six dependent gathers, no LDS, no wave-wide scan.  So 64 walkers share a wave and
// advance in lock-step; the lanes whose step is not decided are served one after the other by
// the whole wave running unit_draw (the routine of the wave-per-walker kernel, unchanged: same
// bits).  Dyadic p, q only (the row sum is then an integer combination of the counts); p == q
// == 1 needs no counts at all (every step is `pick`).
#ifndef N2V_LANES_WAVES
#define N2V_LANES_WAVES 5
#endif
// kHops: the hop table (n2v_hops_build) is at hand -- the entry that names the next vertex also
// carries its row pointer, its degree and the class counts of the edge just walked (needed at
// the next step), so the quick phase of a step is ONE 16-byte gather plus the membership test.
template <bool kHops>
__global__ __launch_bounds__(kWavesPerBlock * 64, N2V_LANES_WAVES) void walk_exact_unit_lanes_kernel(
    n2v_graph g, const int32_t *__restrict__ start_ids, int64_t n_start, int32_t num_walks,
    int32_t walk_length, double p, double q, UnitConsts K, uint64_t seed,
    int32_t *__restrict__ walks_out, uint8_t *__restrict__ valid_out,
    uint32_t *__restrict__ status) {
  __shared__ UnitLds lds_all[kWavesPerBlock];
  const int lane = threadIdx.x & 63;
  UnitLds &L = lds_all[threadIdx.x >> 6];
  const int64_t total = n_start * (int64_t)num_walks;
  const int L1 = walk_length + 1;
  const bool biased = !(p == 1.0 && q == 1.0);
  const bool need_mem = q != 1.0;
  // wedge table (n2v_wedge_build): the class of every slot of a step's table by position
  const bool have_w = g.wedge_off != nullptr && g.wedge_pos != nullptr;
#ifdef N2V_CHECK
  n2v_check_status = status;
  const int dbg = g.reserved;  // bit 0: no per-lane pairing, 1: no list classification, 2: no list membership
#else
  constexpr int dbg = 0;
#endif
  UnitStep c;
  c.need_mem = need_mem;
#ifdef N2V_STATS
  WaveStats WS;
  for (int i = 0; i < 40; ++i) WS.v[i] = 0;
  const unsigned long long t_kernel0 = __builtin_readcyclecounter();
#endif
  for (;;) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(&status[1], 64u);
    const int64_t base = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    if (base >= total) break;
    const int64_t r = base + lane;
    const bool have = r < total;
    int32_t start = -1;
    uint64_t h0 = 0;
    bool alive = have;
    if (have) {
      start = start_ids[r / num_walks];
      const int32_t ordinal = (int32_t)(r % num_walks) + 1;
      h0 = walker_stream(seed, (uint64_t)start * (uint64_t)num_walks + (uint64_t)(ordinal - 1));
      if (start < 0 || (int64_t)start >= g.n_vertices) {
        atomicOr(status, N2V_ST_RANGE);
        alive = false;
      }
    }
    int64_t vb = 0, sb = 0, e_prev = 0;
    uint32_t ec_prev = 0;  // kHops: class counts of the edge (s -> v), from the hop that walked it
    int n = 0, m = 0;
    if (alive) {
      vb = g.rowptr[start];
      n = (int)(g.rowptr[start + 1] - vb);
      alive = n > 0;  // fugue.py:132
    }
    int32_t *out = walks_out + r * L1;
    if (have) out[0] = alive ? start : -1;
    const bool started = alive;
    int32_t s = -1, v = start;
    bool walking = alive;
    for (int step = 0; step < walk_length; ++step) {
      if (ballot64(walking) == 0ull) break;
      int idx = 0;
      int32_t x = -1;
      bool unresolved = false;
      uint32_t u1 = 0, u2 = 0;
      n2v_hop h;
      h.col = -1;
      h.classes = 0;
      h.row = 0;
      int64_t w_off = 0;
      int w_rpos = 0, w_nR = 0, w_nM = 0;
      bool w_ok = false;
#ifdef N2V_STATS
      const unsigned long long t_q0 = __builtin_readcyclecounter();
#endif
      if (walking) {
        const uint64_t bits = step_bits(h0, (uint32_t)step);
        u1 = (uint32_t)(bits >> 32);
        u2 = (uint32_t)bits;
        const int pick = pick_index(u1, n);
        idx = pick;
        const bool w_wide = wedge_row_wide(g.wedge_wide, n);  // (a mixed table: by the row stood on)
        // the class counts of this step's table are known before any load (they came with the
        // hop that walked the edge), so the offset of the edge's wedge list is requested FIRST and
        // travels together with the hop gather: two independent loads, one latency
        const bool step_biased = s >= 0 && biased;
        uint32_t fR = 0, fM = 0;
        if (step_biased) {
          const uint32_t ec = kHops ? ec_prev : g.edge_classes[e_prev];
          fR = ec >> N2V_EC_RETURN_SHIFT;
          fM = ec & N2V_EC_SHARED_MASK;
        }
        const bool counts_ok = step_biased && fR != N2V_EC_RETURN_SAT && fM != N2V_EC_SHARED_MASK;
        uint64_t wraw = 0;
        if (have_w && counts_ok && ((need_mem && fM > 0) || fR > 0)) {
          N2V_CHECK_RANGE(3, e_prev, (int64_t)0, g.n_edges);
          wraw = g.wedge_off[e_prev];  // list offset | return position << 40
        }
        if (kHops) {
          h = load_hop(g.hops + vb + pick);
          x = h.col;
        } else {
          x = g.col[vb + pick];
        }
        if (step_biased) {
          if (!counts_ok) {
            unresolved = true;  // a count that did not fit: classify the row
          } else {
            const int nR = (int)fR, nM = need_mem ? (int)fM : 0, nO = n - nR - nM;
            const int64_t isum = (int64_t)nR * K.TR + (int64_t)nM * K.TM + (int64_t)nO * K.TO;
            const double avg = ((double)isum * (1.0 / 1048576.0)) / (double)n;  // :172
            const bool isR = x == s;
            bool isM = false;
            w_off = (int64_t)(wraw & N2V_WEDGE_OFF_MASK);
            w_rpos = (int)(wraw >> N2V_WEDGE_RPOS_SHIFT);
            int lo_pick = 0;  // entries of the edge's list below `pick`
            if (need_mem && !isR && nM > 0) {  // :226
              if (have_w && !(dbg & 4))
                lo_pick = wedge_lower(g.wedge_pos, w_off, nM, pick, w_wide, isM);
              else
                isM = member_sorted_lane(g.col + sb, m, x);
            }
            const double p_pick = pick3(isR, isM, K.bR, K.bM, K.bO) / avg;      // :173
            const double r2 = (double)u2 * (1.0 / 4294967296.0);
            if (!(p_pick < 1.0 && r2 < p_pick)) {
              const double vR = K.bR / avg, vM = K.bM / avg, vO = K.bO / avg;
              const bool uR = vR < 1.0, uM = vM < 1.0, uO = vO < 1.0;
              const bool any_under = (nR && uR) || (nM && uM) || (nO && uO);
              const bool any_over = (nR && !uR) || (nM && !uM) || (nO && !uO);
              int jres = -1;
              if (any_under && any_over && have_w && uO && !(nR && uR) && !(nM && uM) &&
                  !(dbg & (16 | 4))) {
                // plain branches on the (uniform) list width: never a select between two loads
                if (w_wide)
                  jres = lane_case_a_jump<uint32_t>(
                      n, pick, r2, K, nR, w_rpos, nM,
                      reinterpret_cast<const uint32_t *>(g.wedge_pos) + w_off, isR, isM, lo_pick);
                else
                  jres = lane_case_a_jump<uint16_t>(
                      n, pick, r2, K, nR, w_rpos, nM,
                      reinterpret_cast<const uint16_t *>(g.wedge_pos) + w_off, isR, isM, lo_pick);
              }
              if (!any_under || !any_over) {  // the loop of :182 never runs
                if (!(r2 < p_pick)) idx = 0;
              } else if (jres >= 0) {
                idx = jres;  // closed form: no loop at all
                N2V_CHECK_RANGE(7, idx, 0, n);
              } else if (have_w && n <= 64 && !(dbg & 1)) {
                // a short row: this lane replays the pairing itself from the two class masks
                uint64_t Rm = 0ull, Mm = 0ull;
                if (nR) Rm = ((nR >= 64) ? ~0ull : ((1ull << nR) - 1ull)) << w_rpos;
                const uint64_t Mm_list = wedge_mask(g.wedge_pos, w_off, nM, w_wide);
                Mm = Mm_list;
                idx = lane_pairing(n, Rm, Mm, pick, r2, vR, vM, vO);
                N2V_CHECK_RANGE(2, idx, 0, n);
              } else if (have_w && uO && !(nR && uR) && !(nM && uM) && !(dbg & 8)) {
                // a longer row whose only underfull class is "other": still this lane's work
                if (w_wide)
                  idx = lane_case_a<uint32_t>(n, pick, r2, vR, vM, vO, nR, w_rpos, nM,
                                              reinterpret_cast<const uint32_t *>(g.wedge_pos) + w_off,
                                              isR, isM, reinterpret_cast<uint32_t *>(L.cls), lane);
                else
                  idx = lane_case_a<uint16_t>(n, pick, r2, vR, vM, vO, nR, w_rpos, nM,
                                              reinterpret_cast<const uint16_t *>(g.wedge_pos) + w_off,
                                              isR, isM, reinterpret_cast<uint16_t *>(L.cls), lane);
                N2V_CHECK_RANGE(6, idx, 0, n);
              } else {
                unresolved = true;
                w_nR = nR;
                w_nM = nM;
                w_ok = have_w && !(dbg & 2);
              }
              if (!unresolved && idx != pick) {
                if (kHops) {
                  h = load_hop(g.hops + vb + idx);
                  x = h.col;
                } else {
                  x = g.col[vb + idx];
                }
              }
            }
          }
        }
      }
      // the steps that need the pairing: the whole wave, one walker at a time
      uint64_t fb = ballot64(unresolved);
#ifdef N2V_STATS
      const unsigned long long t_f0 = __builtin_readcyclecounter();
      WS.v[26] += t_f0 - t_q0;
      WS.v[28] += __popcll(ballot64(walking));
      WS.v[29] += __popcll(fb);
#endif
      while (fb != 0ull) {
        const int l = (int)__builtin_ctzll(fb);
        fb &= fb - 1ull;
        const int64_t vb_l = readfirstlane_i64(__shfl(vb, l, 64));
        const int64_t sb_l = readfirstlane_i64(__shfl(sb, l, 64));
        c.vcol = g.col + vb_l;
        c.scol = g.col + sb_l;
        c.n = __builtin_amdgcn_readlane(n, l);
        c.nch = (c.n + 63) >> 6;
        c.m = __builtin_amdgcn_readlane(m, l);
        c.iters = 32 - __clz(c.m);
        c.s = __builtin_amdgcn_readlane(s, l);
        c.w_have = __builtin_amdgcn_readlane((int)w_ok, l) != 0;
        c.w_wide = wedge_row_wide(g.wedge_wide, c.n);
        c.w_nR = __builtin_amdgcn_readlane(w_nR, l);
        c.w_nM = __builtin_amdgcn_readlane(w_nM, l);
        c.w_rpos = __builtin_amdgcn_readlane(w_rpos, l);
        {
          const int64_t wo = readfirstlane_i64(__shfl(w_off, l, 64));
          c.w_pos = c.w_wide ? (const void *)(reinterpret_cast<const uint32_t *>(g.wedge_pos) + wo)
                           : (const void *)(reinterpret_cast<const uint16_t *>(g.wedge_pos) + wo);
        }
        const uint32_t u1_l = (uint32_t)__builtin_amdgcn_readlane((int)u1, l);
        const uint32_t u2_l = (uint32_t)__builtin_amdgcn_readlane((int)u2, l);
#ifdef N2V_STATS
        const unsigned long long t_d0 = __builtin_readcyclecounter();
#endif
        const int res = __builtin_amdgcn_readfirstlane(unit_draw<true>(c, K, u1_l, u2_l, lane, L N2V_STATS_PASS));
        __builtin_amdgcn_wave_barrier();
        if (lane == l) {
          idx = res;
          N2V_CHECK_RANGE(1, idx, 0, n);
        }
#ifdef N2V_STATS
        {  // fallback draws and their cycles by deg(v) bucket
          const int bk = c.n <= 64 ? 0 : (c.n <= 1024 ? 1 : (c.n <= 4096 ? 2 : 3));
          WS.v[32 + bk] += 1;
          WS.v[36 + bk] += __builtin_readcyclecounter() - t_d0;
        }
#endif
      }
#ifdef N2V_STATS
      WS.v[27] += __builtin_readcyclecounter() - t_f0;
#endif
      if (unresolved) {
        if (kHops) {
          h = load_hop(g.hops + vb + idx);
          x = h.col;
        } else {
          x = g.col[vb + idx];
        }
      }
      if (walking) {
        out[step + 1] = x;
        e_prev = vb + idx;
        ec_prev = h.classes;
        s = v;  // the row of the new previous vertex is the row just walked
        sb = vb;
        m = n;
        v = x;
        if (step + 1 < walk_length) {
          if (kHops) {
            vb = hop_row(h);
            n = hop_deg(h);
          } else {
            vb = g.rowptr[v];
            n = (int)(g.rowptr[v + 1] - vb);
          }
          if (n == 0) {  // fugue.py:147: the walker vanishes at a sink
            walking = false;
            alive = false;
            for (int tt = step + 2; tt < L1; ++tt) out[tt] = -1;
          }
        }
      }
    }
    if (have) {
      if (!started)  // no such vertex / no out-edges: the row is all -1, like the other kernels
        for (int tt = 1; tt < L1; ++tt) out[tt] = -1;
      valid_out[r] = alive ? 1 : 0;
    }
  }
#ifdef N2V_STATS
  WS.v[23] = __builtin_readcyclecounter() - t_kernel0;
  if (lane == 0)
    for (int i = 0; i < 40; ++i) atomicAdd(&n2v_stats[i], WS.v[i]);
#endif
}

}  // namespace n2v

// true when x * 2^20 is an exact integer in [0, 2^31): the integer-sum argument holds
static bool scales_exactly(double x, int64_t *t_out) {
  const double t = x * 1048576.0;
  if (!(t >= 0.0) || !(t < 2147483648.0) || t != (double)(int64_t)t) return false;
  *t_out = (int64_t)t;
  return true;
}

// the per-(p, q) constants of the unit-weight kernels; false = the pair is outside their range
static bool unit_consts(double p, double q, n2v::UnitConsts &K, bool &dyadic) {
  K.bR = 1.0 / p;  // the reference's weight / return_param with weight == 1.0
  K.bM = 1.0;
  K.bO = 1.0 / q;
  K.TR = K.TM = K.TO = 0;
  // dyadic 1/p, 1/q: the row sum is an integer combination of three counts; otherwise it is
  // added up in the reference's order, run by run (needs ordinary magnitudes)
  dyadic = scales_exactly(K.bR, &K.TR) && scales_exactly(K.bM, &K.TM) &&
           scales_exactly(K.bO, &K.TO) && K.TR != 0 && K.TO != 0;
  K.gR = K.TR;
  K.gM = K.TM;
  K.gO = K.TO;
  if (dyadic) {
    int64_t a = K.TR, b = K.TM;
    while (b) { const int64_t t = a % b; a = b; b = t; }
    b = K.TO;
    while (b) { const int64_t t = a % b; a = b; b = t; }
    if (a > 0) {
      K.gR = K.TR / a;
      K.gM = K.TM / a;
      K.gO = K.TO / a;
    }
  }
  K.fR = (double)K.gR;
  K.fM = (double)K.gM;
  K.fO = (double)K.gO;
  K.dyadic = dyadic ? 1 : 0;
  const bool ordinary = K.bR >= 0x1p-20 && K.bR <= 0x1p20 && K.bO >= 0x1p-20 && K.bO <= 0x1p20;
  return dyadic || ordinary;
}

int n2v_edge_row_sums_launch(const n2v_graph *g, const n2v::UnitConsts &K, double q, const int64_t *edges, int64_t k,
                             double *sums_out, void *stream);

extern "C" int n2v_edge_row_sums_build(const n2v_graph *g, double p, double q, const int64_t *edges, int64_t k,
                                       double *sums_out, void *stream) {
  if (!g || k < 0 || p == 0.0 || q == 0.0) return N2V_EINVAL;
  if (k == 0) return N2V_OK;
  if (g->w || g->w64 || !g->rowptr || !g->col || !g->edge_classes || !g->wedge_off || !g->wedge_pos || !edges ||
      !sums_out || g->wedge_wide < 0 || g->wedge_wide > 65536)
    return N2V_EINVAL;
  n2v::UnitConsts K;
  bool dyadic = false;
  if (!unit_consts(p, q, K, dyadic)) return N2V_EINVAL;
  return n2v_edge_row_sums_launch(g, K, q, edges, k, sums_out, stream);
}

// returns 1 when the unit-weight kernel applies (and was launched), 0 when the caller
// must use the generic kernel, < 0 on error
// workspace n2v_walk_ws can use for (g, p, q): the record lists of n2v_walk_wedge2.hip, 0 = none
extern "C" int64_t n2v_walk_exact_unit_workspace(const n2v_graph *g, int64_t total,
                                                 int32_t walk_length, double p, double q) {
  if (g->w != nullptr || g->w64 != nullptr || (p == 1.0 && q == 1.0)) return 0;
  if (!g->hops || !g->wedge_off || !g->wedge_pos || (g->reserved & 1)) return 0;
  n2v::UnitConsts K;
  bool dyadic = false;
  if (!unit_consts(p, q, K, dyadic) || !dyadic) return 0;
  if (total <= 0 || total >= 0xffffff00ll || walk_length >= 0xfffff0) return 0;
#ifdef N2V_WITH_WEDGE2
  return n2v_walk_wedge2_workspace(total);
#else
  return 0;  // the passes over a workspace (n2v_walk_wedge2.hip) are not part of this build: `make WEDGE2=1`
#endif
}

extern "C" int n2v_walk_exact_unit_try(const n2v_graph *g, const int32_t *start_ids,
                                       int64_t n_start, int32_t num_walks,
                                       int32_t walk_length, double p, double q, uint64_t seed,
                                       int32_t *walks_out, uint8_t *valid_out, uint32_t *status,
                                       void *workspace, int64_t workspace_bytes, void *stream) {
  if (g->w != nullptr || g->w64 != nullptr) return 0;
  n2v::UnitConsts K;
  bool dyadic = false;
  if (!unit_consts(p, q, K, dyadic)) return 0;
  const int64_t total = n_start * (int64_t)num_walks;
  if (total == 0) return 1;
  // status[1] is the kernels' walker counter: start it at zero on the same stream
  if (hipMemsetAsync(status + 1, 0, sizeof(uint32_t), (hipStream_t)stream) != hipSuccess)
    return N2V_ELAUNCH;
  // lanes kernel: dyadic p, q with the per-edge class counts at hand (p == q == 1 needs none),
  // when the "other" class -- nearly every slot of a row -- is the underfull one: 1/q <= 1 and
  // 1/p >= 1/q.  Then ~80 % of the steps leave through the per-lane exits.  For the other
  // arrangements (q < 1, or p > q) the usual `pick` is an overfull slot, every step needs the
  // pairing, and one wave per walker (below) is the better shape (measured: cfg 2, p = 4,
  // q = 0.25: 344 against 241 M steps/s; p = 2, q = 1: 938 against 711).
  const bool lanes_regime = (p == 1.0 && q == 1.0) || (K.bO <= 1.0 && K.bR >= K.bO);
  // every dyadic (p, q) has its pairing in closed form (n2v_unit_core.h): "other" alone on its
  // stack -- the only underfull class (1/q <= 1, 1/p >= 1/q) or the only overfull one (1/q >= 1,
  // 1/p <= 1/q; with p == q the return slot is an "other" slot) --, or sharing it with the return
  // run (q > 1 with p > q; q < 1 with p < q)
  // (values that are not dyadic: the same kernel adds the row up in the reference's order and
  // replays every pairing run by run)
  if (!(p == 1.0 && q == 1.0) && !(g->reserved & 1)) {
    // every per-edge table is at hand and the caller lent a workspace: closed forms in the main
    // launches, declined steps replayed out of line (n2v_walk_wedge2.hip)
#ifdef N2V_WITH_WEDGE2  // (measured slower than the one-launch kernel on every BASELINE graph: a build option)
    if (workspace) {
      int rounds = 4;
      if (const char *e = getenv("N2V_WEDGE2_ROUNDS")) rounds = atoi(e);
      const int r2 = n2v_walk_wedge2_try(g, start_ids, n_start, num_walks, walk_length, p, q, K, seed,
                                         walks_out, valid_out, status, workspace, workspace_bytes,
                                         rounds, stream);
      if (r2 != 0) return r2;
    }
#else
    (void)workspace;
    (void)workspace_bytes;
#endif
    // every per-edge table is at hand: the kernel in which no step needs the wave
    const int rw = n2v_walk_wedge_try(g, start_ids, n_start, num_walks, walk_length, p, q, K, seed,
                                      walks_out, valid_out, status, stream);
    if (rw != 0) return rw;
  }
  // (a hop table with inline return positions is read by the slots kernel alone)
  if (g->hops && (g->reserved2 & N2V_HOPS_INLINE_RPOS) && !(p == 1.0 && q == 1.0)) return N2V_EINVAL;
  if (dyadic && lanes_regime && total < 0xffffff00ll &&
      (g->edge_classes || g->hops || (p == 1.0 && q == 1.0))) {
    int64_t blocks = (total + 255) / 256;
    const void *lfn = g->hops ? (const void *)n2v::walk_exact_unit_lanes_kernel<true>
                              : (const void *)n2v::walk_exact_unit_lanes_kernel<false>;
    const int64_t cap = n2v::resident_blocks(lfn, n2v::kWavesPerBlock * 64, 0);
    if (blocks > cap) blocks = cap;
    if (g->hops)
      hipLaunchKernelGGL(n2v::walk_exact_unit_lanes_kernel<true>, dim3((unsigned)blocks),
                         dim3(n2v::kWavesPerBlock * 64), 0, (hipStream_t)stream, *g, start_ids,
                         n_start, num_walks, walk_length, p, q, K, seed, walks_out, valid_out,
                         status);
    else
      hipLaunchKernelGGL(n2v::walk_exact_unit_lanes_kernel<false>, dim3((unsigned)blocks),
                         dim3(n2v::kWavesPerBlock * 64), 0, (hipStream_t)stream, *g, start_ids,
                         n_start, num_walks, walk_length, p, q, K, seed, walks_out, valid_out,
                         status);
    if (hipGetLastError() != hipSuccess) return N2V_ELAUNCH;
    return 1;
  }
  // persistent grid: exactly the resident capacity, walkers are grid-strided
  int64_t blocks = (total + n2v::kWavesPerBlock - 1) / n2v::kWavesPerBlock;
  const void *fn = dyadic ? (const void *)n2v::walk_exact_unit_kernel<true>
                          : (const void *)n2v::walk_exact_unit_kernel<false>;
  const int64_t cap = n2v::resident_blocks(fn, n2v::kWavesPerBlock * 64, 0);
  if (blocks > cap) blocks = cap;
  if (dyadic)
    hipLaunchKernelGGL(n2v::walk_exact_unit_kernel<true>, dim3((unsigned)blocks),
                       dim3(n2v::kWavesPerBlock * 64), 0, (hipStream_t)stream, *g, start_ids,
                       n_start, num_walks, walk_length, p, q, K, seed, walks_out, valid_out,
                       status);
  else
    hipLaunchKernelGGL(n2v::walk_exact_unit_kernel<false>, dim3((unsigned)blocks),
                       dim3(n2v::kWavesPerBlock * 64), 0, (hipStream_t)stream, *g, start_ids,
                       n_start, num_walks, walk_length, p, q, K, seed, walks_out, valid_out,
                       status);
  if (hipGetLastError() != hipSuccess) return N2V_ELAUNCH;
  return 1;
}

// the unit-weight instance of n2v_partition_step (n2v_walk.hip calls it first): 1 = launched,
// 0 = (p, q) outside the unit kernels' range, < 0 on error.  status[1] was zeroed by the caller.
extern "C" int n2v_partition_step_unit_try(const int64_t *rowptr, const int32_t *col, int64_t lo,
                                           int64_t n_local, const int64_t *head, int32_t head_cols,
                                           const int64_t *src_ptr, const int32_t *src_ids,
                                           int32_t wedge_lists, int64_t k, double p, double q,
                                           uint64_t seed, int32_t *next_out, int64_t *edge_out,
                                           uint32_t *status, void *stream) {
  n2v::UnitConsts K;
  bool dyadic = false;
  if (!unit_consts(p, q, K, dyadic)) return 0;
  if (wedge_lists)  // one lane per walker, the closed forms of the wedge kernel
    return n2v_partition_step_wedge_launch(rowptr, col, lo, n_local, head, head_cols, src_ptr,
                                           src_ids, wedge_lists == 2, k, p, q, K, seed, next_out,
                                           edge_out, status, stream);
  int64_t blocks = (k + n2v::kWavesPerBlock - 1) / n2v::kWavesPerBlock;
#define N2V_PART_LAUNCH(D)                                                                        \
  do {                                                                                           \
    const int64_t cap = n2v::resident_blocks((const void *)n2v::partition_step_unit_kernel<D>,   \
                                             n2v::kWavesPerBlock * 64, 0);                       \
    if (blocks > cap) blocks = cap;                                                              \
    hipLaunchKernelGGL(n2v::partition_step_unit_kernel<D>, dim3((unsigned)blocks),               \
                       dim3(n2v::kWavesPerBlock * 64), 0, (hipStream_t)stream, rowptr, col, lo,  \
                       n_local, head, (int)head_cols, src_ptr, src_ids, k, p, q, K, seed,        \
                       next_out, edge_out, status);                                              \
  } while (0)
  if (dyadic)
    N2V_PART_LAUNCH(true);
  else
    N2V_PART_LAUNCH(false);
#undef N2V_PART_LAUNCH
  if (hipGetLastError() != hipSuccess) return N2V_ELAUNCH;
  return 1;
}

#ifdef N2V_STATS
extern "C" int n2v_debug_stats_unit(unsigned long long *out_host, int reset) {
  if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(n2v::n2v_stats), sizeof(unsigned long long) * 40) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[40] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(n2v::n2v_stats), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
