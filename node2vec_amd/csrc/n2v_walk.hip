// n2v_walk.hip -- K2, the p/q-biased second-order walk sampler for gfx950.
//
// Replaces, per walker and per step, the body of next_step_random_walk
// (reference randomwalk.py:316-339): decode both neighbour lists, apply the
// p/q bias (generate_edge_alias_tables, :193-232), rebuild the Walker alias
// table (generate_alias_tables, :157-190), draw with two uniforms
// (sampling_from_alias, :86-99), append (RandomPath.append, :123-153) -- and the
// loop around it (fugue.py:137-153).
//
// EXACT mode design (one wave64 per walker, walker resident for all L steps):
//   pass 1  lanes stream N(v) = {col, w} in 64-element chunks (coalesced),
//           classify each neighbour x as return (x == s) / shared (x in N(s),
//           binary search over the sorted row of s) / other, form the biased
//           weight b in fp64 and sum it.  The reference sums left to right in
//           fp64; when every b is a multiple of 2^-20 below 2^11 every partial
//           sum is exactly representable, so any order gives the same bits and
//           the wave reduces in int64; otherwise it falls back to a serial
//           left-to-right fp64 sum.  Class ballots are cached in LDS.
//   draw    pick = floor(r1 * n).  The reference only ever looks at alias[pick]
//           and the final probs[pick], so the table is never materialised:
//           an untouched underfull slot is decided at once; otherwise
//   pass 2  the LIFO pairing of :182-189 is replayed as two descending streams
//           (underfull / overfull candidates, 64 per refill, consumed through
//           ballots + v_readlane), with the same fp64 operations in the same
//           order, and stops as soon as slot `pick` has been paired.
#include "n2v_alias_core.h"

namespace n2v {

constexpr int kBqCap = 512;       // scaled biased weights of the LAST kBqCap neighbours

struct WaveLds {
  uint64_t cls[2 * kLdsChunks];  // per 64-neighbour chunk: ballot(return), ballot(shared)
  int32_t bq[kBqCap];            // b * 2^20 at position n-1-i (valid in exact-sum mode)
  uint32_t bits[kBitWordsMax];   // hashed id filter of N(s)
  int32_t mlist[kMaybeCap];      // indices into N(v) that hit the filter
  uint8_t flags[64];             // lst_chunk_mask scratch (zero between calls)
};

struct SumState {
  int64_t isum;
  bool exact;
  double b_pick, bmin, bmax;
  // (round 5) the row sum when its additions are exact in ANY order: every addend a multiple of
  // G = 2^e_min (the lowest set bit of any addend) and the whole sum below 2^53 G -- then every partial sum
  // of the reference's left-to-right loop is representable, nothing is ever rounded, and the sum of the
  // lanes' partial sums is the reference's sum bit for bit.  fp32 weights with p, q powers of two (24-bit
  // addends), integer and short decimal weights: the 10^3 - 10^4 serial additions of a step (a third of
  // this kernel's cycles, profiles/r03g_walk_stats_generic_cfg2.log) become one wave reduction.
  double psum;
  int e_min;     // min over the addends > 0 of the exponent of their lowest set bit
  bool grid_ok;  // no addend negative / infinite / NaN
  bool track;    // wave-uniform: still worth tracking (the test can only get harder as the row goes by)
};

__device__ __forceinline__ void account(SumState &st, WaveLds &L, const StepCtx &c, int i,
                                        bool act, double b, int lane, int pick) {
  const double t = b * 1048576.0;
  const bool ok = (t >= 0.0) && (t < 2147483648.0) && (t == trunc(t));
  st.exact = st.exact && (ok || !act);
  st.isum += (ok && act) ? (int64_t)t : 0;
  if (act && st.track) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(b);
    const int ex = (int)((bits >> 52) & 0x7ffull);
    if ((bits >> 63) != 0ull || ex == 0x7ff) {
      st.grid_ok = st.grid_ok && b == 0.0;  // (-0.0 is a harmless addend)
    } else if (b != 0.0) {
      const unsigned long long man = (bits & 0xfffffffffffffull) | (ex ? (1ull << 52) : 0ull);
      const int e_lsb = (ex ? ex : 1) - 1075 + (int)__builtin_ctzll(man);
      st.e_min = e_lsb < st.e_min ? e_lsb : st.e_min;
      st.psum += b;
    }
  }
  if (act) {
    const int pos = c.n - 1 - i;
    if (pos < kBqCap) L.bq[pos] = ok ? (int32_t)t : 0;
    st.bmin = fmin(st.bmin, b);
    st.bmax = fmax(st.bmax, b);
  }
  const uint64_t pm = ballot64(act && i == pick);
  if (pm) st.b_pick = readlane_f64(b, __ffsll((long long)pm) - 1);
}

// are the additions of this row's sum exact in any order?  (wave-uniform answer)
__device__ __forceinline__ bool sum_any_order(const SumState &st, int n) {
  if (!st.track || ballot64(!st.grid_ok) != 0ull) return false;
  int e_min = st.e_min;
  double bmax = st.bmax;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int eo = __shfl_xor(e_min, off, 64);
    e_min = eo < e_min ? eo : e_min;
    bmax = fmax(bmax, __shfl_xor(bmax, off, 64));
  }
  if (e_min == 0x7fffffff) return true;        // every addend is 0
  if (e_min < -1000 || !(bmax < 0x1p1000)) return false;
  // n addends of at most bmax: the sum stays below 2^52 G (one bit to spare)
  return (double)n * bmax < ldexp(1.0, 52 + e_min);
}

// exact membership for the filter hits collected in L.mlist[0, count)
__device__ __forceinline__ void verify_maybes(const StepCtx &c, WaveLds &L, int count, int lane,
                                              int pick, SumState &st) {
  for (int k = 0; k < count; k += 64) {
    const bool act = k + lane < count;
    const int i = act ? L.mlist[k + lane] : 0;
    const int32_t x = act ? c.vcol[i] : -1;
    const double wt = act ? weight_at(c, i) : 0.0;
    const bool mem = member_sorted(c.scol, c.m, x, c.iters) && act;
    const double b = mem ? wt : wt / c.q;  // :226-230
    if (mem && (i >> 6) < kLdsChunks)
      atomicOr(reinterpret_cast<unsigned long long *>(&L.cls[2 * (i >> 6) + 1]),
               1ull << (i & 63));
    account(st, L, c, i, act, b, lane, pick);
  }
}

// probs0 of chunk `chunk` (w_biased / avg); from the LDS cache when it covers the chunk
__device__ __forceinline__ double load_vals(const StepCtx &c, WaveLds &L, int chunk, int lane,
                                            bool bq_valid, double avg, bool &valid) {
  const int i = chunk * 64 + lane;
  if (bq_valid && c.n - 1 - chunk * 64 < kBqCap) {  // wave-uniform: whole chunk cached
    valid = i < c.n;
    const int pos = c.n - 1 - i;
    const double b = valid ? (double)L.bq[pos] * (1.0 / 1048576.0) : 0.0;
    return b / avg;
  }
  return chunk_bias<true>(c, chunk, lane, L.cls, valid) / avg;
}

// The while-loop of generate_alias_tables (:182-189), replayed until alias[pick] and
// probs[pick] are final.  Equivalent formulation of the two Python stacks: the top
// of `overfull` is always the current `over` (it is re-pushed while >= 1.0), and an
// `over` that drops below 1.0 is pushed on `underfull` and is therefore the very
// next `under`.  So: outer loop = one overfull slot per iteration (descending
// index), inner loop = that slot absorbing underfull slots (descending index) with
// the reference's two fp64 operations, until it is demoted (carry).  Candidates
// come 64 at a time from ballots over a chunk of probs0 = b / avg.
template <bool kCached>
__device__ __forceinline__ int pairing(const StepCtx &c, WaveLds &L, int lane, int pick,
                                       double avg, double p_pick, double r2, bool bq_valid N2V_STATS_ARG) {
  const int n = c.n;
  // Rows longer than the LDS cache re-read their weights from HBM/L2.  The loop
  // below is serial, so that latency would be fully exposed at every refill: each
  // stream therefore keeps the NEXT chunk's weights in flight (wnext) while the
  // current 64 candidates are consumed.
  auto fetch_w = [&](int chunk) -> double {
    const int i = chunk * 64 + lane;
    return (chunk >= 0 && i < n) ? weight_at(c, i) : 0.0;
  };
  auto load_s = [&](int chunk, double &wnext, int &wnext_chunk, bool &valid) -> double {
    const int i = chunk * 64 + lane;
    if (kCached || (bq_valid && n - 1 - chunk * 64 < kBqCap)) {  // wave-uniform
      valid = i < n;
      const double b = valid ? (double)L.bq[n - 1 - i] * (1.0 / 1048576.0) : 0.0;
      return b / avg;
    }
    if (c.need_cls && chunk >= kLdsChunks)  // classes not cached: search again
      return chunk_bias<true>(c, chunk, lane, L.cls, valid) / avg;
    valid = i < n;
    const double wf = (wnext_chunk == chunk) ? wnext : fetch_w(chunk);
    wnext = fetch_w(chunk - 1);
    wnext_chunk = chunk - 1;
    const double wt = valid ? wf : 0.0;
    double b = wt;
    if (c.need_cls) {
      const bool is_ret = (L.cls[2 * chunk] >> lane) & 1ull;
      const bool is_mem = (L.cls[2 * chunk + 1] >> lane) & 1ull;
      if (is_ret)
        b = wt / c.p;
      else if (!(is_mem || !c.need_mem))
        b = wt / c.q;
    }
    return b / avg;
  };
  double wu_next = 0.0, wo_next = 0.0;
  int wu_chunk = -2, wo_chunk = -2;
  int cu = c.nch, co = c.nch;
  uint64_t um = 0, om = 0;
  double uval = 0.0, oval = 0.0;
  bool carry = false;
  double carry_r = 0.0;
  int carry_idx = 0;
  double fin_prob = p_pick;
  int fin_alias = 0;
  for (;;) {
    N2V_STAT(9, 1);
    while (om == 0ull && co > 0) {  // next overfull candidates
      --co;
      bool valid;
      oval = load_s(co, wo_next, wo_chunk, valid);
      om = ballot64(valid && !(oval < 1.0));
    }
    if (om == 0ull) {  // `overfull` empty: a demoted slot keeps alias 0
      if (carry && carry_idx == pick) fin_prob = carry_r;
      break;
    }
    const int lo = 63 - __clzll((long long)om);
    om ^= 1ull << lo;
    double r = readlane_f64(oval, lo);
    const int o_idx = co * 64 + lo;
    if (carry) {  // under = the slot demoted last iteration
      if (carry_idx == pick) {
        fin_prob = carry_r;
        fin_alias = o_idx;
        break;
      }
      r = readfirstlane_f64(r + carry_r - 1.0);
      carry = false;
      if (r < 1.0) {
        carry = true;
        carry_r = r;
        carry_idx = o_idx;
        continue;
      }
    }
    bool finished = false;
    for (;;) {  // `over` absorbs underfull slots
      while (um == 0ull && cu > 0) {
        --cu;
        bool valid;
        uval = load_s(cu, wu_next, wu_chunk, valid);
        um = ballot64(valid && uval < 1.0);
      }
      if (um == 0ull) {  // `underfull` empty
        if (o_idx == pick) fin_prob = r;
        finished = true;
        break;
      }
      // Candidates of this chunk that come BEFORE slot `pick` (higher index) are absorbed in
      // a minimal loop: find bit, v_readlane, the reference's two fp64 operations, compare.
      // Whether `pick` is the next candidate is decided once per chunk, not per slot.
      const int pl = pick & 63;
      const bool stop_here = (pick >> 6) == cu && ((um >> pl) & 1ull);
      uint64_t run = stop_here ? (um & ~((2ull << pl) - 1ull)) : um;
      um &= ~run;
      bool demoted = false;
      while (run != 0ull) {
        N2V_STAT(8, 1);
        const int l = 63 - __clzll((long long)run);
        run ^= 1ull << l;
        r = r + readlane_f64(uval, l) - 1.0;  // probs[over] = probs[over] + probs[under] - 1.0
        if (r < 1.0) {
          demoted = true;
          break;
        }
      }
      um |= run;  // candidates not consumed yet
      r = readfirstlane_f64(r);
      if (demoted) {  // it is the next `under`
        carry = true;
        carry_r = r;
        carry_idx = o_idx;
        break;
      }
      if (stop_here) {  // alias[under] = over; probs[under] is final
        fin_prob = readlane_f64(uval, pl);
        fin_alias = o_idx;
        finished = true;
        break;
      }
    }
    if (finished) break;
  }
  return (r2 < fin_prob) ? pick : fin_alias;  // :95-99
}

// Index drawn by sampling_from_alias(r1, r2) on the table that
// generate_edge_alias_tables would build.  Returns -1 on ZeroDivisionError.
__device__ __forceinline__ int exact_draw(const StepCtx &c, uint32_t u1, uint32_t u2,
                                          int lane, WaveLds &L N2V_STATS_ARG) {
  const int n = c.n;
  const int pick = pick_index(u1, n);  // int(r1 * n), r1 = u1 / 2^32
  const double r2 = (double)u2 * (1.0 / 4294967296.0);

  N2V_T0
  // ---- pass 0: hashed-id filter of N(s) in LDS ---------------------------------
  // Membership "x in N(s)" (:226) costs a dependent chain of ~log2(m) gathers per
  // 64 neighbours when searched directly.  Instead every y of N(s) sets one bit;
  // pass 1 tests one bit per x; only the hits (true members + a few false
  // positives) are verified by binary search, batched once per step.
#if defined(N2V_ABLATE) && (N2V_ABLATE & 8)  // timing-only: no filter, no search at all
  const bool use_filter = false;
#else
  const bool use_filter = c.lst == nullptr && c.need_mem && c.m <= 4096 && c.m <= 8 * n + 64;
#endif
  int shift = 32;
  if (use_filter) {
    int words = 64;
    while (words < kBitWordsMax && words * 32 < 16 * c.m) words <<= 1;
    shift = 32 - (5 + (31 - __clz(words)));
    for (int wv = lane; wv < words; wv += 64) L.bits[wv] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int yb = 0; yb < c.m; yb += 256) {  // 4 loads in flight per lane
      int32_t y[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = yb + u * 64 + lane;
        y[u] = j < c.m ? c.scol[j] : -1;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (yb + u * 64 + lane < c.m) {
          const uint32_t h = hash_id(y[u], shift);
          atomicOr(&L.bits[h >> 5], 1u << (h & 31));
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  N2V_T(16);
  N2V_STAT(0, 1); N2V_STAT(1, use_filter ? 1 : 0); N2V_STAT(2, (c.need_mem && !use_filter) ? 1 : 0);
  // ---- pass 1: stream N(v): classify, bias, sum ----------------------------------
  SumState st;
  st.isum = 0;
  st.exact = true;
  st.psum = 0.0;
  st.e_min = 0x7fffffff;
  st.grid_ok = true;
  st.track = true;
  st.b_pick = 0.0;
  st.bmin = __builtin_huge_val();
  st.bmax = -__builtin_huge_val();
  int mcount = 0;
  int lm_fwd = 0;  // cursor into the shared-position list (table classes)
  const bool tables = c.lst != nullptr && c.need_cls;
  constexpr int kU = 4;  // chunks per iteration: 2 * kU global loads in flight per lane
  for (int chunk0 = 0; chunk0 < c.nch; chunk0 += kU) {
    double wf[kU];
    int32_t xs[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int i = (chunk0 + u) * 64 + lane;
      const bool valid = i < n;
      wf[u] = valid ? weight_at(c, i) : 0.0;
      xs[u] = (valid && c.need_cls && !tables) ? c.vcol[i] : -1;
    }
    bool memv[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) memv[u] = false;
    uint64_t tmask[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) tmask[u] = 0ull;
    if (tables && c.need_mem) {
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (chunk0 + u < c.nch) tmask[u] = lst_chunk_mask(c, chunk0 + u, lane, lm_fwd, L.flags);
    }
    if (c.need_mem && !use_filter && !tables) {
#if !(defined(N2V_ABLATE) && (N2V_ABLATE & 2))
      member_sorted_x4(c.scol, c.m, xs, c.iters, memv);  // 4 interleaved searches
#endif
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int chunk = chunk0 + u;
      if (chunk >= c.nch) break;  // wave-uniform
      const int i = chunk * 64 + lane;
      const bool valid = i < n;
      const double wt = wf[u];
      bool is_ret = false, is_mem = false, maybe = false;
      if (tables) {
        is_ret = valid && i >= c.rpos && i < c.rpos + c.n_ret;
        is_mem = valid && !is_ret && ((tmask[u] >> lane) & 1ull);
        if (chunk < kLdsChunks) {
          const uint64_t rm = ballot64(is_ret), mm = ballot64(is_mem);
          if (lane == 0) {
            L.cls[2 * chunk] = rm;
            L.cls[2 * chunk + 1] = mm;
          }
        }
      } else if (c.need_cls) {
        const int32_t x = xs[u];
        is_ret = valid && x == c.s;
        if (use_filter) {
          const uint32_t h = hash_id(x, shift);
          maybe = valid && !is_ret && ((L.bits[h >> 5] >> (h & 31)) & 1u);
        } else if (c.need_mem) {
          is_mem = memv[u] && valid && !is_ret;
        }
        if (chunk < kLdsChunks) {
          const uint64_t rm = ballot64(is_ret), mm = ballot64(is_mem);
          if (lane == 0) {
            L.cls[2 * chunk] = rm;
            L.cls[2 * chunk + 1] = mm;
          }
        }
      }
      double b;
      if (is_ret)
        b = wt / c.p;  // :223-224
      else if (is_mem || !c.need_mem)
        b = wt;        // :226-227 (and q == 1)
      else
        b = wt / c.q;  // :229-230
      account(st, L, c, i, valid && !maybe, b, lane, pick);
      const uint64_t mm = ballot64(maybe);
      if (mm) {
        const int cnt = __popcll(mm);
        if (mcount + cnt > kMaybeCap) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          verify_maybes(c, L, mcount, lane, pick, st);
          mcount = 0;
        }
        if (maybe) L.mlist[mcount + __popcll(mm & ((1ull << lane) - 1ull))] = i;
        mcount += cnt;
      }
    }
    // full-width addends (w / q with q not a power of two, fp64 weights): hopeless after the first chunks --
    // e_min only falls and bmax only grows -- so the bookkeeping stops (a uniform branch)
    if (chunk0 == 0 && st.track && n > 256) st.track = sum_any_order(st, n);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  N2V_T(17);
#if !(defined(N2V_ABLATE) && (N2V_ABLATE & 4))  // timing-only: no verification of filter hits
  if (mcount) verify_maybes(c, L, mcount, lane, pick, st);
#endif
  N2V_STAT(3, mcount); N2V_STAT(4, (mcount + 63) / 64);
  N2V_T(18);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // ---- sum in the reference's order (:172) ---------------------------------------
  const bool bq_valid = ballot64(!st.exact) == 0ull && n <= (1 << 21);
  double total, b_pick = st.b_pick;
  if (bq_valid) {
    // every b is a multiple of 2^-20 below 2^11: all partial sums are exactly
    // representable, any order gives the reference's bits
    // (a shuffle reduction is lane-varying to the compiler: make the sum a scalar so that
    // avg, the shortcuts and the pairing loops below stay wave-uniform)
    total = (double)readfirstlane_i64(wave_sum_i64(st.isum)) * (1.0 / 1048576.0);
  } else if (sum_any_order(st, n)) {
    double ps = st.psum;  // exact additions: any tree gives the reference's bits
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ps += __shfl_xor(ps, off, 64);
    total = readfirstlane_f64(ps);
  } else {
    total = 0.0;  // left to right, one rounding per add
    for (int chunk = 0; chunk < c.nch; ++chunk) {
      bool valid;
      const double b = chunk_bias<true>(c, chunk, lane, L.cls, valid);
      const int cnt = min(64, n - chunk * 64);
      for (int j = 0; j < cnt; ++j) total = total + readlane_f64(b, j);
    }
  }
  const double avg = readfirstlane_f64(total / (double)n);  // :172
  if (avg == 0.0) return -1;
  const double p_pick = readfirstlane_f64(b_pick / avg);    // :173

  N2V_T(19);
  // untouched underfull slot: probs[pick] never changes, alias irrelevant
  if (p_pick < 1.0 && r2 < p_pick) return pick;
  N2V_STAT(5, 1);
#if defined(N2V_ABLATE) && (N2V_ABLATE & 1)  // timing-only build: no pairing pass
  return pick;
#endif
  // x -> x / avg is monotone, so the extreme weights decide whether either stack
  // is empty; then the pairing loop (:182) never runs: alias stays 0, probs stay.
#if defined(N2V_ABLATE) && (N2V_ABLATE & 64)  // timing-only: stop before the min/max shortcut
  return pick;
#endif
  double bmin = st.bmin, bmax = st.bmax;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    bmin = fmin(bmin, __shfl_xor(bmin, off, 64));
    bmax = fmax(bmax, __shfl_xor(bmax, off, 64));
  }
  N2V_T(20);
  bmin = readfirstlane_f64(bmin);
  bmax = readfirstlane_f64(bmax);
  if (!(bmin / avg < 1.0) || (bmax / avg < 1.0)) return (r2 < p_pick) ? pick : 0;

  N2V_STAT(6, 1); N2V_STAT(7, n <= 64 ? 1 : 0); N2V_STAT(10, (bq_valid && n <= kBqCap) ? 0 : 1);
  // ---- pass 2: LIFO pairing (:182-189) until slot `pick` is final ------------
#if defined(N2V_ABLATE) && (N2V_ABLATE & 16)  // timing-only: run the pairing twice
  if (bq_valid && n <= kBqCap) {
    int tmp = pairing<true>(c, L, lane, pick, avg, p_pick, r2, bq_valid N2V_STATS_PASS);
    asm volatile("" ::"v"(tmp));
  }
#endif
  if (bq_valid && n <= kBqCap) {
    const int res = pairing<true>(c, L, lane, pick, avg, p_pick, r2, bq_valid N2V_STATS_PASS);
    N2V_T(21);
    return res;
  }
#if defined(N2V_ABLATE) && (N2V_ABLATE & 32)  // timing-only: no pairing on uncached rows
  return pick;
#endif
  const int res2 = pairing<false>(c, L, lane, pick, avg, p_pick, r2, bq_valid N2V_STATS_PASS);
  N2V_T(22);
  return res2;
}

// the per-edge tables of the edge walked last into the step context (wave-uniform), when the graph has them
__device__ __forceinline__ void step_tables(StepCtx &c, const n2v_graph &g, int64_t e_prev) {
  c.lst = nullptr;
  if (!c.need_cls || !g.edge_classes || !g.wedge_off || !g.wedge_pos || e_prev < 0 || e_prev >= g.n_edges) return;
  const uint32_t ec = (uint32_t)__builtin_amdgcn_readfirstlane((int)g.edge_classes[e_prev]);
  const uint32_t fR = ec >> N2V_EC_RETURN_SHIFT, fM = ec & N2V_EC_SHARED_MASK;
  if (fR == N2V_EC_RETURN_SAT || fM == N2V_EC_SHARED_MASK) return;  // a saturated count: classes by search
  const uint64_t wraw = readfirstlane_u64(g.wedge_off[e_prev]);
  const int rpos = (int)(wraw >> N2V_WEDGE_RPOS_SHIFT);
  if ((int64_t)fR + (int64_t)fM > c.n || rpos + (int)fR > c.n) return;  // not a table of this row
  const uint64_t off = wraw & N2V_WEDGE_OFF_MASK;
  c.lst_wide = wedge_row_wide(g.wedge_wide, c.n) ? 1 : 0;
  c.lst = c.lst_wide ? (const void *)(reinterpret_cast<const uint32_t *>(g.wedge_pos) + off)
                     : (const void *)(reinterpret_cast<const uint16_t *>(g.wedge_pos) + off);
  c.lst_n = (int)fM;
  c.rpos = rpos;
  c.n_ret = (int)fR;
}

__global__ __launch_bounds__(kWavesPerBlock * 64, 7) void walk_exact_kernel(
    n2v_graph g, const int32_t *__restrict__ start_ids, int64_t n_start, int32_t num_walks,
    int32_t walk_length, double p, double q, uint64_t seed, int32_t *__restrict__ walks_out,
    uint8_t *__restrict__ valid_out, uint32_t *__restrict__ status) {
  __shared__ WaveLds lds_all[kWavesPerBlock];
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  WaveLds &L = lds_all[wave_in_block];
  if (lane < 16) reinterpret_cast<uint32_t *>(L.flags)[lane] = 0u;
  const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t total = n_start * (int64_t)num_walks;
  const int L1 = walk_length + 1;

  StepCtx c;
  c.p = p;
  c.q = q;
#ifdef N2V_STATS
  WaveStats WS;
  for (int i = 0; i < 40; ++i) WS.v[i] = 0;
  const unsigned long long t_kernel0 = __builtin_readcyclecounter();
#endif

  // walkers are taken from a shared counter (status[1], zero at launch), not by a fixed
  // stride: their costs differ widely and no wave should sit on a long queue while others idle
  // (n2v_walk_unit.hip); results are addressed by walker row
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
    int32_t *out = walks_out + r * L1;
    // the path lives in registers (lane t holds vertices t and 64 + t) and is stored as
    // whole rows at the end; walks longer than 128 vertices fall back to direct stores
    const bool buffered = L1 <= 128;
    int32_t path0 = -1, path1 = -1;
    if (!buffered)
      for (int t = lane; t < L1; t += 64) out[t] = -1;
    const int32_t start = __builtin_amdgcn_readfirstlane(start_ids[r / num_walks]);
    const int32_t ordinal = (int32_t)(r % num_walks) + 1;  // randomwalk.py:294
    bool alive = true;
    if (start < 0 || (int64_t)start >= g.n_vertices) {
      if (lane == 0) atomicOr(status, N2V_ST_RANGE);
      alive = false;
    }
    int32_t s = -1, v = start;
    int64_t e_prev = -1;  // the edge walked last (its per-edge tables say which slots are which)
    if (alive) {
      // fugue.py:132: only vertices with an adjacency row start walks
      int64_t vb = readfirstlane_i64(g.rowptr[v]);
      int64_t ve = readfirstlane_i64(g.rowptr[v + 1]);
      alive = ve > vb;
    }
    if (alive) {
      const uint64_t key = (uint64_t)start * (uint64_t)num_walks + (uint64_t)(ordinal - 1);
      const uint64_t h0 = walker_stream(seed, key);
      __builtin_amdgcn_wave_barrier();
      if (buffered) {
        if (lane == 0) path0 = start;
      } else if (lane == 0) {
        out[0] = start;
      }  // path[0] after the first-step rule (:146-147)
      for (int step = 0; step < walk_length; ++step) {
        const int64_t vb = readfirstlane_i64(g.rowptr[v]);
        const int64_t ve = readfirstlane_i64(g.rowptr[v + 1]);
        const int n = (int)(ve - vb);
        if (n == 0) {  // fugue.py:147 inner join on dst: the walker vanishes
          alive = false;
          break;
        }
        c.vcol = g.col + vb;
        c.vw = g.w ? g.w + vb : nullptr;
        c.vw64 = g.w64 ? g.w64 + vb : nullptr;
        c.n = n;
        c.nch = (n + 63) >> 6;
        c.s = s;
        const bool first = s < 0;  // randomwalk.py:320-321: unbiased table
        c.need_cls = !first && !(p == 1.0 && q == 1.0);
        c.need_mem = c.need_cls && q != 1.0;
        c.scol = g.col;
        c.m = 1;
        c.iters = 1;
        if (c.need_mem) {
          const int64_t sb = readfirstlane_i64(g.rowptr[s]);
          const int64_t se = readfirstlane_i64(g.rowptr[s + 1]);
          c.scol = g.col + sb;
          c.m = (int)(se - sb);
          c.iters = 32 - __clz(c.m);
        }
        step_tables(c, g, e_prev);
        const uint64_t bits = step_bits(h0, (uint32_t)step);
        const int idx = exact_draw(c, (uint32_t)(bits >> 32), (uint32_t)bits, lane, L N2V_STATS_PASS);
        if (idx < 0) {
          if (lane == 0) atomicOr(status, N2V_ST_ZERODIV);
          alive = false;
          break;
        }
        e_prev = vb + idx;
        const int32_t next = __builtin_amdgcn_readfirstlane(c.vcol[idx]);
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
        s = v;  // :339 src = path[-2], dst = path[-1]
        v = next;
      }
    }
    if (buffered) {
      if (lane < L1) out[lane] = path0;
      if (64 + lane < L1) out[64 + lane] = path1;
    }
    if (lane == 0) valid_out[r] = alive ? 1 : 0;
  }
#ifdef N2V_STATS
  WS.v[23] = __builtin_readcyclecounter() - t_kernel0;
  if (lane == 0)
    for (int i = 0; i < 40; ++i) atomicAdd(&n2v_stats[i], WS.v[i]);
#endif
}


// ---- one step of a batch of MIGRATING walkers (graph-partitioned walking) --------
// The reference joins every walker row with the adjacency row of its current vertex and carries
// the row of its previous vertex along (fugue.py:146-149), then calls next_step_random_walk on
// the joined row.  This is that call for the walkers resident on one part of a vertex-range
// partition: N(v) is read from the part's own CSR (rows [lo, lo + n_local)), N(s) from the
// packed rows that travelled with the walkers, and the draw is exact_draw above -- the same
// table, the same uniforms (keyed by walker and step, not by where the walker is), hence the
// same vertex as n2v_walk on the whole graph.  One wave per walker, taken from a counter.
// head: int64 [k, 4] = (output row, RNG key, s << 32 | v, step); s == -1 on the first step.
__global__ __launch_bounds__(kWavesPerBlock * 64, 7) void partition_step_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ w, const double *__restrict__ w64, int64_t lo, int64_t n_local,
    const int64_t *__restrict__ head, int head_cols, const int64_t *__restrict__ src_ptr,
    const int32_t *__restrict__ src_ids, int64_t k, double p, double q, uint64_t seed,
    int32_t *__restrict__ next_out, int64_t *__restrict__ edge_out,
    uint32_t *__restrict__ status) {
  __shared__ WaveLds lds_all[kWavesPerBlock];
  const int lane = threadIdx.x & 63;
  WaveLds &L = lds_all[threadIdx.x >> 6];
  if (lane < 16) reinterpret_cast<uint32_t *>(L.flags)[lane] = 0u;
#ifdef N2V_STATS
  WaveStats WS;
  for (int i = 0; i < 40; ++i) WS.v[i] = 0;
#endif
  StepCtx c;
  c.p = p;
  c.q = q;
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
      const bool first = s < 0;  // randomwalk.py:320-321: unbiased table
      c.vcol = col + vb;
      c.vw = w ? w + vb : nullptr;
      c.vw64 = w64 ? w64 + vb : nullptr;
      c.n = n;
      c.nch = (n + 63) >> 6;
      c.s = s;
      c.need_cls = !first && !(p == 1.0 && q == 1.0);
      c.need_mem = c.need_cls && q != 1.0;
      c.scol = col;
      c.m = 1;
      c.iters = 1;
      bool ok = n > 0;  // (arrivals at a sink were dropped by the caller, fugue.py:147)
      if (ok && c.need_mem) {
        const int64_t sb = readfirstlane_i64(src_ptr[i]);
        c.m = (int)(readfirstlane_i64(src_ptr[i + 1]) - sb);
        c.scol = src_ids + sb;
        c.iters = 32 - __clz(c.m > 0 ? c.m : 1);
        if (c.m <= 0) {  // the previous vertex had out-edges: its row must have travelled
          if (lane == 0) atomicOr(status, N2V_ST_RANGE);
          ok = false;
        }
      }
      if (ok) {
        const uint64_t bits = step_bits(walker_stream(seed, key), step);
        const int idx = exact_draw(c, (uint32_t)(bits >> 32), (uint32_t)bits, lane, L N2V_STATS_PASS);
        if (idx < 0) {
          if (lane == 0) atomicOr(status, N2V_ST_ZERODIV);
        } else {
          next = __builtin_amdgcn_readfirstlane(c.vcol[idx]);
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

// ---- one step of the walkers of a batch that stand on LONG rows (n2v_walk_weighted_step: the
// lane-per-walker kernel of n2v_walk_wlanes.hip takes the rows of up to `min_n` slots, a lane that
// walked a row of 10^4 slots alone would be the launch's tail).  The walkers come ordered by the length
// of the row they stand on, descending; waves take them from a counter (status[1], zero at launch) and a
// wave leaves when it meets a row that is the lanes'.  The draw is exact_draw above: N(s) read from the
// graph, the classes by membership search -- no per-edge table needed.
__global__ __launch_bounds__(kWavesPerBlock * 64, 7) void weighted_step_wave_kernel(
    n2v_graph g, const int32_t *__restrict__ start_ids, int32_t num_walks,
    const int64_t *__restrict__ order, int64_t n_rows, int32_t min_n, int32_t step, int32_t walk_length,
    double p, double q, uint64_t seed, int64_t *__restrict__ edge_state, int32_t *__restrict__ walks,
    uint8_t *__restrict__ valid, uint32_t *__restrict__ status) {
  __shared__ WaveLds lds_all[kWavesPerBlock];
  const int lane = threadIdx.x & 63;
  WaveLds &L = lds_all[threadIdx.x >> 6];
  if (lane < 16) reinterpret_cast<uint32_t *>(L.flags)[lane] = 0u;
#ifdef N2V_STATS
  WaveStats WS;
  for (int i = 0; i < 40; ++i) WS.v[i] = 0;
#endif
  const int L1 = walk_length + 1;
  StepCtx c;
  c.p = p;
  c.q = q;
  // min_n < 0: `order` is a LIST with its length in front of it (order[-1]; n2v_walk_weighted_step: what the margin
  // kernels left undecided -- nothing, as a rule).  A wave with no entry of its own leaves before it touches the shared
  // counter: 4 096 waves taking one look each were 4 096 same-address atomics, 48 us of every step whatever the batch
  // (profiles/r10h_kernel_stats_47k_walkers.csv).
  if (min_n < 0) {
    const int64_t listed = readfirstlane_i64(order[-1]);
    const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (wave >= listed) return;
    min_n = 0;
  }
  for (;;) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(&status[1], 1u);
    const int64_t i = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    if (i >= n_rows) break;
    const int64_t r = readfirstlane_i64(order[i]);
    if (r < 0 || r >= n_rows) break;  // (the lane kernel flags it)
    int32_t *row = walks + r * (int64_t)L1;
    const int32_t v = __builtin_amdgcn_readfirstlane(row[step]);
    if (v < 0 || (int64_t)v >= g.n_vertices || !valid[r]) break;  // vanished walkers come last in the order
    const int64_t vb = readfirstlane_i64(g.rowptr[v]);
    const int n = (int)(readfirstlane_i64(g.rowptr[v + 1]) - vb);
    if (n <= min_n) break;  // this row and every later one: the lane kernel's
    const int32_t s = step > 0 ? __builtin_amdgcn_readfirstlane(row[step - 1]) : -1;
    c.vcol = g.col + vb;
    c.vw = g.w ? g.w + vb : nullptr;
    c.vw64 = g.w64 ? g.w64 + vb : nullptr;
    c.n = n;
    c.nch = (n + 63) >> 6;
    c.s = s;
    const bool first = s < 0;  // randomwalk.py:320-321: unbiased table
    c.need_cls = !first && !(p == 1.0 && q == 1.0);
    c.need_mem = c.need_cls && q != 1.0;
    c.scol = g.col;
    c.m = 1;
    c.iters = 1;
    if (c.need_mem) {
      const int64_t sb = readfirstlane_i64(g.rowptr[s]);
      const int64_t se = readfirstlane_i64(g.rowptr[s + 1]);
      c.scol = g.col + sb;
      c.m = (int)(se - sb);
      c.iters = 32 - __clz(c.m > 0 ? c.m : 1);
    }
    step_tables(c, g, first ? -1 : readfirstlane_i64(edge_state[r]));
    const uint64_t key = (uint64_t)start_ids[r / num_walks] * (uint64_t)num_walks + (uint64_t)(r % num_walks);
    const uint64_t bits = step_bits(walker_stream(seed, key), (uint32_t)step);
    const int idx = exact_draw(c, (uint32_t)(bits >> 32), (uint32_t)bits, lane, L N2V_STATS_PASS);
    if (lane == 0) {
      if (idx < 0) {  // ZeroDivisionError (:172-173)
        atomicOr(status, N2V_ST_ZERODIV);
        valid[r] = 0;
      } else {
        const int32_t x = c.vcol[idx];
        row[step + 1] = x;
        edge_state[r] = vb + idx;
        if (step + 1 < walk_length &&
            (x < 0 || (int64_t)x >= g.n_vertices || g.rowptr[x + 1] == g.rowptr[x]))
          valid[r] = 0;  // fugue.py:147
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace n2v

extern "C" int n2v_walk_exact_launch(const n2v_graph *g, const int32_t *start_ids,
                                     int64_t n_start, int32_t num_walks, int32_t walk_length,
                                     double p, double q, uint64_t seed, int32_t *walks_out,
                                     uint8_t *valid_out, uint32_t *status, void *stream) {
  const int64_t total = n_start * (int64_t)num_walks;
  if (total == 0) return N2V_OK;
  int64_t blocks = (total + n2v::kWavesPerBlock - 1) / n2v::kWavesPerBlock;
  const int64_t cap = n2v::resident_blocks((const void *)n2v::walk_exact_kernel,
                                           n2v::kWavesPerBlock * 64, 0);
  if (blocks > cap) blocks = cap;
  // status[1] is the kernel's walker counter: start it at zero on the same stream
  if (hipMemsetAsync(status + 1, 0, sizeof(uint32_t), (hipStream_t)stream) != hipSuccess)
    return N2V_ELAUNCH;
  hipLaunchKernelGGL(n2v::walk_exact_kernel, dim3((unsigned)blocks),
                     dim3(n2v::kWavesPerBlock * 64), 0, (hipStream_t)stream, *g, start_ids,
                     n_start, num_walks, walk_length, p, q, seed, walks_out, valid_out, status);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}

// the generic (any weights) instance of n2v_partition_step (n2v_partition.hip dispatches):
// status[1] was zeroed by the caller
extern "C" int n2v_partition_step_generic_launch(const int64_t *rowptr, const int32_t *col,
                                                 const float *w, const double *w64, int64_t lo,
                                                 int64_t n_local, const int64_t *head,
                                                 int32_t head_cols, const int64_t *src_ptr,
                                                 const int32_t *src_ids, int64_t k, double p,
                                                 double q, uint64_t seed, int32_t *next_out,
                                                 int64_t *edge_out, uint32_t *status, void *stream) {
  int64_t blocks = (k + n2v::kWavesPerBlock - 1) / n2v::kWavesPerBlock;
  const int64_t cap = n2v::resident_blocks((const void *)n2v::partition_step_kernel,
                                           n2v::kWavesPerBlock * 64, 0);
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(n2v::partition_step_kernel, dim3((unsigned)blocks),
                     dim3(n2v::kWavesPerBlock * 64), 0, (hipStream_t)stream, rowptr, col, w, w64,
                     lo, n_local, head, (int)head_cols, src_ptr, src_ids, k, p, q, seed, next_out,
                     edge_out, status);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}

#ifdef N2V_STATS
extern "C" int n2v_debug_stats(unsigned long long *out_host, int reset) {
  if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(n2v::n2v_stats), sizeof(unsigned long long) * 40) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[40] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(n2v::n2v_stats), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

// the long rows of n2v_walk_weighted_step (n2v_walk_wlanes.hip): status[1] must be zero at launch
extern "C" int n2v_weighted_step_wave_launch(const n2v_graph *g, const int32_t *start_ids, int32_t num_walks,
                                             const int64_t *order, int64_t n_rows, int32_t min_n,
                                             int32_t step, int32_t walk_length, double p, double q,
                                             uint64_t seed, int64_t *edge_state, int32_t *walks,
                                             uint8_t *valid, uint32_t *status, void *stream) {
  int64_t blocks = (n_rows + n2v::kWavesPerBlock - 1) / n2v::kWavesPerBlock;
  int64_t cap = n2v::resident_blocks((const void *)n2v::weighted_step_wave_kernel, n2v::kWavesPerBlock * 64, 0);
  // (min_n == 0: `order` is the short list of the walkers the margin kernels left undecided, -1 behind the last --
  // a full grid of waves that each take one look at it cost 0.08 ms per step)
  if (min_n <= 0 && cap > 1024) cap = 1024;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(n2v::weighted_step_wave_kernel, dim3((unsigned)blocks), dim3(n2v::kWavesPerBlock * 64), 0,
                     (hipStream_t)stream, *g, start_ids, num_walks, order, n_rows, min_n, step, walk_length, p, q,
                     seed, edge_state, walks, valid, status);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}
