// n2v_unit_core.h -- per-LANE replay of the pairing loop of generate_alias_tables
// (reference randomwalk.py:175-189) for ONE slot of a unit-weight step table whose slots take
// three class values (return 1/p, shared 1, other 1/q; randomwalk.py:223-230), given the classes
// by position (wedge table, include/n2v_hip.h).  Shared by the class-count kernels
// (n2v_walk_unit.hip: lanes kernel with wave fallback; n2v_walk_wedge.hip: the kernel that needs
// no wave at all).  Every routine returns sampling_from_alias (:86-99) for slot `pick`:
//   lane_case_a_jump  closed form (exact integer bucket arithmetic), -1 when fp64 rounding decides
//   lane_pairing      rows of <= 64 slots: the two stacks as bit masks
//   lane_case_a       any row whose only underfull class is "other": O(return + shared) steps
//   lane_pairing_list any row, any arrangement: O(n), classes read off the list on the way
#pragma once
#include "n2v_common.h"

namespace n2v {

// diagnostic build only (-DN2V_CHECK): out-of-range values are recorded in status[0] (bits 8..)
// and clamped instead of being used as addresses
#ifdef N2V_CHECK
static __device__ uint32_t *n2v_check_status;
#define N2V_CHECK_RANGE(code, val, lo_, hi_)                                  \
  do {                                                                         \
    if ((val) < (lo_) || (val) >= (hi_)) {                                     \
      atomicOr(n2v_check_status, 1u << (8 + (code)));                          \
      (val) = (lo_);                                                           \
    }                                                                          \
  } while (0)
#else
#define N2V_CHECK_RANGE(code, val, lo_, hi_) do { } while (0)
#endif

template <typename T>
__device__ __forceinline__ T pick3(bool first, bool second, T a, T b, T c) {
  return first ? a : (second ? b : c);
}

// diagnostic build (-DN2V_DECLINE_STATS): why lane_case_a_jump (codes 1 ..) / lane_case_b_jump (11 ..) return -1 on rows of
// more than 64 slots (words 8 + code of the launch's status) and of 4096 and more (words 40 + code); scripts/r6/decline_stats.py
// lends the launch 128 words.  Round 6, cfg 4 trimmed at 100 000 (profiles/r11d_decline_stats.log): every decline on such a
// row is a TIE of the exact process (codes 3, 5, 15, 20) -- the reference's rounding decides it and only a replay knows.
#ifdef N2V_DECLINE_STATS
static __device__ uint32_t *n2v_decline_words;
#define N2V_DECLINE(code)                                            \
  do {                                                               \
    atomicAdd(n2v_decline_words + 256 + (code), 1u); /* rows of any length */                       \
    if (n > 64) atomicAdd(n2v_decline_words + ((code) < 32 ? (code) : 96 + (code)), 1u);           \
    if (n >= 4096 && (code) < 32) atomicAdd(n2v_decline_words + 32 + (code), 1u);   \
    return -1;                                                       \
  } while (0)
#else
#define N2V_DECLINE(code) return -1
#endif

struct UnitConsts {
  double bR, bM, bO;     // 1/p, 1, 1/q
  int64_t TR, TM, TO;    // the same times 2^20 (exact integers; dyadic kernel only)
  int64_t gR, gM, gO;    // TR, TM, TO divided by their greatest common divisor
  double fR, fM, fO;     // the same as fp64 (exact): the closed forms compute in doubles
  int dyadic;            // 1/p, 1/q are multiples of 2^-20: TR .. fO are valid
};

__device__ __forceinline__ int biased_exp(double x) {
  return (int)((__double_as_longlong(x) >> 52) & 0x7ff);
}

// ---- absorbing a long run of equal under values in O(1) per binade ----------------------
// The inner statement of the pairing, probs[over] = probs[over] + probs[under] - 1.0 (:186),
// applied to a run of slots with the same value v < 1: r' = fl(r + v) - 1.0.  The subtraction
// is always exact (t = fl(r + v) >= 2 while the over is not demoted, and t - 1 is a multiple of
// ulp(t) below 2 t), so the only rounding is that of r + v, and while t stays inside one
// binade [2^e, 2^(e+1)) it is the same at every step: r is a multiple of u = ulp(t) (it is a
// previous t minus 1), v = k u + f, and fl rounds f the same way each time -- a tie f == u/2
// goes to the even neighbour, after which t and r are even multiples of u and the choice is
// constant as well.  So after three real steps inside one binade (the first brings r onto the
// grid, the second may be the tie, the third measures the decrement d = r2 - r3, exact) the
// next n steps are r - n d exactly, for every n that keeps t strictly above 2^e (below the
// edge the grid is finer and the argument ends).  Demotion (r < 1) is t < 2, the lower edge of
// binade 1, so it is never skipped: the caller's step-by-step loop finds it.
// Advances (r, j) by real steps and exact jumps, never past `limit` slots and never past a
// demotion; what is left is finished by the caller's loop.
__device__ __forceinline__ void absorb_skip(double &r, double val, int &j, int limit) {
  while (j + 3 < limit) {
    const double t1 = r + val, r1 = t1 - 1.0;
    const double t2 = r1 + val, r2 = t2 - 1.0;
    const double t3 = r2 + val, r3 = t3 - 1.0;
    if (r3 < 1.0) return;  // demoted within three steps: the exact loop takes over
    const int e1 = biased_exp(t1), e3 = biased_exp(t3);
    j += 3;
    r = r3;
    if (e1 != e3) continue;  // crossed into a lower binade: measure again there
    const double edge = __longlong_as_double((long long)e3 << 52);  // 2^e
    const double d = r2 - r3;  // exact
    const int room = limit - j;
    if (d == 0.0) {  // val vanishes against r: every remaining slot is absorbed unchanged
      j += room;
      return;
    }
    if (!(t3 > edge)) continue;
    double nn = floor((t3 - edge) / d) - 1.0;
    nn = fmin(nn, (double)room);
    while (nn >= 1.0 && !(t3 - nn * d > edge)) nn -= 1.0;  // exact products: stay above the edge
    if (nn >= 1.0) {
      r = r3 - nn * d;  // exact: a multiple of ulp(t) inside the binade
      j += (int)nn;
    }
    if (nn < 4.0) return;  // at the edge of the binade: let the exact loop cross it
  }
}

// ---- the reference's row sum when 1/p or 1/q is not dyadic, by one lane -----------------
// sum(node_weights) (:172) adds left to right in fp64; a run of k equal addends needs no loop
// (the argument of rep_add, n2v_walk_unit.hip, of which this is the per-lane form): while s stays
// inside one binade the rounded step of s + c is the same at every addition.
__device__ __forceinline__ double rep_add_lane(double s, double c, int k) {
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
      s = s + c;  // a tie: one real addition makes s an even multiple of ulp
      --k;
      if (k == 0 || biased_exp(s) != es) continue;
    }
    const double t = s + c;
    if (biased_exp(t) != es) {  // this addition leaves the binade: the real operation
      s = t;
      --k;
      continue;
    }
    const double step = t - s;  // exact
    if (step == 0.0) return s;  // c vanishes against s
    const double need = __longlong_as_double((long long)(es + 1) << 52) - s;  // exact
    if ((double)k * step < need) return s + (double)k * step;  // the run stays in the binade: exact
    double j = ceil(need / step) - 1.0;
    while ((j + 1.0) * step < need) j += 1.0;
    while (j * step >= need) j -= 1.0;
    j = fmin(j, (double)k);
    s = s + j * step;  // exact: a multiple of ulp below 2^(e+1)
    k -= (int)j;
  }
  return s;
}

// the row sum of a unit-weight row by classes, left to right: runs of "other" slots between the
// listed (shared) positions and the return run
template <typename P>
__device__ __forceinline__ double lane_row_sum(int n, const UnitConsts &K, int nR, int rpos, int nM,
                                               ListRef<P> list) {
  double sum = 0.0;
  int pos = 0;
  bool run_done = nR == 0;
  for (int k0 = 0; k0 < nM; k0 += 8) {  // eight independent loads per round trip
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (k0 + u < nM) ? (int)list[k0 + u] : n;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int pm = v[u];
      if (pm >= n) break;
      if (!run_done && rpos < pm) {
        sum = rep_add_lane(sum, K.bO, rpos - pos);
        sum = rep_add_lane(sum, K.bR, nR);
        pos = rpos + nR;
        run_done = true;
      }
      sum = rep_add_lane(sum, K.bO, pm - pos);
      sum = sum + K.bM;
      pos = pm + 1;
    }
  }
  if (!run_done) {
    sum = rep_add_lane(sum, K.bO, rpos - pos);
    sum = rep_add_lane(sum, K.bR, nR);
    pos = rpos + nR;
  }
  return rep_add_lane(sum, K.bO, n - pos);
}

// OR of 1 << list[k] for k < cnt (a row of at most 64 neighbours); one lane
template <typename P>
__device__ __forceinline__ uint64_t wedge_mask_t(const void *base, int64_t off, int cnt) {
  const P *a = reinterpret_cast<const P *>(base) + off;
  uint64_t mk = 0ull;
  for (int k = 0; k < cnt; k += 8) {  // eight independent loads per round trip
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (k + u < cnt) ? (int)a[k + u] : 0;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k + u < cnt) mk |= 1ull << (v[u] & 63);
  }
  return mk;
}

// the same for a list as the pairing routines hold it (plain or folded)
template <typename P>
__device__ __forceinline__ uint64_t wedge_mask_l(const ListRef<P> &L, int cnt) {
  uint64_t mk = 0ull;
  for (int k = 0; k < cnt; k += 8) {  // eight independent loads per round trip
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (k + u < cnt) ? L[k + u] : 0;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k + u < cnt) mk |= 1ull << (v[u] & 63);
  }
  return mk;
}

// ---- the pairing loop for ONE slot of a row of any length, by one lane, when "other" is the
// ONLY underfull class (return and shared slots overfull or absent: what p <= q, q > 1 gives on
// every row that is not nearly a clique).  Every slot the loop absorbs from `underfull` then has
// the same value vO -- or is the residual of the overfull slot demoted just before, which sits
// on top of the stack -- so the fp64 sequence of :185 depends on how MANY slots an overfull
// slot absorbs, not on which; which slot is absorbed k-th is its rank from the top among the
// "other" slots.  The overfull slots are exactly the listed positions (wedge list + return run),
// taken in descending order as list.pop() does.  O(return + shared) iterations, each absorbing
// its run of equal values in closed form (absorb_skip): no pass over the row.
// `list` = the shared positions, ascending; returns sampling_from_alias.
// The list is consumed from its end, one entry per iteration; read straight from memory that
// is one dependent round trip per iteration (the kernel was latency-bound on it), so the lane
// stages kStage entries at a time in its own LDS column (`stage`: [kStage][64] entries of this
// wave, idle outside the wave fallback): kStage independent loads per round trip.
template <typename P>
__device__ __forceinline__ int lane_case_a(int n, int pick, double r2, double vR, double vM,
                                           double vO, int nR, int rpos, int nM, ListRef<P> list,
                                           bool pickR, bool pickM, P *stage, int lane) {
  constexpr int kStage = sizeof(P) == 2 ? 16 : 8;
  int st_hi = -1;  // stage slot u holds list[st_hi - u]; nothing staged yet
  auto list_at = [&](int k) -> int {
    if (st_hi < 0 || k > st_hi || k < st_hi - (kStage - 1)) {
      st_hi = k;
      P v[kStage];
#pragma unroll
      for (int u = 0; u < kStage; ++u) v[u] = (k - u >= 0) ? list.p[k - u] : (P)0;  // raw: a folded entry is fixed on the way out
#pragma unroll
      for (int u = 0; u < kStage; ++u) stage[u * 64 + lane] = v[u];
    }
    return list.fix(k, (int)stage[(st_hi - k) * 64 + lane]);
  };
  const int nO = n - nR - nM;
  int km = nM - 1, kr = nR - 1;  // next shared / return slot, descending
  int rank = -1;                 // pick is the (rank + 1)-th "other" slot from the top
  if (!pickR && !pickM) {
    const int lo = list_lower_bound<P>(list, nM, pick + 1);  // entries of the list at or below pick
    int above_r = rpos + nR - 1 - pick;
    above_r = above_r < 0 ? 0 : (above_r > nR ? nR : above_r);
    rank = (n - 1 - pick) - (nM - lo) - above_r;
  }
  int used = 0;  // "other" slots absorbed so far
  bool have_carry = false;
  int carry_i = 0, alias_pick = 0;
  double carry_v = 0.0;
  double p_pick = pick3(pickR, pickM, vR, vM, vO);
  for (;;) {
    if (!have_carry && used >= nO) break;  // underfull is empty (:182)
    const int pm = km >= 0 ? list_at(km) : -1;
    const int pr = kr >= 0 ? rpos + kr : -1;
    if (pm < 0 && pr < 0) break;  // overfull is empty (:182)
    int oi;
    double ov;
    if (pm > pr) {
      oi = pm;
      ov = vM;
      --km;
    } else {
      oi = pr;
      ov = vR;
      --kr;
    }
    if (have_carry) {  // the slot demoted last is on top of underfull
      if (carry_i == pick) {  // alias[under] = over: pick is final
        alias_pick = oi;
        p_pick = carry_v;
        break;
      }
      ov = ov + carry_v - 1.0;  // :185
      have_carry = false;
      if (ov < 1.0) {
        if (oi == pick) p_pick = ov;
        have_carry = true;
        carry_i = oi;
        carry_v = ov;
        continue;
      }
    }
    // `oi` absorbs "other" slots until it drops below 1, `pick` is next, or none is left
    int limit = nO - used;
    if (rank >= used && rank - used < limit) limit = rank - used;
    int j = 0;
    absorb_skip(ov, vO, j, limit);
    bool demoted = false;
    while (j < limit) {
      ov = ov + vO - 1.0;  // :185
      ++j;
      if (ov < 1.0) {
        demoted = true;
        break;
      }
    }
    used += j;
    if (oi == pick) p_pick = ov;
    if (demoted) {
      have_carry = true;
      carry_i = oi;
      carry_v = ov;
      continue;
    }
    if (rank >= 0 && used == rank && used < nO) {  // the next underfull slot is pick itself
      alias_pick = oi;
      p_pick = vO;
    }
    break;  // pick paired, or underfull exhausted with `oi` still >= 1
  }
  return (r2 < p_pick) ? pick : alias_pick;
}

// Position of the T-th slot from the top of a row of n slots (position n - 1 first) that is NOT in the ascending list
// list[0, nM): the largest pos with (n - pos) - (listed entries >= pos) >= T, 1 <= T <= n - nM.  h(k) = list[k] - k
// never decreases (the list ascends strictly), and with i = the number of k with h(k) <= c, c = n - T - nM, the slot
// lies between list[i - 1] and list[i], at c + i: ONE search of the list.  (Rounds 2 - 5 iterated pos = n - T - cnt,
// cnt = listed entries >= pos, to its fixed point -- a full search of the list per iteration, 64 at most, then a
// count slot by slot.  Where the listed slots are dense -- the low positions of a hub's row are the other hubs -- that
// takes dozens of iterations: on cfg 4 trimmed at 100 000 it was 27 of the 72 ms of the (4, 0.25) kernel in the
// replays alone, profiles/r13e_time.log.)
template <typename P>
__device__ __forceinline__ int unlisted_from_top(int n, int nM, const ListRef<P> &list, int T) {
  const int c = n - T - nM;
  int lo = 0, hi = nM;  // the number of k with list[k] - k <= c
  while (hi - lo > N2V_LIST_KARY_MIN) {  // seven independent probes per round trip (as list_lower_bound)
    const int step = (hi - lo) >> 3;
    int v[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) v[k] = list[lo + (k + 1) * step] - (lo + (k + 1) * step);
    int le = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) le += v[k] <= c ? 1 : 0;
    const int nlo = le == 0 ? lo : lo + le * step + 1;
    const int nhi = le == 7 ? hi : lo + (le + 1) * step;
    lo = nlo;
    hi = nhi;
  }
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (list[mid] - mid <= c)
      lo = mid + 1;
    else
      hi = mid;
  }
  return c + lo;
}

// the same with a run of nR more slots [rpos, rpos + nR) taken out (the return run: never listed): position of the
// t-th "other" slot from the top, 1 <= t <= n - nM - nR.  Either it lies above the run -- then the run does not count --
// or below it -- then all of the run does.
template <typename P>
__device__ __forceinline__ int other_from_top(int n, int nR, int rpos, int nM, const ListRef<P> &list, int t) {
  int pos = unlisted_from_top<P>(n, nM, list, t);
  if (nR > 0 && pos < rpos + nR) pos = unlisted_from_top<P>(n, nM, list, t + nR);
  return pos;
}

// The closed forms below do exact INTEGER arithmetic in fp64: every quantity is an integer below
// 2^53 (guarded), for which +, -, * and fma are exact, and a quotient is one fp64 division plus an
// exact correction.  (As int64 the same code cost ~5x the instructions -- 64-bit multiply, divide
// and int64 <-> fp64 conversion are emulated on this hardware -- and the kernel is bound by
// instruction issue: profiles/r02n_cfg4_summary.json, 6 waves x 18 % issuing.)
__device__ __forceinline__ double floor_div(double a, double b) {  // 0 <= a, 0 < b, integers
  double qd = floor(a / b);
  const double r = fma(-qd, b, a);  // exact
  if (r < 0.0) qd -= 1.0;
  if (r >= b) qd += 1.0;
  return qd;
}

// ---- the same draw in closed form ---------------------------------------------------------
// In exact arithmetic the loop of :182-189 under the "other is the only underfull class"
// arrangement is a bucket process.  Scale every value by isum / n (the dyadic class values are
// integers gR, gM, gO then, and 1.0 is isum = nR gR + nM gM + nO gO): an overfull slot brings the
// excess E = g n - isum, an absorbed "other" slot removes D = isum - gO n, a demoted slot carries
// its (negative) rest to its successor.  So after the first i overfull slots (descending
// position) X_i = sum of their excesses has been offered and floor(X_i / D) + 1 "other" slots
// have been absorbed: the slot of rank r (from the top) is absorbed by the first i with
// X_i >= r D, and overfull slot i is demoted at 1 + (X_i - (floor(X_i / D) + 1) D) / isum and
// paired with slot i + 1.  X_i is piecewise linear in i (shared slots above the return run,
// the return run, shared slots below), so i is an integer division -- no loop at all.
// The reference computes in fp64, one rounding per operation; its decisions can differ from the
// exact ones only where an exact quantity is 0 (a tie) or within the accumulated rounding
// (< 4 n 1e-15) of the decision point, and nonzero exact quantities are >= 1 / isum apart.  Ties,
// rows where that margin does not hold, and a final comparison too close to r2 return -1: the
// caller then replays the loop step by step (lane_pairing / lane_case_a).  Checked against the
// reference loop in Python (0 mismatches in 85 k short and 1 k long rows, 5-20 % returned -1).
template <typename P>
__device__ __forceinline__ int lane_case_a_jump(int n, int pick, double r2, const UnitConsts &K,
                                                int nR, int rpos, int nM, ListRef<P> list,
                                                bool pickR, bool pickM, int lo_pick, int below = -1) {
  const int nO = n - nR - nM;
  const double dn = (double)n;
  const double isum = (double)nR * K.fR + (double)nM * K.fM + (double)nO * K.fO;
  const double EM = K.fM * dn - isum, ER = K.fR * dn - isum, D = isum - K.fO * dn;
  // (A shared class exactly on the average -- p = 1/2, q = 2 with two "other" slots per return edge,
  // 5 % of that regime's pairings -- could be decided here as lane_case_b2_jump decides its
  // zero-excess rows (scripts/models/flat_a.py: 4.7 M draws, 0 mismatches).  Built and measured in
  // round 4: same walks, no gain -- such rows are short and their replay is the bit-mask loop --
  // so this instance, whose registers are the flagship configuration's, was left as it was.)
  if (!(D > 0.0) || (nM > 0 && !(EM > 0.0)) || (nR > 0 && !(ER > 0.0))) N2V_DECLINE(1);
  // exactness of the arithmetic (products < 2^52) and of the decisions (4 n 1e-15 < 1 / isum)
  if (dn * isum > 2.0e14 || dn * dn * fmax(K.fR, K.fM) > 4.0e15) N2V_DECLINE(2);
  int mA = nM;  // shared slots above the return run come first in descending order
  if (nR > 0 && nM > 0) {
    int lo = below;  // entries of the list below the return position: stored with the wedge slot,
    if (lo < 0) {    // else searched (probes of the list are requests the kernel is bound by)
      lo = list_lower_bound<P>(list, nM, rpos);
    }
    mA = nM - lo;
  }
  const int N = nM + nR;
  const double dmA = (double)mA, dnR = (double)nR;
  auto X_of = [&](double i) -> double {
    if (i <= dmA) return i * EM;
    if (i <= dmA + dnR) return dmA * EM + (i - dmA) * ER;
    return dmA * EM + dnR * ER + (i - dmA - dnR) * EM;
  };
  auto pos_of = [&](int i) -> int {  // position of the i-th overfull slot, i = 1 .. N
    if (i <= mA) return (int)list[nM - i];
    if (i <= mA + nR) return rpos + nR - (i - mA);
    return (int)list[nM - (i - nR)];
  };
  if (!pickR && !pickM) {
    int above_r = rpos + nR - 1 - pick;
    above_r = above_r < 0 ? 0 : (above_r > nR ? nR : above_r);
    const double r = (double)((n - 1 - pick) - (nM - lo_pick) - above_r);
    const double T = r * D;
    double i;
    if (T <= 0.0) {
      i = 1.0;
    } else if (mA > 0 && dmA * EM >= T) {
      i = floor_div(T + EM - 1.0, EM);
    } else {
      const double X1 = dmA * EM;
      if (nR > 0 && X1 + dnR * ER >= T)
        i = dmA + floor_div(T - X1 + ER - 1.0, ER);
      else
        i = dmA + dnR + floor_div(T - X1 - dnR * ER + EM - 1.0, EM);
    }
    if (!(i >= 1.0) || i > (double)N || X_of(i) == T) N2V_DECLINE(3);  // a tie: fp64 rounding decides
    return pos_of((int)i);  // r2 >= probs[pick] here: the caller's quick exit took the other case
  }
  int i0;
  if (pickR) {
    i0 = mA + (nR - (pick - rpos));
  } else {
    const int d = nM - lo_pick;  // pick is the d-th shared slot from the top
    i0 = d <= mA ? d : d + nR;
  }
  if (i0 < 1 || i0 > N) N2V_DECLINE(4);
  // the last overfull slot ends at exactly 1.0 in exact arithmetic (mass balance), i.e. within the
  // accumulated rounding of it in fp64 -- or is never reached and keeps its value >= 1; either way
  // probs[pick] > 1 - 1e-9 > r2 (r2 <= 1 - 2^-32): sampling_from_alias returns pick
  if (i0 == N) return pick;
  const double X = X_of((double)i0);
  const double xq = floor_div(X, D);
  const double rem0 = fma(-xq, D, X);  // X mod D, exact
  if (rem0 == 0.0) N2V_DECLINE(5);
  const double prob = 1.0 + (rem0 - D) / isum;  // demoted at 1 + (X - (xq + 1) D) / isum
  if (fabs(prob - r2) < 1e-9) N2V_DECLINE(6);
  return (r2 < prob) ? pick : pos_of(i0 + 1);
}

// ---- the mirror arrangement in closed form: "other" is the OVERFULL class and every return /
// shared slot is underfull (what q < 1 gives: cfg 5's p = 4, q = 0.25).  Now the few listed slots
// are the underfull stack (popped from the highest position) and the long runs of "other" slots
// the overfull one: an underfull slot of deficit d = isum - g n is paired with the current
// "other" slot, whose excess e = gO n - isum is tiny, so that slot is demoted and the rest of the
// deficit cascades down the "other" slots until their cumulative excess covers it.  In exact
// arithmetic, with Y_j the cumulative deficit of the first j listed slots (descending position):
//   * listed slot j is paired with the "other" slot of rank ceil(Y_(j-1) / e) (rank 1 for j = 1);
//   * the "other" slot of rank t < nO is demoted while slot j(t) = min { j : Y_j > t e } is being
//     absorbed, ends at 1 + (t e - Y_j(t)) / isum and is paired with the "other" slot of rank
//     t + 1; the last one ends at exactly 1.0 (mass balance: Y_S = nO e) -- fp64 decides.
// Same exactness argument and the same -1 cases as lane_case_a_jump; checked against the
// reference loop in Python (0 mismatches in 180 k short and 5.7 k long rows, 2-10 % return -1).
template <typename P, bool kNextSlot>
__device__ __forceinline__ int lane_case_b_jump(int n, int pick, double r2, const UnitConsts &K,
                                                int nR, int rpos, int nM, ListRef<P> list,
                                                bool pickR, bool pickM, int lo_pick, int below = -1) {
  const int nO = n - nR - nM;
  const double dn = (double)n;
  const double isum = (double)nR * K.fR + (double)nM * K.fM + (double)nO * K.fO;
  const double e = K.fO * dn - isum, dR = isum - K.fR * dn, dM = isum - K.fM * dn;
  if (nO <= 0 || !(e > 0.0) || (nM > 0 && !(dM > 0.0)) || (nR > 0 && !(dR > 0.0))) N2V_DECLINE(11);
  if (dn * isum > 2.0e14 || dn * dn * K.fO > 4.0e15) N2V_DECLINE(12);
  auto list_lower = [&](int pos) -> int {  // entries of the list below pos
    return list_lower_bound<P>(list, nM, pos);
  };
  int mA = nM;  // shared slots above the return run come first in descending order
  if (nR > 0 && nM > 0) mA = nM - (below >= 0 ? below : list_lower(rpos));
  const int S = nM + nR;
  const double dmA = (double)mA, dnR = (double)nR;
  auto Y_of = [&](double j) -> double {
    if (j <= dmA) return j * dM;
    if (j <= dmA + dnR) return dmA * dM + (j - dmA) * dR;
    return dmA * dM + dnR * dR + (j - dmA - dnR) * dM;
  };
  auto other_pos = [&](int t) -> int {  // position of the t-th "other" slot from the top
    return other_from_top<P>(n, nR, rpos, nM, list, t);
  };
  if (pickR || pickM) {  // underfull: r2 >= its value here (the caller's quick exit took the rest)
    int j;
    if (pickR) {
      j = mA + (nR - (pick - rpos));
    } else {
      const int d = nM - lo_pick;
      j = d <= mA ? d : d + nR;
    }
    if (j < 1 || j > S) N2V_DECLINE(14);
    double t = 1.0;
    if (j > 1) {
      const double Yp = Y_of((double)(j - 1));
      t = floor_div(Yp + e - 1.0, e);
      if (t * e == Yp) N2V_DECLINE(15);  // that slot holds exactly 1.0: fp64 decides whether it was demoted
    }
    if (!(t >= 1.0) || t > (double)nO) N2V_DECLINE(16);
    return other_pos((int)t);
  }
  int ar = rpos + nR - 1 - pick;  // return slots above pick
  ar = ar < 0 ? 0 : (ar > nR ? nR : ar);
  const int t = (n - pick) - (nM - lo_pick) - ar;  // rank of pick among "other", from 1
  if (t < 1 || t > nO) N2V_DECLINE(17);
  // the last "other" slot ends at exactly 1.0 in exact arithmetic (mass balance: Y_S = nO e), i.e.
  // within the accumulated rounding of it in fp64 -- or is never reached and keeps its value
  // >= 1; either way probs[pick] > 1 - 1e-9 > r2 (r2 <= 1 - 2^-32): the draw returns pick
  if (t == nO) return pick;
  const double T = (double)t * e;
  double j;  // smallest j with Y_j > T
  if (mA > 0 && dmA * dM > T) {
    j = floor_div(T, dM) + 1.0;
  } else {
    const double Y1 = dmA * dM;
    if (nR > 0 && Y1 + dnR * dR > T)
      j = dmA + floor_div(T - Y1, dR) + 1.0;
    else if (dM > 0.0)
      j = dmA + dnR + floor_div(T - Y1 - dnR * dR, dM) + 1.0;
    else
      N2V_DECLINE(18);
  }
  if (!(j >= 1.0) || j > (double)S) N2V_DECLINE(19);
  if (j > 1.0 && Y_of(j - 1.0) == T) N2V_DECLINE(20);  // exactly 1.0 after the previous listed slot
  const double prob = 1.0 + (T - Y_of(j)) / isum;
  if (fabs(prob - r2) < 1e-9) N2V_DECLINE(21);
  if (r2 < prob) return pick;
  // the "other" slot of rank t + 1 is the next one below pick: usually pick - 1, found from the
  // rank of pick in the list (no search); other_pos() if the row runs out (it cannot: t < nO)
  int cpos = kNextSlot ? pick - 1 : -1, k = lo_pick - 1;
  // (at most four steps down: where the listed slots crowd -- the low positions of a hub's row -- the walk down a run of
  // them is a chain of dependent loads as long as the run; the search below finds the slot whatever lies between)
  for (int tries = 0; cpos >= 0 && tries < 4; ++tries) {
    if (nR > 0 && cpos >= rpos && cpos < rpos + nR) {
      cpos = rpos - 1;
      continue;
    }
    while (k >= 0 && (int)list[k] > cpos) --k;
    if (k >= 0 && (int)list[k] == cpos) {
      --cpos;
      --k;
      continue;
    }
    return cpos;
  }
  return other_pos(t + 1);
}

__device__ __forceinline__ uint64_t wedge_mask(const void *base, int64_t off, int cnt, bool wide) {
  if (wide) return wedge_mask_t<uint32_t>(base, off, cnt);
  return wedge_mask_t<uint16_t>(base, off, cnt);
}

// ---- the pairing loop of generate_alias_tables (randomwalk.py:175-189) for ONE slot of a row
// of at most 64 neighbours, by one lane: the two stacks are bit masks (popped from the highest
// index, as list.pop() does on ascending lists), a demoted overfull slot is the next underfull
// one, an overfull slot that stays >= 1 is the next overfull one.  Slot values are the three
// class values; only probs[pick] and alias[pick] are tracked, and the loop stops when `pick`
// has been paired as an underfull slot (it never changes afterwards).  Returns
// sampling_from_alias: pick if r2 < probs[pick] else alias[pick].
__device__ __forceinline__ int lane_pairing(int n, uint64_t Rm, uint64_t Mm, int pick, double r2,
                                            double vR, double vM, double vO) {
  const uint64_t valid = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
  Rm &= valid;
  Mm &= valid & ~Rm;
  const uint64_t Om = valid & ~(Rm | Mm);
  uint64_t under = ((vR < 1.0) ? Rm : 0ull) | ((vM < 1.0) ? Mm : 0ull) | ((vO < 1.0) ? Om : 0ull);
  uint64_t over = valid & ~under;
  auto val0 = [&](int i) -> double {
    return pick3(((Rm >> i) & 1ull) != 0, ((Mm >> i) & 1ull) != 0, vR, vM, vO);
  };
  double p_pick = val0(pick);
  int alias_pick = 0;
  bool have_carry = false, have_cur = false;
  int carry_i = 0, cur_i = 0;
  double carry_v = 0.0, cur_v = 0.0;
  while ((have_carry || under != 0ull) && (have_cur || over != 0ull)) {  // :182
    int ui, oi;
    double uv, ov;
    if (have_carry) {
      ui = carry_i;
      uv = carry_v;
      have_carry = false;
    } else {
      ui = 63 - __clzll((long long)under);
      under &= ~(1ull << ui);
      uv = val0(ui);
    }
    if (have_cur) {
      oi = cur_i;
      ov = cur_v;
      have_cur = false;
    } else {
      oi = 63 - __clzll((long long)over);
      over &= ~(1ull << oi);
      ov = val0(oi);
    }
    if (ui == pick) {  // alias[under] = over (:184); probs[under] is final
      alias_pick = oi;
      p_pick = uv;
      break;
    }
    ov = ov + uv - 1.0;  // :185, two roundings
    if (oi == pick) p_pick = ov;
    if (ov < 1.0) {  // :186-189
      have_carry = true;
      carry_i = oi;
      carry_v = ov;
    } else {
      have_cur = true;
      cur_i = oi;
      cur_v = ov;
    }
  }
  return (r2 < p_pick) ? pick : alias_pick;
}

// ---- the mirror arrangement replayed run by run (the fallback of lane_case_b_jump on long
// rows): the listed slots are the underfull stack, the "other" slots ONE run of equal overfull
// values vO in [1, 2).  A listed slot is absorbed by the current "other" slot; if that drops
// below 1 its rest cascades down the run, and a cascade through equal values has a closed form
// (the one of unit_draw, n2v_walk_unit.hip): the first slot is fl(fl(vO + a) - 1) -- the
// reference's two operations --, and while slots keep being demoted vO + a < 2, every operand is
// a multiple of 2^-52 below 2 and the sums are exact: slot i holds a1 + (i - 1)(vO - 1) EXACTLY,
// so the number of demoted slots is an integer search seeded by one multiplication and fixed up
// with exact products, and only the slot that settles (its sum reaches [2, 3) and may round) is
// recomputed with the two real operations.  O(listed slots) iterations, no pass over the row.
// `list` = the shared positions, ascending.
template <typename P>
__device__ __forceinline__ int lane_case_b(int n, int pick, double r2, double vR, double vM,
                                           double vO, int nR, int rpos, int nM, ListRef<P> list,
                                           bool pickR, bool pickM, int lo_pick = -1, int below = -1) {
  const int nO = n - nR - nM;
#if defined(N2V_ABLATE_B) && N2V_ABLATE_B == 1  // timing only: the replay does nothing
  return pick;
#endif
  auto list_lower = [&](int pos) -> int {  // entries of the list below pos
    return list_lower_bound<P>(list, nM, pos);
  };
  auto specials_ge = [&](int pos) -> int {
    int r = rpos + nR - pos;
    r = r < 0 ? 0 : (r > nR ? nR : r);
    return (nM - list_lower(pos)) + r;
  };
  auto other_pos = [&](int t) -> int {  // position of the t-th "other" slot from the top; 0 if none
#if defined(N2V_ABLATE_B) && N2V_ABLATE_B == 2  // timing only: no search for the slot of a rank
    return pick;
#endif
    if (t < 1 || t > nO) return 0;
    return other_from_top<P>(n, nR, rpos, nM, list, t);
  };
  // rank of pick among the "other" slots (1 = highest position), 0 if pick is listed
  int pick_rank = 0;
  if (!pickR && !pickM) pick_rank = (n - pick) - specials_ge(pick + 1);
  const double d = vO - 1.0;  // exact (vO in [1, 2))
  const double inv = d > 0.0 ? 1.0 / d : 0.0;
  // The underfull stack is popped from the highest position: the mA listed slots above the return run, the run,
  // the listed slots below it -- so the i-th slot popped is a listed one unless mA < i <= mA + nR, and which i is
  // `pick` follows from its rank in the list (lo_pick).  The loop therefore reads NO list entry (round 6: it read
  // list[km] at every iteration, one more dependent load in a chain that one lane runs while its wave waits; on
  // cfg 4 trimmed at 100 000 these replays are a third of the (4, 0.25) kernel, profiles/r12c_*).
  const int lo_p = (pickM && lo_pick < 0) ? list_lower(pick) : lo_pick;
  const int mA = nM - ((nR > 0 && nM > 0) ? (below >= 0 ? below : list_lower(rpos)) : 0);
  int i_pick = 0;  // the place of `pick` on the underfull stack, from 1; 0: an "other" slot
  if (pickR) {
    i_pick = mA + (nR - (pick - rpos));
  } else if (pickM) {
    const int dd = nM - lo_p;
    i_pick = dd <= mA ? dd : dd + nR;
  }
  const int S = nM + nR;
  int i_next = 1;                // next slot of the underfull stack
  int t_used = 0;                // "other" slots of rank <= t_used have been demoted
  bool have_cur = false;         // rank t_used + 1 is the current overfull slot, at cur_val >= 1
  double cur_val = 0.0;
  for (;;) {
    if (i_next > S) break;  // underfull is empty (:182)
    if (!have_cur && t_used >= nO) break;  // overfull is empty (:182)
    const int i_cur = i_next++;
#ifdef N2V_DECLINE_STATS
    atomicAdd(n2v_decline_words + 81, 1u);  // iterations of this loop
#endif
    const double uv = (i_cur > mA && i_cur <= mA + nR) ? vR : vM;
    const int over_rank = t_used + 1;
    if (i_cur == i_pick) return other_pos(over_rank);  // alias[pick]; r2 >= probs[pick] here
    double a = (have_cur ? cur_val : vO) + uv - 1.0;  // :185
    if (!(a < 1.0)) {
      cur_val = a;
      have_cur = true;
      continue;
    }
    // rank `over_rank` is demoted at a, paired next with rank over_rank + 1 (if any)
    if (pick_rank == over_rank) return (r2 < a) ? pick : other_pos(over_rank + 1);
    t_used = over_rank;
    have_cur = false;
    const int avail = nO - t_used;  // untouched "other" slots, all at vO
    if (avail <= 0) break;          // overfull is empty: the rest stays on underfull
    const double a1 = vO + a - 1.0;  // slot 1 of the cascade, the reference's two operations
    if (!(a1 < 1.0)) {
      cur_val = a1;
      have_cur = true;
      continue;
    }
    const double need = 1.0 - a1;  // exact, > 0
    const double m1 = (double)(avail - 1);
    if (m1 * d < need) {  // every remaining slot is demoted: slot i holds a1 + (i - 1) d exactly
      if (pick_rank > t_used) {
        const double pv = a1 + (double)(pick_rank - t_used - 1) * d;
        return (r2 < pv) ? pick : other_pos(pick_rank + 1);  // the last one keeps alias 0
      }
      break;
    }
    // j = the first i >= 1 with a1 + i d >= 1: slots 1 .. j are demoted, slot j + 1 settles
    double j = fmin(fmax(ceil(need * inv), 1.0), m1);
#ifdef N2V_DECLINE_STATS
    {
      double jj = j;
      unsigned fix = 0;
      while (jj * d < need) { jj += 1.0; ++fix; }
      while (jj >= 2.0 && (jj - 1.0) * d >= need) { jj -= 1.0; ++fix; }
      atomicAdd(n2v_decline_words + 82, fix);  // corrections of the cascade length
    }
#endif
    while (j * d < need) j += 1.0;
    while (j >= 2.0 && (j - 1.0) * d >= need) j -= 1.0;
    const int jd = (int)j;
    if (pick_rank > t_used && pick_rank <= t_used + jd) {
      const double pv = a1 + (double)(pick_rank - t_used - 1) * d;
      return (r2 < pv) ? pick : other_pos(pick_rank + 1);
    }
    const double a_prev = a1 + (j - 1.0) * d;  // slot j: exact, < 1
    cur_val = vO + a_prev - 1.0;               // the slot that settles: real operations
    t_used += jd;
    have_cur = true;
  }
  // pick was never paired: an "other" slot keeps a value >= 1 (returns pick: r2 < 1), a listed
  // slot keeps its value <= r2 and alias 0
  return (pickR || pickM) ? 0 : pick;
}

// ---- two classes on one stack ----------------------------------------------------------------
// q > 1 with p > q (the return slot is the smallest value, "other" the middle one: p = 4, q = 2)
// puts the return run AND the "other" slots on the underfull stack of every row with a shared
// neighbour; q < 1 with p < q (p = 1/4, q = 1/2) puts both on the overfull stack.  The bucket
// process is the same, one of its two cumulative functions is now piecewise linear in the rank:
// the stack is the "other" slots in descending position with the return run inserted after the
// rho "other" slots above it.  The listed (shared) slots are the whole opposite stack.

// geometry shared by the four functions below
template <typename P>
struct TwoOnStack {
  int n, nR, rpos, nM, nO, rho, nS;  // nS = nO + nR slots on the mixed stack
  ListRef<P> list;
  __device__ __forceinline__ int list_lower(int pos) const {  // entries of the list below pos
    return list_lower_bound<P>(list, nM, pos);
  }
  // `below`: the number of listed slots below the return position when the caller has it stored
  // (wedge slots), else -1 and the list is searched
  __device__ __forceinline__ TwoOnStack(int n_, int nR_, int rpos_, int nM_, ListRef<P> list_, int below = -1)
      : n(n_), nR(nR_), rpos(rpos_), nM(nM_), nO(n_ - nR_ - nM_), list(list_) {
    const int mA = nM - (below >= 0 ? below : list_lower(rpos));  // shared slots above the return run
    rho = (n - rpos - nR) - mA;            // "other" slots above the return run
    nS = nO + nR;
  }
  __device__ __forceinline__ bool in_run(int t) const { return t > rho && t <= rho + nR; }  // 1-based
  // the mixed stack is every slot that is not listed, in descending position:
  // position of its t-th slot (0 if none)
  __device__ __forceinline__ int stack_pos(int t) const {
    if (t < 1 || t > nS) return 0;
    return unlisted_from_top<P>(n, nM, list, t);
  }
  // 1-based rank on the mixed stack of a return slot / of an "other" slot with lo_pick listed
  // slots below it
  __device__ __forceinline__ int stack_rank(int pick, bool pickR, int lo_pick) const {
    if (pickR) return rho + (rpos + nR - pick);
    return (n - pick) - (nM - lo_pick);
  }
};

// q > 1, p > q in closed form: the mixed stack is UNDERFULL ("other" deficit D, return deficit DR),
// the listed slots are the overfull stack (excess EM each).  Def(k) = deficit of the first k slots
// of the mixed stack.  Slot of rank k + 1 is paired with listed slot i = min { i : i EM > Def(k) };
// listed slot i0 < nM is demoted while slot k = min { k : Def(k) > i0 EM } is absorbed, at
// 1 + (i0 EM - Def(k)) / isum, and paired with listed slot i0 + 1.  Ties and thin margins: -1.
// Checked against the reference loop in Python (0 mismatches in 70 k short and 1.9 k long rows).
template <typename P>
__device__ __forceinline__ int lane_case_a2_jump(int n, int pick, double r2, const UnitConsts &K,
                                                 int nR, int rpos, int nM, ListRef<P> list,
                                                 bool pickR, bool pickM, int lo_pick, int below = -1) {
  const int nO = n - nR - nM;
  const double dn = (double)n;
  const double isum = (double)nR * K.fR + (double)nM * K.fM + (double)nO * K.fO;
  const double EM = K.fM * dn - isum, D = isum - K.fO * dn, DR = isum - K.fR * dn;
  if (nR <= 0 || nM <= 0 || !(D > 0.0) || !(DR > 0.0) || !(EM > 0.0)) N2V_DECLINE(41);
  if (dn * isum > 2.0e14 || dn * dn * K.fM > 4.0e15) N2V_DECLINE(42);
  const TwoOnStack<P> G(n, nR, rpos, nM, list, below);
  const double drho = (double)G.rho, dnR = (double)nR;
  auto Def = [&](double k) -> double {
    if (k <= drho) return k * D;
    if (k <= drho + dnR) return drho * D + (k - drho) * DR;
    return drho * D + dnR * DR + (k - drho - dnR) * D;
  };
  if (!pickM) {  // underfull: r2 >= its value here (the caller's quick exit took the rest)
    const int k = G.stack_rank(pick, pickR, lo_pick) - 1;  // slots of the stack above pick
    if (k < 0 || k >= G.nS) N2V_DECLINE(43);
    const double T = Def((double)k);
    double i = 1.0;
    if (T > 0.0) {
      i = floor_div(T + EM - 1.0, EM);
      if (i * EM == T) N2V_DECLINE(44);  // that listed slot holds exactly 1.0: fp64 decides
    }
    if (!(i >= 1.0) || i > (double)nM) N2V_DECLINE(45);
    return (int)list[nM - (int)i];
  }
  const int i0 = nM - lo_pick;  // pick is the i0-th listed slot from the top
  if (i0 < 1 || i0 > nM) N2V_DECLINE(46);
  if (i0 == nM) return pick;  // the last overfull slot: 1.0 within rounding, or never reached
  const double X = (double)i0 * EM;
  double k;  // smallest k with Def(k) > X
  if (drho * D > X) {
    k = floor_div(X, D) + 1.0;
  } else {
    const double Y1 = drho * D;
    if (Y1 + dnR * DR > X)
      k = drho + floor_div(X - Y1, DR) + 1.0;
    else
      k = drho + dnR + floor_div(X - Y1 - dnR * DR, D) + 1.0;
  }
  if (!(k >= 1.0) || k > (double)G.nS) N2V_DECLINE(47);
  if (Def(k - 1.0) == X) N2V_DECLINE(48);
  const double prob = 1.0 + (X - Def(k)) / isum;
  if (fabs(prob - r2) < 1e-9) N2V_DECLINE(49);
  return (r2 < prob) ? pick : (int)list[nM - (i0 + 1)];
}

// the same arrangement replayed run by run (the fallback of lane_case_a2_jump on long rows):
// lane_case_a with the overfull stack = the list alone and the run of equal underfull values
// split at the return run (three segments: vO, vR, vO).
template <typename P>
__device__ __forceinline__ int lane_case_a2(int n, int pick, double r2, double vR, double vM,
                                            double vO, int nR, int rpos, int nM, ListRef<P> list,
                                            bool pickR, bool pickM, P *stage, int lane) {
  constexpr int kStage = sizeof(P) == 2 ? 16 : 8;
  int st_hi = -1;
  auto list_at = [&](int k) -> int {
    if (st_hi < 0 || k > st_hi || k < st_hi - (kStage - 1)) {
      st_hi = k;
      P v[kStage];
#pragma unroll
      for (int u = 0; u < kStage; ++u) v[u] = (k - u >= 0) ? list.p[k - u] : (P)0;  // raw: a folded entry is fixed on the way out
#pragma unroll
      for (int u = 0; u < kStage; ++u) stage[u * 64 + lane] = v[u];
    }
    return list.fix(k, (int)stage[(st_hi - k) * 64 + lane]);
  };
  const TwoOnStack<P> G(n, nR, rpos, nM, list);
  const int nU = G.nS, rho = G.rho;
  int rank = -1;  // slots of the underfull stack above pick
  if (!pickM) rank = G.stack_rank(pick, pickR, G.list_lower(pick)) - 1;
  int km = nM - 1, used = 0;
  bool have_carry = false;
  int carry_i = 0, alias_pick = 0;
  double carry_v = 0.0;
  double p_pick = pick3(pickR, pickM, vR, vM, vO);
  for (;;) {
    if (!have_carry && used >= nU) break;  // underfull is empty (:182)
    if (km < 0) break;                     // overfull is empty (:182)
    const int oi = list_at(km);
    double ov = vM;
    --km;
    if (have_carry) {  // the slot demoted last is on top of underfull
      if (carry_i == pick) {
        alias_pick = oi;
        p_pick = carry_v;
        break;
      }
      ov = ov + carry_v - 1.0;  // :185
      have_carry = false;
      if (ov < 1.0) {
        if (oi == pick) p_pick = ov;
        have_carry = true;
        carry_i = oi;
        carry_v = ov;
        continue;
      }
    }
    bool demoted = false;
    while (used < nU && !(rank >= 0 && used == rank)) {
      const bool run = used >= rho && used < rho + nR;
      const double val = run ? vR : vO;
      int limit = (used < rho ? rho : (run ? rho + nR : nU)) - used;
      if (rank >= used && rank - used < limit) limit = rank - used;
      int j = 0;
      absorb_skip(ov, val, j, limit);
      while (j < limit) {
        ov = ov + val - 1.0;  // :185
        ++j;
        if (ov < 1.0) {
          demoted = true;
          break;
        }
      }
      used += j;
      if (demoted) break;
    }
    if (oi == pick) p_pick = ov;
    if (demoted) {
      have_carry = true;
      carry_i = oi;
      carry_v = ov;
      continue;
    }
    if (rank >= 0 && used == rank && used < nU) {  // the next underfull slot is pick itself
      alias_pick = oi;
      p_pick = (rank >= rho && rank < rho + nR) ? vR : vO;
    }
    break;
  }
  return (r2 < p_pick) ? pick : alias_pick;
}

// q < 1, p < q in closed form: the mixed stack is OVERFULL ("other" excess e, return excess eR),
// the listed slots are the underfull stack (deficit dM each).  Xo(t) = excess of the first t slots
// of the mixed stack.  Listed slot j is paired with stack slot t = min { t : Xo(t) >= (j - 1) dM }
// (slot 1 for j = 1); stack slot t < nS is demoted while listed slot j = floor(Xo(t) / dM) + 1 is
// absorbed, at 1 + (Xo(t) - j dM) / isum, and paired with stack slot t + 1.
// Checked against the reference loop in Python (0 mismatches in 76 k short and 3.7 k long rows).
template <typename P>
__device__ __forceinline__ int lane_case_b2_jump(int n, int pick, double r2, const UnitConsts &K,
                                                 int nR, int rpos, int nM, ListRef<P> list,
                                                 bool pickR, bool pickM, int lo_pick, int below = -1) {
  const int nO = n - nR - nM;
  const double dn = (double)n;
  const double isum = (double)nR * K.fR + (double)nM * K.fM + (double)nO * K.fO;
  const double e = K.fO * dn - isum, eR = K.fR * dn - isum, dM = isum - K.fM * dn;
  // "other" exactly ON the average (p = 1/4, q = 1/2 with two shared neighbours per return edge:
  // 10 % of that regime's pairings): its slots are overfull with excess 0.  The average then IS the
  // "other" weight, a power of two, so every value of the table is one and the reference's loop is
  // exact arithmetic -- no rounding decides anything: a slot that reaches exactly 1.0 stays
  // overfull (:187), the zero-excess slots above the return run pass the first deficit on, those
  // below it are never reached.  The formulas below hold with the ties taken that way (checked
  // against the reference loop in Python: 4.7 M draws, 0 mismatches).
  const bool flat = e == 0.0;
  if (nR <= 0 || nM <= 0 || nO <= 0 || !(e > 0.0 || flat) || !(eR > 0.0) || !(dM > 0.0)) N2V_DECLINE(61);
  if (dn * isum > 2.0e14 || dn * dn * fmax(K.fR, K.fO) > 4.0e15) N2V_DECLINE(62);
  const TwoOnStack<P> G(n, nR, rpos, nM, list, below);
  const double drho = (double)G.rho, dnR = (double)nR;
  auto Xo = [&](double t) -> double {
    if (t <= drho) return t * e;
    if (t <= drho + dnR) return drho * e + (t - drho) * eR;
    return drho * e + dnR * eR + (t - drho - dnR) * e;
  };
  if (pickM) {  // underfull: r2 >= its value here
    const int j = nM - lo_pick;
    if (j < 1 || j > nM) N2V_DECLINE(63);
    double t = 1.0;
    if (j > 1) {
      const double Yp = (double)(j - 1) * dM;  // smallest t with Xo(t) >= Yp
      if (!flat && drho * e >= Yp) {
        t = floor_div(Yp + e - 1.0, e);
      } else {
        const double X1 = drho * e;
        if (X1 + dnR * eR >= Yp)
          t = drho + floor_div(Yp - X1 + eR - 1.0, eR);
        else if (!flat)
          t = drho + dnR + floor_div(Yp - X1 - dnR * eR + e - 1.0, e);
        else
          N2V_DECLINE(64);  // (mass balance: the return run covers every listed slot)
      }
      if (!flat && Xo(t) == Yp) N2V_DECLINE(65);  // that slot holds exactly 1.0: fp64 decides
    }
    if (!(t >= 1.0) || t > (double)G.nS) N2V_DECLINE(66);
    return G.stack_pos((int)t);
  }
  const int t = G.stack_rank(pick, pickR, lo_pick);
  if (t < 1 || t > G.nS) N2V_DECLINE(67);
  if (t == G.nS) return pick;  // the last overfull slot: 1.0 within rounding, or never reached
  const double T = Xo((double)t);
  const double j = floor_div(T, dM) + 1.0;
  if (flat && j > (double)nM) return pick;  // never demoted: the end of the return run, "other" below it
  if (!(j >= 1.0) || j > (double)nM) N2V_DECLINE(68);
  if (!flat && j > 1.0 && (j - 1.0) * dM == T) N2V_DECLINE(69);
  const double prob = 1.0 + (T - j * dM) / isum;
  if (fabs(prob - r2) < 1e-9) N2V_DECLINE(70);
  if (r2 < prob) return pick;
  if (!pickR) {  // the next slot of the stack is the next position below pick that is not listed
    int cpos = pick - 1, k = lo_pick - 1;
    for (int tries = 0; cpos >= 0 && tries < 4; ++tries) {  // (at most four steps down, then the search)
      while (k >= 0 && (int)list[k] > cpos) --k;
      if (k >= 0 && (int)list[k] == cpos) {
        --cpos;
        --k;
        continue;
      }
      return cpos;
    }
  }
  return G.stack_pos(t + 1);
}

// the same arrangement replayed run by run: lane_case_b with the underfull stack = the list alone
// and the cascade through the overfull slots stopped at the return run (whose slots start at vR,
// not vO: they take the two real operations each).
template <typename P>
__device__ __forceinline__ int lane_case_b2(int n, int pick, double r2, double vR, double vM,
                                            double vO, int nR, int rpos, int nM, ListRef<P> list,
                                            bool pickR, bool pickM) {
  const TwoOnStack<P> G(n, nR, rpos, nM, list);
  const int nV = G.nS, rho = G.rho;
  const int pick_rank = pickM ? 0 : G.stack_rank(pick, pickR, G.list_lower(pick));
  const double d = vO - 1.0;  // exact (vO in [1, 2))
  const double inv = d > 0.0 ? 1.0 / d : 0.0;
  int km = nM - 1;
  int t_used = 0;         // stack slots of rank <= t_used have been demoted
  bool have_cur = false;  // rank t_used + 1 is the current overfull slot, at cur_val >= 1
  double cur_val = 0.0;
  for (;;) {
    if (km < 0) break;                      // underfull is empty (:182)
    if (!have_cur && t_used >= nV) break;   // overfull is empty (:182)
    const int ui = (int)list[km];
    --km;
    const int over_rank = t_used + 1;
    if (ui == pick) return G.stack_pos(over_rank);  // alias[pick]; r2 >= probs[pick] here
    double a = (have_cur ? cur_val : (G.in_run(over_rank) ? vR : vO)) + vM - 1.0;  // :185
    if (!(a < 1.0)) {
      cur_val = a;
      have_cur = true;
      continue;
    }
    if (pick_rank == over_rank) return (r2 < a) ? pick : G.stack_pos(over_rank + 1);
    t_used = over_rank;
    have_cur = false;
    while (t_used < nV) {  // the rest of the demoted slot cascades down the stack
      const int r = t_used + 1;
      if (G.in_run(r)) {
        const double val = vR + a - 1.0;
        if (!(val < 1.0)) {
          cur_val = val;
          have_cur = true;
          break;
        }
        if (pick_rank == r) return (r2 < val) ? pick : G.stack_pos(r + 1);
        t_used = r;
        a = val;
        continue;
      }
      const int avail = (t_used < rho ? rho : nV) - t_used;  // untouched "other" slots up to the run
      const double a1 = vO + a - 1.0;  // slot 1 of the cascade, the reference's two operations
      if (!(a1 < 1.0)) {
        cur_val = a1;
        have_cur = true;
        break;
      }
      const double need = 1.0 - a1;  // exact, > 0
      const double m1 = (double)(avail - 1);
      if (m1 * d < need) {  // all of them are demoted: slot i holds a1 + (i - 1) d exactly
        if (pick_rank > t_used && pick_rank <= t_used + avail) {
          const double pv = a1 + (double)(pick_rank - t_used - 1) * d;
          return (r2 < pv) ? pick : G.stack_pos(pick_rank + 1);
        }
        a = a1 + m1 * d;
        t_used += avail;
        continue;
      }
      double j = fmin(fmax(ceil(need * inv), 1.0), m1);
      while (j * d < need) j += 1.0;
      while (j >= 2.0 && (j - 1.0) * d >= need) j -= 1.0;
      const int jd = (int)j;
      if (pick_rank > t_used && pick_rank <= t_used + jd) {
        const double pv = a1 + (double)(pick_rank - t_used - 1) * d;
        return (r2 < pv) ? pick : G.stack_pos(pick_rank + 1);
      }
      const double a_prev = a1 + (j - 1.0) * d;  // slot j: exact, < 1
      cur_val = vO + a_prev - 1.0;               // the slot that settles: real operations
      t_used += jd;
      have_cur = true;
      break;
    }
  }
  // pick was never paired: a stack slot keeps a value >= 1 (returns pick: r2 < 1), a listed slot
  // keeps its value <= r2 and alias 0
  return pickM ? 0 : pick;
}

// ---- the return run alone on the overfull stack (p < q < 1 on rows with few shared neighbours:
// 1/p is overfull, "other" and shared are both underfull).  The nR return slots absorb the other
// slots from the top of the row down: a slot whose predecessors' deficits sum to T is paired with
// return slot floor(T / ER) + 1, counted from the top of the run, if there is one.  An overfull
// `pick` is the last return slot on every graph without multi-edges: mass balance, as above.
template <typename P>
__device__ __forceinline__ int lane_case_a3_jump(int n, int pick, double r2, const UnitConsts &K,
                                                 int nR, int rpos, int nM, bool pickR, bool pickM,
                                                 int lo_pick) {
  const int nO = n - nR - nM;
  const double dn = (double)n;
  const double isum = (double)nR * K.fR + (double)nM * K.fM + (double)nO * K.fO;
  const double ER = K.fR * dn - isum, D = isum - K.fO * dn, DM = isum - K.fM * dn;
  if (nR <= 0 || nM <= 0 || !(ER > 0.0) || !(D > 0.0) || !(DM > 0.0)) N2V_DECLINE(81);
  if (dn * isum > 2.0e14 || dn * dn * K.fR > 4.0e15) N2V_DECLINE(82);
  if (pickR) return (rpos + nR - pick == nR) ? pick : -1;
  const int m_above = nM - lo_pick - (pickM ? 1 : 0);
  int ar = rpos + nR - 1 - pick;  // return slots above pick
  ar = ar < 0 ? 0 : (ar > nR ? nR : ar);
  const int o_above = (n - 1 - pick) - ar - m_above;
  const double T = (double)o_above * D + (double)m_above * DM;
  const double iq = floor_div(T, ER);
  if (T > 0.0 && fma(-iq, ER, T) == 0.0) N2V_DECLINE(83);  // that return slot holds exactly 1.0
  if (iq >= (double)nR) return 0;  // never paired: alias stays 0 (:170)
  return rpos + nR - 1 - (int)iq;
}

// the same arrangement replayed run by run: lane_case_a with the overfull stack = the return run
// alone and the underfull values = runs of vO split by the listed slots (vM).
template <typename P>
__device__ __forceinline__ int lane_case_a3(int n, int pick, double r2, double vR, double vM,
                                            double vO, int nR, int rpos, int nM, ListRef<P> list,
                                            bool pickR, bool pickM) {
  const int nU = n - nR;
  auto urank = [&](int pos) -> int {  // underfull slots above position pos
    int ar = rpos + nR - 1 - pos;
    ar = ar < 0 ? 0 : (ar > nR ? nR : ar);
    return (n - 1 - pos) - ar;
  };
  const int rank = pickR ? -1 : urank(pick);
  int km = nM - 1, kr = nR - 1, used = 0;
  bool have_carry = false;
  int carry_i = 0, alias_pick = 0;
  double carry_v = 0.0;
  double p_pick = pick3(pickR, pickM, vR, vM, vO);
  for (;;) {
    if (!have_carry && used >= nU) break;  // underfull is empty (:182)
    if (kr < 0) break;                     // overfull is empty (:182)
    const int oi = rpos + kr;
    double ov = vR;
    --kr;
    if (have_carry) {
      if (carry_i == pick) {
        alias_pick = oi;
        p_pick = carry_v;
        break;
      }
      ov = ov + carry_v - 1.0;  // :185
      have_carry = false;
      if (ov < 1.0) {
        if (oi == pick) p_pick = ov;
        have_carry = true;
        carry_i = oi;
        carry_v = ov;
        continue;
      }
    }
    bool demoted = false;
    while (used < nU && !(rank >= 0 && used == rank)) {
      const int mrank = km >= 0 ? urank((int)list[km]) : nU;
      double val = vO;
      int limit = mrank - used;
      if (used == mrank) {
        val = vM;
        limit = 1;
        --km;
      }
      if (rank >= used && rank - used < limit) limit = rank - used;
      int j = 0;
      absorb_skip(ov, val, j, limit);
      while (j < limit) {
        ov = ov + val - 1.0;  // :185
        ++j;
        if (ov < 1.0) {
          demoted = true;
          break;
        }
      }
      used += j;
      if (demoted) break;
    }
    if (oi == pick) p_pick = ov;
    if (demoted) {
      have_carry = true;
      carry_i = oi;
      carry_v = ov;
      continue;
    }
    if (rank >= 0 && used == rank && used < nU) {  // the next underfull slot is pick itself
      alias_pick = oi;
      p_pick = pickM ? vM : vO;
    }
    break;
  }
  return (r2 < p_pick) ? pick : alias_pick;
}

// ---- any row, any arrangement of the classes: the loop of :175-189 slot by slot, by one lane.
// The two stacks are walked as two descending cursors over the positions of the row; the class
// of a position is read off the (ascending) shared list and the return run on the way down, so
// a draw costs O(n) steps and O(shared) loads.  The rare case (a row of more than 64 slots whose
// shared or return class is underfull: nearly a clique); correctness first.
template <typename P>
__device__ __forceinline__ int lane_pairing_list(int n, int pick, double r2, double vR, double vM,
                                                 double vO, int nR, int rpos, int nM,
                                                 ListRef<P> list) {
  const bool uR = vR < 1.0, uM = vM < 1.0, uO = vO < 1.0;
  // class of position i (0 return, 1 shared, 2 other); k walks down the list with i
  auto cls_at = [&](int i, int &k) -> int {
    while (k >= 0 && (int)list[k] > i) --k;
    if (i >= rpos && i < rpos + nR) return 0;
    if (k >= 0 && (int)list[k] == i) return 1;
    return 2;
  };
  auto val_of = [&](int c) -> double { return pick3(c == 0, c == 1, vR, vM, vO); };
  auto is_under = [&](int c) -> bool { return pick3(c == 0, c == 1, uR, uM, uO); };
  int pu = n - 1, ku = nM - 1;  // next candidate position of the underfull / overfull stack
  int po = n - 1, ko = nM - 1;
  int kp = nM - 1;
  double p_pick = val_of(cls_at(pick, kp));
  int alias_pick = 0;
  bool have_carry = false, have_cur = false;
  int carry_i = 0, cur_i = 0;
  double carry_v = 0.0, cur_v = 0.0;
  for (;;) {
    int ui = -1, oi = -1;
    double uv = 0.0, ov = 0.0;
    if (have_carry) {
      ui = carry_i;
      uv = carry_v;
    } else {
      while (pu >= 0) {
        const int c = cls_at(pu, ku);
        if (is_under(c)) {
          ui = pu;
          uv = val_of(c);
          break;
        }
        --pu;
      }
    }
    if (have_cur) {
      oi = cur_i;
      ov = cur_v;
    } else {
      while (po >= 0) {
        const int c = cls_at(po, ko);
        if (!is_under(c)) {
          oi = po;
          ov = val_of(c);
          break;
        }
        --po;
      }
    }
    if (ui < 0 || oi < 0) break;  // :182: one of the stacks is empty
    if (have_carry)
      have_carry = false;
    else
      --pu;  // popped
    if (have_cur)
      have_cur = false;
    else
      --po;
    if (ui == pick) {  // alias[under] = over; probs[under] is final
      alias_pick = oi;
      p_pick = uv;
      break;
    }
    ov = ov + uv - 1.0;  // :185
    if (oi == pick) p_pick = ov;
    if (ov < 1.0) {
      have_carry = true;
      carry_i = oi;
      carry_v = ov;
    } else {
      have_cur = true;
      cur_i = oi;
      cur_v = ov;
    }
  }
  return (r2 < p_pick) ? pick : alias_pick;
}

}  // namespace n2v

// the kernel of n2v_walk_wedge.hip (hop table + wedge table present): 1 = launched, 0 = does not apply
int n2v_walk_wedge_try(const n2v_graph *g, const int32_t *start_ids, int64_t n_start,
                       int32_t num_walks, int32_t walk_length, double p, double q,
                       const n2v::UnitConsts &K, uint64_t seed, int32_t *walks_out,
                       uint8_t *valid_out, uint32_t *status, void *stream);

// the same walk in passes over a caller-lent workspace (n2v_walk_wedge2.hip): closed forms in the
// main launches, declined steps replayed out of line.  1 = enqueued, 0 = does not apply
int64_t n2v_walk_wedge2_workspace(int64_t total);
int n2v_walk_wedge2_try(const n2v_graph *g, const int32_t *start_ids, int64_t n_start,
                        int32_t num_walks, int32_t walk_length, double p, double q,
                        const n2v::UnitConsts &K, uint64_t seed, int32_t *walks_out,
                        uint8_t *valid_out, uint32_t *status, void *workspace,
                        int64_t workspace_bytes, int32_t rounds, void *stream);

// the wedge-list instance of n2v_partition_step (n2v_walk_wedge.hip): 1 = launched, < 0 on error
int n2v_partition_step_wedge_launch(const int64_t *rowptr, const int32_t *col, int64_t lo,
                                    int64_t n_local, const int64_t *head, int32_t head_cols,
                                    const int64_t *src_ptr, const int32_t *src_ids, int32_t src_at,
                                    int64_t k, double p, double q, const n2v::UnitConsts &K, uint64_t seed,
                                    int32_t *next_out, int64_t *edge_out, uint32_t *status,
                                    void *stream);
