// n2v_unit_near.h -- the closed forms of n2v_unit_core.h for class values that are NOT dyadic
// (round 4).  lane_case_*_jump decide the draw of generate_alias_tables' pairing loop
// (randomwalk.py:175-189) by bucket arithmetic, and are exact because dyadic class values make
// every quantity an integer: the reference's fp64 loop can differ from the exact process only at an
// exact tie.  With other values (1/p or 1/q not a power of two) nothing is an integer, but the
// argument is the same with a margin in the place of the tie:
//
//   * the reference's loop works on the table's values v = fl(b / avg) and does, per absorbed slot,
//     probs[over] += probs[under] - 1.0: two roundings of at most 2^-53 |value| each, at most n
//     absorptions that can reach one slot, values <= vmax: what any slot holds deviates from the
//     exact-real process on the same v by at most 2.3e-16 n vmax;
//   * these functions are given v computed from the row AVERAGE IN ANY ORDER (the counts times the
//     class weights: no pass over the row), which is within (n + 8) 4.5e-16 relatively of the
//     reference's left-to-right sum: cumulative sums of up to n values deviate by at most
//     4.5e-16 (n + 8) n vmax;
//   * the formulas themselves are a handful of fp64 operations on quantities <= n vmax.
//
// So every decision of the exact-real process -- which slot absorbs which, where a slot is demoted,
// the final r2 < probs[pick] -- that clears  mg = 5e-15 n (n + 8) vmax  (about ten times the sum of
// the three bounds) is the reference's decision, and one that does not returns -1: the caller adds
// the row up in the reference's order and replays the loop (lane_row_sum, lane_case_*), as it did
// for every such step before.  A decision of the process only depends on the cumulative sums at
// the slot asked for (the bucket argument of n2v_unit_core.h), so only those are checked.
// Same geometry and symbols as the functions they restate; values in units of 1.0 (isum == 1).
#pragma once
#include "n2v_unit_core.h"

namespace n2v {

struct NearVals {
  double vR, vM, vO;  // the table's values of the three classes (b / avg, avg from the counts)
  double mg;          // decisions closer than this to their threshold are not made here
};

// smallest integer i with i * step >= T  (T > 0, step > 0), as a double
__device__ __forceinline__ double near_ceil_div(double T, double step) {
  double q = floor(T / step);
  double r = fma(-q, step, T);
  if (r < 0.0) {
    q -= 1.0;
    r += step;
  }
  if (r >= step) {
    q += 1.0;
    r -= step;
  }
  return r > 0.0 ? q + 1.0 : q;
}
// largest integer q with q * step <= T  (T >= 0, step > 0)
__device__ __forceinline__ double near_floor_div(double T, double step) {
  double q = floor(T / step);
  const double r = fma(-q, step, T);
  if (r < 0.0) q -= 1.0;
  if (r >= step) q += 1.0;
  return q;
}

// "other" is the only underfull class (lane_case_a_jump)
template <typename P>
__device__ __forceinline__ int lane_case_a_near(int n, int pick, double r2, const NearVals &V, int nR,
                                                int rpos, int nM, ListRef<P> list, bool pickR, bool pickM,
                                                int lo_pick, int below) {
  const double mg = V.mg;
  const double EM = V.vM - 1.0, ER = V.vR - 1.0, D = 1.0 - V.vO;
  if (!(D > mg) || (nM > 0 && !(EM > mg)) || (nR > 0 && !(ER > mg))) return -1;
  int mA = nM;
  if (nR > 0 && nM > 0) {
    int lo = below;
    if (lo < 0) {
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
  auto pos_of = [&](int i) -> int {
    if (i <= mA) return (int)list[nM - i];
    if (i <= mA + nR) return rpos + nR - (i - mA);
    return (int)list[nM - (i - nR)];
  };
  if (!pickR && !pickM) {
    int above_r = rpos + nR - 1 - pick;
    above_r = above_r < 0 ? 0 : (above_r > nR ? nR : above_r);
    const double r = (double)((n - 1 - pick) - (nM - lo_pick) - above_r);
    const double T = r * D;
    if (!(r >= 1.0)) return pos_of(1);  // the top "other" slot: absorbed by the first overfull slot
    double i;  // smallest i with X_i >= T
    if (mA > 0 && dmA * EM >= T) {
      i = near_ceil_div(T, EM);
    } else {
      const double X1 = dmA * EM;
      if (nR > 0 && X1 + dnR * ER >= T)
        i = dmA + near_ceil_div(T - X1, ER);
      else if (nM > 0)
        i = dmA + dnR + near_ceil_div(T - X1 - dnR * ER, EM);
      else
        return -1;
    }
    if (!(i >= 1.0) || i > (double)N) return -1;
    if (!(X_of(i) - T > mg)) return -1;                // X_i >= T, not within the margin
    if (i > 1.0 && !(T - X_of(i - 1.0) > mg)) return -1;  // X_(i-1) < T, not within the margin
    return pos_of((int)i);
  }
  int i0;
  if (pickR) {
    i0 = mA + (nR - (pick - rpos));
  } else {
    const int d = nM - lo_pick;
    i0 = d <= mA ? d : d + nR;
  }
  if (i0 < 1 || i0 > N) return -1;
  if (i0 == N) return (r2 < 1.0 - mg - 1e-9) ? pick : -1;  // ends at 1.0 within the margin, or is never reached
  const double X = X_of((double)i0);
  const double xq = near_floor_div(X, D);
  const double rem0 = fma(-xq, D, X);
  if (!(rem0 > mg) || !(D - rem0 > mg)) return -1;
  const double prob = 1.0 + (rem0 - D);
  if (fabs(prob - r2) < 1e-9 + mg) return -1;
  return (r2 < prob) ? pick : pos_of(i0 + 1);
}

// "other" is the only overfull class (lane_case_b_jump)
template <typename P>
__device__ __forceinline__ int lane_case_b_near(int n, int pick, double r2, const NearVals &V, int nR,
                                                int rpos, int nM, ListRef<P> list, bool pickR, bool pickM,
                                                int lo_pick, int below) {
  const int nO = n - nR - nM;
  const double mg = V.mg;
  const double e = V.vO - 1.0, dR = 1.0 - V.vR, dM = 1.0 - V.vM;
  if (nO <= 0 || !(e > mg) || (nM > 0 && !(dM > mg)) || (nR > 0 && !(dR > mg))) return -1;
  auto list_lower = [&](int pos) -> int {
    return list_lower_bound<P>(list, nM, pos);
  };
  int mA = nM;
  if (nR > 0 && nM > 0) mA = nM - (below >= 0 ? below : list_lower(rpos));
  const int S = nM + nR;
  const double dmA = (double)mA, dnR = (double)nR;
  auto Y_of = [&](double j) -> double {
    if (j <= dmA) return j * dM;
    if (j <= dmA + dnR) return dmA * dM + (j - dmA) * dR;
    return dmA * dM + dnR * dR + (j - dmA - dnR) * dM;
  };
  auto other_pos = [&](int t) -> int {
    return other_from_top<P>(n, nR, rpos, nM, list, t);
  };
  if (pickR || pickM) {
    int j;
    if (pickR) {
      j = mA + (nR - (pick - rpos));
    } else {
      const int d = nM - lo_pick;
      j = d <= mA ? d : d + nR;
    }
    if (j < 1 || j > S) return -1;
    double t = 1.0;
    if (j > 1) {
      const double Yp = Y_of((double)(j - 1));
      t = near_ceil_div(Yp, e);  // smallest t with t e >= Yp
      if (!(t * e - Yp > mg)) return -1;
      if (t > 1.0 && !(Yp - (t - 1.0) * e > mg)) return -1;
    }
    if (!(t >= 1.0) || t > (double)nO) return -1;
    return other_pos((int)t);
  }
  int ar = rpos + nR - 1 - pick;
  ar = ar < 0 ? 0 : (ar > nR ? nR : ar);
  const int t = (n - pick) - (nM - lo_pick) - ar;
  if (t < 1 || t > nO) return -1;
  if (t == nO) return (r2 < 1.0 - mg - 1e-9) ? pick : -1;
  const double T = (double)t * e;
  double j;  // smallest j with Y_j > T
  if (mA > 0 && dmA * dM > T) {
    j = near_floor_div(T, dM) + 1.0;
  } else {
    const double Y1 = dmA * dM;
    if (nR > 0 && Y1 + dnR * dR > T)
      j = dmA + near_floor_div(T - Y1, dR) + 1.0;
    else if (nM > 0 && T - Y1 - dnR * dR >= 0.0)
      j = dmA + dnR + near_floor_div(T - Y1 - dnR * dR, dM) + 1.0;
    else
      return -1;
  }
  if (!(j >= 1.0) || j > (double)S) return -1;
  if (!(Y_of(j) - T > mg)) return -1;
  if (j > 1.0 && !(T - Y_of(j - 1.0) > mg)) return -1;
  const double prob = 1.0 + (T - Y_of(j));
  if (fabs(prob - r2) < 1e-9 + mg) return -1;
  if (r2 < prob) return pick;
  // the "other" slot of rank t + 1 is the next one below pick: usually pick - 1, found from the
  // rank of pick in the list (no search); other_pos() if the row runs out (it cannot: t < nO)
  int cpos = pick - 1, k = lo_pick - 1;
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

// return + "other" underfull, the listed slots overfull (lane_case_a2_jump)
template <typename P>
__device__ __forceinline__ int lane_case_a2_near(int n, int pick, double r2, const NearVals &V, int nR,
                                                 int rpos, int nM, ListRef<P> list, bool pickR, bool pickM,
                                                 int lo_pick, int below) {
  const double mg = V.mg;
  const double EM = V.vM - 1.0, D = 1.0 - V.vO, DR = 1.0 - V.vR;
  if (nR <= 0 || nM <= 0 || !(D > mg) || !(DR > mg) || !(EM > mg)) return -1;
  const TwoOnStack<P> G(n, nR, rpos, nM, list, below);
  const double drho = (double)G.rho, dnR = (double)nR;
  auto Def = [&](double k) -> double {
    if (k <= drho) return k * D;
    if (k <= drho + dnR) return drho * D + (k - drho) * DR;
    return drho * D + dnR * DR + (k - drho - dnR) * D;
  };
  if (!pickM) {
    const int k = G.stack_rank(pick, pickR, lo_pick) - 1;
    if (k < 0 || k >= G.nS) return -1;
    const double T = Def((double)k);
    double i = 1.0;
    if (k > 0) {
      i = near_ceil_div(T, EM);  // smallest i with i EM >= T  (slot k + 1 goes to min { i : i EM > Def(k) })
      if (!(i * EM - T > mg)) return -1;
      if (i > 1.0 && !(T - (i - 1.0) * EM > mg)) return -1;
    }
    if (!(i >= 1.0) || i > (double)nM) return -1;
    return (int)list[nM - (int)i];
  }
  const int i0 = nM - lo_pick;
  if (i0 < 1 || i0 > nM) return -1;
  if (i0 == nM) return (r2 < 1.0 - mg - 1e-9) ? pick : -1;
  const double X = (double)i0 * EM;
  double k;  // smallest k with Def(k) > X
  if (drho * D > X) {
    k = near_floor_div(X, D) + 1.0;
  } else {
    const double Y1 = drho * D;
    if (Y1 + dnR * DR > X)
      k = drho + near_floor_div(X - Y1, DR) + 1.0;
    else if (X - Y1 - dnR * DR >= 0.0)
      k = drho + dnR + near_floor_div(X - Y1 - dnR * DR, D) + 1.0;
    else
      return -1;
  }
  if (!(k >= 1.0) || k > (double)G.nS) return -1;
  if (!(Def(k) - X > mg)) return -1;
  if (!(X - Def(k - 1.0) > mg)) return -1;
  const double prob = 1.0 + (X - Def(k));
  if (fabs(prob - r2) < 1e-9 + mg) return -1;
  return (r2 < prob) ? pick : (int)list[nM - (i0 + 1)];
}

// return + "other" overfull, the listed slots underfull (lane_case_b2_jump)
template <typename P>
__device__ __forceinline__ int lane_case_b2_near(int n, int pick, double r2, const NearVals &V, int nR,
                                                 int rpos, int nM, ListRef<P> list, bool pickR, bool pickM,
                                                 int lo_pick, int below) {
  const int nO = n - nR - nM;
  const double mg = V.mg;
  const double e = V.vO - 1.0, eR = V.vR - 1.0, dM = 1.0 - V.vM;
  if (nR <= 0 || nM <= 0 || nO <= 0 || !(e > mg) || !(eR > mg) || !(dM > mg)) return -1;
  const TwoOnStack<P> G(n, nR, rpos, nM, list, below);
  const double drho = (double)G.rho, dnR = (double)nR;
  auto Xo = [&](double t) -> double {
    if (t <= drho) return t * e;
    if (t <= drho + dnR) return drho * e + (t - drho) * eR;
    return drho * e + dnR * eR + (t - drho - dnR) * e;
  };
  if (pickM) {
    const int j = nM - lo_pick;
    if (j < 1 || j > nM) return -1;
    double t = 1.0;
    if (j > 1) {
      const double Yp = (double)(j - 1) * dM;  // smallest t with Xo(t) >= Yp
      if (drho * e >= Yp) {
        t = near_ceil_div(Yp, e);
      } else {
        const double X1 = drho * e;
        if (X1 + dnR * eR >= Yp)
          t = drho + near_ceil_div(Yp - X1, eR);
        else
          t = drho + dnR + near_ceil_div(Yp - X1 - dnR * eR, e);
      }
      if (!(t >= 1.0) || t > (double)G.nS) return -1;
      if (!(Xo(t) - Yp > mg)) return -1;
      if (t > 1.0 && !(Yp - Xo(t - 1.0) > mg)) return -1;
    }
    if (!(t >= 1.0) || t > (double)G.nS) return -1;
    return G.stack_pos((int)t);
  }
  const int t = G.stack_rank(pick, pickR, lo_pick);
  if (t < 1 || t > G.nS) return -1;
  if (t == G.nS) return (r2 < 1.0 - mg - 1e-9) ? pick : -1;
  const double T = Xo((double)t);
  const double j = near_floor_div(T, dM) + 1.0;  // smallest j with j dM > T
  if (!(j >= 1.0) || j > (double)nM) return -1;
  if (!(j * dM - T > mg)) return -1;
  if (!(T - (j - 1.0) * dM > mg)) return -1;  // (j == 1: T itself, the excess of t >= 1 slots)
  const double prob = 1.0 + (T - j * dM);
  if (fabs(prob - r2) < 1e-9 + mg) return -1;
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

// the return run alone overfull (lane_case_a3_jump)
__device__ __forceinline__ int lane_case_a3_near(int n, int pick, double r2, const NearVals &V, int nR,
                                                 int rpos, int nM, bool pickR, bool pickM, int lo_pick) {
  const double mg = V.mg;
  const double ER = V.vR - 1.0, D = 1.0 - V.vO, DM = 1.0 - V.vM;
  if (nR <= 0 || nM <= 0 || !(ER > mg) || !(D > mg) || !(DM > mg)) return -1;
  if (pickR) return (rpos + nR - pick == nR && r2 < 1.0 - mg - 1e-9) ? pick : -1;
  const int m_above = nM - lo_pick - (pickM ? 1 : 0);
  int ar = rpos + nR - 1 - pick;
  ar = ar < 0 ? 0 : (ar > nR ? nR : ar);
  const int o_above = (n - 1 - pick) - ar - m_above;
  const double T = (double)o_above * D + (double)m_above * DM;
  const double iq = near_floor_div(T, ER);
  if (o_above + m_above > 0) {
    const double rem = fma(-iq, ER, T);
    if (!(rem > mg) || !(ER - rem > mg)) return -1;
  }
  if (iq >= (double)nR) return 0;  // never paired: alias stays 0 (:170)
  return rpos + nR - 1 - (int)iq;
}

// the closed-form half of a step whose class values are not dyadic: the slot the draw returns, or
// -1 (the caller adds the row up in the reference's order and replays)
template <typename P>
__device__ __forceinline__ int near_listed(int arr, int n, int pick, double r2, const NearVals &V, int nR,
                                           int rpos, int nM, ListRef<P> list, bool isR, bool isM,
                                           int lo_pick, int below) {
  if (arr == 1) return lane_case_a_near<P>(n, pick, r2, V, nR, rpos, nM, list, isR, isM, lo_pick, below);
  if (arr == 2) return lane_case_b_near<P>(n, pick, r2, V, nR, rpos, nM, list, isR, isM, lo_pick, below);
  if (arr == 3) return lane_case_a2_near<P>(n, pick, r2, V, nR, rpos, nM, list, isR, isM, lo_pick, below);
  if (arr == 4) return lane_case_b2_near<P>(n, pick, r2, V, nR, rpos, nM, list, isR, isM, lo_pick, below);
  if (arr == 5) return lane_case_a3_near(n, pick, r2, V, nR, rpos, nM, isR, isM, lo_pick);
  return -1;
}

// The closed forms once more, AFTER the row has been added up in the reference's order (round 5).  near_step
// works on values from the class counts and must allow for the order of the row sum: its margin grows with
// n^2 (5e-15 n (n + 8) vmax: 5e-5 at n = 10^5, the size of an excess or a deficit there), so on the long rows
// of a graph trimmed at the reference's own cap (100 000, constants.py:6) it declines several per cent of
// the pairings and every one of those is an O(list) replay.  Given the reference's `avg` bit for bit
// (lane_row_sum) the values v = fl(b / avg) ARE the reference's, which side of 1.0 each class lies on and
// whether `pick` was accepted are exact comparisons the caller has made, and what is left is
//   * the reference's loop: two roundings per iteration, fewer than n iterations, values <= vmax + 1:
//     <= 2.2e-16 n (vmax + 1) at any lattice point of the staircase ("Why a margin ...");
//   * the formulas: products and sums of up to n values, a handful of operations: <= 1e-15 n vmax;
// so a margin of 2e-14 n (vmax + 1) -- LINEAR in n, twelve times the sum -- decides.  -1: replay.
template <typename P>
__device__ __forceinline__ int near_listed_exact(int arr, int n, int pick, double r2, const UnitConsts &K,
                                                 double avg, int nR, int rpos, int nM, ListRef<P> list, bool isR,
                                                 bool isM, int lo_pick, int below) {
  const int nO = n - nR - nM;
  NearVals V;
  V.vR = K.bR / avg, V.vM = K.bM / avg, V.vO = K.bO / avg;
  double vmax = nO ? V.vO : 0.0;
  if (nR) vmax = fmax(vmax, V.vR);
  if (nM) vmax = fmax(vmax, V.vM);
  V.mg = 2e-14 * (double)n * (vmax + 1.0);
  return near_listed<P>(arr, n, pick, r2, V, nR, rpos, nM, list, isR, isM, lo_pick, below);
}
constexpr int kNearExactMin = 256;  // rows from this length on try the forms again with the exact average

// One step whose class values are not dyadic, before the row is added up in the reference's order:
// taken when every decision on the way clears the margin -- which side of the average every class is
// on, that `pick` was not accepted (the caller's quick exit accepts with its own margin), and the
// decisions inside the closed form.  Returns the slot of the draw, or -1: the exact row sum and the
// replays decide.  `below`: listed slots below the return position, or -1 (searched).
template <typename P>
__device__ __forceinline__ int near_step(int n, int pick, double r2, const UnitConsts &K, int nR, int rpos,
                                         int nM, ListRef<P> list, bool isR, bool isM, int lo_pick, int below) {
  const int nO = n - nR - nM;
  // one class only (a leaf's single return edge; a row without shared neighbours when p == q): every
  // value is b / avg = 1 within n 2^-53, and sampling_from_alias returns `pick` on either side of 1.0
  // -- below it is accepted (r2 <= 1 - 2^-32 is smaller), at or above it the loop of :182 never runs
  if ((nR == n || nM == n || nO == n) && n < (1 << 20)) return pick;
  const double b_pick = pick3(isR, isM, K.bR, K.bM, K.bO);
  const double approx = ((double)nR * K.bR + (double)nM * K.bM + (double)nO * K.bO) / (double)n;
  const double eps = ((double)n + 8.0) * 4.5e-16;  // any order of the addends against the reference's
  const double lo_f = 1.0 - 2.0 * eps, hi_f = 1.0 + 2.0 * eps;
  const bool decR = nR == 0 || K.bR < approx * lo_f || K.bR > approx * hi_f;
  const bool decM = nM == 0 || K.bM < approx * lo_f || K.bM > approx * hi_f;
  const bool decO = nO == 0 || K.bO < approx * lo_f || K.bO > approx * hi_f;
  const double pp = b_pick / approx;
  const bool not_accepted = b_pick > approx * hi_f || (b_pick < approx * lo_f && r2 > pp * hi_f);
  if (!(decR && decM && decO && not_accepted)) return -1;
  const bool uR = K.bR < approx, uM = K.bM < approx, uO = K.bO < approx;
  const bool any_under = (nR && uR) || (nM && uM) || (nO && uO);
  const bool any_over = (nR && !uR) || (nM && !uM) || (nO && !uO);
  if (!any_under || !any_over) return -1;
  // the stacks, as the kernels number them
  int arr = 0;
  if (uO && !(nR && uR) && !(nM && uM)) arr = 1;
  else if (!uO && nO > 0 && (!nR || uR) && (!nM || uM)) arr = 2;
  else if (uO && nR && uR && nM && !uM) arr = 3;
  else if (!uO && nO > 0 && nR && !uR && nM && uM) arr = 4;
  else if (uO && nR && !uR && nM && uM) arr = 5;
  NearVals V;
  V.vR = K.bR / approx, V.vM = K.bM / approx, V.vO = K.bO / approx;
  double vmax = nO ? V.vO : 0.0;
  if (nR) vmax = fmax(vmax, V.vR);
  if (nM) vmax = fmax(vmax, V.vM);
  V.mg = 5e-15 * (double)n * ((double)n + 8.0) * vmax;
  return near_listed<P>(arr, n, pick, r2, V, nR, rpos, nM, list, isR, isM, lo_pick, below);
}

}  // namespace n2v
