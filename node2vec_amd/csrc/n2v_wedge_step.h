// n2v_wedge_step.h -- one step of an exact biased walk on a unit-weight graph from the per-edge
// tables (hop table with class counts + wedge table, include/n2v_hip.h), by one lane: shared by the
// kernels of n2v_walk_wedge.hip (one launch, replays inline) and n2v_walk_wedge2.hip (closed
// forms in the main launches, declined steps replayed out of line).  Reference: the table
// generate_edge_alias_tables builds at (s, v) and sampling_from_alias on it, randomwalk.py:86-99,
// :157-232.
#pragma once
#include "n2v_unit_core.h"
#include "n2v_unit_near.h"

#ifndef N2V_DEFER_HOP
#define N2V_DEFER_HOP 2  // 0: every step gathers the hop entry of `pick` first (rounds 4 - 5); 1: deferred on edges
                         // with an inline return position only (A/B builds: profiles/r10o_time_defer_hop_ab.log)
#endif
#ifndef N2V_NEAR_FORMS
#define N2V_NEAR_FORMS 1  // 0: values that are not dyadic replay every pairing (rounds 2 - 3; A/B builds)
#endif

namespace n2v {

#ifdef N2V_NEAR_COUNT
__device__ uint32_t *n2v_count_words;  // = status of the launch (set by the kernel's first lines)
#endif
#if defined(N2V_BIG_STATS) && defined(N2V_BIG_DECLINES)
__device__ uint32_t *n2v_big_words;
#endif

// the pairing loop for slot `pick` by one lane: closed form by arrangement `arr` (see the kernel),
// else -- fp64 rounding decides the draw: a tie or a thin margin -- the replays.
// kMode 0 / 3: the (p, q) that leave "other" alone on its stack on ordinary rows, underfull (0) or
// overfull (3: the mirror closed form then runs at every step and finds the next slot without a
// search -- code instance 0 does without: its registers are the flagship configuration's); 1: those for which
// the return run shares a stack with it (arrangements 3-5: compiled out of instance 0, whose
// registers they would cost); 2: 1/p or 1/q not dyadic -- no exact integer arithmetic, so no
// closed form: every pairing is replayed run by run in fp64.
template <typename P, int kMode>
__device__ __forceinline__ int pair_listed(int arr, int n, int pick, double r2, const UnitConsts &K,
                                           double avg, int nR, int rpos, int nM, ListRef<P> list,
                                           bool isR, bool isM, int lo_pick, P *stage, int lane,
                                           int below = -1) {
  constexpr bool kShared = kMode == 1 || kMode == 2;
  int res = -1;
  if constexpr (kMode != 2) {
    if (arr == 1)
      res = lane_case_a_jump<P>(n, pick, r2, K, nR, rpos, nM, list, isR, isM, lo_pick, below);
    else if (arr == 2)
      res = lane_case_b_jump<P, kMode != 0>(n, pick, r2, K, nR, rpos, nM, list, isR, isM, lo_pick, below);
  }
  if constexpr (kMode == 1) {
    if (arr == 3)
      res = lane_case_a2_jump<P>(n, pick, r2, K, nR, rpos, nM, list, isR, isM, lo_pick, below);
    else if (arr == 4)
      res = lane_case_b2_jump<P>(n, pick, r2, K, nR, rpos, nM, list, isR, isM, lo_pick, below);
    else if (arr == 5)
      res = lane_case_a3_jump<P>(n, pick, r2, K, nR, rpos, nM, isR, isM, lo_pick);
  }
#ifdef N2V_ABLATE_W
  if (N2V_ABLATE_W == 1 && arr == 2) res = pick;  // timing-only: no closed form at all
#endif
#ifdef N2V_NEAR_COUNT  // diagnostic build: pairings ([2]) / pairings the closed forms declined ([3])
  if (kMode != 2) {
    atomicAdd(n2v_count_words + 2, 1u);
    if (res < 0) atomicAdd(n2v_count_words + 3, 1u);
  }
#endif
#if defined(N2V_BIG_STATS) && defined(N2V_BIG_DECLINES)  // diagnostic: [2] then counts the big-row pairings the closed forms DECLINED
  if (n >= N2V_BIG_STATS && res < 0) atomicAdd(n2v_big_words + 2, 1u);
#endif
#ifdef N2V_DECLINE_STATS  // diagnostic: declined pairings on rows of more than 64 / of 4096 slots and more, by arrangement
  if (res < 0 && n > 64) atomicAdd(n2v_decline_words + 24 + arr, 1u);
  if (res < 0 && n >= 4096) atomicAdd(n2v_decline_words + 56 + arr, 1u);
  if (n > 64) atomicAdd(n2v_decline_words + 30, 1u);  // pairings on such rows
  if (n >= 4096) atomicAdd(n2v_decline_words + 62, 1u);
#endif
#ifdef N2V_FORCE_REPLAY  // test build: every pairing on a row of more than 64 slots is REPLAYED (the closed forms unused)
  res = -1;
#endif
  if (res >= 0) return res;
#ifdef N2V_ABLATE_STEP  // timing only: 4 = a pairing the closed forms decline keeps `pick` (no replay); 5 = on rows > 64
  if (N2V_ABLATE_STEP == 4 || (N2V_ABLATE_STEP == 5 && n > 64)) return pick;
#endif
  if constexpr (kMode == 2) {
    // a long row whose values are not dyadic: the closed forms on the reference's own values (the exact
    // row sum has been taken), with a margin that grows with n instead of n^2
#ifndef N2V_FORCE_REPLAY
    if (N2V_NEAR_FORMS && n >= kNearExactMin && arr >= 1) {
      res = near_listed_exact<P>(arr, n, pick, r2, K, avg, nR, rpos, nM, list, isR, isM, lo_pick, below);
      if (res >= 0) return res;
    }
#endif
  }
  const double vR = K.bR / avg, vM = K.bM / avg, vO = K.bO / avg;
  if (n <= 64) {  // a short row: the two stacks as bit masks
    uint64_t Rm = 0ull;
    if (nR) Rm = ((nR >= 64) ? ~0ull : ((1ull << nR) - 1ull)) << rpos;
    const uint64_t Mm = wedge_mask_l<P>(list, nM);
    return lane_pairing(n, Rm, Mm, pick, r2, vR, vM, vO);
  }
#ifdef N2V_DECLINE_STATS  // diagnostic: the cycles / 256 of the replays of long rows, and their number, by the length of the
  // list (words 64 + 2 b, 65 + 2 b of the decline words; b = 0: <= 64 entries, 1: <= 256, 2: <= 1024, 3: <= 4096, 4: more)
  const unsigned long long rep_t0 = __builtin_readcyclecounter();
  auto timed = [&](int result) -> int {
    const int b = nM <= 64 ? 0 : nM <= 256 ? 1 : nM <= 1024 ? 2 : nM <= 4096 ? 3 : 4;
    atomicAdd(n2v_decline_words + 64 + 2 * b, (uint32_t)((__builtin_readcyclecounter() - rep_t0) >> 8));
    atomicAdd(n2v_decline_words + 65 + 2 * b, 1u);
    return result;
  };
#else
  auto timed = [&](int result) -> int { return result; };
#endif
  if (arr == 1) return timed(lane_case_a<P>(n, pick, r2, vR, vM, vO, nR, rpos, nM, list, isR, isM, stage, lane));
  if (arr == 2) return timed(lane_case_b<P>(n, pick, r2, vR, vM, vO, nR, rpos, nM, list, isR, isM, lo_pick, below));
  if constexpr (kShared) {
    if (arr == 3) return lane_case_a2<P>(n, pick, r2, vR, vM, vO, nR, rpos, nM, list, isR, isM, stage, lane);
    if (arr == 4) return lane_case_b2<P>(n, pick, r2, vR, vM, vO, nR, rpos, nM, list, isR, isM);
    if (arr == 5) return lane_case_a3<P>(n, pick, r2, vR, vM, vO, nR, rpos, nM, list, isR, isM);
  }
  return lane_pairing_list<P>(n, pick, r2, vR, vM, vO, nR, rpos, nM, list);
}


// the closed-form half of pair_listed alone: the slot the draw returns, or -1 when fp64 rounding
// decides it (a tie, a thin margin, an arrangement without a closed form) -- the caller then has
// the step replayed (pair_listed) somewhere else.  kMode 0, 1, 3 (dyadic p, q).
template <typename P, int kMode>
__device__ __forceinline__ int jump_listed(int arr, int n, int pick, double r2, const UnitConsts &K,
                                           int nR, int rpos, int nM, ListRef<P> list, bool isR,
                                           bool isM, int lo_pick, int below = -1) {
  static_assert(kMode != 2, "values that are not dyadic have no closed form");
  if (arr == 1) return lane_case_a_jump<P>(n, pick, r2, K, nR, rpos, nM, list, isR, isM, lo_pick, below);
  if (arr == 2)
    return lane_case_b_jump<P, kMode != 0>(n, pick, r2, K, nR, rpos, nM, list, isR, isM, lo_pick, below);
  if constexpr (kMode == 1) {
    if (arr == 3) return lane_case_a2_jump<P>(n, pick, r2, K, nR, rpos, nM, list, isR, isM, lo_pick, below);
    if (arr == 4) return lane_case_b2_jump<P>(n, pick, r2, K, nR, rpos, nM, list, isR, isM, lo_pick, below);
    if (arr == 5) return lane_case_a3_jump<P>(n, pick, r2, K, nR, rpos, nM, isR, isM, lo_pick);
  }
  return -1;
}

// ---- per-edge wedge slots (n2v_wedge_slots_build, include/n2v_hip.h): 16 halfwords per edge ----
// A biased step reads the return position and the shared-position list of the edge it came along.
// Through wedge_off that is two DEPENDENT gathers (offset, then list); the slot of edge e sits at
// a fixed place, is requested together with the hop, and holds for a list of n <= kSlotShort
// entries the list itself:  [0] return position  [1] entries below the return position
//   n <= 14:  [2 .. 2 + n) the positions           n > 14:  [4 .. 8) offset of the list in wedge_pos
//                                                           (64 bits), [8 .. 16) eight pivots,
//                                                           list[((k + 1) n) / 9], k = 0 .. 7
// so that ~80 % of the steps that need a list (cfg 4: 39 % of all steps) take one sector instead
// of two, and a long list is entered at the right ninth (one more sector up to ~290 entries).
// (kSlotShort, kSlotPivots, slot_half, slot_lower: n2v_common.h -- the fast kernel reads slots too)

// what a kernel needs of (p, q) beyond UnitConsts, computed once
struct StepFlags {
  bool need_mem, always_pair, merge_r, w_wide, inline_rpos, folded;
};
__device__ __forceinline__ StepFlags step_flags(const n2v_graph &g, const UnitConsts &K, double q) {
  StepFlags f;
  f.need_mem = q != 1.0;
  f.always_pair = K.bO > 1.0;  // 1/q > 1: "other" overfull, an overfull `pick` has no quick exit
  f.merge_r = K.bR == K.bO;    // p == q: the return slot IS an "other" slot (:223-230)
  f.w_wide = g.wedge_wide == 1;  // (a mixed table, wedge_wide >= 2: by the row, in wedge_step)
  f.inline_rpos = (g.reserved2 & N2V_HOPS_INLINE_RPOS) != 0;  // (the slots kernel's hop table only)
  f.folded = (g.reserved2 & N2V_SLOTS_FOLDED) != 0;  // the edges into wide rows have folded lists and slots
  return f;
}

// One biased step (s >= 0) of a walker standing on v (row vb, n slots) that came along edge e_prev
// with class counts ec_prev: the slot sampling_from_alias returns, and in `h` the hop entry of that
// slot.  kJumpOnly: closed forms only; returns -1 when the step has to be replayed (h is then the
// entry of `pick`).  Otherwise the replays run here (pair_listed) and the result is always >= 0.
// kSlots: the edge's list comes from g.wedge_slots (16-bit positions only) instead of wedge_off.
// A saturated class count (tables that do not belong to this graph) flags N2V_ST_RANGE and keeps
// `pick`, as the one-launch kernel does.  kMode 0, 1, 3 (dyadic p, q); with kSlots also 2 (values
// that are not dyadic: reference-order row sum, every pairing replayed).
template <int kMode, bool kJumpOnly, bool kSlots>
__device__ __forceinline__ int wedge_step(const n2v_graph &g, const UnitConsts &K, const StepFlags &F,
                                          uint32_t u1, uint32_t u2, int32_t s, int64_t vb, int n,
                                          int64_t e_prev, uint32_t ec_prev, n2v_hop &h,
                                          uint32_t *stage, int lane, uint32_t *status,
                                          uint16_t *lds_list = nullptr);

// the step of a walker standing on a WIDE row of a mixed wedge table, out of line (N2V_WIDE_NOINLINE):
// it is taken by a few per cent of the steps at most, and inlined into the slots kernel its 32-bit
// instances of the closed forms cost every step registers (scratch 24 -> 40 B per lane in <0>, 100 -> 144 in <1>)
template <int kMode, bool kJumpOnly>
#ifdef N2V_WIDE_NOINLINE
__device__ __attribute__((noinline)) int
#else
__device__ __forceinline__ int
#endif
wedge_step_wide(const n2v_graph &g, const UnitConsts &K, const StepFlags &F, uint32_t u1, uint32_t u2, int32_t s,
                int64_t vb, int n, int64_t e_prev, uint32_t ec_prev, n2v_hop &h, uint32_t *stage, int lane,
                uint32_t *status) {
  StepFlags Fw = F;
  Fw.w_wide = true;
  return wedge_step<kMode, kJumpOnly, false>(g, K, Fw, u1, u2, s, vb, n, e_prev, ec_prev, h, stage, lane, status,
                                             nullptr);
}

template <int kMode, bool kJumpOnly, bool kSlots>
__device__ __forceinline__ int wedge_step(const n2v_graph &g, const UnitConsts &K, const StepFlags &F,
                                          uint32_t u1, uint32_t u2, int32_t s, int64_t vb, int n,
                                          int64_t e_prev, uint32_t ec_prev, n2v_hop &h,
                                          uint32_t *stage, int lane, uint32_t *status,
                                          uint16_t *lds_list) {
  constexpr bool kShared = kMode == 1 || kMode == 2;
  bool folded = false;  // this lane stands on a wide row and reads a folded slot / list
  if constexpr (kSlots) {
    // mixed wedge table (g.wedge_wide = T >= 2): the edges into a row of T entries or more have
    // uint32 lists and no slot -- the step of a walker standing on such a row goes through wedge_off
    // (a per-lane branch: the other lanes of the wave keep their slots)
    if (g.wedge_wide >= 2 && n >= g.wedge_wide) {
#if defined(N2V_ABLATE_WIDE) && N2V_ABLATE_WIDE == 1  // timing only: a wide step is a plain uniform draw
      const int pk = pick_index(u1, n);
      h = load_hop(g.hops + vb + pk);
      return pk;
#endif
      // (round 6) a wide row: its edges have FOLDED lists and slots (ListRef, n2v_wedge_slots_fold) and it takes the
      // step below like every other row.  (The launcher gives a mixed table without folded slots to the kernel that
      // reads wedge_off; rounds 4 - 5 stepped such a row here through a second, 32-bit instance of this function,
      // wedge_step_wide: every wave with one lane on a wide row executed both.)
      if (!(F.folded && n - g.wedge_wide <= 65536)) {  // slots that do not belong to this graph
        atomicOr(status, N2V_ST_RANGE);
        const int pk = pick_index(u1, n);
        h = load_hop(g.hops + vb + pk);
        return pk;
      }
      folded = true;
    }
  }
  const int pick = pick_index(u1, n);
  int idx = pick;
  const bool w_wide = F.w_wide || wedge_row_wide(g.wedge_wide, n);  // width of this step's list (!kSlots)
  // an edge without shared neighbours may carry its return position in the class word itself
  // (N2V_EC_INLINE, slots kernel only): its slot is then never fetched
  const bool inl = kSlots && F.inline_rpos && ec_prev != 0xffffffffu && (ec_prev & N2V_EC_INLINE) != 0u;
  const uint32_t fR = inl ? ((ec_prev >> N2V_EC_RETURN_SHIFT) & 0x7fu) : (ec_prev >> N2V_EC_RETURN_SHIFT);
  const uint32_t fM = inl ? 0u : (ec_prev & N2V_EC_SHARED_MASK);
  const bool counts_ok = inl || (fR != N2V_EC_RETURN_SAT && fM != N2V_EC_SHARED_MASK);
  // this step's wedge list: its offset (its slot) is requested before the hop so both loads
  // overlap; steps whose edge has no shared neighbour need it only if the pairing runs (lazy)
  uint64_t wraw = 0;
  int4 sa = make_int4(0, 0, 0, 0), sb = make_int4(0, 0, 0, 0);
  bool w_loaded = inl;
  if (inl) sa.x = (int)(ec_prev & 0xffffu);  // halfword 0: the return position (16-bit positions), 1: below = 0
  const uint16_t *slot = nullptr;
  if constexpr (kSlots) slot = reinterpret_cast<const uint16_t *>(g.wedge_slots) + e_prev * 16;
  // what the slot says once its first 16 bytes are in `sa`.  A FOLDED slot (n2v_wedge_slots_fold: the edge leads into
  // a wide row) packs three more numbers: whether the return position lies in the upper part, the entries of the list
  // in the lower part (ListRef::nlow), and the high bits of counts that a list of up to 2^20 entries needs --
  // a list that lives in the slot (<= 14 entries): halfword [1] = below | nlow << 4 | upper << 8; a longer one:
  // [1] = below, [2] = nlow (low 16 bits each), [3] = upper | (below >> 16) << 4 | (nlow >> 16) << 8
  const bool fshort = fM <= (uint32_t)kSlotShort;
  auto slot_rpos = [&]() -> int {
    int r = (int)((uint32_t)sa.x & 0xffffu);
    if (folded && (((fshort ? (uint32_t)sa.x >> 24 : (uint32_t)sa.y >> 16) & 1u) != 0u)) r += g.wedge_wide;
    return r;
  };
  auto slot_below = [&]() -> int {
    uint32_t b = (uint32_t)sa.x >> 16;
    if (folded) b = fshort ? (b & 0xfu) : (b | ((((uint32_t)sa.y >> 20) & 0xfu) << 16));
    return (int)b;
  };
  auto slot_nlow = [&]() -> int {
    if (!folded) return 0x7fffffff;
    return fshort ? (int)(((uint32_t)sa.x >> 20) & 0xfu)
                  : (int)(((uint32_t)sa.y & 0xffffu) | ((((uint32_t)sa.y >> 24) & 0xfu) << 16));
  };
  // (not dyadic: the steps past the quick accept need the return position and the list -- for the
  // closed forms with margins, else for the row sum --: fetched there if not here)
  if (!inl && counts_ok &&
      ((F.need_mem && fM > 0) || (F.always_pair && (fM > 0 || fR > 0)))) {
    if constexpr (kSlots) {
      // (asking for the second half only when the list has more than six entries was measured
      // and changes nothing: -3 .. +4 % by (p, q), profiles/r4i_time_slots_on_demand.log)
      sa = reinterpret_cast<const int4 *>(slot)[0];
      sb = reinterpret_cast<const int4 *>(slot)[1];
    } else {
      wraw = g.wedge_off[e_prev];
    }
    w_loaded = true;
  }
  // An edge whose class word carries its return position (`inl`: no shared neighbours) has everything the decision
  // needs in registers already -- the return run is the slots [rpos, rpos + nR), every other slot is "other" -- so the
  // hop entry of `pick` is not a gather the step has to wait for, and when the draw returns another slot it was a
  // gather for nothing (a fifth of the steps at (0.5, 2), most of them where "other" is overfull).  Such a step
  // decides first and gathers the entry of its RESULT: one gather, always.  (N2V_DEFER_HOP 0: rounds 4 - 5.)
  // N2V_DEFER_HOP 2 (the default): also the steps that have asked for their slot -- the hop entry then waits for the
  // slot, two dependent gathers where there were two parallel ones, and that costs nothing: the kernel is bound by
  // the NUMBER of random sectors, not by their latency (+2 - 3.5 % over 1 on every (p, q), two runs).
  // (A list that is not inside the slot is searched in memory, a chain of dependent probes: there the entry of `pick`
  // is requested at once, as before, and arrives behind them -- waiting with it cost the graph trimmed at the
  // reference's cap, whose hub steps are such searches, 10 %: 14.0 -> 12.6 G.)
  const bool defer = N2V_DEFER_HOP && (inl || (N2V_DEFER_HOP == 2 && kSlots && w_loaded && fM <= (uint32_t)kSlotShort));
  if (!defer) h = load_hop(g.hops + vb + pick);
  if (!counts_ok) {
    atomicOr(status, N2V_ST_RANGE);
    return idx;
  }
  const int nR = F.merge_r ? 0 : (int)fR, nM = F.need_mem ? (int)fM : 0, nO = n - nR - nM;
  int64_t w_off = (int64_t)(wraw & N2V_WEDGE_OFF_MASK);
  // (rows are sorted by neighbour: the slots that lead back to s are one run)
  const int rp0 = slot_rpos();
  const bool isR = defer ? (nR > 0 && pick >= rp0 && pick < rp0 + nR) : (!F.merge_r && h.col == s);
  bool isM = false;
  int lo_pick = 0;  // entries of the edge's list below `pick`
#ifdef N2V_ABLATE_STEP  // timing only: 1 = no search of a list that is not inside its slot
  if (N2V_ABLATE_STEP == 1 && nM > kSlotShort) {
  } else
#endif
  if (F.need_mem && !isR && nM > 0) {  // :226
    if constexpr (kSlots)
      lo_pick = slot_lower(sa, sb, nM, pick, reinterpret_cast<const uint16_t *>(g.wedge_pos), isM, slot_nlow(),
                           g.wedge_wide);
    else
      lo_pick = wedge_lower(g.wedge_pos, w_off, nM, pick, w_wide, isM);
  }
  const double r2 = (double)u2 * (1.0 / 4294967296.0);
  double avg;  // :172
  if constexpr (kMode == 2) {
    static_assert(!kJumpOnly, "values that are not dyadic: replays inline");
    // the reference's sum is rounded at every addition; any order of the same positive addends
    // agrees with it to (n - 1) 2^-53 relatively, so an underfull `pick` whose acceptance clears
    // that margin is decided from the counts alone; otherwise the row is added up in the
    // reference's order, run by run (lane_row_sum), the short list read from this lane's LDS row
    const double b_pick = pick3(isR, isM, K.bR, K.bM, K.bO);
    const double approx = ((double)nR * K.bR + (double)nM * K.bM + (double)nO * K.bO) / (double)n;
    const double eps = ((double)n + 8.0) * 4.5e-16;
    if (b_pick < approx * (1.0 - eps) && r2 < (b_pick / approx) * (1.0 - 2.0 * eps)) {
      avg = approx;  // only the (decided) comparison below reads it
    } else {
      // (round 4) Before the row is added up in the reference's order: the closed forms on the values
      // the COUNTS give, with margins (near_step, n2v_unit_near.h).  Anything closer than the margin
      // goes on to the exact row sum and the replays below.
#ifdef N2V_NEAR_COUNT
      atomicAdd(status + 2, 1u);
#endif
      if constexpr (kSlots) {
        if (!w_loaded) {  // an edge without shared neighbours whose step got here: its return position
          sa = reinterpret_cast<const int4 *>(slot)[0];
          sb = reinterpret_cast<const int4 *>(slot)[1];
          w_loaded = true;
        }
        if (N2V_NEAR_FORMS) {
          const uint16_t *nlist = slot + 2;
          if (nM > kSlotShort)
            nlist = reinterpret_cast<const uint16_t *>(g.wedge_pos) +
                    ((uint64_t)(uint32_t)sa.z | ((uint64_t)(uint32_t)sa.w << 32));
          const int res = near_step<uint16_t>(n, pick, r2, K, nR, slot_rpos(), nM,
                                              ListRef<uint16_t>(nlist, slot_nlow(), g.wedge_wide), isR, isM, lo_pick,
                                              slot_below());
          if (res >= 0) {
            if (defer || res != pick) h = load_hop(g.hops + vb + res);
            return res;
          }
#ifdef N2V_NEAR_COUNT  // diagnostic build: steps past the quick accept ([2]) / declined by the closed forms ([3])
          atomicAdd(status + 3, 1u);
#endif
#ifdef N2V_ABLATE_STEP  // timing only: 6 = a step the closed forms with margins decline keeps `pick` (no row sum, no replay)
          if (N2V_ABLATE_STEP == 6 || (N2V_ABLATE_STEP == 7 && n >= 4096)) {
            h = load_hop(g.hops + vb + pick);
            return pick;
          }
#endif
        }
        if (g.row_sums != nullptr && n >= g.row_sums_from) {
          // a long row: the sum of this edge's table was added up once (n2v_edge_row_sums_build: this routine, these bits)
          avg = g.row_sums[e_prev] / (double)n;
        } else {
        const uint16_t *sum_list = slot + 2;
        if (nM > kSlotShort) {
          sum_list = reinterpret_cast<const uint16_t *>(g.wedge_pos) +
                     ((uint64_t)(uint32_t)sa.z | ((uint64_t)(uint32_t)sa.w << 32));
        } else if (lds_list != nullptr) {
          uint32_t *row = reinterpret_cast<uint32_t *>(lds_list);
          row[0] = (uint32_t)sa.y;
          row[1] = (uint32_t)sa.z;
          row[2] = (uint32_t)sa.w;
          row[3] = (uint32_t)sb.x;
          row[4] = (uint32_t)sb.y;
          row[5] = (uint32_t)sb.z;
          row[6] = (uint32_t)sb.w;
          sum_list = lds_list;
        }
        avg = lane_row_sum<uint16_t>(n, K, nR, slot_rpos(), nM, ListRef<uint16_t>(sum_list, slot_nlow(), g.wedge_wide)) /
              (double)n;
        }
      } else {
        // through wedge_off (the wide rows of a mixed table): the same two stages on the list in memory
        if (!w_loaded) {
          wraw = g.wedge_off[e_prev];
          w_off = (int64_t)(wraw & N2V_WEDGE_OFF_MASK);
          w_loaded = true;
        }
        const int rp = (int)(wraw >> N2V_WEDGE_RPOS_SHIFT);
        int res = -1;
        if (N2V_NEAR_FORMS) {
          if (w_wide)
            res = near_step<uint32_t>(n, pick, r2, K, nR, rp, nM,
                                      reinterpret_cast<const uint32_t *>(g.wedge_pos) + w_off, isR, isM, lo_pick, -1);
          else
            res = near_step<uint16_t>(n, pick, r2, K, nR, rp, nM,
                                      reinterpret_cast<const uint16_t *>(g.wedge_pos) + w_off, isR, isM, lo_pick, -1);
          if (res >= 0) {
            if (res != pick) h = load_hop(g.hops + vb + res);
            return res;
          }
        }
        double sum;
        if (g.row_sums != nullptr && n >= g.row_sums_from)
          sum = g.row_sums[e_prev];
        else if (w_wide)
          sum = lane_row_sum<uint32_t>(n, K, nR, rp, nM, reinterpret_cast<const uint32_t *>(g.wedge_pos) + w_off);
        else
          sum = lane_row_sum<uint16_t>(n, K, nR, rp, nM, reinterpret_cast<const uint16_t *>(g.wedge_pos) + w_off);
        avg = sum / (double)n;
      }
    }
  } else {
    const int64_t isum = (int64_t)nR * K.TR + (int64_t)nM * K.TM + (int64_t)nO * K.TO;
    avg = ((double)isum * (1.0 / 1048576.0)) / (double)n;
#if defined(N2V_ABLATE_STEP) && N2V_ABLATE_STEP == 8  // timing only: neither division of the quick path (with 2: no pairing either)
    avg = ((double)isum * (1.0 / 1048576.0)) * 1e-3;
#endif
  }
#if defined(N2V_ABLATE_STEP) && N2V_ABLATE_STEP == 8
  const double p_pick = pick3(isR, isM, K.bR, K.bM, K.bO) * 0.9 + avg * 1e-30;
#else
  const double p_pick = pick3(isR, isM, K.bR, K.bM, K.bO) / avg;  // :173
#endif
  if (p_pick < 1.0 && r2 < p_pick) {  // an accepted underfull slot is final
    if (defer) h = load_hop(g.hops + vb + idx);
    return idx;
  }
#if defined(N2V_ABLATE_WIDE) && N2V_ABLATE_WIDE == 2  // timing only: a wide step never pairs
  if (!kSlots) return idx;
#endif
#ifdef N2V_ABLATE_STEP  // timing only: 2 = no pairing at all, 3 = none on rows of 4096 slots and more
  if (N2V_ABLATE_STEP == 2 || N2V_ABLATE_STEP == 8 || (N2V_ABLATE_STEP == 3 && n >= 4096)) {
    if (defer) h = load_hop(g.hops + vb + idx);
    return idx;
  }
#endif
  // underfull / overfull by class without dividing: fl(b / avg) < 1.0 <=> b < avg
  const bool uR = K.bR < avg, uM = K.bM < avg, uO = K.bO < avg;
  const bool any_under = (nR && uR) || (nM && uM) || (nO && uO);
  const bool any_over = (nR && !uR) || (nM && !uM) || (nO && !uO);
  if (!any_under || !any_over) {  // the loop of :182 never runs
    if (!(r2 < p_pick)) idx = 0;
  } else {
#ifdef N2V_BIG_STATS  // diagnostic build: pairings on rows of >= N2V_BIG_STATS slots ([2]) and the cycles / 256 they take ([3])
    const unsigned long long big_t0 = __builtin_readcyclecounter();
#endif
    if (!w_loaded) {  // the return position (and an empty list)
      if constexpr (kSlots) {
        sa = reinterpret_cast<const int4 *>(slot)[0];
      } else {
        wraw = g.wedge_off[e_prev];
        w_off = (int64_t)(wraw & N2V_WEDGE_OFF_MASK);
      }
    }
    int w_rpos;
    if constexpr (kSlots)
      w_rpos = slot_rpos();
    else
      w_rpos = (int)(wraw >> N2V_WEDGE_RPOS_SHIFT);
    // the stacks: 1 = "other" alone underfull, 2 = "other" alone overfull, 3 = return + "other"
    // underfull, 4 = return + "other" overfull, 5 = return alone overfull, 0 = else
    int arr = 0;
    if (uO && !(nR && uR) && !(nM && uM)) arr = 1;
    else if (!uO && nO > 0 && (!nR || uR) && (!nM || uM)) arr = 2;
    else if (kShared && uO && nR && uR && nM && !uM) arr = 3;
    else if (kShared && !uO && nO > 0 && nR && !uR && nM && uM) arr = 4;
    else if (kShared && uO && nR && !uR && nM && uM) arr = 5;
    if constexpr (kSlots) {
      // entries of the list below the return position (stored: saves the routines a search)
      const int w_below = slot_below();
      // the list as the pairing routines read it: inside the slot, or in wedge_pos
      const uint16_t *list_p = slot + 2;
      if (nM > kSlotShort)
        list_p = reinterpret_cast<const uint16_t *>(g.wedge_pos) +
                 ((uint64_t)(uint32_t)sa.z | ((uint64_t)(uint32_t)sa.w << 32));
      const ListRef<uint16_t> list(list_p, slot_nlow(), g.wedge_wide);
      if constexpr (kJumpOnly) {
        idx = jump_listed<uint16_t, kMode>(arr, n, pick, r2, K, nR, w_rpos, nM, list, isR, isM, lo_pick,
                                           w_below);
        if (idx < 0) {
          if (defer) h = load_hop(g.hops + vb + pick);
          return -1;
        }
      } else {
        bool done = false;
        if (kMode != 2 && lds_list != nullptr && nM <= kSlotShort) {
          // (optional, measured and NOT used by the kernels: a short list copied from the slot's
          // registers to 32 bytes of LDS per lane for the closed forms to read -- 20.3 against
          // 23.2 G steps/s at (0.5, 2), 16.4 against 18.6 G at (4, 0.25): the flat loads and the
          // LDS it takes (5 waves per SIMD instead of 6) cost more than probes of a sector that
          // is still cached, profiles/r4k_time_lds_list_cfg4.log)
          uint32_t *row = reinterpret_cast<uint32_t *>(lds_list);
          row[0] = (uint32_t)sa.y;
          row[1] = (uint32_t)sa.z;
          row[2] = (uint32_t)sa.w;
          row[3] = (uint32_t)sb.x;
          row[4] = (uint32_t)sb.y;
          row[5] = (uint32_t)sb.z;
          row[6] = (uint32_t)sb.w;
          if constexpr (kMode != 2)
            idx = jump_listed<uint16_t, kMode>(arr, n, pick, r2, K, nR, w_rpos, nM,
                                               ListRef<uint16_t>(lds_list, slot_nlow(), g.wedge_wide), isR, isM,
                                               lo_pick, w_below);
          done = idx >= 0;
        }
        if (!done)  // long lists; a tie or a thin margin: the replays read the list in memory
          idx = pair_listed<uint16_t, kMode>(arr, n, pick, r2, K, avg, nR, w_rpos, nM, list, isR, isM,
                                             lo_pick, reinterpret_cast<uint16_t *>(stage), lane, w_below);
      }
    } else if constexpr (kJumpOnly) {
      // a plain branch on the (uniform) list width: never a select between two loads
      if (w_wide)
        idx = jump_listed<uint32_t, kMode>(arr, n, pick, r2, K, nR, w_rpos, nM,
                                           reinterpret_cast<const uint32_t *>(g.wedge_pos) + w_off,
                                           isR, isM, lo_pick);
      else
        idx = jump_listed<uint16_t, kMode>(arr, n, pick, r2, K, nR, w_rpos, nM,
                                           reinterpret_cast<const uint16_t *>(g.wedge_pos) + w_off,
                                           isR, isM, lo_pick);
      if (idx < 0) return -1;
    } else {
      if (w_wide)
        idx = pair_listed<uint32_t, kMode>(arr, n, pick, r2, K, avg, nR, w_rpos, nM,
                                           reinterpret_cast<const uint32_t *>(g.wedge_pos) + w_off,
                                           isR, isM, lo_pick, stage, lane);
      else
        idx = pair_listed<uint16_t, kMode>(arr, n, pick, r2, K, avg, nR, w_rpos, nM,
                                           reinterpret_cast<const uint16_t *>(g.wedge_pos) + w_off,
                                           isR, isM, lo_pick, reinterpret_cast<uint16_t *>(stage), lane);
    }
#ifdef N2V_BIG_STATS
    if (n >= N2V_BIG_STATS && idx >= 0) {
#ifndef N2V_BIG_DECLINES
      atomicAdd(status + 2, 1u);
#endif
      atomicAdd(status + 3, (uint32_t)((__builtin_readcyclecounter() - big_t0) >> 8));
    }
#endif
  }
  if (defer || idx != pick) h = load_hop(g.hops + vb + idx);
  return idx;
}

}  // namespace n2v
