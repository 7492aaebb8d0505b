// n2v_walk_wlanes.hip -- K2 exact mode on WEIGHTED graphs with p or q != 1: ONE STEP of every
// walker of a batch (n2v_walk_weighted_step, include/n2v_hip.h), the walkers ordered by the length
// of the row they stand on.  Three kernels, in the order they were written:
//   1. walk_weighted_step_kernel        a LANE per walker, the pairing loop REPLAYED with O(1) state (below);
//                                       what runs when the caller lends no scratch / row sums
//   2. walk_weighted_margin_kernel      a WAVE per walker on the long rows (768 slots and more): the pairing is
//                                       not replayed but DECIDED -- the one slot the draw asks for, from sums over
//                                       the row, every comparison with a margin that covers the roundings of the
//                                       reference's loop; what the margins cannot decide gets a second chance on
//                                       the reference-order row sum (the kSeq instance), then the exact wave
//                                       kernel of n2v_walk.hip
//   3. walk_weighted_lane_margin_kernel the same decision with a LANE per walker on the rows below 768 slots
// With 2 and 3 weighted cfg 2 walks at 0.79 G steps/s at (0.5, 2) where 1 alone reached 0.06 G and the
// wave-per-walker kernel of n2v_walk.hip 0.04 G (DESIGN.md 5, K2 exact, weighted graphs).
//
// The reference rebuilds the whole table of the row a walker stands on at every step
// (generate_edge_alias_tables + generate_alias_tables, randomwalk.py:157-232): bias every weight by
// the class of its slot (:219-231), add the row up left to right in fp64 (:172), divide, and pair
// underfull with overfull slots from the TOP of two stacks until both run out (:182-189).  With
// arbitrary weights none of that has a closed form -- every value of the table is different -- so
// the work per step is O(row), and two parts of it are inherently serial: the left-to-right sum
// (one rounding per addition) and the pairing.  n2v_walk.hip gives a walker a whole wave: the
// streaming passes use 64 lanes, the two serial parts use one (cfg 2 weighted: mean visited row
// 947 entries, 70 % of the steps pair, 171 pairings on average: 36 - 53 M steps/s).
//
// Here a walker gets ONE lane and the 64 serial chains of a wave run side by side.  That only pays
// when the lanes of a wave have rows of about the same length, so the walk is STEP-SYNCHRONOUS: the
// host orders the walkers of a step by the degree of the vertex they stand on (one sort of 4-byte
// keys per step: a fraction of a per cent of the step) and lane i of the launch takes walker
// order[i].  Per step and lane:
//   * the classes of the slots come from the per-edge tables of the edge walked last -- return run
//     (edge_classes, wedge_off >> 40) and shared positions (the wedge list) -- which depend on the
//     ids alone and are built for weighted graphs exactly as for unit ones: no search over N(s), no
//     pass over col;
//   * SUM: one forward pass over the weights, 8 per load group, the reference's additions in the
//     reference's order; avg = sum / n; an accepted underfull `pick` and a row with an empty stack
//     leave here (:182 never runs);
//   * PAIRING: the loop of :182-189 replayed with O(1) state by two cursors that run DOWN the row,
//     one yielding the underfull slots, one the overfull ones (an overfull slot that is demoted is
//     the next `under`, so nothing is ever pushed): the same fp64 operations in the same order as the
//     reference's, until slot `pick` has its final (alias, probs).  Each cursor keeps the biased
//     weights of its current 8 slots in the lane's own LDS column.
// Same uniform stream, same table, same draw as every other exact kernel: bit-identical walks
// (tests/test_weighted_lanes_gpu.py: the fp64 goldens, the oracle, walk_exact_kernel over a batch).
#include "n2v_common.h"

namespace n2v {

// Two instances: rows of up to N2V_WLANES_SHORT slots in groups of 8 (blocks of 256 lanes, 4 waves
// per SIMD); longer rows in groups of 32 (blocks of 64 lanes): a lane reads its row group by group and
// each group is a round trip to memory that nothing hides but the next group's load, so on a row of
// 10^4 - 10^5 slots the group must be long (8-slot groups: 75 ms per step for the wave that stands on
// the hubs of cfg 2, whatever the batch; profiles/r7f_time_wlanes_first.log).

struct WlConsts {
  double p, q, inv_p, inv_q;
  int p_pow2, q_pow2;  // w / p == w * (1 / p) bit for bit when p is a power of two
  int coef_bits;       // significant bits of 1 - 1/q and 1/p - 1/q (powers of two p, q; see wm_draw)
};

// what a lane knows about the row it stands on
struct WlRow {
  int n;          // slots
  int nR, rpos;   // return run [rpos, rpos + nR)
  int nM;         // shared positions: list[0, nM), ascending
  const void *list;
  bool wide;      // uint32 list entries (a wide row of a mixed wedge table), else uint16
  bool first;     // first step: the unbiased table of the row (:320-321)
};

__device__ __forceinline__ int wl_list_at(const WlRow &R, int k) {
  // (a plain branch on the width: never a select between two loads)
  if (R.wide) return (int)reinterpret_cast<const uint32_t *>(R.list)[k];
  return (int)reinterpret_cast<const uint16_t *>(R.list)[k];
}

// A stream over the shared positions of the edge walked last, ascending (kFwd) or descending.  The
// obvious form -- load list[k] when position list[k - 1] has gone by -- puts a memory round trip on the
// critical path of EVERY slot: a lane meets a listed position every ~20 slots, but with 64 lanes in
// step some lane meets one at nearly every slot and the whole wave waits for its load (measured: 78 ms
// per step on cfg 2 whatever the batch, ~1 us per slot of the longest row; profiles/r7j_*).  So the
// entries come four at a time (one 8-byte load of uint16 positions), and the NEXT four are requested
// when a window is opened: by the time they are needed -- ~80 slots later -- they have arrived.  The
// last 1 - 3 entries of a list (no full window left) and the lists of wide rows (uint32) are read one
// by one.
template <bool kFwd>
struct WlList {
  const WlRow *R;
  int idx;       // index of the entry `cur` holds (kFwd: ascending from 0; else descending from nM - 1)
  int cur;       // its value; past the end: 0x7fffffff (kFwd) / -1
  uint64_t win;  // the window idx lies in: entries [base, base + 4)
  uint64_t nxt;  // the following window (requested when `win` was opened)
  int have_nxt;  // nxt was requested

  __device__ __forceinline__ static uint64_t load4(const WlRow &R, int k) {  // entries [k, k + 4), uint16
    struct __attribute__((packed, aligned(2))) Q {
      uint16_t v[4];
    };
    const Q q = *reinterpret_cast<const Q *>(reinterpret_cast<const uint16_t *>(R.list) + k);
    return (uint64_t)q.v[0] | ((uint64_t)q.v[1] << 16) | ((uint64_t)q.v[2] << 32) | ((uint64_t)q.v[3] << 48);
  }
  __device__ __forceinline__ int base_of(int i) const {  // first entry of the window entry i lies in
    return kFwd ? (i & ~3) : (R->nM - 1 - ((R->nM - 1 - i) & ~3)) - 3;
  }
  __device__ __forceinline__ bool full(int b) const { return b >= 0 && b + 4 <= R->nM; }
  __device__ __forceinline__ void init(const WlRow &row) {
    R = &row;
    win = nxt = 0ull;
    have_nxt = 0;
#if defined(N2V_WL_ABLATE) && (N2V_WL_ABLATE & 2)  // timing only: no shared positions
    const int n = 0;
#else
    const int n = row.first ? 0 : row.nM;
#endif
    idx = kFwd ? 0 : n - 1;
    if (n == 0) {
      cur = kFwd ? 0x7fffffff : -1;
      idx = kFwd ? 0 : -1;
      return;
    }
    open();
  }
  // idx entered a new window (or the stream starts): fetch it, request the one after it
  __device__ __forceinline__ void open() {
    const int b = base_of(idx);
    if (!R->wide && full(b)) {
      win = have_nxt ? nxt : load4(*R, b);
      const int b2 = kFwd ? b + 4 : b - 4;
      have_nxt = full(b2) ? 1 : 0;
      if (have_nxt) nxt = load4(*R, b2);
      cur = (int)((win >> (16 * (idx - b))) & 0xffffull);
    } else {
      have_nxt = 0;
      cur = wl_list_at(*R, idx);  // the tail of the list / a wide row: one by one
    }
  }
  // a stream that starts at entry `at` (descending: the entries above it are not looked at)
  __device__ __forceinline__ void init_at(const WlRow &row, int at) {
    R = &row;
    win = nxt = 0ull;
    have_nxt = 0;
    const int n = row.first ? 0 : row.nM;
    idx = at;
    if (n == 0 || at < 0 || at >= n) {
      cur = kFwd ? 0x7fffffff : -1;
      idx = kFwd ? n : -1;
      return;
    }
    open();
  }
  __device__ __forceinline__ void advance() {
    const int n = R->nM;
    idx += kFwd ? 1 : -1;
    if (idx < 0 || idx >= n) {
      cur = kFwd ? 0x7fffffff : -1;
      return;
    }
    const int b = base_of(idx);
    const bool same = kFwd ? (idx != b) : (idx != b + 3);  // still inside the window opened last
    if (same && !R->wide && full(b))
      cur = (int)((win >> (16 * (idx - b))) & 0xffffull);
    else
      open();
  }
};

// biased weight of a slot (:219-231): cls 0 = other (w / q), 1 = shared or first step (w), 2 = return (w / p).
// kPow2 (p and q powers of two): w / x == w * (1 / x) bit for bit, one multiplication; else ONE division by
// the selected divisor (cls 1 divides by 1.0: exact) -- never the instructions of two divisions.
template <bool kPow2>
__device__ __forceinline__ double wl_bias(double w, int cls, const WlConsts &K) {
  if constexpr (kPow2) {
    const double f = cls == 1 ? 1.0 : (cls == 2 ? K.inv_p : K.inv_q);
    return w * f;
  } else {
    const double d = cls == 1 ? 1.0 : (cls == 2 ? K.p : K.q);
    return w / d;
  }
}

// 8 consecutive weights of a row as stored (slots at or beyond n: 0)
template <typename WT, int CH>
struct WlRaw {
  WT v[CH];
};
template <typename WT, int CH>
__device__ __forceinline__ WlRaw<WT, CH> wl_load_raw(const WT *w, int c0, int n) {
  WlRaw<WT, CH> out;
#if defined(N2V_WL_ABLATE) && (N2V_WL_ABLATE & 1)  // timing only: no weight loads
#pragma unroll
  for (int k = 0; k < CH; ++k) out.v[k] = (WT)(1 + ((c0 + k) & 3));
  return out;
#endif
  if (c0 + CH <= n) {
    // whole group inside the row: wide loads (dword-aligned only: rows start anywhere)
    struct __attribute__((packed, aligned(4))) Pack {
      WT v[CH];
    };
    const Pack pk = *reinterpret_cast<const Pack *>(w + c0);
#pragma unroll
    for (int k = 0; k < CH; ++k) out.v[k] = pk.v[k];
  } else {
#pragma unroll
    for (int k = 0; k < CH; ++k) out.v[k] = (c0 + k < n) ? w[c0 + k] : (WT)0;
  }
  return out;
}

// one of the two cursors of the pairing: runs down the row and yields, in descending position, the
// slots that are underfull (kUnder) / not underfull.  The biased weights of its current group of 8
// sit in the lane's LDS column `tile[k][tid]`; the weights of the group below are already on their
// way (a lane alone cannot hide the latency of its own loads: one group ahead nearly does).
template <typename WT, int CH>
struct WlCursor {
  int chunk;        // group loaded last (groups above it are done)
  uint32_t mask;    // slots of that group still to yield
  WlList<false> lst;  // the shared positions, descending
  int nxt_chunk;    // the group whose weights `nxt` holds (requested, maybe not arrived), or -1
  WlRaw<WT, CH> nxt;
};

template <typename WT, int CH, int TH, bool kPow2, bool kUnder>
__device__ __forceinline__ void wl_refill(WlCursor<WT, CH> &C, const WlRow &R, const WT *w, const WlConsts &K,
                                          double avg, double *tile, int tid) {
  while (C.mask == 0u && C.chunk > 0) {
    --C.chunk;
    const int c0 = C.chunk * CH;
    WlRaw<WT, CH> raw;
    if (C.nxt_chunk == C.chunk)
      raw = C.nxt;
    else
      raw = wl_load_raw<WT, CH>(w, c0, R.n);
    if (C.chunk > 0) {  // the group below: requested now, read at the next refill
      C.nxt = wl_load_raw<WT, CH>(w, c0 - CH, R.n);
      C.nxt_chunk = C.chunk - 1;
    }
    uint32_t mm = 0u;  // shared slots of this group
    while (C.lst.cur >= c0) {
      mm |= 1u << (C.lst.cur - c0);
      C.lst.advance();
    }
    uint32_t mask = 0u;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int j = c0 + k;
      int cls = 1;
      if (!R.first) cls = ((mm >> k) & 1u) ? 1 : ((j >= R.rpos && j < R.rpos + R.nR) ? 2 : 0);
      const double b = wl_bias<kPow2>((double)raw.v[k], cls, K);
      tile[k * TH + tid] = b;
      // probs[i] < 1.0 (:175-180) <=> fl(b / avg) < 1.0 <=> b < avg for a correctly rounded quotient
      const bool under = b < avg;
      if (j < R.n && under == kUnder) mask |= 1u << k;
    }
    C.mask = mask;
  }
}

// index sampling_from_alias(r1, r2) returns on the table of this row, or -1: ZeroDivisionError (:172-173)
#ifdef N2V_WL_STATS  // diagnostic build: the longest sum pass / pairing of a launch (cycles >> 8) and its row
__device__ uint32_t *wl_stats_words;
#endif

template <typename WT, int CH, int TH, bool kPow2>
__device__ __forceinline__ int wl_draw(const WlRow &R, const WT *w, const WlConsts &K, int pick, double r2,
                                       double *tU, double *tO, int tid) {
  const int n = R.n;
#ifdef N2V_WL_STATS
  const unsigned long long st0 = __builtin_readcyclecounter();
#endif
  // ---- the row sum in the reference's order (:172) -----------------------------------------------
  double total = 0.0, b_pick = 0.0;
  double bmin = __builtin_huge_val(), bmax = -__builtin_huge_val();
  WlList<true> fwd;
  fwd.init(R);
  WlRaw<WT, CH> ahead = wl_load_raw<WT, CH>(w, 0, n);
  for (int c0 = 0; c0 < n; c0 += CH) {
    const WlRaw<WT, CH> raw = ahead;
    if (c0 + CH < n) ahead = wl_load_raw<WT, CH>(w, c0 + CH, n);  // one group ahead
    // the shared slots of this group, ONCE per group: inside the slot loop the list would be advanced at
    // every slot at which ANY of the 64 lanes meets a listed position -- nearly every slot
    uint32_t mm = 0u;
    while (fwd.cur < c0 + CH) {
      mm |= 1u << (fwd.cur - c0);
      fwd.advance();
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int j = c0 + k;
      int cls = 1;
      if (!R.first) cls = ((mm >> k) & 1u) ? 1 : ((j >= R.rpos && j < R.rpos + R.nR) ? 2 : 0);
      const double b = wl_bias<kPow2>((double)raw.v[k], cls, K);
      if (j < n) {
        total = total + b;  // one rounding per addition, left to right
        bmin = fmin(bmin, b);
        bmax = fmax(bmax, b);
        b_pick = j == pick ? b : b_pick;
      }
    }
  }
#ifdef N2V_WL_STATS
  const unsigned long long st1 = __builtin_readcyclecounter();
  atomicMax(wl_stats_words + 2, (uint32_t)((st1 - st0) >> 8));
#endif
  const double avg = total / (double)n;  // :172
  if (avg == 0.0) return -1;
  const double p_pick = b_pick / avg;    // :173
  if (p_pick < 1.0 && r2 < p_pick) return pick;  // an untouched underfull slot: final
  // x -> x / avg is monotone: the extreme weights say whether a stack is empty (:182 never runs)
  if (!(bmin / avg < 1.0) || (bmax / avg < 1.0)) return (r2 < p_pick) ? pick : 0;

  // ---- the pairing loop (:182-189) until slot `pick` is final -------------------------------------
  const int nch = (n + CH - 1) / CH;
  WlCursor<WT, CH> U, O;
  U.chunk = O.chunk = nch;
  U.mask = O.mask = 0u;
  U.lst.init(R);
  O.lst.init(R);
  U.nxt_chunk = O.nxt_chunk = -1;
  bool carry = false;  // the slot demoted last is the next `under`
  double carry_r = 0.0;
  int carry_idx = 0;
  double fin_prob = p_pick;
  int fin_alias = 0;
  for (;;) {
    wl_refill<WT, CH, TH, kPow2, false>(O, R, w, K, avg, tO, tid);
    if (O.mask == 0u) {  // `overfull` is empty: a demoted slot keeps alias 0
      if (carry && carry_idx == pick) fin_prob = carry_r;
      break;
    }
    const int ko = 31 - __clz(O.mask);
    O.mask ^= 1u << ko;
    const int o_idx = O.chunk * CH + ko;
    double r = tO[ko * TH + tid] / avg;  // probs[over]
    if (carry) {
      if (carry_idx == pick) {  // alias[under] = over; probs[under] is final
        fin_prob = carry_r;
        fin_alias = o_idx;
        break;
      }
      r = r + carry_r - 1.0;  // probs[over] = probs[over] + probs[under] - 1.0  (:187)
      carry = false;
      if (r < 1.0) {
        carry = true;
        carry_r = r;
        carry_idx = o_idx;
        continue;
      }
    }
    bool finished = false;
    for (;;) {  // `over` absorbs underfull slots while it stays >= 1.0
      wl_refill<WT, CH, TH, kPow2, true>(U, R, w, K, avg, tU, tid);
      if (U.mask == 0u) {  // `underfull` is empty
        if (o_idx == pick) fin_prob = r;
        finished = true;
        break;
      }
      const int ku = 31 - __clz(U.mask);
      U.mask ^= 1u << ku;
      const int u_idx = U.chunk * CH + ku;
      const double pu = tU[ku * TH + tid] / avg;  // probs[under]
      if (u_idx == pick) {
        fin_prob = pu;
        fin_alias = o_idx;
        finished = true;
        break;
      }
      r = r + pu - 1.0;
      if (r < 1.0) {
        carry = true;
        carry_r = r;
        carry_idx = o_idx;
        break;
      }
    }
    if (finished) break;
  }
#ifdef N2V_WL_STATS
  {
    const uint32_t dt = (uint32_t)((__builtin_readcyclecounter() - st1) >> 8);
    const uint32_t old = atomicMax(wl_stats_words + 3, dt);
    if (dt > old) wl_stats_words[1] = (uint32_t)n;  // (racy: the row of a longest pairing)
  }
#endif
  return (r2 < fin_prob) ? pick : fin_alias;  // :95-99
}

template <typename WT, int CH, int TH, bool kPow2>
__global__ __launch_bounds__(TH) void walk_weighted_step_kernel(
    n2v_graph g, const WT *__restrict__ w, const int32_t *__restrict__ start_ids, int32_t num_walks,
    const int64_t *__restrict__ order, int64_t n_rows, int32_t min_n, int32_t max_n, int32_t step,
    int32_t walk_length, WlConsts K,
    uint64_t seed, int64_t *__restrict__ edge_state, int32_t *__restrict__ walks,
    uint8_t *__restrict__ valid, uint32_t *__restrict__ status) {
  __shared__ double tU[CH * TH], tO[CH * TH];
  const int tid = threadIdx.x;
  const int L1 = walk_length + 1;
  const bool biased = !(K.p == 1.0 && K.q == 1.0);
#ifdef N2V_WL_STATS
  wl_stats_words = status;
#endif
  for (int64_t i = (int64_t)blockIdx.x * TH + tid; i < n_rows; i += (int64_t)gridDim.x * TH) {
    const int64_t r = order ? order[i] : i;
    if (r < 0 || r >= n_rows) {
      atomicOr(status, N2V_ST_RANGE);
      continue;
    }
    int32_t *row = walks + r * (int64_t)L1;
    const int32_t v = row[step];
    if (v < 0 || !valid[r]) continue;  // a walker that has vanished (or never started)
    if ((int64_t)v >= g.n_vertices) {
      atomicOr(status, N2V_ST_RANGE);
      continue;
    }
    const int32_t s = step > 0 ? row[step - 1] : -1;
    const int64_t vb = g.rowptr[v];
    WlRow R;
    R.n = (int)(g.rowptr[v + 1] - vb);
    if (R.n <= min_n && order && min_n > 0) break;  // ordered by row length: every later row is shorter still
    if (R.n <= min_n || R.n > max_n) continue;  // (the other instance's rows)
    R.first = s < 0 || !biased;
    R.nR = R.nM = R.rpos = 0;
    R.list = nullptr;
    R.wide = false;
    bool ok = true;
    if (!R.first) {
      const int64_t e_prev = edge_state[r];
      if (e_prev < 0 || e_prev >= g.n_edges) {
        ok = false;
      } else {
        const uint32_t ec = g.edge_classes[e_prev];
        const uint64_t wraw = g.wedge_off[e_prev];
        const uint32_t fR = ec >> N2V_EC_RETURN_SHIFT, fM = ec & N2V_EC_SHARED_MASK;
        R.nR = (int)fR;
        R.nM = (int)fM;
        R.rpos = (int)(wraw >> N2V_WEDGE_RPOS_SHIFT);
        R.wide = wedge_row_wide(g.wedge_wide, R.n);
        const uint64_t off = wraw & N2V_WEDGE_OFF_MASK;
        R.list = R.wide ? (const void *)(reinterpret_cast<const uint32_t *>(g.wedge_pos) + off)
                        : (const void *)(reinterpret_cast<const uint16_t *>(g.wedge_pos) + off);
        // a saturated count, or counts that cannot belong to this row: tables of another graph
        ok = fR != N2V_EC_RETURN_SAT && fM != N2V_EC_SHARED_MASK && (int64_t)fR + (int64_t)fM <= R.n &&
             R.rpos + (int)fR <= R.n;
      }
    }
    if (!ok) {
      atomicOr(status, N2V_ST_RANGE);
      continue;
    }
    const uint64_t key = (uint64_t)start_ids[r / num_walks] * (uint64_t)num_walks + (uint64_t)(r % num_walks);
    const uint64_t bits = step_bits(walker_stream(seed, key), (uint32_t)step);
    const uint32_t u1 = (uint32_t)(bits >> 32), u2 = (uint32_t)bits;
    const int pick = pick_index(u1, R.n);
    const double r2 = (double)u2 * (1.0 / 4294967296.0);
    const int idx = wl_draw<WT, CH, TH, kPow2>(R, w + vb, K, pick, r2, tU, tO, tid);
    if (idx < 0) {  // ZeroDivisionError (:172-173): the walk ends here, the caller raises
      atomicOr(status, N2V_ST_ZERODIV);
      valid[r] = 0;
      continue;
    }
    const int64_t e = vb + idx;
    const int32_t x = g.col[e];
    row[step + 1] = x;
    edge_state[r] = e;
    if (step + 1 < walk_length) {
      // fugue.py:147: a walker that reaches a vertex without out-edges vanishes
      if (x < 0 || (int64_t)x >= g.n_vertices || g.rowptr[x + 1] == g.rowptr[x]) valid[r] = 0;
    }
  }
}


// ---- long rows: one WAVE per walker, the pairing DECIDED, not replayed ------------------------------
// The pairing of :182-189 is serial, but what a draw needs of it is little: the final (alias, probs) of
// ONE slot.  Take the slots in the order the stacks are popped (descending position) and write
// d_i = 1 - probs[i] for the underfull ones, e_i = probs[i] - 1 for the others, D_j / E_k for their
// running sums from the top.  In exact arithmetic the loop is a merge of the two sums: the k-th
// overfull slot stays `over` while E_k - D_j >= 0, i.e. it absorbs underfull slots until D_j > E_k, is then
// demoted with probs = 1 + E_k - D_j and is the next `under` of the (k + 1)-th; so
//   * an underfull `pick` (the j-th) keeps its probs and gets alias = the first k with E_k >= D_{j-1};
//   * an overfull `pick` (the k-th) ends with probs = 1 + E_k - D_j at the first j with D_j > E_k and
//     alias = the next overfull slot below it -- or is never demoted (probs >= 1: the draw returns it).
// Sums, not a replay: a wave streams the row twice (the row sum; the chunk sums of d and e) and looks at
// one or two chunks again.  In floating point the reference's loop rounds twice per pairing
// (probs[over] + probs[under] - 1.0) and this kernel adds in another order and from an average that was
// summed in another order, so every comparison is made with a MARGIN M = 16 n^2 2^-52 that covers both
//   (the loop's residual differs from E_k - D_j by at most 2 n 2^-53 (n + 1) whatever path it took: each
//    pairing rounds twice at magnitude <= probs[over] <= n; the sums here differ from the exact ones by
//    at most (2 n + 4) 2^-53 of sum(probs) = n for the average, plus log2 n roundings of the tree, plus
//    2 (2 n + 4) 2^-53 per slot whose side of 1.0 cannot be told: together < 6 n^2 2^-53)
// and a draw that any comparison cannot decide by that margin -- r2 against probs, a sum against the
// sum it crosses, `pick` against 1.0 -- is NOT decided here: the walker goes on a list and the exact
// wave kernel (n2v_weighted_step_wave_launch) steps it.  With real-valued weights that is one step in
// ~10^6; rows whose sums tie exactly (few distinct weights) are undecided often and walk at the exact
// kernel's rate.  Same draws either way.
constexpr int kWmUndecided = -2;
// running sums kept per wave: one entry per 4 blocks of 256 slots (a sum over the wave per entry, not per block;
// 2 and 8 blocks measured the same, profiles/r8y_wm_variants.log), more blocks per entry on rows of more than
// 128 x 1 024 slots
constexpr int kWmEntries = 128;
#ifndef N2V_WM_MIN_BLOCKS_PER_ENTRY
#define N2V_WM_MIN_BLOCKS_PER_ENTRY 4
#endif
constexpr int kWmMinBlocksPerEntry = N2V_WM_MIN_BLOCKS_PER_ENTRY;
struct WmLds {
  double cd[kWmEntries], cx[kWmEntries];  // sum of d / of x = probs - 1 over the slots up to the end of entry i
  int lm0[kWmEntries];                    // how many shared positions lie below the first slot of entry i
  uint32_t flags[64];                     // one byte per slot of a block of 256: the slot is a shared position
};

__device__ __forceinline__ int64_t readlane_i64(int64_t v, int lane) {
  const int lo = __builtin_amdgcn_readlane((int)(uint32_t)v, lane);
  const int hi = __builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), lane);
  return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
// one DPP step on a 64-bit value: the value of the lane selected by kCtrl (no LDS round trip)
template <int kCtrl>
__device__ __forceinline__ uint64_t wm_dpp_u64(uint64_t x) {
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)x, kCtrl, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(x >> 32), kCtrl, 0xF, 0xF, true);
  return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
template <int kCtrl>
__device__ __forceinline__ double wm_dpp_f64(double x) {
  return __longlong_as_double((long long)wm_dpp_u64<kCtrl>((uint64_t)__double_as_longlong(x)));
}
// Sum over the wave as a balanced tree (the order does not matter to the caller: margins): distances 1, 2,
// 4, 8 inside a row of 16 lanes are DPP moves, the four row sums are read with v_readlane -- six dependent
// ds_bpermute round trips per sum made this kernel 10 x slower (profiles/r7y_wm_kernel_stats.csv).  The
// result is uniform.
__device__ __forceinline__ double wm_wave_sum(double x) {
  x = x + wm_dpp_f64<0xB1>(x);   // quad_perm [1,0,3,2]: lane ^ 1
  x = x + wm_dpp_f64<0x4E>(x);   // quad_perm [2,3,0,1]: lane ^ 2
  x = x + wm_dpp_f64<0x141>(x);  // row_half_mirror: the other quad
  x = x + wm_dpp_f64<0x140>(x);  // row_mirror: the other half row
  return (readlane_f64(x, 0) + readlane_f64(x, 16)) + (readlane_f64(x, 32) + readlane_f64(x, 48));
}
// inclusive sum over the lanes at or ABOVE this one (the order the stacks are popped in), without LDS: a
// Hillis-Steele scan inside every row of 16 lanes by DPP row shifts, the sums of the rows above added as scalars
template <int kCtrl>
__device__ __forceinline__ double wm_dpp_f64_or0(double x) {  // the selected lane's value, 0 where the row ends
  const uint64_t u = (uint64_t)__double_as_longlong(x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, kCtrl, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), kCtrl, 0xF, 0xF, true);
  return __longlong_as_double((long long)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo));
}
__device__ __forceinline__ double wm_scan_down(double x, int lane) {
  x = x + wm_dpp_f64_or0<0x101>(x);  // row_shl:1  (lane l reads lane l + 1 of its row)
  x = x + wm_dpp_f64_or0<0x102>(x);  // row_shl:2
  x = x + wm_dpp_f64_or0<0x104>(x);  // row_shl:4
  x = x + wm_dpp_f64_or0<0x108>(x);  // row_shl:8
  const double t1 = readlane_f64(x, 16), t2 = readlane_f64(x, 32), t3 = readlane_f64(x, 48);
  const double s2 = t2 + t3, s1 = t1 + s2;
  const int row = lane >> 4;
  return x + (row == 0 ? s1 : (row == 1 ? s2 : (row == 2 ? t3 : 0.0)));
}

// The shared positions of the edge walked last, taken in ASCENDING order by blocks of 256 slots.  The wave
// holds a window of 64 list entries in registers (lane l: entry base + l) and reloads it only when the blocks
// have consumed it -- a list load per block would put a memory round trip on the path of every block.
struct WmWindow {
  int base;  // first entry of the window (wave-uniform); entries below `lm` are consumed
  int lm;
  int pos;   // this lane's entry (0x7fffffff past the end of the list)
};
__device__ __forceinline__ void wm_window_load(WmWindow &W, const WlRow &R, int lane) {
  W.base = W.lm;
  const int k = W.base + lane;
  W.pos = k < R.nM ? wl_list_at(R, k) : 0x7fffffff;
}
__device__ __forceinline__ void wm_window_init(WmWindow &W, const WlRow &R, int lane, int lm) {
  W.lm = lm;
  W.pos = 0x7fffffff;
  W.base = lm;
  if (!R.first && R.nM > 0) wm_window_load(W, R, lane);
}
// which of the lane's 4 slots [c0 + 4 lane, + 4) of the block starting at slot c0 are shared positions: byte k
// of the result.  Entries below c0 are passed over.  The lanes that hold an entry of the block set its byte in
// the wave's LDS, every lane reads the dword of its own slots and clears it.
__device__ __forceinline__ uint32_t wm_block_flags(WmWindow &W, const WlRow &R, int c0, int lane, WmLds &L) {
  if (R.first || R.nM == 0) return 0u;
  bool any = false;
  for (;;) {
    const bool in = W.base + lane >= W.lm && W.pos < c0 + 256;
    const uint64_t bal = ballot64(in);
    if (bal) {
      if (in && W.pos >= c0) {
        reinterpret_cast<uint8_t *>(L.flags)[W.pos - c0] = 1;
        any = true;
      }
      W.lm += __popcll(bal);
    }
    if (W.lm < W.base + 64 || W.lm >= R.nM) break;
    wm_window_load(W, R, lane);  // the window is used up and the list goes on
  }
  if (!ballot64(any)) return 0u;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const uint32_t m4 = L.flags[lane];
  __builtin_amdgcn_wave_barrier();
  L.flags[lane] = 0u;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  return m4;
}
// first list entry at or beyond position `pos` (wave-uniform binary search)
__device__ __forceinline__ int wm_list_lower(const WlRow &R, int pos) {
  int lo = 0, hi = R.nM;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (wl_list_at(R, mid) < pos)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

// 4 consecutive stored weights of a row, widened (slots at or beyond n: 0)
template <typename WT>
__device__ __forceinline__ void wm_load4(const WT *w, int j0, int n, WT (&out)[4]) {  // (kept as stored: registers)
  if (j0 + 4 <= n) {
    struct __attribute__((packed, aligned(4))) Pack {
      WT v[4];
    };
    const Pack pk = *reinterpret_cast<const Pack *>(w + j0);
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = pk.v[k];
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = (j0 + k < n) ? w[j0 + k] : (WT)0;
  }
}

// x = probs - 1 of the lane's 4 slots [j0, j0 + 4) from their stored weights and the block's flags (m4: byte k
// set = slot j0 + k is a shared position).  The value only has to be within a few roundings of b / avg - 1
// (margins): with p and q powers of two the factor and 1 / avg are one product chosen per slot and x one fma.
// kWhole: every slot of the block exists (all blocks but the last).
template <typename WT, bool kPow2, bool kWhole>
__device__ __forceinline__ void wm_block_x(const WlRow &R, const WlConsts &K, const WT (&raw)[4], uint32_t m4, int j0,
                                           int c0, double inv, double (&x)[4]) {
  const bool biased = !R.first;
  const bool has_ret = biased && R.rpos < c0 + 256 && R.rpos + R.nR > c0;  // (uniform: the block holds return slots)
  if constexpr (kPow2) {
    // (three uniform cases, so that the common block pays no select it does not need: this loop is bound by
    // the number of vector instructions -- profiles/r7y_wm_pmc.txt)
    const double f_shared = inv, f_other = biased ? K.inv_q * inv : inv, f_ret = K.inv_p * inv;
    if (!has_ret && !ballot64(m4 != 0u)) {  // every slot of the block is "other" (or the row unbiased)
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = __fma_rn((double)raw[k], f_other, -1.0);
    } else if (!has_ret) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        x[k] = __fma_rn((double)raw[k], (m4 & (1u << (8 * k))) ? f_shared : f_other, -1.0);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool shared = (m4 >> (8 * k)) & 1u;
        double f = shared ? f_shared : f_other;
        if (!shared && j0 + k >= R.rpos && j0 + k < R.rpos + R.nR) f = f_ret;
        x[k] = __fma_rn((double)raw[k], f, -1.0);
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int cls = 1;
      if (biased)
        cls = ((m4 >> (8 * k)) & 1u) ? 1 : ((has_ret && j0 + k >= R.rpos && j0 + k < R.rpos + R.nR) ? 2 : 0);
      x[k] = __fma_rn(wl_bias<false>((double)raw[k], cls, K), inv, -1.0);
    }
  }
  if constexpr (!kWhole) {
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = j0 + k < R.n ? x[k] : 0.0;
  }
}

// probs - 1 (x) of the lane's 4 slots of block `blk`, looked at again after the pass over the row (0 for slots
// beyond the row); the list cursor of the block's entry was recorded by the pass
template <typename WT, bool kPow2>
__device__ __forceinline__ void wm_block_again(const WlRow &R, const WT *w, const WlConsts &K, double inv, int blk,
                                               int g, int lane, WmLds &L, double (&x)[4]) {
  WT raw[4];
  const int c0 = blk << 8, j0 = c0 + 4 * lane;
  wm_load4<WT>(w, j0, R.n, raw);
  WmWindow W;
  wm_window_init(W, R, lane, (!R.first && R.nM > 0) ? L.lm0[blk / g] : 0);
  const uint32_t m4 = wm_block_flags(W, R, c0, lane, L);
  wm_block_x<WT, kPow2, false>(R, K, raw, m4, j0, c0, inv, x);
}

// The first crossing of `target` by the running sum, from the top of the row, of the d (kDeficit) or e of
// the slots: returns its slot (or -1: the sum of the whole row stays below target), `before` / `at` = the
// running sum without / with that slot; kWmUndecided when the sums per entry and per slot disagree.
// L.cd / L.cx hold the sums from the BOTTOM of the row up to the end of every entry (g blocks of 256 slots
// each), so the sum from the top down to the start of entry i is total - c[i - 1].
template <typename WT, bool kPow2, bool kDeficit>
__device__ __forceinline__ int wm_crossing(const WlRow &R, const WT *w, const WlConsts &K, double inv, double target,
                                           int nent, int g, double tot_d, double tot_x, int lane, WmLds &L,
                                           double &before, double &at) {
  const double tot = kDeficit ? tot_d : tot_x + tot_d;
  int ent = -1;
  double base = 0.0;  // the sum over the entries above `ent`
  for (int top = nent - 1; top >= 0 && ent < 0; top -= 64) {
    const int e = top - lane;  // lane 0 holds the topmost entry
    double below = 0.0, upto = 0.0;  // sums from the bottom to the start / to the end of entry e
    if (e >= 0) {
      upto = kDeficit ? L.cd[e] : L.cx[e] + L.cd[e];
      if (e > 0) below = kDeficit ? L.cd[e - 1] : L.cx[e - 1] + L.cd[e - 1];
    }
    const uint64_t hit = ballot64(e >= 0 && tot - below >= target);
    if (hit) {
      const int l = __builtin_ctzll(hit);
      ent = top - l;
      base = tot - readlane_f64(upto, l);
    }
  }
  if (ent < 0) return -1;
  // inside the entry: its blocks from the top; in a block the lanes from the top, in a lane its 4 slots
  const int nblk = (R.n + 255) >> 8;
  for (int blk = min(nblk, (ent + 1) * g) - 1; blk >= ent * g; --blk) {
    double x[4], v[4];
    wm_block_again<WT, kPow2>(R, w, K, inv, blk, g, lane, L, x);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = kDeficit ? fmax(-x[k], 0.0) : fmax(x[k], 0.0);
    const double mine = (v[0] + v[1]) + (v[2] + v[3]);
    const double inc = wm_scan_down(mine, lane);
    const uint64_t hit = ballot64(base + inc >= target);
    if (hit) {
      const int l = 63 - __builtin_clzll(hit);
      double run = readlane_f64(base + inc - mine, l);  // everything above the lane's slots
#pragma unroll
      for (int k = 3; k >= 0; --k) {
        const double vk = readlane_f64(v[k], l);
        if (run + vk >= target) {
          before = run;
          at = run + vk;
          return (blk << 8) + 4 * l + k;
        }
        run += vk;
      }
      return kWmUndecided;  // (the lane's sum crossed, its slots one by one did not: rounding)
    }
    base = readlane_f64(base + inc, 0);
  }
  return kWmUndecided;  // (rounding between the entry sums and the slot sums: nobody crossed inside the entry)
}

// the first slot below position `top` that is overfull, every slot skipped on the way underfull -- by the
// margin delta on both sides, else kWmUndecided
template <typename WT, bool kPow2>
__device__ __forceinline__ int wm_next_over_below(const WlRow &R, const WT *w, const WlConsts &K, double inv,
                                                  double delta, int top, int g, int lane, WmLds &L) {
  for (int blk = (top - 1) >> 8; blk >= 0; --blk) {
    double x[4];
    wm_block_again<WT, kPow2>(R, w, K, inv, blk, g, lane, L, x);
    int best = -1;  // the lane's highest slot below `top` that is not decidedly underfull (bit 2: not decidedly overfull)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = (blk << 8) + 4 * lane + k;
      const bool below = j < R.n && j < top;
      const bool over = x[k] > 2.0 * delta, unsure = !over && !(x[k] < -2.0 * delta);
      if (below && (over || unsure)) best = k | (unsure ? 4 : 0);
    }
    const uint64_t any = ballot64(best >= 0);
    if (any) {
      const int l = 63 - __builtin_clzll(any);
      const int code = __builtin_amdgcn_readlane(best, l);
      if (code & 4) return kWmUndecided;
      return (blk << 8) + 4 * l + (code & 3);
    }
  }
  return kWmUndecided;  // no overfull slot left: the slot asked for would keep alias 0
}

// index sampling_from_alias(r1, r2) returns on the table of this row, or kWmUndecided.  row_sum = the sum of
// the STORED weights of the row (any order; n2v_row_weight_sums).
// The biased row sum in the REFERENCE's order (:172: left to right, one rounding per addition), by the wave: the
// lanes bias a block of 256 slots (4 each) into LDS, lane 0 adds them up in order.  ~50 us on a row of 26 786
// slots against the 1 - 5 ms of an exact replay: the second chance of a walker the general margins left undecided.
template <typename WT, bool kPow2>
__device__ __forceinline__ double wm_sequential_total(const WlRow &R, const WT *w, const WlConsts &K, int lane,
                                                      WmLds &L) {
  const int n = R.n, nblk = (n + 255) >> 8;
  double *buf = L.cd;  // 256 doubles: cd and cx lie back to back (the pass that follows rewrites them)
  static_assert(kWmEntries == 128, "wm_sequential_total uses cd + cx as one array of 256 doubles");
  WmWindow W;
  wm_window_init(W, R, lane, 0);
  double total = 0.0;
  for (int blk = 0; blk < nblk; ++blk) {
    const int c0 = blk << 8, j0 = c0 + 4 * lane;
    WT raw[4];
    wm_load4<WT>(w, j0, n, raw);
    const uint32_t m4 = wm_block_flags(W, R, c0, lane, L);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = j0 + k;
      int cls = 1;
      if (!R.first) cls = ((m4 >> (8 * k)) & 1u) ? 1 : ((j >= R.rpos && j < R.rpos + R.nR) ? 2 : 0);
      buf[4 * lane + k] = wl_bias<kPow2>((double)raw[k], cls, K);  // the reference's own operation (:219-231)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      const int m = min(256, n - c0);
      for (int t = 0; t < m; ++t) total = total + buf[t];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  return readlane_f64(total, 0);
}

// The sums of a HUB row without the pass (n2v_weighted_hubs): two thirds of the slots a step stands on belong to a
// few hundred rows on which thousands of walkers make nearly the same pass -- the same weights, the one factor of
// the "other" slots times a 1 / avg that differs from walker to walker.  For every block of 256 slots of such a row
// the graph keeps the weights SORTED and their prefix sums: the block's sum of d and of x at the walker's own
// threshold is one binary search (a lane per block: 64 blocks per round), as if every slot were "other"; the shared
// positions and the return run are then CORRECTED from the list (O(list), LDS atomics per entry).  Fills what the
// pass fills -- the running sums per entry, the list cursor of every entry, the sums below pick.
template <typename WT, bool kPow2>
__device__ __forceinline__ void wm_hub_sums(const WlRow &R, const WT *w, const WlConsts &K, double inv, const WT *hs,
                                            const double *hp, int pick, bool under, int nblk, int nent, int lane,
                                            WmLds &L, double &tot_d, double &tot_x, double &below, int pos0,
                                            double w0) {
  const int n = R.n;
  const bool biased = !R.first;
  const double cq = kPow2 ? K.inv_q : 1.0 / K.q, cp = kPow2 ? K.inv_p : 1.0 / K.p;
  const double f_other = biased ? cq * inv : inv, f_shared = inv, f_ret = cp * inv;
  for (int e = lane; e < nent; e += 64) {
    L.cd[e] = 0.0;
    L.cx[e] = 0.0;
    L.lm0[e] = 0;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  double lo_d = 0.0, lo_x = 0.0;  // per lane: over the slots below pick
  if (biased) {
    for (int k = lane; k < R.nM; k += 64) {
      // (the first 64 entries and their weights were read for the row sum and kept: wm_draw)
      const int pos = k < 64 ? pos0 : wl_list_at(R, k);
      const double wv = k < 64 ? w0 : (double)w[pos < n ? pos : 0];
      const double xo = __fma_rn(wv, f_other, -1.0), xs = __fma_rn(wv, f_shared, -1.0);
      const double dx = xs - xo, dd = fmax(-xs, 0.0) - fmax(-xo, 0.0);
      const int e = pos >> 10;
      atomicAdd(&L.cx[e], dx);
      atomicAdd(&L.cd[e], dd);
      atomicAdd(&L.lm0[e], 1);
      if (pos < pick) {
        lo_x += dx;
        lo_d += dd;
      }
    }
    for (int j = R.rpos + lane; j < R.rpos + R.nR; j += 64) {
      const double wv = (double)w[j];
      const double xo = __fma_rn(wv, f_other, -1.0), xr = __fma_rn(wv, f_ret, -1.0);
      const double dx = xr - xo, dd = fmax(-xr, 0.0) - fmax(-xo, 0.0);
      atomicAdd(&L.cx[j >> 10], dx);
      atomicAdd(&L.cd[j >> 10], dd);
      if (j < pick) {
        lo_x += dx;
        lo_d += dd;
      }
    }
  }
  const int bp = pick >> 8;
  for (int c = 0; c < nblk; c += 64) {
    const int blk = c + lane;
    if (blk < nblk) {
      const int cnt = min(256, n - (blk << 8));
      const WT *srt = hs + (int64_t)blk * 256;
      const double *pre = hp + (int64_t)blk * 257;
      int lo = 0, hi = cnt;  // the weights below the walker's threshold: x < 0 as "other"
#pragma unroll 1
      for (int it = 0; it < 9; ++it) {
        const int mid = (lo + hi) >> 1;
        const bool go = lo < hi;
        const double v = (double)srt[go ? mid : 0];
        const bool neg = __fma_rn(v, f_other, -1.0) < 0.0;
        lo = (go && neg) ? mid + 1 : lo;
        hi = (go && !neg) ? mid : hi;
      }
      const double s_below = pre[lo], s_all = pre[cnt];
      const double d_blk = (double)lo - f_other * s_below;  // sum of 1 - w f over the slots below the threshold
      const double x_blk = f_other * s_all - (double)cnt;
      atomicAdd(&L.cd[blk >> 2], d_blk);
      atomicAdd(&L.cx[blk >> 2], x_blk);
      if (blk < bp) {
        lo_d += d_blk;
        lo_x += x_blk;
      }
    }
  }
  {  // the slots of pick's block below pick, one by one (as "other": the corrections above did the rest)
    WT raw[4];
    const int j0 = (bp << 8) + 4 * lane;
    wm_load4<WT>(w, j0, n, raw);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (j0 + k < pick && j0 + k < n) {
        const double xo = __fma_rn((double)raw[k], f_other, -1.0);
        lo_x += xo;
        lo_d += fmax(-xo, 0.0);
      }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {
    double run = 0.0;
    for (int e = 0; e < nent; ++e) {
      run += L.cd[e];
      L.cd[e] = run;
    }
  } else if (lane == 1) {
    double run = 0.0;
    for (int e = 0; e < nent; ++e) {
      run += L.cx[e];
      L.cx[e] = run;
    }
  } else if (lane == 2) {
    int run = 0;
    for (int e = 0; e < nent; ++e) {
      const int t = L.lm0[e];
      L.lm0[e] = run;
      run += t;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  tot_d = readfirstlane_f64(L.cd[nent - 1]);
  tot_x = readfirstlane_f64(L.cx[nent - 1]);
  below = wm_wave_sum(under ? lo_d : lo_x + lo_d);
}

template <typename WT, bool kPow2>
__device__ __forceinline__ int wm_decide(const WlRow &R, const WT *w, const WlConsts &K, double total, bool exact_total,
                                         double kfac, double w_max, double cmax, double b_pick, int pick, double r2,
                                         int lane, WmLds &L, const WT *hub_sorted, const double *hub_prefix, int pos0, double w0);

// kSeq: the second chance of a walker the general margins left undecided -- the row sum in the reference's own
// order, hence the exact-sum margins (a launch of its own over the list of those walkers: inside the first one
// the second attempt cost every walker registers, 485 -> 335 M steps/s)
template <typename WT, bool kPow2, bool kSeq>
__device__ __forceinline__ int wm_draw(const WlRow &R, const WT *w, const WlConsts &K, double row_sum, double w_grid,
                                       double w_max, int pick, double r2, int lane, WmLds &L, const WT *hub_sorted,
                                       const double *hub_prefix) {
  const int n = R.n;
  const bool biased = !R.first;
  // ---- the row sum WITHOUT a pass over the row: every slot is "other" (w / q) but the shared positions and
  //      the return run, so sum = W / q + (1 - 1 / q) sum(shared w) + (1 / p - 1 / q) sum(return w): O(list) ----
  const double cq = kPow2 ? K.inv_q : 1.0 / K.q, cp = kPow2 ? K.inv_p : 1.0 / K.p;
  double total = row_sum, kfac = 1.0;
  bool pick_shared = false;
  int pos0 = -1;     // this lane's entry of the first 64 of the list and its weight: the corrections of a row
  double w0 = 0.0;   // with block summaries read them again (wm_hub_sums) -- from here, not from memory
  if (biased) {
    double acc_s = 0.0, acc_r = 0.0;
    for (int k = lane; k < R.nM; k += 64) {
      const int pos = wl_list_at(R, k);
      pick_shared = pick_shared || pos == pick;
      const double wv = (double)w[pos < n ? pos : 0];
      acc_s += wv;
      if (k < 64) {
        pos0 = pos;
        w0 = wv;
      }
    }
    for (int j = R.rpos + lane; j < R.rpos + R.nR; j += 64) acc_r += (double)w[j];
    const double ss = R.nM > 0 ? wm_wave_sum(acc_s) : 0.0;
    const double sr = R.nR > 1 ? wm_wave_sum(acc_r) : (R.nR == 1 ? readlane_f64(acc_r, 0) : 0.0);
    total = cq * row_sum + (1.0 - cq) * ss + (cp - cq) * sr;
    const double cmax = fmax(fmax(cp, cq), 1.0), cmin = fmin(fmin(cp, cq), 1.0);
    kfac = 1.0 + 2.0 * (cmax / cmin);  // the three terms are up to cmax / cmin times the sum each
  }
  if (!(total > 0.0) || !(total < 1.0e300) || !(row_sum > 0.0)) return kWmUndecided;
  int cls_pick = 1;
  if (biased) cls_pick = ballot64(pick_shared) ? 1 : ((pick >= R.rpos && pick < R.rpos + R.nR) ? 2 : 0);
  const double b_pick = wl_bias<kPow2>((double)w[pick], cls_pick, K);
  const double nn = (double)n;
  const double cmax = biased ? fmax(fmax(cp, cq), 1.0) : 1.0, cmin = biased ? fmin(fmin(cp, cq), 1.0) : 1.0;
  bool exact_total = (kPow2 || !biased) && w_grid > 0.0 &&
                     nn * w_max * cmax < w_grid * cmin * ldexp(1.0, 52 - (biased ? K.coef_bits : 0));
  if constexpr (kSeq) {
    if (exact_total) return kWmUndecided;  // (the exact-sum margins have spoken already: a tie, or r2 on a threshold)
    total = wm_sequential_total<WT, kPow2>(R, w, K, lane, L);
    if (!(total > 0.0) || !(total < 1.0e300)) return kWmUndecided;
    exact_total = true;
  }
  return wm_decide<WT, kPow2>(R, w, K, total, exact_total, kfac, w_max, cmax, b_pick, pick, r2, lane, L, hub_sorted,
                              hub_prefix, pos0, w0);
}

// what follows the row sum: `total` within (n + 2) 2^-53 kfac of the reference's, or (exact_total) the reference's
template <typename WT, bool kPow2>
__device__ __forceinline__ int wm_decide(const WlRow &R, const WT *w, const WlConsts &K, double total, bool exact_total,
                                         double kfac, double w_max, double cmax, double b_pick, int pick, double r2,
                                         int lane, WmLds &L, const WT *hub_sorted, const double *hub_prefix, int pos0,
                                         double w0) {
  const int n = R.n;
  const bool biased = !R.first;
  const double nn = (double)n;
  const double inv = nn / total;  // 1 / avg
  const double eps = 2.220446049250313e-16;
  // EXACT row sum: when every biased weight is a multiple of one power of two G (fp32 weights: w_grid = the
  // place of the last mantissa bit of the smallest one; factors powers of two) and n max < 2^52 G, the
  // reference's left-to-right sum rounds nowhere, and neither do the three products and two additions above
  // (coef_bits of room): `total` IS the reference's.  Then what separates b * inv from probs[i] is 4 roundings,
  // and the margin covers the loop (2 roundings per pairing, each of a value <= max(probs) + 1) and the sums
  // taken here (a lane adds n / 256 blocks, the tree 16 more, each of a value <= 4 D + 4) -- linear in n,
  // not 16 n^2: on the hubs, where n^2 2^-52 is 10^-6 and a replay costs a millisecond, that is the difference
  // between four walkers per step left to the exact kernel and none.
  const double delta = exact_total ? 8.0 * eps
                                   : kfac * (2.0 * nn + 16.0) * eps;  // relative distance of b * inv from probs[i]
  double M = kfac * 16.0 * nn * nn * eps;
  const double p_pick = b_pick * inv;
  const bool under = p_pick < 1.0 - 2.0 * delta;
#if defined(N2V_WM_ABLATE) && N2V_WM_ABLATE == 1  // timing only: the walk ends with the row sum
  return pick;
#endif
  if (!under && !(p_pick > 1.0 + 2.0 * delta)) return kWmUndecided;
  if (under) {
    if (r2 < p_pick * (1.0 - delta)) return pick;  // an underfull slot keeps its probs: accepted
    if (!(r2 > p_pick * (1.0 + delta))) return kWmUndecided;
  }
  // ---- ONE pass over the row, 4 slots per lane and block of 256: x = probs - 1 and d = max(-x, 0) summed per
  //      lane; the running sums over the wave are taken at the end of every entry (e = x + d).  The loop body
  //      is the kernel: whole blocks only (the last, partial block apart), no copy of prefetched weights, the
  //      factor of a slot one select (the return run looked for only in the block that holds it) ------------
  const int nblk = (n + 255) >> 8;
  const int g = max((nblk + kWmEntries - 1) / kWmEntries, kWmMinBlocksPerEntry);  // blocks per entry
  const int nent = (nblk + g - 1) / g;
  const int bp = pick >> 8;
  double run_d = 0.0, run_x = 0.0;    // per lane, the whole row so far
  double pre_d = 0.0, pre_x = 0.0;    // per lane: the same over the slots BELOW pick
  WmWindow W;
  wm_window_init(W, R, lane, 0);
  int in_entry = 0, ent = 0;
  double tot_d = 0.0, tot_x = 0.0, below = 0.0;
  const bool summaries = hub_sorted != nullptr && g == 4;  // (a hub row: no pass)
  if (summaries)
    wm_hub_sums<WT, kPow2>(R, w, K, inv, hub_sorted, hub_prefix, pick, under, nblk, nent, lane, L, tot_d, tot_x, below,
                           pos0, w0);
  for (int blk = summaries ? nblk : 0; blk < nblk; ++blk) {
    const int c0 = blk << 8, j0 = c0 + 4 * lane;
    if (in_entry == 0 && lane == 0) L.lm0[ent] = W.lm;  // (looked at again: wm_block_again)
    if (blk == bp) {
      pre_d = run_d;
      pre_x = run_x;
    }
    WT raw[4];
    double x[4];
    if (c0 + 256 <= n) {  // (uniform) a whole block: one 16-byte load per lane, no slot beyond the row
      struct __attribute__((packed, aligned(4))) Pack {
        WT v[4];
      };
      const Pack pk = *reinterpret_cast<const Pack *>(w + j0);
#pragma unroll
      for (int k = 0; k < 4; ++k) raw[k] = pk.v[k];
      const uint32_t m4 = wm_block_flags(W, R, c0, lane, L);
      wm_block_x<WT, kPow2, true>(R, K, raw, m4, j0, c0, inv, x);
    } else {
      wm_load4<WT>(w, j0, n, raw);
      const uint32_t m4 = wm_block_flags(W, R, c0, lane, L);
      wm_block_x<WT, kPow2, false>(R, K, raw, m4, j0, c0, inv, x);
    }
    if (blk == bp) {  // (uniform) the slots of pick's block below pick
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (j0 + k < pick) {
          pre_x += x[k];
          pre_d += fmax(-x[k], 0.0);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      run_x += x[k];
      run_d += fmax(-x[k], 0.0);
    }
    if (++in_entry == g || blk + 1 == nblk) {
      tot_d = wm_wave_sum(run_d);
      tot_x = wm_wave_sum(run_x);
      if (lane == 0) {
        L.cd[ent] = tot_d;
        L.cx[ent] = tot_x;
      }
      in_entry = 0;
      ++ent;
    }
  }
  // over the slots below pick: the d for an underfull pick, the e = x + d for an overfull one (one sum over the wave)
  if (!summaries) below = wm_wave_sum(under ? pre_d : pre_x + pre_d);
  if (exact_total)  // (+ 64: a block summary is the difference of two sums of up to 256 weights)
    M = 8.0 * eps * (nn * (w_max * cmax * inv + 12.0 + 64.0) + (nn * (1.0 / 256.0) + 16.0) * (4.0 * tot_d + 4.0));
#if defined(N2V_WM_ABLATE) && N2V_WM_ABLATE == 2  // timing only: the walk ends with the pass over the row
  return below > 1.0e300 ? 0 : pick;
#endif
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const double x_pick = p_pick - 1.0;
  double before = 0.0, at = 0.0;
  if (under) {
    // rejected: alias = the overfull slot on top when `pick` is popped -- the first k with E_k >= D_{j-1},
    // D_{j-1} = the d of the slots ABOVE pick
    const double d_above = tot_d - below - (-x_pick);
    if (!(d_above > M)) {
      // (next to) nothing underfull above pick: the topmost overfull slot is still `over` when pick is popped
      // if its excess outlasts whatever was popped before -- at most d_above + M <= 2 M
      const int t = wm_next_over_below<WT, kPow2>(R, w, K, inv, delta, n, g, lane, L);
      if (t < 0) return kWmUndecided;
      int cls_t = 1;
      if (biased) {
        const int lo = wm_list_lower(R, t);
        const bool shared = lo < R.nM && wl_list_at(R, lo) == t;
        cls_t = shared ? 1 : ((t >= R.rpos && t < R.rpos + R.nR) ? 2 : 0);
      }
      const double p_t = wl_bias<kPow2>((double)w[t], cls_t, K) * inv;
      return p_t - 1.0 >= 3.0 * M ? t : kWmUndecided;
    }
    const int k = wm_crossing<WT, kPow2, false>(R, w, K, inv, d_above, nent, g, tot_d, tot_x, lane, L, before, at);
    if (k < 0 || !(before <= d_above - M) || !(at >= d_above + M)) return kWmUndecided;
    return k;
  }
  // overfull: demoted by the first underfull slot j with D_j > E_k (E_k = the e of the slots at or above
  // pick), then probs = 1 + E_k - D_j
  const double e_from = (tot_x + tot_d) - below;
  if (tot_d <= e_from - M) return pick;  // never demoted
  if (tot_d <= e_from + M)               // demoted, if at all, with probs >= 1 - 2 M
    return r2 < 1.0 - 3.0 * M ? pick : kWmUndecided;
  const int j = wm_crossing<WT, kPow2, true>(R, w, K, inv, e_from, nent, g, tot_d, tot_x, lane, L, before, at);
  if (j < 0 || !(before <= e_from - M) || !(at >= e_from + M)) return kWmUndecided;
  const double resid = 1.0 + e_from - at;
  if (r2 < resid - M) return pick;
  if (!(r2 > resid + M)) return kWmUndecided;
  return wm_next_over_below<WT, kPow2>(R, w, K, inv, delta, pick, g, lane, L);  // alias = the next `over`
}


// ---- rows below the wave kernel's: the SAME decision, one LANE per walker -------------------------------------
// The decision of wm_draw is three plain loops over the row -- the sums of x = probs - 1 and of d = max(-x, 0), the
// descending scan to the crossing, the look for the next overfull slot -- with no state machine: 64 walkers of a
// wave run them side by side and diverge in nothing but the trip counts (the walkers are sorted by row length).  A
// wave per walker pays ~700 vector instructions per walker before the first slot; a lane pays the row: ~0.6 n.
// (scripts/models/weighted_margins.py is this function, line by line.)
template <typename WT, bool kPow2>
__device__ __forceinline__ void lm_group_x(const WlRow &R, const WlConsts &K, const WT *w, int c0, uint32_t mm,
                                           double inv, double (&x)[4]) {
  const WlRaw<WT, 4> raw = wl_load_raw<WT, 4>(w, c0, R.n);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = c0 + k;
    int cls = 1;
    if (!R.first) cls = ((mm >> k) & 1u) ? 1 : ((j >= R.rpos && j < R.rpos + R.nR) ? 2 : 0);
    const double b = wl_bias<kPow2>((double)raw.v[k], cls, K);
    x[k] = j < R.n ? __fma_rn(b, inv, -1.0) : 0.0;
  }
}

// the first slot below position `top` that is overfull, every slot skipped on the way underfull (by delta)
template <typename WT, bool kPow2>
__device__ __forceinline__ int lm_next_over_below(const WlRow &R, const WlConsts &K, const WT *w, double inv,
                                                  double delta, int top, double &x_found) {
  if (top <= 0) return kWmUndecided;
  WlList<false> bwd;
  int at = R.nM - 1;
  if (!R.first && R.nM > 0 && top < R.n) {  // the last entry below `top`
    int lo = 0, hi = R.nM;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (wl_list_at(R, mid) < top)
        lo = mid + 1;
      else
        hi = mid;
    }
    at = lo - 1;
  }
  bwd.init_at(R, at);
  for (int c0 = (top - 1) & ~3; c0 >= 0; c0 -= 4) {
    uint32_t mm = 0u;
    while (bwd.cur >= c0) {
      mm |= 1u << (bwd.cur - c0);
      bwd.advance();
    }
    double x[4];
    lm_group_x<WT, kPow2>(R, K, w, c0, mm & 0xfu, inv, x);
#pragma unroll
    for (int k = 3; k >= 0; --k) {
      const int j = c0 + k;
      if (j >= top || j >= R.n) continue;
      if (x[k] > 2.0 * delta) {
        x_found = x[k];
        return j;
      }
      if (!(x[k] < -2.0 * delta)) return kWmUndecided;
    }
  }
  return kWmUndecided;  // no overfull slot left: the slot asked for would keep alias 0
}

template <typename WT, bool kPow2>
__device__ __forceinline__ int lm_draw(const WlRow &R, const WT *w, const WlConsts &K, double row_sum, double w_grid,
                                       double w_max, int pick, double r2) {
  const int n = R.n;
  const bool biased = !R.first;
  // a row of one slot: probs = [w / w] = [1.0], nothing is paired, r2 < 1.0 returns the slot (a weight of 0 or a
  // NaN is the exact kernels' business)
  if (n == 1) return (double)w[0] > 0.0 && (double)w[0] < 1.0e300 ? 0 : kWmUndecided;
  const double cq = kPow2 ? K.inv_q : 1.0 / K.q, cp = kPow2 ? K.inv_p : 1.0 / K.p;
  double total = row_sum, kfac = 1.0;
  bool pick_shared = false;
  if (biased) {
    double ss = 0.0, sr = 0.0;
    // (four entries at a time: the four positions, then the four weights behind them, are independent loads --
    // one by one every entry was two dependent round trips of a lane that has nothing else to do)
    for (int k = 0; k < R.nM; k += 4) {
      int pos[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) pos[t] = k + t < R.nM ? wl_list_at(R, k + t) : -1;
      double wt[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) wt[t] = (pos[t] >= 0 && pos[t] < n) ? (double)w[pos[t]] : 0.0;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        pick_shared = pick_shared || pos[t] == pick;
        ss += wt[t];
      }
    }
    for (int j = R.rpos; j < R.rpos + R.nR; ++j) sr += (double)w[j];
    total = cq * row_sum + (1.0 - cq) * ss + (cp - cq) * sr;
    const double cmax0 = fmax(fmax(cp, cq), 1.0), cmin0 = fmin(fmin(cp, cq), 1.0);
    kfac = 1.0 + 2.0 * (cmax0 / cmin0);
  }
  if (!(total > 0.0) || !(total < 1.0e300) || !(row_sum > 0.0)) return kWmUndecided;
  int cls_pick = 1;
  if (biased) cls_pick = pick_shared ? 1 : ((pick >= R.rpos && pick < R.rpos + R.nR) ? 2 : 0);
  const double b_pick = wl_bias<kPow2>((double)w[pick], cls_pick, K);
  const double nn = (double)n, inv = nn / total, eps = 2.220446049250313e-16;
  const double cmax = biased ? fmax(fmax(cp, cq), 1.0) : 1.0, cmin = biased ? fmin(fmin(cp, cq), 1.0) : 1.0;
  const bool exact_total = (kPow2 || !biased) && w_grid > 0.0 &&
                           nn * w_max * cmax < w_grid * cmin * ldexp(1.0, 52 - (biased ? K.coef_bits : 0));
  const double delta = exact_total ? 8.0 * eps : kfac * (2.0 * nn + 16.0) * eps;
  double M = kfac * 16.0 * nn * nn * eps;
  const double p_pick = b_pick * inv;
  const bool under = p_pick < 1.0 - 2.0 * delta;
  if (!under && !(p_pick > 1.0 + 2.0 * delta)) return kWmUndecided;
  if (under) {
    if (r2 < p_pick * (1.0 - delta)) return pick;
    if (!(r2 > p_pick * (1.0 + delta))) return kWmUndecided;
  }
  // the sums over the row, and over the slots below pick
  double tot_d = 0.0, tot_x = 0.0, pre_d = 0.0, pre_x = 0.0;
  {
    WlList<true> fwd;
    fwd.init(R);
    for (int c0 = 0; c0 < n; c0 += 4) {
      uint32_t mm = 0u;
      while (fwd.cur < c0 + 4) {
        mm |= 1u << (fwd.cur - c0);
        fwd.advance();
      }
      double x[4];
      lm_group_x<WT, kPow2>(R, K, w, c0, mm, inv, x);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double d = fmax(-x[k], 0.0);
        tot_x += x[k];
        tot_d += d;
        const bool below = c0 + k < pick;
        pre_x += below ? x[k] : 0.0;
        pre_d += below ? d : 0.0;
      }
    }
  }
  if (exact_total)  // (a lane adds the n slots one by one: n roundings of a value <= 4 D + 4, not n / 256 + 16)
    M = 8.0 * eps * (nn * (w_max * cmax * inv + 12.0) + (nn + 16.0) * (4.0 * tot_d + 4.0));
  const double x_pick = p_pick - 1.0;
  double target;
  if (under) {
    target = tot_d - pre_d - (-x_pick);  // the d of the slots above pick
    if (!(target > M)) {
      double x_t = 0.0;
      const int t = lm_next_over_below<WT, kPow2>(R, K, w, inv, delta, n, x_t);
      if (t < 0) return kWmUndecided;
      return x_t >= 3.0 * M ? t : kWmUndecided;
    }
  } else {
    target = (tot_x + tot_d) - (pre_x + pre_d);  // the e of the slots at or above pick
    if (tot_d <= target - M) return pick;  // never demoted
    if (tot_d <= target + M) return r2 < 1.0 - 3.0 * M ? pick : kWmUndecided;
  }
  // the crossing: from the top, the running sum of e (underfull pick) / of d (overfull pick) up to `target`
  double run = 0.0, before = 0.0, at = 0.0;
  int found = -1;
  {
    WlList<false> bwd;
    bwd.init(R);
    for (int c0 = (n - 1) & ~3; c0 >= 0 && found < 0; c0 -= 4) {
      uint32_t mm = 0u;
      while (bwd.cur >= c0) {
        mm |= 1u << (bwd.cur - c0);
        bwd.advance();
      }
      double x[4];
      lm_group_x<WT, kPow2>(R, K, w, c0, mm & 0xfu, inv, x);
#pragma unroll
      for (int k = 3; k >= 0; --k) {
        const double v = under ? fmax(x[k], 0.0) : fmax(-x[k], 0.0);
        if (found < 0 && c0 + k < n) {
          if (run + v >= target) {
            before = run;
            at = run + v;
            found = c0 + k;
          } else {
            run += v;
          }
        }
      }
    }
  }
  if (found < 0 || !(before <= target - M) || !(at >= target + M)) return kWmUndecided;
  if (under) return found;
  const double resid = 1.0 + target - at;
  if (r2 < resid - M) return pick;
  if (!(r2 > resid + M)) return kWmUndecided;
  double x_t = 0.0;
  return lm_next_over_below<WT, kPow2>(R, K, w, inv, delta, pick, x_t);
}

#ifndef N2V_LM_WAVES_PER_SIMD
#define N2V_LM_WAVES_PER_SIMD 6
#endif
template <typename WT, bool kPow2>
__global__ __launch_bounds__(256, N2V_LM_WAVES_PER_SIMD) void walk_weighted_lane_margin_kernel(
    n2v_graph g, const WT *__restrict__ w, const int32_t *__restrict__ start_ids, int32_t num_walks,
    const int64_t *__restrict__ order, int64_t n_rows, int32_t max_n, int32_t step, int32_t walk_length,
    WlConsts K, uint64_t seed, int64_t *__restrict__ edge_state, int32_t *__restrict__ walks,
    uint8_t *__restrict__ valid, uint32_t *__restrict__ status, int64_t *__restrict__ undecided,
    const double *__restrict__ row_sums) {
  const int L1 = walk_length + 1;
  const bool biased = !(K.p == 1.0 && K.q == 1.0);
  const bool tables = biased && g.edge_classes && g.wedge_off;
  const double w_grid = row_sums[g.n_vertices], w_max = row_sums[g.n_vertices + 1];
  const bool small = n_rows <= 0x7fffffffll;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_rows; i += (int64_t)gridDim.x * 256) {
    const int64_t r = order[i];
    if (r < 0 || r >= n_rows) {
      atomicOr(status, N2V_ST_RANGE);
      continue;
    }
    int32_t *row = walks + r * (int64_t)L1;
    const int32_t v = row[step];
    if (v < 0 || !valid[r]) continue;  // a walker that has vanished (or never started)
    if ((int64_t)v >= g.n_vertices) {
      atomicOr(status, N2V_ST_RANGE);
      continue;
    }
    const int32_t s = step > 0 ? row[step - 1] : -1;
    const int64_t vb = g.rowptr[v];
    WlRow R;
    R.n = (int)(g.rowptr[v + 1] - vb);
    if (R.n > max_n || R.n <= 0) continue;  // (the wave kernel's rows)
    R.first = s < 0 || !biased;
    R.nR = R.nM = R.rpos = 0;
    R.list = nullptr;
    R.wide = false;
    bool ok = true;
    if (!R.first) {
      const int64_t e_prev = edge_state[r];
      if (!tables || e_prev < 0 || e_prev >= g.n_edges) {
        ok = false;
      } else {
        const uint32_t ec = g.edge_classes[e_prev];
        const uint64_t wraw = g.wedge_off[e_prev];
        const uint32_t fR = ec >> N2V_EC_RETURN_SHIFT, fM = ec & N2V_EC_SHARED_MASK;
        R.nR = (int)fR;
        R.nM = (int)fM;
        R.rpos = (int)(wraw >> N2V_WEDGE_RPOS_SHIFT);
        R.wide = wedge_row_wide(g.wedge_wide, R.n);
        const uint64_t off = wraw & N2V_WEDGE_OFF_MASK;
        R.list = R.wide ? (const void *)(reinterpret_cast<const uint32_t *>(g.wedge_pos) + off)
                        : (const void *)(reinterpret_cast<const uint16_t *>(g.wedge_pos) + off);
        ok = fR != N2V_EC_RETURN_SAT && fM != N2V_EC_SHARED_MASK && (int64_t)fR + (int64_t)fM <= R.n &&
             R.rpos + (int)fR <= R.n;
      }
    }
    if (!ok) {
      atomicOr(status, N2V_ST_RANGE);
      continue;
    }
    const int64_t r_start = small ? (int64_t)((uint32_t)r / (uint32_t)num_walks) : r / num_walks;
    const int64_t r_ord = small ? (int64_t)((uint32_t)r % (uint32_t)num_walks) : r % num_walks;
    const uint64_t key = (uint64_t)start_ids[r_start] * (uint64_t)num_walks + (uint64_t)r_ord;
    const uint64_t bits = step_bits(walker_stream(seed, key), (uint32_t)step);
    const uint32_t u1 = (uint32_t)(bits >> 32), u2 = (uint32_t)bits;
    const int pick = pick_index(u1, R.n);
    const double r2 = (double)u2 * (1.0 / 4294967296.0);
    const int idx = lm_draw<WT, kPow2>(R, w + vb, K, row_sums[v], w_grid, w_max, pick, r2);
    if (idx < 0) {  // not decided by the margins: the second chance, then the exact wave kernel
      const unsigned long long at = atomicAdd(reinterpret_cast<unsigned long long *>(undecided), 1ull);
      undecided[1 + at] = r;
      continue;
    }
    const int64_t e = vb + idx;
    const int32_t x = g.col[e];
    row[step + 1] = x;
    edge_state[r] = e;
    if (step + 1 < walk_length &&
        (x < 0 || (int64_t)x >= g.n_vertices || g.rowptr[x + 1] == g.rowptr[x]))
      valid[r] = 0;  // fugue.py:147
  }
}

constexpr int kWmWaves = 4;
#ifndef N2V_WM_WAVES_PER_SIMD
#define N2V_WM_WAVES_PER_SIMD 5
#endif

template <typename WT, bool kPow2, bool kSeq>
__global__ __launch_bounds__(kWmWaves * 64, N2V_WM_WAVES_PER_SIMD) void walk_weighted_margin_kernel(
    n2v_graph g, const WT *__restrict__ w, const int32_t *__restrict__ start_ids, int32_t num_walks,
    const int64_t *__restrict__ order, int64_t n_rows, int32_t min_n, int32_t step, int32_t walk_length,
    WlConsts K, uint64_t seed, int64_t *__restrict__ edge_state, int32_t *__restrict__ walks,
    uint8_t *__restrict__ valid, uint32_t *__restrict__ status, int64_t *__restrict__ undecided,
    const double *__restrict__ row_sums, n2v_weighted_hubs hubs) {
  __shared__ WmLds lds_all[kWmWaves];
  const int lane = threadIdx.x & 63;
  WmLds &L = lds_all[threadIdx.x >> 6];
  L.flags[lane] = 0u;
  // (behind the sums of the rows: the grid of the stored weights -- 0: unknown -- and their maximum)
  const double w_grid = readfirstlane_f64(row_sums[g.n_vertices]);
  const double w_max = readfirstlane_f64(row_sums[g.n_vertices + 1]);
  const int L1 = walk_length + 1;
  const bool biased = !(K.p == 1.0 && K.q == 1.0);
  // Walkers are dealt to the waves round robin (the order is by row length, descending: every wave gets the
  // same mix, and stops at the first row that is the lane kernel's).  A shared counter -- one atomic per
  // walker on ONE address -- took 12 ns per walker, 32 of the 36 ms of a step (profiles/r7z_wm_ablation.log).
  const int64_t n_waves = (int64_t)gridDim.x * kWmWaves;
  // What a wave needs to know of a walker sits behind three dependent round trips (order -> the walker's
  // state -> the row and the tables of the edge walked last); everything of one level is requested at once,
  // on indices clamped into range, BEFORE anything of it is looked at -- read one by one between the range
  // checks it was eight round trips per walker, the larger part of a step (profiles/r7z_wm_ablation.log: "sum").
  const bool tables = biased && g.edge_classes && g.wedge_off;
  int64_t i = (int64_t)blockIdx.x * kWmWaves + (threadIdx.x >> 6);
  int64_t r_raw = i < n_rows ? order[i] : -1;
  for (; i < n_rows; i += n_waves) {
    const int64_t r = readfirstlane_i64(r_raw);
    if (i + n_waves < n_rows) r_raw = order[i + n_waves];  // (the next walker of this wave: one level ahead)
    if (r < 0 || r >= n_rows) break;  // (the lane kernel flags it)
    int32_t *row = walks + r * (int64_t)L1;
    // level 2: the walker
    const int32_t v_ld = row[step], s_ld = step > 0 ? row[step - 1] : -1;
    const uint8_t valid_ld = valid[r];
    const int64_t e_ld = edge_state[r];
    // (a 64-bit division is ~150 instructions and this kernel is bound by their number: 32 bits when they do)
    const bool small = n_rows <= 0x7fffffffll;
    const int64_t r_start = small ? (int64_t)((uint32_t)r / (uint32_t)num_walks) : r / num_walks;
    const int64_t r_ord = small ? (int64_t)((uint32_t)r % (uint32_t)num_walks) : r % num_walks;
    const int32_t sid_ld = start_ids[r_start];
    const int32_t v = __builtin_amdgcn_readfirstlane(v_ld);
    if (v < 0 || (int64_t)v >= g.n_vertices || !__builtin_amdgcn_readfirstlane((int)valid_ld))
      break;  // vanished walkers come last in the order
    const int32_t s = __builtin_amdgcn_readfirstlane(s_ld);
    const int64_t e_prev = readfirstlane_i64(e_ld);
    // level 3: the row, and the tables of the edge walked last
    const bool first = s < 0 || !biased;
    const int64_t e_at = (!first && tables && e_prev >= 0 && e_prev < g.n_edges) ? e_prev : 0;
    const int64_t vb_ld = g.rowptr[v], ve_ld = g.rowptr[v + 1];
    const double rs_ld = row_sums[v];
    const int32_t hub_ld = hubs.block0 ? hubs.block0[v] : -1;
    const uint32_t ec_ld = tables ? g.edge_classes[e_at] : 0u;
    const uint64_t wraw_ld = tables ? g.wedge_off[e_at] : 0ull;
    const int64_t vb = readfirstlane_i64(vb_ld);
    WlRow R;
    R.n = (int)(readfirstlane_i64(ve_ld) - vb);
    if (!kSeq && R.n <= min_n) break;  // this row and every later one: the lane kernel's
    const double row_sum = readfirstlane_f64(rs_ld);
    R.first = first;
    R.nR = R.nM = R.rpos = 0;
    R.list = nullptr;
    R.wide = false;
    bool ok = true;
    if (!R.first) {
      if (!tables || e_prev < 0 || e_prev >= g.n_edges) {
        ok = false;
      } else {
        const uint32_t ec = (uint32_t)__builtin_amdgcn_readfirstlane((int)ec_ld);
        const uint64_t wraw = readfirstlane_u64(wraw_ld);
        const uint32_t fR = ec >> N2V_EC_RETURN_SHIFT, fM = ec & N2V_EC_SHARED_MASK;
        R.nR = (int)fR;
        R.nM = (int)fM;
        R.rpos = (int)(wraw >> N2V_WEDGE_RPOS_SHIFT);
        R.wide = wedge_row_wide(g.wedge_wide, R.n);
        const uint64_t off = wraw & N2V_WEDGE_OFF_MASK;
        R.list = R.wide ? (const void *)(reinterpret_cast<const uint32_t *>(g.wedge_pos) + off)
                        : (const void *)(reinterpret_cast<const uint16_t *>(g.wedge_pos) + off);
        ok = fR != N2V_EC_RETURN_SAT && fM != N2V_EC_SHARED_MASK && (int64_t)fR + (int64_t)fM <= R.n &&
             R.rpos + (int)fR <= R.n;
      }
    }
    if (!ok) {
      if (lane == 0) atomicOr(status, N2V_ST_RANGE);
      continue;
    }
    const uint64_t key = (uint64_t)__builtin_amdgcn_readfirstlane(sid_ld) * (uint64_t)num_walks + (uint64_t)r_ord;
    const uint64_t bits = step_bits(walker_stream(seed, key), (uint32_t)step);
    const uint32_t u1 = (uint32_t)(bits >> 32), u2 = (uint32_t)bits;
    const int pick = pick_index(u1, R.n);
    const double r2 = (double)u2 * (1.0 / 4294967296.0);
    const int64_t hub_b0 = (int64_t)__builtin_amdgcn_readfirstlane(hub_ld);
    const WT *hub_sorted = hub_b0 >= 0 ? reinterpret_cast<const WT *>(hubs.sorted) + hub_b0 * 256 : nullptr;
    const double *hub_prefix = hub_b0 >= 0 ? hubs.prefix + hub_b0 * 257 : nullptr;
    const int idx = wm_draw<WT, kPow2, kSeq>(R, w + vb, K, row_sum, w_grid, w_max, pick, r2, lane, L, hub_sorted,
                                             hub_prefix);
    if (lane == 0) {
      if (idx < 0) {  // not decided by the margins: the exact wave kernel steps this walker
        const unsigned long long at = atomicAdd(reinterpret_cast<unsigned long long *>(undecided), 1ull);
        undecided[1 + at] = r;
      } else {
        const int64_t e = vb + idx;
        const int32_t x = g.col[e];
        row[step + 1] = x;
        edge_state[r] = e;
        if (step + 1 < walk_length &&
            (x < 0 || (int64_t)x >= g.n_vertices || g.rowptr[x + 1] == g.rowptr[x]))
          valid[r] = 0;  // fugue.py:147
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace n2v

extern "C" int n2v_weighted_step_wave_launch(const n2v_graph *g, const int32_t *start_ids, int32_t num_walks,
                                             const int64_t *order, int64_t n_rows, int32_t min_n,
                                             int32_t step, int32_t walk_length, double p, double q,
                                             uint64_t seed, int64_t *edge_state, int32_t *walks,
                                             uint8_t *valid, uint32_t *status, void *stream);

// rows of up to this many slots: the 8-slot instance; longer ones: the 32-slot instance.  Measured on
// cfg 2 (profiles/r7h_time_wlanes_two_instances.log, r7k_time_wlanes.log): the 32-slot instance (255 VGPRs,
// one wave per SIMD) LOSES -- 15 - 25 M steps/s with the cut at 1 024 / 4 096 slots against 51 - 55 M with
// every row in the 8-slot instance -- and so does a whole wave per walker on the long rows (37 - 47 M), so the
// default sends every row to the 8-slot instance; the other two stay as build-time variants.
#ifndef N2V_WLANES_SHORT
#define N2V_WLANES_SHORT 0x7fffffff
#endif
#ifndef N2V_WLANES_WAVE_FROM
#define N2V_WLANES_WAVE_FROM 0x7fffffff
#endif

namespace n2v {
template <typename WT, int CH, int TH>
static int wl_launch(const n2v_graph *g, const WT *w, const int32_t *start_ids, int32_t num_walks,
                     const int64_t *order, int64_t n_rows, int min_n, int max_n, int32_t step,
                     int32_t walk_length, const WlConsts &K, uint64_t seed, int64_t *edge_state,
                     int32_t *walks, uint8_t *valid, uint32_t *status, hipStream_t st) {
  int64_t blocks = (n_rows + TH - 1) / TH;
  const bool pow2 = K.p_pow2 && K.q_pow2;
  const void *fn = pow2 ? (const void *)walk_weighted_step_kernel<WT, CH, TH, true>
                        : (const void *)walk_weighted_step_kernel<WT, CH, TH, false>;
  const int64_t cap = resident_blocks(fn, TH, 0);
  if (blocks > cap) blocks = cap;
  if (pow2)
    hipLaunchKernelGGL((walk_weighted_step_kernel<WT, CH, TH, true>), dim3((unsigned)blocks), dim3(TH), 0, st, *g,
                       w, start_ids, num_walks, order, n_rows, min_n, max_n, step, walk_length, K, seed,
                       edge_state, walks, valid, status);
  else
    hipLaunchKernelGGL((walk_weighted_step_kernel<WT, CH, TH, false>), dim3((unsigned)blocks), dim3(TH), 0, st, *g,
                       w, start_ids, num_walks, order, n_rows, min_n, max_n, step, walk_length, K, seed,
                       edge_state, walks, valid, status);
  return hipGetLastError() == hipSuccess ? N2V_OK : N2V_ELAUNCH;
}
}  // namespace n2v

// rows of at least this many slots: a wave per walker that decides the pairing with margins
// (walk_weighted_margin_kernel), given an order and a scratch list for the walkers it leaves undecided
#ifndef N2V_WLANES_MARGIN_FROM
#define N2V_WLANES_MARGIN_FROM 768
#endif
// small batches: the cut is n_rows / this, at least N2V_WLANES_MARGIN_FROM_MIN
#ifndef N2V_WLANES_WALKERS_PER_CUT_SLOT
#define N2V_WLANES_WALKERS_PER_CUT_SLOT 2048
#endif
#ifndef N2V_WLANES_MARGIN_FROM_MIN
#define N2V_WLANES_MARGIN_FROM_MIN 48
#endif
// the rows below that: 1 = the same decision with a lane per walker (walk_weighted_lane_margin_kernel), 0 = the
// exact lane kernel (the pairing replayed: round 5's first form)
#ifndef N2V_WLANES_LANE_MARGINS
#define N2V_WLANES_LANE_MARGINS 1
#endif

namespace n2v {
template <typename WT, bool kSeq>
static int wm_launch(const n2v_graph *g, const WT *w, const int32_t *start_ids, int32_t num_walks,
                     const int64_t *order, int64_t n_rows, int min_n, int32_t step, int32_t walk_length,
                     const WlConsts &K, uint64_t seed, int64_t *edge_state, int32_t *walks, uint8_t *valid,
                     uint32_t *status, int64_t *undecided, const double *row_sums, const n2v_weighted_hubs &hubs,
                     hipStream_t st) {
  const bool pow2 = K.p_pow2 && K.q_pow2;
  const void *fn = pow2 ? (const void *)walk_weighted_margin_kernel<WT, true, kSeq>
                        : (const void *)walk_weighted_margin_kernel<WT, false, kSeq>;
  int64_t blocks = (n_rows + kWmWaves - 1) / kWmWaves;
  int64_t cap = resident_blocks(fn, kWmWaves * 64, 0);
  if (kSeq && cap > 1024) cap = 1024;  // (a list of a few walkers as a rule)
  if (blocks > cap) blocks = cap;
  if (pow2)
    hipLaunchKernelGGL((walk_weighted_margin_kernel<WT, true, kSeq>), dim3((unsigned)blocks), dim3(kWmWaves * 64), 0,
                       st, *g, w, start_ids, num_walks, order, n_rows, min_n, step, walk_length, K, seed, edge_state,
                       walks, valid, status, undecided, row_sums, hubs);
  else
    hipLaunchKernelGGL((walk_weighted_margin_kernel<WT, false, kSeq>), dim3((unsigned)blocks), dim3(kWmWaves * 64), 0,
                       st, *g, w, start_ids, num_walks, order, n_rows, min_n, step, walk_length, K, seed, edge_state,
                       walks, valid, status, undecided, row_sums, hubs);
  return hipGetLastError() == hipSuccess ? N2V_OK : N2V_ELAUNCH;
}
template <typename WT>
static int lm_launch(const n2v_graph *g, const WT *w, const int32_t *start_ids, int32_t num_walks,
                     const int64_t *order, int64_t n_rows, int max_n, int32_t step, int32_t walk_length,
                     const WlConsts &K, uint64_t seed, int64_t *edge_state, int32_t *walks, uint8_t *valid,
                     uint32_t *status, int64_t *undecided, const double *row_sums, hipStream_t st) {
  const bool pow2 = K.p_pow2 && K.q_pow2;
  const void *fn = pow2 ? (const void *)walk_weighted_lane_margin_kernel<WT, true>
                        : (const void *)walk_weighted_lane_margin_kernel<WT, false>;
  int64_t blocks = (n_rows + 255) / 256;
  const int64_t cap = resident_blocks(fn, 256, 0);
  if (blocks > cap) blocks = cap;
  if (pow2)
    hipLaunchKernelGGL((walk_weighted_lane_margin_kernel<WT, true>), dim3((unsigned)blocks), dim3(256), 0, st, *g, w,
                       start_ids, num_walks, order, n_rows, max_n, step, walk_length, K, seed, edge_state, walks, valid,
                       status, undecided, row_sums);
  else
    hipLaunchKernelGGL((walk_weighted_lane_margin_kernel<WT, false>), dim3((unsigned)blocks), dim3(256), 0, st, *g, w,
                       start_ids, num_walks, order, n_rows, max_n, step, walk_length, K, seed, edge_state, walks, valid,
                       status, undecided, row_sums);
  return hipGetLastError() == hipSuccess ? N2V_OK : N2V_ELAUNCH;
}
}  // namespace n2v

namespace n2v {
__global__ __launch_bounds__(256) void weighted_keys_kernel(const int32_t *__restrict__ walks,
                                                            const uint8_t *__restrict__ valid,
                                                            const int32_t *__restrict__ rank_of, int64_t n_vertices,
                                                            int64_t n_rows, int32_t step, int32_t L1,
                                                            int32_t *__restrict__ keys) {
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * 256) {
    const int32_t v = walks[r * (int64_t)L1 + step];
    keys[r] = (valid[r] && v >= 0 && (int64_t)v < n_vertices) ? rank_of[v] : 0x7fffffff;
  }
}
}  // namespace n2v

extern "C" int n2v_walk_weighted_keys(const int32_t *walks, const uint8_t *valid, const int32_t *rank_of,
                                      int64_t n_vertices, int64_t n_rows, int32_t step, int32_t walk_length,
                                      int32_t *keys, void *stream) {
  if (n_rows < 0 || step < 0 || step > walk_length || n_vertices < 0) return N2V_EINVAL;
  if (n_rows == 0) return N2V_OK;
  if (!walks || !valid || !rank_of || !keys) return N2V_EINVAL;
  int64_t blocks = (n_rows + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(n2v::weighted_keys_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, walks, valid,
                     rank_of, n_vertices, n_rows, step, walk_length + 1, keys);
  return hipGetLastError() == hipSuccess ? N2V_OK : N2V_ELAUNCH;
}

extern "C" int n2v_walk_weighted_step(const n2v_graph *g, const int32_t *start_ids, int32_t num_walks,
                                      const int64_t *order, int64_t n_rows, int32_t step,
                                      int32_t walk_length, double return_param, double inout_param,
                                      uint64_t seed, int64_t *edge_state, int32_t *walks, uint8_t *valid,
                                      uint32_t *status, int64_t *scratch, const double *row_sums,
                                      const n2v_weighted_hubs *hubs, void *stream) {
  if (!g || !g->rowptr || !g->col || n_rows < 0 || num_walks < 1 || walk_length < 0) return N2V_EINVAL;
  if (step < 0 || step >= walk_length) return N2V_EINVAL;
  if (return_param == 0.0 || inout_param == 0.0) return N2V_EINVAL;  // randomwalk.py:214-217
  if ((!g->w && !g->w64) || (g->w && g->w64)) return N2V_EINVAL;      // weighted graphs, one storage form
  if (n_rows == 0) return N2V_OK;
  if (!start_ids || !edge_state || !walks || !valid || !status) return N2V_EINVAL;
  const bool biased = !(return_param == 1.0 && inout_param == 1.0);
  if (biased && step > 0 && (!g->edge_classes || !g->wedge_off || !g->wedge_pos)) return N2V_EINVAL;
  if (g->wedge_wide < 0 || g->wedge_wide > 65536) return N2V_EINVAL;
  n2v::WlConsts K;
  K.p = return_param;
  K.q = inout_param;
  K.inv_p = 1.0 / return_param;
  K.inv_q = 1.0 / inout_param;
  int ex = 0;
  K.p_pow2 = frexp(return_param, &ex) == 0.5 && ex > -500 && ex < 500;
  K.q_pow2 = frexp(inout_param, &ex) == 0.5 && ex > -500 && ex < 500;
  K.coef_bits = 0;
  if (K.p_pow2 && K.q_pow2) {  // significant bits of the coefficients 1 - 1/q and 1/p - 1/q of wm_draw's row sum
    const double co[2] = {fabs(1.0 - K.inv_q), fabs(K.inv_p - K.inv_q)};
    for (double c : co) {
      if (c == 0.0) continue;
      double m = frexp(c, &ex);
      int bits = 0;
      while (m != 0.0 && bits < 60) {  // mantissa bits until nothing is left
        m *= 2.0;
        m -= floor(m);
        ++bits;
      }
      K.coef_bits = bits > K.coef_bits ? bits : K.coef_bits;
    }
  }
  hipStream_t st = (hipStream_t)stream;
  // Without an order every row is the 8-slot instance's.  With one (rows sorted by the length of the row
  // stood on, descending): the rows above N2V_WLANES_SHORT slots first, in 32-slot groups; a whole wave per
  // walker for the rows from N2V_WLANES_WAVE_FROM slots on (both off by default: see above).
  const int short_n = order ? N2V_WLANES_SHORT : 0x7fffffff;
  const int wave_from = order ? N2V_WLANES_WAVE_FROM : 0x7fffffff;
  int rc = N2V_OK;
  int lanes_max = 0x7fffffff;  // the lane kernel's rows: up to this many slots
  if (order && scratch && row_sums && N2V_WLANES_MARGIN_FROM > 1) {
    // long rows first (the order is by row length, descending): a wave per walker, the pairing decided with
    // margins; scratch[0] = how many walkers it left undecided, scratch[1 ..] = those, -1 behind the last:
    // the exact wave kernel steps them
    // The cut between the two: a lane pays the row (~0.45 us per slot of its longest row, serial: 340 us of every step
    // at 768 slots whatever the batch), a wave ~700 vector instructions per walker.  A full batch (millions of walkers)
    // is throughput-bound and best at 768 (profiles/r8k_wm_cut.log); a small one is bound by that critical path, so the
    // cut comes down with the number of walkers (profiles/r10i_wm_cut_by_batch.log).  hubs->lane_cut > 0 overrides.
    int from = N2V_WLANES_MARGIN_FROM;
    if (hubs && hubs->lane_cut > 0) {
      from = hubs->lane_cut;
    } else {
      const int64_t by_batch = n_rows / N2V_WLANES_WALKERS_PER_CUT_SLOT;
      if (by_batch < from) from = by_batch < N2V_WLANES_MARGIN_FROM_MIN ? N2V_WLANES_MARGIN_FROM_MIN : (int)by_batch;
    }
    // scratch: two lists of n_rows + 2 words each -- [0] how many, [1 ..] the rows, -1 behind the last: what the
    // first launch (row sum in any order: general margins unless the sum is exact anyway) leaves undecided, and
    // what the second (row sum in the reference's order: exact-sum margins) still does; the exact wave kernel
    // steps those
    int64_t *second = scratch, *last = scratch + (n_rows + 2);
    n2v_weighted_hubs hb;  // (no summaries: every row of the wave kernel makes its pass)
    hb.block0 = nullptr;
    hb.sorted = nullptr;
    hb.prefix = nullptr;
    hb.min_slots = hb.lane_cut = 0;
    // (the sorted weights are stored as the graph's: fp32 beside g->w, fp64 beside g->w64)
    if (hubs && hubs->block0 && hubs->sorted && hubs->prefix) hb = *hubs;
    if (hipMemsetAsync(scratch, 0xff, sizeof(int64_t) * (size_t)(2 * (n_rows + 2)), st) != hipSuccess ||
        hipMemsetAsync(second, 0, sizeof(int64_t), st) != hipSuccess ||
        hipMemsetAsync(last, 0, sizeof(int64_t), st) != hipSuccess)
      return N2V_ELAUNCH;
    rc = g->w64 ? n2v::wm_launch<double, false>(g, g->w64, start_ids, num_walks, order, n_rows, from - 1, step,
                                                walk_length, K, seed, edge_state, walks, valid, status, second,
                                                row_sums, hb, st)
                : n2v::wm_launch<float, false>(g, g->w, start_ids, num_walks, order, n_rows, from - 1, step,
                                               walk_length, K, seed, edge_state, walks, valid, status, second, row_sums,
                                               hb, st);
    if (rc != N2V_OK) return rc;
#if N2V_WLANES_LANE_MARGINS
    // the rows below the cut: the same decision, a lane per walker (its undecided walkers join the list)
    rc = g->w64 ? n2v::lm_launch<double>(g, g->w64, start_ids, num_walks, order, n_rows, from - 1, step, walk_length, K,
                                         seed, edge_state, walks, valid, status, second, row_sums, st)
                : n2v::lm_launch<float>(g, g->w, start_ids, num_walks, order, n_rows, from - 1, step, walk_length, K,
                                        seed, edge_state, walks, valid, status, second, row_sums, st);
    if (rc != N2V_OK) return rc;
#endif
    rc = g->w64 ? n2v::wm_launch<double, true>(g, g->w64, start_ids, num_walks, second + 1, n_rows, 0, step,
                                               walk_length, K, seed, edge_state, walks, valid, status, last, row_sums,
                                               hb, st)
                : n2v::wm_launch<float, true>(g, g->w, start_ids, num_walks, second + 1, n_rows, 0, step, walk_length,
                                              K, seed, edge_state, walks, valid, status, last, row_sums, hb, st);
    if (rc != N2V_OK) return rc;
    if (hipMemsetAsync(status + 1, 0, sizeof(uint32_t), st) != hipSuccess) return N2V_ELAUNCH;
    rc = n2v_weighted_step_wave_launch(g, start_ids, num_walks, last + 1, n_rows, -1, step, walk_length,
                                       return_param, inout_param, seed, edge_state, walks, valid, status, stream);
    if (rc != N2V_OK) return rc;
#if N2V_WLANES_LANE_MARGINS
    return N2V_OK;  // (every row was one of the two margin kernels')
#endif
    lanes_max = from - 1;
  } else if (order && wave_from != 0x7fffffff) {
    if (hipMemsetAsync(status + 1, 0, sizeof(uint32_t), st) != hipSuccess) return N2V_ELAUNCH;
    rc = n2v_weighted_step_wave_launch(g, start_ids, num_walks, order, n_rows, wave_from - 1, step, walk_length,
                                       return_param, inout_param, seed, edge_state, walks, valid, status, stream);
    if (rc != N2V_OK) return rc;
  }
  if (g->w64) {
    if (order && short_n != 0x7fffffff)
      rc = n2v::wl_launch<double, 32, 64>(g, g->w64, start_ids, num_walks, order, n_rows, short_n, wave_from - 1,
                                          step, walk_length, K, seed, edge_state, walks, valid, status, st);
    if (rc == N2V_OK)
      rc = n2v::wl_launch<double, 8, 256>(g, g->w64, start_ids, num_walks, order, n_rows, 0,
                                          short_n < lanes_max ? short_n : lanes_max, step,
                                          walk_length, K, seed, edge_state, walks, valid, status, st);
  } else {
    if (order && short_n != 0x7fffffff)
      rc = n2v::wl_launch<float, 32, 64>(g, g->w, start_ids, num_walks, order, n_rows, short_n, wave_from - 1,
                                         step, walk_length, K, seed, edge_state, walks, valid, status, st);
    if (rc == N2V_OK)
      rc = n2v::wl_launch<float, 8, 256>(g, g->w, start_ids, num_walks, order, n_rows, 0,
                                         short_n < lanes_max ? short_n : lanes_max, step,
                                         walk_length, K, seed, edge_state, walks, valid, status, st);
  }
  return rc;
}

