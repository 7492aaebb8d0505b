// n2v_walk_wlanes.hip -- K2 exact mode on WEIGHTED graphs with p or q != 1: ONE STEP of every
// walker of a batch, one LANE per walker (n2v_walk_weighted_step, include/n2v_hip.h).
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

extern "C" int n2v_walk_weighted_step(const n2v_graph *g, const int32_t *start_ids, int32_t num_walks,
                                      const int64_t *order, int64_t n_rows, int32_t step,
                                      int32_t walk_length, double return_param, double inout_param,
                                      uint64_t seed, int64_t *edge_state, int32_t *walks, uint8_t *valid,
                                      uint32_t *status, void *stream) {
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
  hipStream_t st = (hipStream_t)stream;
  // Without an order every row is the 8-slot instance's.  With one (rows sorted by the length of the row
  // stood on, descending): the rows above N2V_WLANES_SHORT slots first, in 32-slot groups; a whole wave per
  // walker for the rows from N2V_WLANES_WAVE_FROM slots on (both off by default: see above).
  const int short_n = order ? N2V_WLANES_SHORT : 0x7fffffff;
  const int wave_from = order ? N2V_WLANES_WAVE_FROM : 0x7fffffff;
  int rc = N2V_OK;
  if (order && wave_from != 0x7fffffff) {
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
      rc = n2v::wl_launch<double, 8, 256>(g, g->w64, start_ids, num_walks, order, n_rows, 0, short_n, step,
                                          walk_length, K, seed, edge_state, walks, valid, status, st);
  } else {
    if (order && short_n != 0x7fffffff)
      rc = n2v::wl_launch<float, 32, 64>(g, g->w, start_ids, num_walks, order, n_rows, short_n, wave_from - 1,
                                         step, walk_length, K, seed, edge_state, walks, valid, status, st);
    if (rc == N2V_OK)
      rc = n2v::wl_launch<float, 8, 256>(g, g->w, start_ids, num_walks, order, n_rows, 0, short_n, step,
                                         walk_length, K, seed, edge_state, walks, valid, status, st);
  }
  return rc;
}

