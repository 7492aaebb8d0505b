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

constexpr int kWlThreads = 256;
constexpr int kWlChunk = 8;  // slots per load group of a cursor

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

// biased weight of a slot (:219-231): cls 0 = other (w / q), 1 = shared or first step (w), 2 = return (w / p)
__device__ __forceinline__ double wl_bias(double w, int cls, const WlConsts &K) {
  if (cls == 1) return w;
  if (cls == 2) return K.p_pow2 ? w * K.inv_p : w / K.p;
  return K.q_pow2 ? w * K.inv_q : w / K.q;
}

// 8 consecutive weights of a row as fp64 (slots at or beyond n: 0)
template <typename WT>
__device__ __forceinline__ void wl_load8(const WT *w, int c0, int n, double (&out)[kWlChunk]) {
  if (c0 + kWlChunk <= n) {
    // whole group inside the row: wide loads (dword-aligned only: rows start anywhere)
    struct __attribute__((packed, aligned(4))) Pack {
      WT v[kWlChunk];
    };
    const Pack pk = *reinterpret_cast<const Pack *>(w + c0);
#pragma unroll
    for (int k = 0; k < kWlChunk; ++k) out[k] = (double)pk.v[k];
  } else {
#pragma unroll
    for (int k = 0; k < kWlChunk; ++k) out[k] = (c0 + k < n) ? (double)w[c0 + k] : 0.0;
  }
}

// one of the two cursors of the pairing: runs down the row and yields, in descending position, the
// slots that are underfull (kUnder) / not underfull.  The biased weights of its current group of 8
// sit in the lane's LDS column `tile[k][tid]`.
struct WlCursor {
  int chunk;        // group loaded last (groups above it are done)
  uint32_t mask;    // slots of that group still to yield
  int lm, next_m;   // backward cursor into the shared-position list: list[lm] = next_m (or -1)
};

template <typename WT, bool kUnder>
__device__ __forceinline__ void wl_refill(WlCursor &C, const WlRow &R, const WT *w, const WlConsts &K,
                                          double avg, double *tile, int tid) {
  while (C.mask == 0u && C.chunk > 0) {
    --C.chunk;
    const int c0 = C.chunk * kWlChunk;
    double wv[kWlChunk];
    wl_load8<WT>(w, c0, R.n, wv);
    uint32_t mm = 0u;  // shared slots of this group
    while (C.next_m >= c0) {
      mm |= 1u << (C.next_m - c0);
      --C.lm;
      C.next_m = C.lm >= 0 ? wl_list_at(R, C.lm) : -1;
    }
    uint32_t mask = 0u;
#pragma unroll
    for (int k = 0; k < kWlChunk; ++k) {
      const int j = c0 + k;
      int cls = 1;
      if (!R.first) cls = ((mm >> k) & 1u) ? 1 : ((j >= R.rpos && j < R.rpos + R.nR) ? 2 : 0);
      const double b = wl_bias(wv[k], cls, K);
      tile[k * kWlThreads + tid] = b;
      // probs[i] < 1.0 (:175-180) <=> fl(b / avg) < 1.0 <=> b < avg for a correctly rounded quotient
      const bool under = b < avg;
      if (j < R.n && under == kUnder) mask |= 1u << k;
    }
    C.mask = mask;
  }
}

// index sampling_from_alias(r1, r2) returns on the table of this row, or -1: ZeroDivisionError (:172-173)
template <typename WT>
__device__ __forceinline__ int wl_draw(const WlRow &R, const WT *w, const WlConsts &K, int pick, double r2,
                                       double *tU, double *tO, int tid) {
  const int n = R.n;
  // ---- the row sum in the reference's order (:172) -----------------------------------------------
  double total = 0.0, b_pick = 0.0;
  double bmin = __builtin_huge_val(), bmax = -__builtin_huge_val();
  int lm = 0, next_m = (!R.first && R.nM > 0) ? wl_list_at(R, 0) : 0x7fffffff;
  for (int c0 = 0; c0 < n; c0 += kWlChunk) {
    double wv[kWlChunk];
    wl_load8<WT>(w, c0, n, wv);
#pragma unroll
    for (int k = 0; k < kWlChunk; ++k) {
      const int j = c0 + k;
      if (j < n) {
        int cls = 1;
        if (!R.first) {
          cls = 0;
          if (j == next_m) {
            cls = 1;
            ++lm;
            next_m = lm < R.nM ? wl_list_at(R, lm) : 0x7fffffff;
          } else if (j >= R.rpos && j < R.rpos + R.nR) {
            cls = 2;
          }
        }
        const double b = wl_bias(wv[k], cls, K);
        total = total + b;  // one rounding per addition, left to right
        bmin = fmin(bmin, b);
        bmax = fmax(bmax, b);
        if (j == pick) b_pick = b;
      }
    }
  }
  const double avg = total / (double)n;  // :172
  if (avg == 0.0) return -1;
  const double p_pick = b_pick / avg;    // :173
  if (p_pick < 1.0 && r2 < p_pick) return pick;  // an untouched underfull slot: final
  // x -> x / avg is monotone: the extreme weights say whether a stack is empty (:182 never runs)
  if (!(bmin / avg < 1.0) || (bmax / avg < 1.0)) return (r2 < p_pick) ? pick : 0;

  // ---- the pairing loop (:182-189) until slot `pick` is final -------------------------------------
  const int nch = (n + kWlChunk - 1) / kWlChunk;
  const int lm_top = R.first ? -1 : R.nM - 1;
  const int m_top = lm_top >= 0 ? wl_list_at(R, lm_top) : -1;
  WlCursor U{nch, 0u, lm_top, m_top}, O{nch, 0u, lm_top, m_top};
  bool carry = false;  // the slot demoted last is the next `under`
  double carry_r = 0.0;
  int carry_idx = 0;
  double fin_prob = p_pick;
  int fin_alias = 0;
  for (;;) {
    wl_refill<WT, false>(O, R, w, K, avg, tO, tid);
    if (O.mask == 0u) {  // `overfull` is empty: a demoted slot keeps alias 0
      if (carry && carry_idx == pick) fin_prob = carry_r;
      break;
    }
    const int ko = 31 - __clz(O.mask);
    O.mask ^= 1u << ko;
    const int o_idx = O.chunk * kWlChunk + ko;
    double r = tO[ko * kWlThreads + tid] / avg;  // probs[over]
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
      wl_refill<WT, true>(U, R, w, K, avg, tU, tid);
      if (U.mask == 0u) {  // `underfull` is empty
        if (o_idx == pick) fin_prob = r;
        finished = true;
        break;
      }
      const int ku = 31 - __clz(U.mask);
      U.mask ^= 1u << ku;
      const int u_idx = U.chunk * kWlChunk + ku;
      const double pu = tU[ku * kWlThreads + tid] / avg;  // probs[under]
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
  return (r2 < fin_prob) ? pick : fin_alias;  // :95-99
}

template <typename WT>
__global__ __launch_bounds__(kWlThreads, 4) void walk_weighted_step_kernel(
    n2v_graph g, const WT *__restrict__ w, const int32_t *__restrict__ start_ids, int32_t num_walks,
    const int64_t *__restrict__ order, int64_t n_rows, int32_t step, int32_t walk_length, WlConsts K,
    uint64_t seed, int64_t *__restrict__ edge_state, int32_t *__restrict__ walks,
    uint8_t *__restrict__ valid, uint32_t *__restrict__ status) {
  __shared__ double tU[kWlChunk * kWlThreads], tO[kWlChunk * kWlThreads];
  const int tid = threadIdx.x;
  const int L1 = walk_length + 1;
  const bool biased = !(K.p == 1.0 && K.q == 1.0);
  for (int64_t i = (int64_t)blockIdx.x * kWlThreads + tid; i < n_rows; i += (int64_t)gridDim.x * kWlThreads) {
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
    if (R.n <= 0) continue;
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
    const int idx = wl_draw<WT>(R, w + vb, K, pick, r2, tU, tO, tid);
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
  int64_t blocks = (n_rows + n2v::kWlThreads - 1) / n2v::kWlThreads;
  const void *fn = g->w64 ? (const void *)n2v::walk_weighted_step_kernel<double>
                          : (const void *)n2v::walk_weighted_step_kernel<float>;
  const int64_t cap = n2v::resident_blocks(fn, n2v::kWlThreads, 0);
  if (blocks > cap) blocks = cap;
  if (g->w64)
    hipLaunchKernelGGL(n2v::walk_weighted_step_kernel<double>, dim3((unsigned)blocks), dim3(n2v::kWlThreads), 0,
                       (hipStream_t)stream, *g, g->w64, start_ids, num_walks, order, n_rows, step, walk_length, K,
                       seed, edge_state, walks, valid, status);
  else
    hipLaunchKernelGGL(n2v::walk_weighted_step_kernel<float>, dim3((unsigned)blocks), dim3(n2v::kWlThreads), 0,
                       (hipStream_t)stream, *g, g->w, start_ids, num_walks, order, n_rows, step, walk_length, K,
                       seed, edge_state, walks, valid, status);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}
