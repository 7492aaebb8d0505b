// n2v_walk_wedge.hip -- K2 exact mode, biased, on a unit-weight graph that carries all three
// per-edge tables: class counts (inside the hop table), hop table and wedge table
// (include/n2v_hip.h; built once by n2v_edge_classes_build, n2v_wedge_build, n2v_hops_build).
//
// Same contract and same bits as the other exact kernels: per step the index
// sampling_from_alias(r1, r2) returns on the table generate_edge_alias_tables builds (reference
// randomwalk.py:86-99, :157-232).  With the tables at hand NO step needs the wave: one lane per
// walker, and per step
//   1. hop gather: the drawn neighbour, its row pointer and degree, the class counts of the edge
//      walked (used at the next step); the offset of this step's wedge list is requested first,
//      so both loads share one latency;
//   2. the table by counts: avg = (nR / p + nM + nO / q) / n in exact integers * 2^-20; the class
//      of slot `pick`: return if the neighbour is s, shared if `pick` is in the edge's wedge list;
//   3. exits without pairing (~80 % of the steps): an accepted underfull `pick`; an empty stack;
//   4. the pairing loop (:182-189) for slot `pick`, by the lane itself (n2v_unit_core.h): closed
//      form by the arrangement of the classes on the two stacks -- "other" alone underfull or
//      alone overfull; with the return run beside it (instance <1>: q > 1 with p > q, q < 1
//      with p < q) -- in exact integer bucket arithmetic; ties and thin margins fall through to
//      bit masks (rows of <= 64 slots), else the run-by-run replay over the list, else slot by slot.
// The path is written as whole 64-byte sectors through an LDS tile (a 4-byte store into a
// 324-byte-pitch row costs a 32-byte write request each: 12x write amplification measured on the
// lanes kernel).  Any dyadic p, q (the row sum is then an exact integer combination of the class
// counts); graphs without the tables keep the kernels of n2v_walk_unit.hip.
#include "n2v_wedge_step.h"

namespace n2v {

constexpr int kWedgeThreads = 256;
#ifndef N2V_WEDGE_WAVES
#define N2V_WEDGE_WAVES 6
#endif

template <int kMode>
__global__ __launch_bounds__(kWedgeThreads, N2V_WEDGE_WAVES) void walk_exact_wedge_kernel(
    n2v_graph g, const int32_t *__restrict__ start_ids, int64_t n_start, int32_t num_walks,
    int32_t walk_length, double p, double q, UnitConsts K, uint64_t seed,
    int32_t *__restrict__ walks_out, uint8_t *__restrict__ valid_out,
    uint32_t *__restrict__ status) {
  __shared__ int32_t path_tile[16][kWedgeThreads];             // word k of thread t at [k][t]
  __shared__ uint32_t stage_all[kWedgeThreads / 64][16 * 32];  // 2 KB per wave (lane_case_a)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  uint32_t *stage = stage_all[tid >> 6];
  const int64_t total = n_start * (int64_t)num_walks;
  const int L1 = walk_length + 1;
  const bool biased = !(p == 1.0 && q == 1.0);
  const bool need_mem = q != 1.0;
  // 1/q > 1: "other" is overfull, an overfull `pick` has no quick exit, so nearly every step runs
  // the pairing and needs the return position: request the wedge offset with the hop, always
  const bool always_pair = K.bO > 1.0;
  const bool merge_r = K.bR == K.bO;
  constexpr bool kShared = kMode == 1 || kMode == 2;
  const bool base_aligned = (reinterpret_cast<uintptr_t>(walks_out) & 63u) == 0;
#ifdef N2V_CHECK
  n2v_check_status = status;
#endif

  int64_t w0 = 0;  // absolute word index of path position 0 of the current walker
  int lo = 0;      // first word of the current sector that belongs to this row
  auto flush = [&](int64_t a) {  // words [sector(a) + lo, a] are complete: store them
    const int k = (int)(a & 15);
    int32_t *sec = walks_out + (a & ~(int64_t)15);
    if (lo == 0 && k == 15 && base_aligned) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        reinterpret_cast<int4 *>(sec)[u] =
            make_int4(path_tile[4 * u][tid], path_tile[4 * u + 1][tid], path_tile[4 * u + 2][tid],
                      path_tile[4 * u + 3][tid]);
    } else {
      for (int kk = lo; kk <= k; ++kk) sec[kk] = path_tile[kk][tid];
    }
    lo = 0;
  };
  auto emit = [&](int pos, int32_t x) {  // path position pos of the current walker
    const int64_t a = w0 + pos;
    path_tile[(int)(a & 15)][tid] = x;
    if ((a & 15) == 15 || pos == walk_length) flush(a);
  };

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
    int64_t vb = 0, e_prev = 0;
    uint32_t ec_prev = 0;  // class counts of the edge (s -> v), from the hop that walked it
    int n = 0;
    if (alive) {
      vb = g.rowptr[start];
      n = (int)(g.rowptr[start + 1] - vb);
      alive = n > 0;  // fugue.py:132
    }
    w0 = r * (int64_t)L1;
    lo = (int)(w0 & 15);
    if (have) {
      emit(0, alive ? start : -1);
      if (!alive)  // no such vertex / no out-edges: the row is all -1, like the other kernels
        for (int tt = 1; tt < L1; ++tt) emit(tt, -1);
    }
    int32_t s = -1, v = start;
    bool walking = alive;
    for (int step = 0; step < walk_length; ++step) {
      if (ballot64(walking) == 0ull) break;
      if (!walking) continue;
      const uint64_t bits = step_bits(h0, (uint32_t)step);
      const uint32_t u1 = (uint32_t)(bits >> 32), u2 = (uint32_t)bits;
      const int pick = pick_index(u1, n);
      int idx = pick;
      const bool w_wide = wedge_row_wide(g.wedge_wide, n);  // (mixed table: by the row stood on)
      const bool step_biased = s >= 0 && biased;
      uint32_t fR = 0, fM = 0;
      if (step_biased) {
        fR = ec_prev >> N2V_EC_RETURN_SHIFT;
        fM = ec_prev & N2V_EC_SHARED_MASK;
      }
      const bool counts_ok = step_biased && fR != N2V_EC_RETURN_SAT && fM != N2V_EC_SHARED_MASK;
      // this step's wedge list: its offset is requested before the hop so both loads overlap.
      // Steps whose edge has no shared neighbour need it only if the pairing runs (lazy).
      uint64_t wraw = 0;
      bool w_loaded = false;
#ifdef N2V_ABLATE_W
      const bool early_ok = N2V_ABLATE_W != 3;
#else
      const bool early_ok = true;
#endif
      // (not dyadic: the row sum needs the list and the return position at every step)
      if (counts_ok && ((need_mem && fM > 0) || ((kMode == 2 || (early_ok && always_pair)) && (fM > 0 || fR > 0)))) {
        N2V_CHECK_RANGE(3, e_prev, (int64_t)0, g.n_edges);
        wraw = g.wedge_off[e_prev];
        w_loaded = true;
      }
      n2v_hop h = load_hop(g.hops + vb + pick);
      int32_t x = h.col;
      if (step_biased) {
        if (!counts_ok) {
          // a saturated count: the tables of this graph must not have been passed (include/
          // n2v_hip.h, n2v_wedge_build); flag it and keep `pick` rather than read garbage
          atomicOr(status, N2V_ST_RANGE);
        } else {
          // p == q: the return slot carries the same value as an "other" slot -- it IS one, as far
          // as the table is concerned (randomwalk.py:223-230 give both w / p == w / q)
          const int nR = merge_r ? 0 : (int)fR, nM = need_mem ? (int)fM : 0, nO = n - nR - nM;
          int64_t w_off = (int64_t)(wraw & N2V_WEDGE_OFF_MASK);
          const bool isR = !merge_r && x == s;
          bool isM = false;
          int lo_pick = 0;  // entries of the edge's list below `pick`
          if (need_mem && !isR && nM > 0)  // :226
            lo_pick = wedge_lower(g.wedge_pos, w_off, nM, pick, w_wide, isM);
          double avg;  // :172
          if constexpr (kMode == 2) {
            // the reference's sum is rounded at every addition; any order of the same positive
            // addends agrees with it to (n - 1) 2^-53 relatively, so an underfull `pick` whose
            // acceptance clears that margin is decided from the counts alone
            const double b_pick = pick3(isR, isM, K.bR, K.bM, K.bO);
            const double approx = ((double)nR * K.bR + (double)nM * K.bM + (double)nO * K.bO) / (double)n;
            const double eps = ((double)n + 8.0) * 4.5e-16;
            const double r2a = (double)u2 * (1.0 / 4294967296.0);
            if (b_pick < approx * (1.0 - eps) && r2a < (b_pick / approx) * (1.0 - 2.0 * eps)) {
              avg = approx;  // only the (decided) comparison below reads it
            } else {
              const int rp = (int)(wraw >> N2V_WEDGE_RPOS_SHIFT);
              double sum;
              if (w_wide)
                sum = lane_row_sum<uint32_t>(n, K, nR, rp, nM,
                                             reinterpret_cast<const uint32_t *>(g.wedge_pos) + w_off);
              else
                sum = lane_row_sum<uint16_t>(n, K, nR, rp, nM,
                                             reinterpret_cast<const uint16_t *>(g.wedge_pos) + w_off);
              avg = sum / (double)n;
            }
          } else {
            const int64_t isum = (int64_t)nR * K.TR + (int64_t)nM * K.TM + (int64_t)nO * K.TO;
            avg = ((double)isum * (1.0 / 1048576.0)) / (double)n;
          }
          const double p_pick = pick3(isR, isM, K.bR, K.bM, K.bO) / avg;  // :173
          const double r2 = (double)u2 * (1.0 / 4294967296.0);
          if (!(p_pick < 1.0 && r2 < p_pick)) {
            // underfull / overfull by class without dividing: for correctly rounded fp64 division
            // fl(b / avg) < 1.0 <=> b < avg (b < avg gives a quotient <= 1 - ulp(avg) / avg <=
            // 1 - 2^-53, which rounds below 1.0; b >= avg gives >= 1).  The class VALUES are only
            // needed by the step-by-step replays.
            const bool uR = K.bR < avg, uM = K.bM < avg, uO = K.bO < avg;
            const bool any_under = (nR && uR) || (nM && uM) || (nO && uO);
            const bool any_over = (nR && !uR) || (nM && !uM) || (nO && !uO);
            if (!any_under || !any_over) {  // the loop of :182 never runs
              if (!(r2 < p_pick)) idx = 0;
            } else {
              if (!w_loaded) {  // the return position (and an empty list)
                N2V_CHECK_RANGE(3, e_prev, (int64_t)0, g.n_edges);
                wraw = g.wedge_off[e_prev];
                w_off = (int64_t)(wraw & N2V_WEDGE_OFF_MASK);
              }
              const int w_rpos = (int)(wraw >> N2V_WEDGE_RPOS_SHIFT);
              // the stacks: 1 = "other" alone underfull, 2 = "other" alone overfull, 3 = return +
              // "other" underfull, 4 = return + "other" overfull, 5 = return alone overfull, 0 = else
              int arr = 0;
              if (uO && !(nR && uR) && !(nM && uM)) arr = 1;
              else if (!uO && nO > 0 && (!nR || uR) && (!nM || uM)) arr = 2;
              else if (kShared && uO && nR && uR && nM && !uM) arr = 3;
              else if (kShared && !uO && nO > 0 && nR && !uR && nM && uM) arr = 4;
              else if (kShared && uO && nR && !uR && nM && uM) arr = 5;
              int res;
              // a plain branch on the (uniform) list width: never a select between two loads
              if (w_wide)
                res = pair_listed<uint32_t, kMode>(arr, n, pick, r2, K, avg, nR, w_rpos, nM,
                                            reinterpret_cast<const uint32_t *>(g.wedge_pos) + w_off,
                                            isR, isM, lo_pick, stage, lane);
              else
                res = pair_listed<uint16_t, kMode>(arr, n, pick, r2, K, avg, nR, w_rpos, nM,
                                            reinterpret_cast<const uint16_t *>(g.wedge_pos) + w_off,
                                            isR, isM, lo_pick, reinterpret_cast<uint16_t *>(stage), lane);
              idx = res;
              N2V_CHECK_RANGE(2, idx, 0, n);
            }
#ifdef N2V_ABLATE_W
            if (N2V_ABLATE_W == 2) idx = pick;  // timing-only: never a second gather
#endif
            if (idx != pick) {
              h = load_hop(g.hops + vb + idx);
              x = h.col;
            }
          }
        }
      }
      emit(step + 1, x);
      e_prev = vb + idx;
      ec_prev = h.classes;
      s = v;
      v = x;
      if (step + 1 < walk_length) {
        vb = hop_row(h);
        n = hop_deg(h);
        if (n == 0) {  // fugue.py:147: the walker vanishes at a sink, the rest of its row is -1
          walking = false;
          alive = false;
          for (int tt = step + 2; tt < L1; ++tt) emit(tt, -1);
        }
      }
    }
    if (have) valid_out[r] = alive ? 1 : 0;
  }
}


// ---- the same walk with the per-edge wedge SLOTS (n2v_wedge_slots_build): one lane per walker,
// replays inline, the step itself in wedge_step (n2v_wedge_step.h).  The slot of the edge walked
// is requested together with the hop entry and holds the return position and a short list itself,
// so a step that needs its list is two INDEPENDENT gathers instead of hop + offset -> list.
// kMode 0, 1, 3: dyadic p, q (values that are not keep the kernel above); 16-bit positions.
#ifndef N2V_SLOTS_WAVES
#define N2V_SLOTS_WAVES 6
#endif
// Instances <1> (the return run shares a stack) and <2> (values that are not dyadic) carry the most code per
// step; at six waves per SIMD (80 VGPRs) they spill 144 / 132 bytes per lane, at five (96 VGPRs) next to
// nothing.  The kernel is bound by its sectors, not by its waves, so those two run at five.
#ifndef N2V_SLOTS_WAVES_BIG
#define N2V_SLOTS_WAVES_BIG 5
#endif
template <int kMode>
__global__ __launch_bounds__(kWedgeThreads, (kMode == 1 || kMode == 2) ? N2V_SLOTS_WAVES_BIG : N2V_SLOTS_WAVES)
void walk_exact_wedge_slots_kernel(
    n2v_graph g, const int32_t *__restrict__ start_ids, int64_t n_start, int32_t num_walks,
    int32_t walk_length, double q, UnitConsts K, uint64_t seed, int32_t *__restrict__ walks_out,
    uint8_t *__restrict__ valid_out, uint32_t *__restrict__ status) {
  __shared__ int32_t path_tile[16][kWedgeThreads];             // word k of thread t at [k][t]
  __shared__ uint32_t stage_all[kWedgeThreads / 64][16 * 32];  // 2 KB per wave (lane_case_a)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  uint32_t *stage = stage_all[tid >> 6];
  const int64_t total = n_start * (int64_t)num_walks;
  const int L1 = walk_length + 1;
  const StepFlags F = step_flags(g, K, q);
#ifdef N2V_NEAR_COUNT
  n2v_count_words = status;
#endif
#if defined(N2V_BIG_STATS) && defined(N2V_BIG_DECLINES)
  n2v_big_words = status;
#endif
#ifdef N2V_DECLINE_STATS
  n2v_decline_words = status + 8;
#endif
  const bool base_aligned = (reinterpret_cast<uintptr_t>(walks_out) & 63u) == 0;

  int64_t w0 = 0;  // absolute word index of path position 0 of the current walker
  int lo = 0;      // first word of the current sector that belongs to this row
  auto flush = [&](int64_t a) {  // words [sector(a) + lo, a] are complete: store them
    const int k = (int)(a & 15);
    int32_t *sec = walks_out + (a & ~(int64_t)15);
    if (lo == 0 && k == 15 && base_aligned) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        reinterpret_cast<int4 *>(sec)[u] =
            make_int4(path_tile[4 * u][tid], path_tile[4 * u + 1][tid], path_tile[4 * u + 2][tid],
                      path_tile[4 * u + 3][tid]);
    } else {
      for (int kk = lo; kk <= k; ++kk) sec[kk] = path_tile[kk][tid];
    }
    lo = 0;
  };
  auto emit = [&](int pos, int32_t x) {  // path position pos of the current walker
    const int64_t a = w0 + pos;
    path_tile[(int)(a & 15)][tid] = x;
    if ((a & 15) == 15 || pos == walk_length) flush(a);
  };

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
    int64_t vb = 0, e_prev = 0;
    uint32_t ec_prev = 0;  // class counts of the edge (s -> v), from the hop that walked it
    int n = 0;
    if (alive) {
      vb = g.rowptr[start];
      n = (int)(g.rowptr[start + 1] - vb);
      alive = n > 0;  // fugue.py:132
    }
    w0 = r * (int64_t)L1;
    lo = (int)(w0 & 15);
    if (have) {
      emit(0, alive ? start : -1);
      if (!alive)  // no such vertex / no out-edges: the row is all -1, like the other kernels
        for (int tt = 1; tt < L1; ++tt) emit(tt, -1);
    }
    int32_t s = -1, v = start;
    bool walking = alive;
    for (int step = 0; step < walk_length; ++step) {
      if (ballot64(walking) == 0ull) break;
      if (!walking) continue;
      const uint64_t bits = step_bits(h0, (uint32_t)step);
      const uint32_t u1 = (uint32_t)(bits >> 32), u2 = (uint32_t)bits;
      n2v_hop h;
      int idx;
      if (s >= 0) {
        idx = wedge_step<kMode, false, true>(g, K, F, u1, u2, s, vb, n, e_prev, ec_prev, h, stage, lane,
                                             status);
      } else {  // first step: generate_alias_tables of unit weights is the uniform draw (:320-321)
        idx = pick_index(u1, n);
        h = load_hop(g.hops + vb + idx);
      }
      const int32_t x = h.col;
      emit(step + 1, x);
      e_prev = vb + idx;
      ec_prev = h.classes;
      s = v;
      v = x;
      if (step + 1 < walk_length) {
        vb = hop_row(h);
        n = hop_deg(h);
        if (n == 0) {  // fugue.py:147: the walker vanishes at a sink, the rest of its row is -1
          walking = false;
          alive = false;
          for (int tt = step + 2; tt < L1; ++tt) emit(tt, -1);
        }
      }
    }
    if (have) valid_out[r] = alive ? 1 : 0;
  }
}

// ---- one step of the walkers resident on one part of a partitioned graph, wedge lists travelling
// (n2v_partition_step with N2V_SRC_WEDGES, n2v_walk.hip).  A walker that leaves along edge e brings
// the class counts of e, the return position and the wedge list of e -- exactly what the kernel
// above reads from hops[e], wedge_off[e] and wedge_pos -- so its step here is the same per-lane
// work: the row sum from the counts, the class of slot `pick` by one search in the list, the
// two exits, else the closed form of the arrangement (pair_listed).  One LANE per walker; neither
// N(s) nor a pass over N(v).  (The step is restated, not shared with the kernel above: that one's
// registers are the flagship configuration's.)
template <int kMode>
__global__ __launch_bounds__(kWedgeThreads, N2V_WEDGE_WAVES) void partition_step_wedge_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, int64_t lo, int64_t n_local,
    const int64_t *__restrict__ head, int head_cols, const int64_t *__restrict__ src_ptr,
    const int32_t *__restrict__ src_ids, int src_at, int64_t k, double p, double q, UnitConsts K,
    uint64_t seed, int32_t *__restrict__ next_out, int64_t *__restrict__ edge_out,
    uint32_t *__restrict__ status) {
  // src_at: src_ptr[i] is where walker i's list starts (N2V_SRC_WEDGES_AT: lists appended in any
  // order by n2v_partition_forward), its length is the shared count in the header
  __shared__ uint32_t stage_all[kWedgeThreads / 64][16 * 32];  // 2 KB per wave (lane_case_a)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  uint32_t *stage = stage_all[tid >> 6];
  const bool biased = !(p == 1.0 && q == 1.0);
  const bool need_mem = q != 1.0;
  const bool merge_r = K.bR == K.bO;
  constexpr bool kShared = kMode == 1 || kMode == 2;
#ifdef N2V_CHECK
  n2v_check_status = status;
#endif
  // whole waves stay in the loop (lane_case_a stages through the wave's LDS tile)
  const int64_t k_up = (k + 63) & ~(int64_t)63;
  for (int64_t i = (int64_t)blockIdx.x * kWedgeThreads + tid; i < k_up;
       i += (int64_t)gridDim.x * kWedgeThreads) {
    // (a header whose output row is negative is an EMPTY slot of a capacity-bounded mailbox: skipped)
    const bool have = i < k;
    const int64_t *hd = head + (have ? i : 0) * head_cols;
    const bool empty = have && hd[0] < 0;
    const uint64_t key = (uint64_t)hd[1];
    const int64_t sv = hd[2];
    const uint32_t step = (uint32_t)hd[3];
    const int32_t s = (int32_t)(sv >> 32);
    const int64_t local = (int64_t)(uint32_t)sv - lo;
    int32_t next = -1;
    int64_t edge = -1;
    bool ok = have && !empty;
    if (ok && (local < 0 || local >= n_local)) {  // a walker that is not resident here
      atomicOr(status, N2V_ST_RANGE);
      ok = false;
    }
    int64_t vb = 0;
    int n = 0;
    if (ok) {
      vb = rowptr[local];
      n = (int)(rowptr[local + 1] - vb);
      ok = n > 0;  // (arrivals at a sink were dropped by the caller, fugue.py:147)
    }
    if (ok) {
      const uint64_t bits = step_bits(walker_stream(seed, key), step);
      const uint32_t u1 = (uint32_t)(bits >> 32), u2 = (uint32_t)bits;
      const int pick = pick_index(u1, n);
      int idx = pick;
      if (s >= 0 && biased) {
        const uint64_t extra = (uint64_t)hd[4];
        const uint32_t ec = (uint32_t)extra;
        const uint32_t fR = ec >> N2V_EC_RETURN_SHIFT, fM = ec & N2V_EC_SHARED_MASK;
        const int w_rpos = (int)((extra >> 32) & 0xffffffu);
        const int64_t sb = src_ptr[i];
        const uint32_t *list = reinterpret_cast<const uint32_t *>(src_ids) + sb;
        if (fR == N2V_EC_RETURN_SAT || fM == N2V_EC_SHARED_MASK ||
            (!src_at && src_ptr[i + 1] - sb != (need_mem ? (int64_t)fM : 0)) ||  // (q == 1: no list travels)
            (int64_t)fR + (int64_t)fM > n || w_rpos + (int)fR > n) {
          atomicOr(status, N2V_ST_RANGE);  // not a wedge list of an edge into this row
          idx = -1;
        } else {
          const int nR = merge_r ? 0 : (int)fR, nM = need_mem ? (int)fM : 0, nO = n - nR - nM;
          const bool isR = !merge_r && pick >= w_rpos && pick < w_rpos + (int)fR;  // N(v)[pick] == s
          bool isM = false;
          int lo_pick = 0;  // entries of the list below `pick`
          if (need_mem && !isR && nM > 0) lo_pick = wedge_lower_t<uint32_t>(list, 0, nM, pick, isM);
          double avg;  // :172
          int near_idx = -1;
          if constexpr (kMode == 2) {
            const double b_pick = pick3(isR, isM, K.bR, K.bM, K.bO);
            const double approx = ((double)nR * K.bR + (double)nM * K.bM + (double)nO * K.bO) / (double)n;
            const double eps = ((double)n + 8.0) * 4.5e-16;
            const double r2a = (double)u2 * (1.0 / 4294967296.0);
            if (b_pick < approx * (1.0 - eps) && r2a < (b_pick / approx) * (1.0 - 2.0 * eps)) {
              avg = approx;  // only the (decided) comparison below reads it
            } else {
              // the closed forms on the values the counts give, with margins (n2v_unit_near.h), before
              // the row is added up in the reference's order
              near_idx = N2V_NEAR_FORMS ? near_step<uint32_t>(n, pick, r2a, K, nR, w_rpos, nM, list, isR, isM,
                                                              lo_pick, -1)
                                        : -1;
              avg = near_idx >= 0 ? approx : lane_row_sum<uint32_t>(n, K, nR, w_rpos, nM, list) / (double)n;
            }
          } else {
            const int64_t isum = (int64_t)nR * K.TR + (int64_t)nM * K.TM + (int64_t)nO * K.TO;
            avg = ((double)isum * (1.0 / 1048576.0)) / (double)n;
          }
          const double p_pick = pick3(isR, isM, K.bR, K.bM, K.bO) / avg;  // :173
          const double r2 = (double)u2 * (1.0 / 4294967296.0);
          if (near_idx >= 0) {
            idx = near_idx;
          } else if (!(p_pick < 1.0 && r2 < p_pick)) {
            const bool uR = K.bR < avg, uM = K.bM < avg, uO = K.bO < avg;
            const bool any_under = (nR && uR) || (nM && uM) || (nO && uO);
            const bool any_over = (nR && !uR) || (nM && !uM) || (nO && !uO);
            if (!any_under || !any_over) {  // the loop of :182 never runs
              if (!(r2 < p_pick)) idx = 0;
            } else {
              int arr = 0;
              if (uO && !(nR && uR) && !(nM && uM)) arr = 1;
              else if (!uO && nO > 0 && (!nR || uR) && (!nM || uM)) arr = 2;
              else if (kShared && uO && nR && uR && nM && !uM) arr = 3;
              else if (kShared && !uO && nO > 0 && nR && !uR && nM && uM) arr = 4;
              else if (kShared && uO && nR && !uR && nM && uM) arr = 5;
              idx = pair_listed<uint32_t, kMode>(arr, n, pick, r2, K, avg, nR, w_rpos, nM, list, isR,
                                                 isM, lo_pick, stage, lane);
              N2V_CHECK_RANGE(2, idx, 0, n);
            }
          }
        }
      }
      if (idx >= 0) {
        next = col[vb + idx];
        edge = vb + idx;
      }
    }
    if (have) {
      next_out[i] = next;
      if (edge_out) edge_out[i] = edge;
    }
  }
}

}  // namespace n2v

// returns 1 when the kernel applies (and was launched), 0 when it does not, < 0 on error
int n2v_walk_wedge_try(const n2v_graph *g, const int32_t *start_ids, int64_t n_start,
                       int32_t num_walks, int32_t walk_length, double p, double q,
                       const n2v::UnitConsts &K, uint64_t seed, int32_t *walks_out,
                       uint8_t *valid_out, uint32_t *status, void *stream) {
  if (!g->hops || !g->wedge_off || !g->wedge_pos || g->w || g->w64) return 0;
  const int64_t total = n_start * (int64_t)num_walks;
  if (total >= 0xffffff00ll) return 0;
  if (total == 0) return 1;
  // the return run shares a stack with "other" on ordinary rows: q > 1 with p > q, q < 1 with p < q
  const bool alone_under = K.bO <= 1.0 && K.bR >= K.bO, alone_over = K.bO >= 1.0 && K.bR <= K.bO;
  if (g->wedge_wide < 0 || g->wedge_wide > 65536) return N2V_EINVAL;
  // (a mixed table -- rows of wedge_wide slots and more -- goes to the slots kernel with FOLDED slots only)
  if (g->wedge_slots && g->wedge_wide != 1 && !(g->reserved & 2) &&
      (g->wedge_wide == 0 || (g->reserved2 & N2V_SLOTS_FOLDED))) {
    n2v_graph gs = *g;  // the table of row sums counts for the (p, q) it was built for only
    if (K.dyadic || gs.row_sums == nullptr || !(gs.row_sums_p == p && gs.row_sums_q == q) || gs.row_sums_from < 1)
      gs.row_sums = nullptr;
    g = &gs;
    // the wedge slots are at hand: the list of a step arrives with its hop entry
    auto sk = !K.dyadic    ? n2v::walk_exact_wedge_slots_kernel<2>
              : alone_under ? n2v::walk_exact_wedge_slots_kernel<0>
              : alone_over  ? n2v::walk_exact_wedge_slots_kernel<3>
                            : n2v::walk_exact_wedge_slots_kernel<1>;
    int64_t sblocks = (total + n2v::kWedgeThreads - 1) / n2v::kWedgeThreads;
    const int64_t scap = n2v::resident_blocks((const void *)sk, n2v::kWedgeThreads, 0);
    if (sblocks > scap) sblocks = scap;
    hipLaunchKernelGGL(sk, dim3((unsigned)sblocks), dim3(n2v::kWedgeThreads), 0, (hipStream_t)stream,
                       *g, start_ids, n_start, num_walks, walk_length, q, K, seed, walks_out,
                       valid_out, status);
    if (hipGetLastError() != hipSuccess) return N2V_ELAUNCH;
    return 1;
  }
  if (g->reserved2 & N2V_HOPS_INLINE_RPOS) return N2V_EINVAL;  // that hop table is the slots kernel's
  n2v_graph gp = *g;
  gp.row_sums = nullptr;  // (this kernel adds every row up itself)
  g = &gp;
  auto kernel = !K.dyadic    ? n2v::walk_exact_wedge_kernel<2>
                : alone_under ? n2v::walk_exact_wedge_kernel<0>
                : alone_over  ? n2v::walk_exact_wedge_kernel<3>
                              : n2v::walk_exact_wedge_kernel<1>;
  int64_t blocks = (total + n2v::kWedgeThreads - 1) / n2v::kWedgeThreads;
  const int64_t cap = n2v::resident_blocks((const void *)kernel, n2v::kWedgeThreads, 0);
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(n2v::kWedgeThreads), 0,
                     (hipStream_t)stream, *g, start_ids, n_start, num_walks, walk_length, p, q, K, seed,
                     walks_out, valid_out, status);
  if (hipGetLastError() != hipSuccess) return N2V_ELAUNCH;
  return 1;
}

// the wedge-list instance of n2v_partition_step: 1 = launched, < 0 on error
int n2v_partition_step_wedge_launch(const int64_t *rowptr, const int32_t *col, int64_t lo,
                                    int64_t n_local, const int64_t *head, int32_t head_cols,
                                    const int64_t *src_ptr, const int32_t *src_ids, int32_t src_at,
                                    int64_t k, double p, double q, const n2v::UnitConsts &K, uint64_t seed,
                                    int32_t *next_out, int64_t *edge_out, uint32_t *status,
                                    void *stream) {
  const bool biased = !(p == 1.0 && q == 1.0);  // p == q == 1: the header's first four words do
  if (biased && (head_cols < 5 || !src_ptr || !src_ids)) return N2V_EINVAL;
  const bool alone_under = K.bO <= 1.0 && K.bR >= K.bO, alone_over = K.bO >= 1.0 && K.bR <= K.bO;
  auto kernel = !K.dyadic    ? n2v::partition_step_wedge_kernel<2>
                : alone_under ? n2v::partition_step_wedge_kernel<0>
                : alone_over  ? n2v::partition_step_wedge_kernel<3>
                              : n2v::partition_step_wedge_kernel<1>;
  int64_t blocks = (k + n2v::kWedgeThreads - 1) / n2v::kWedgeThreads;
  const int64_t cap = n2v::resident_blocks((const void *)kernel, n2v::kWedgeThreads, 0);
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(n2v::kWedgeThreads), 0, (hipStream_t)stream,
                     rowptr, col, lo, n_local, head, (int)head_cols, src_ptr, src_ids, (int)src_at, k, p, q, K,
                     seed, next_out, edge_out, status);
  if (hipGetLastError() != hipSuccess) return N2V_ELAUNCH;
  return 1;
}

// ---- row sums of the tables of the steps into long rows, for one (p, q) that is not dyadic (include/n2v_hip.h,
// n2v_edge_row_sums_build): lane_row_sum -- the routine a step of walk_exact_wedge_slots_kernel<2> would run -- once
// per edge, one lane per edge of the caller's list
namespace n2v {
__global__ __launch_bounds__(256) void edge_row_sums_kernel(n2v_graph g, UnitConsts K, bool merge_r, bool need_mem,
                                                            const int64_t *__restrict__ edges, int64_t k,
                                                            double *__restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < k; i += (int64_t)gridDim.x * 256) {
    const int64_t e = edges[i];
    if (e < 0 || e >= g.n_edges) continue;
    const int64_t x = g.col[e];
    const int64_t nn = g.rowptr[x + 1] - g.rowptr[x];
    const uint32_t ec = g.edge_classes[e];
    const uint32_t fR = ec >> N2V_EC_RETURN_SHIFT, fM = ec & N2V_EC_SHARED_MASK;
    if (fR == N2V_EC_RETURN_SAT || fM == N2V_EC_SHARED_MASK || nn <= 0 || nn >= (1 << 24)) {
      out[e] = 0.0;  // (the walk kernels flag such an edge before they would read this)
      continue;
    }
    const int n = (int)nn;
    const int nR = merge_r ? 0 : (int)fR, nM = need_mem ? (int)fM : 0;
    const uint64_t wraw = g.wedge_off[e];
    const int64_t off = (int64_t)(wraw & N2V_WEDGE_OFF_MASK);
    const int rp = (int)(wraw >> N2V_WEDGE_RPOS_SHIFT);
    double sum;
    if (wedge_row_wide(g.wedge_wide, n))
      sum = lane_row_sum<uint32_t>(n, K, nR, rp, nM, reinterpret_cast<const uint32_t *>(g.wedge_pos) + off);
    else
      sum = lane_row_sum<uint16_t>(n, K, nR, rp, nM, reinterpret_cast<const uint16_t *>(g.wedge_pos) + off);
    out[e] = sum;
  }
}
}  // namespace n2v

int n2v_edge_row_sums_launch(const n2v_graph *g, const n2v::UnitConsts &K, double q, const int64_t *edges, int64_t k,
                             double *sums_out, void *stream) {
  int64_t blocks = (k + 255) / 256;
  const int64_t cap = 4 * n2v::resident_blocks((const void *)n2v::edge_row_sums_kernel, 256, 0);
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(n2v::edge_row_sums_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *g, K,
                     K.bR == K.bO, q != 1.0, edges, k, sums_out);
  if (hipGetLastError() != hipSuccess) return N2V_ELAUNCH;
  return N2V_OK;
}
