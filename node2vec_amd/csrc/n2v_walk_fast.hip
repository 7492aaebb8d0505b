// n2v_walk_fast.hip -- K2 fast mode: the same transition distribution as
// generate_edge_alias_tables + sampling_from_alias (reference randomwalk.py:193-232,
// :86-99) without rebuilding a table per step.
//
// A candidate x is drawn from the PRECOMPUTED first-order alias table of the
// current vertex v (K1; exactly one 16-byte slot gather, the slot holds both the
// neighbour and its alias's vertex id) and accepted with probability
// beta(x) / beta_max, beta = 1/p if x == s, 1 if x in N(s) (binary search over
// the sorted row of s), 1/q otherwise: P(x) ~ w(v,x) * beta(x), the reference's
// unnormalised probability (:223-230).  The first step (s < 0), and every step when
// p == q == 1, is the unbiased table itself and is draw-for-draw identical to the exact
// mode (n2v_walk dispatches exact walks with p == q == 1 here when the slots exist).
// That is the REJECTION sampler (weighted graphs; unit-weight graphs without the wedge table).  On
// a unit-weight graph with the per-edge tables the step is drawn from the LAYERS of its table
// instead (kClassFirst below): one trial per step, no rejection at q >= 1 with p <= q.
//
// One LANE per walker, walkers resident for all L steps.  Every loop iteration
// each live lane performs ONE trial for its own current step; a lane whose
// candidate is accepted moves to its next step (or its next walker) at once, so
// no lane waits for another's rejections: the wave only drains when the ballot of
// live lanes is empty.  Uniforms: (seed, start vertex, ordinal, step, trial).
#include "n2v_common.h"

namespace n2v {

// The same test through the block-end index (include/n2v_hip.h, n2v_pivots_build).  The
// kernel is bound by the number of cache lines its random probes pull in (every probe of a
// plain binary search is another 128-byte line until the last few), so the search runs
// over pivots[j] = col[32 j + 31] of the aligned 32-entry blocks that END inside the row
// -- (degree / 32) ids, contiguous -- and then inside the one block that can hold x.
__device__ __forceinline__ bool member_pivoted_lane(const int32_t *col, const int32_t *pivots,
                                                    int64_t first, int m, int32_t x) {
  const int64_t last = first + m;  // row = col[first, last)
  int64_t lo = first >> 5, hi = (last - 1) >> 5;  // blocks lo .. hi-1 end inside the row
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (pivots[mid] < x)
      lo = mid + 1;
    else
      hi = mid;
  }
  int64_t a = max(first, lo << 5);
  const int64_t end = min(last, (lo << 5) + 32);
  int64_t b = end;
  while (a < b) {
    const int64_t mid = (a + b) >> 1;
    if (col[mid] < x)
      a = mid + 1;
    else
      b = mid;
  }
  return a < end && col[a] == x;
}

__global__ void pivots_build_kernel(const int32_t *__restrict__ col, int64_t n_edges,
                                    int32_t *__restrict__ pivots) {
  const int64_t n_blocks = (n_edges + 31) >> 5;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_blocks;
       j += (int64_t)gridDim.x * blockDim.x)
    pivots[j] = col[min(32 * j + 31, n_edges - 1)];
}

// position of the first entry >= x in the sorted row a[0, m) (one lane)
__device__ __forceinline__ int lower_bound_lane(const int32_t *a, int m, int32_t x) {
  int lo = 0, hi = m;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < x)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

// kUnit: every weight is 1.0 -- the first-order table of a row is uniform, so a candidate is
// col[row + int(r1 * n)] (4-byte gather, no slots needed, and the index of the edge walked is
// known).  With the per-edge class counts (g.edge_classes, n2v_edge_classes_build) two more
// things follow for the step (s -> v): the multiplicity nR of the return edge, which takes the
// return edge OUT of the rejection envelope.  One trial = with probability rho propose "return"
// (always accepted), otherwise propose one of ALL n entries uniformly, reject it if it is the
// return edge and accept any other x with beta(x) / b', b' = max(1, 1/q).  Per trial
// P(return) = rho and P(x) = (1 - rho) beta(x) / (n b'), so P(return) : P(x) = (nR/p) : beta(x)
// -- the reference's rule, randomwalk.py:219-231 -- needs rho / (1 - rho) = (nR/p) / (n b'):
// rho = (nR/p) / (nR/p + n b').  (Round 2 had (n - nR) b' there, which over-weights the return
// edge by n / (n - nR): the candidate is drawn over n entries, not n - nR.)  p no longer
// inflates the envelope: 3.5 -> ~2 trials per step at p = 0.5, q = 2.  And the number nM of neighbours of
// v that are neighbours of s: when it is 0 the test "x in N(s)" (:226) is known to fail and
// its binary search is skipped.
// kHops (unit weights): candidates come from the hop table (n2v_hops_build) -- the accepted
// entry already holds the row pointer and degree of the next vertex and the class counts of the
// edge, so an accepted step needs no further gather (one sector per trial + the membership test).
// kClassFirst (unit weights, class counts AND wedge table at hand): the table of a step has nR
// slots of weight 1/p, nM of weight 1 and nO of weight 1/q, and the tables say where the first
// two kinds are.  Sort the three weights, w1 <= w2 <= w3 of classes c1, c2, c3, and cut the table
// into LAYERS: every slot carries w1 (layer 1: all n slots), the slots of c2 and c3 carry
// w2 - w1 more (layer 2), those of c3 carry w3 - w2 more (layer 3).  A step draws the layer from
// the three exact masses and then a slot uniformly inside the layer's set:
//   * layer 1 is a plain uniform draw -- no rejection, no table access: at q >= 1 with p <= q
//     (1/q the smallest weight: p = 0.5, q = 2 of the BASELINE configs) that is nearly every
//     step, one gather like the p == q == 1 kernel;
//   * a set without "other" slots (return run and / or wedge list) is drawn by index;
//   * a set with "other" slots is a uniform draw over the row, repeated INSIDE the layer while it
//     hits a slot outside the set: the return run is recognised by x == s, only the exclusion of
//     SHARED slots (q < 1) needs the list.
// P(slot) = sum over the layers that contain it of (layer mass / total) / |set| = weight / total.
// Trials per step: 1.86 -> 1.00x at p = 0.5, q = 2.
template <bool kUnit, bool kHops, bool kClassFirst, bool kSlots = false>
__global__ __launch_bounds__(256, 6) void walk_fast_kernel(
    n2v_graph g, const int32_t *__restrict__ start_ids, int64_t n_start, int32_t num_walks,
    int32_t walk_length, double p, double q, uint64_t seed, int32_t *__restrict__ walks_out,
    uint8_t *__restrict__ valid_out, uint32_t *__restrict__ status,
    unsigned long long *__restrict__ trials_out) {
  const int64_t total = n_start * (int64_t)num_walks;
  const int64_t n_lanes = (int64_t)gridDim.x * blockDim.x;
  const int L1 = walk_length + 1;
  const double inv_p = 1.0 / p, inv_q = 1.0 / q;
  const double b_lo = fmin(1.0, inv_q), b_hi = fmax(1.0, inv_q);  // biases of the non-return entries
  const double beta_max = fmax(b_hi, inv_p);
  const bool biased = !(p == 1.0 && q == 1.0);
  const bool have_ec = kUnit && (kHops || g.edge_classes != nullptr);
  // wedge table (n2v_wedge_build): "x in N(s)" for the candidate at position `pick` of N(v) is
  // "pick is in the list of the edge (s -> v)": one offset gather + a search in a short list
  const bool have_w = have_ec && g.wedge_off != nullptr && g.wedge_pos != nullptr;
  // wedge slots (n2v_wedge_slots_build): return position and list of an edge with ONE gather
  // (its own code instance, launched at q < 1 only -- where every "other" draw is tested against the
  // list: there the slots win 9 - 10 %; at q >= 1 nearly every step is a plain uniform draw and the
  // eight registers of a slot cost 9 - 10 %, profiles/r4r_time_fast_slots_cfg4.log)
  // (a mixed wedge table: the steps of walkers standing on a wide row go through wedge_off)
  const bool slots_tab = kSlots && have_w && g.wedge_wide != 1 && g.wedge_slots != nullptr;
  // hop table with inline return positions (N2V_HOPS_INLINE_RPOS): an edge without shared
  // neighbours says where its return run starts in the class word itself
  const bool inl_tab = kHops && (g.reserved2 & N2V_HOPS_INLINE_RPOS) != 0;
  int rpos_inl = -1;
  int64_t e_prev = 0;  // the edge (s -> v) walked last
  const bool fold_return = have_ec && inv_p > b_hi;
  const bool base_aligned = (reinterpret_cast<uintptr_t>(walks_out) & 63u) == 0;

  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // current walker row
  bool live = false;
  int32_t s = -1, v = 0;
  int step = 0;
  uint32_t trial = 0;
  int64_t vb = 0, sb = 0, w0 = 0;
  int n = 0, m = 0;
  uint64_t h0 = 0, hstep = 0;
  unsigned long long trials = 0;
  double rho = -1.0;   // share of the return branch at this step (< 0: not folded)
  int shared = -1;     // neighbours of v inside N(s) at this step, -1 = unknown
  int n_ret = 0;       // kClassFirst: return slots of this step's table
  int cls = -1;        // kClassFirst: the set of the layer drawn for this step (bits: 1 return,
                       // 2 shared, 4 other), -1 = not drawn yet
  double m_l1 = 0.0, m_l12 = 0.0, m_tot = 0.0;  // kClassFirst: masses of layer 1, layers 1 + 2, all
  // the classes by ascending weight (ties: any order, a layer of mass 0 is never drawn)
  int c1 = 4, c3 = 1;  // bits of the lightest and of the heaviest class
  double w1 = 0.0, d2 = 0.0, d3 = 0.0;
  if (kClassFirst) {
    double wc[3] = {inv_p, 1.0, inv_q};
    int bc[3] = {1, 2, 4};
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2 - a; ++b)
        if (wc[b] > wc[b + 1]) {
          const double tw = wc[b]; wc[b] = wc[b + 1]; wc[b + 1] = tw;
          const int tb = bc[b]; bc[b] = bc[b + 1]; bc[b + 1] = tb;
        }
    c1 = bc[0];
    c3 = bc[2];
    w1 = wc[0];
    d2 = wc[1] - wc[0];
    d3 = wc[2] - wc[1];
  }

  // The path is not stored word by word: a 4-byte store into a 324-byte-pitch row costs a
  // 32-byte write request.  Each lane keeps the 16 words of the 64-byte sector of walks_out
  // it is filling in LDS (word k of lane t at [k][t]: conflict-free) and stores the sector
  // whole: four aligned 16-byte stores per 16 steps.
  __shared__ int32_t path_tile[16][256];
  const int tid = threadIdx.x;
  int lo = 0;  // first word of the current sector that belongs to this row
  auto put = [&](int64_t a, int32_t x) { path_tile[(int)(a & 15)][tid] = x; };
  auto flush = [&](int64_t a) {  // words [sector(a) + lo, a] are complete
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
    put(a, x);
    if ((a & 15) == 15 || pos == walk_length) flush(a);
  };

  // (re)load a walker into this lane; returns false when the lane has none left
  auto begin_walker = [&]() -> bool {
    while (r < total) {
      const int32_t start = start_ids[r / num_walks];
      const int32_t ordinal = (int32_t)(r % num_walks) + 1;
      w0 = r * (int64_t)L1;
      lo = (int)(w0 & 15);
      bool ok = true;
      if (start < 0 || (int64_t)start >= g.n_vertices) {
        atomicOr(status, N2V_ST_RANGE);
        ok = false;
      }
      if (ok) {
        vb = g.rowptr[start];
        n = (int)(g.rowptr[start + 1] - vb);
        ok = n > 0;  // fugue.py:132
      }
      if (!ok || walk_length == 0) {
        // no out-edges (fugue.py:132): the row is all -1; nothing to walk: the row is [start]
        emit(0, ok ? start : -1);
        for (int t = 1; t < L1; ++t) emit(t, -1);
        valid_out[r] = ok ? 1 : 0;
        r += n_lanes;
        continue;
      }
      s = -1;
      v = start;
      step = 0;
      trial = 0;
      rho = -1.0;
      shared = -1;
      cls = -1;
      emit(0, start);
      h0 = walker_stream(seed, (uint64_t)start * (uint64_t)num_walks + (uint64_t)(ordinal - 1));
      hstep = step_bits(h0, 0);
      return true;
    }
    return false;
  };

  live = begin_walker();
  while (__ballot(live) != 0ull) {
    if (!live) continue;
    // ---- one trial of the current step ----------------------------------------
    // unbiased steps (the first one; every one when p == q == 1) never reject and use the exact
    // mode's two uniforms: those draws are bit-identical to exact mode
    const bool plain = s < 0 || !biased;
    const uint64_t bits = plain ? hstep : trial_bits(hstep, trial);
    const uint32_t u1 = (uint32_t)(bits >> 32), u2 = (uint32_t)bits;
    int pick = pick_index(u1, n);
    const bool w_wide = wedge_row_wide(g.wedge_wide, n);  // width of the list of the edge walked last
    const bool have_slots = slots_tab && !w_wide;
    uint64_t wraw = 0;  // kClassFirst: wedge_off of the edge walked last
    int4 ws_a = make_int4(0, 0, 0, 0), ws_b = make_int4(0, 0, 0, 0);  // ... or its wedge slot
    if (kClassFirst && !plain) {
      if (cls < 0) {  // first trial of the step: the layer, then (sets without "other") the slot
        const double uc = (double)u2 * (1.0 / 4294967296.0) * m_tot;
        double ul = uc, wl = w1;
        cls = 7;
        if (!(uc < m_l1)) {
          if (uc < m_l12) {
            cls = 7 & ~c1;
            ul = uc - m_l1;
            wl = d2;
          } else {
            cls = c3;
            ul = uc - m_l12;
            wl = d3;
          }
        }
        if (!(cls & 4)) {  // by index: the return run, then the wedge list
          const int cnt_r = (cls & 1) ? n_ret : 0, cnt_m = (cls & 2) ? shared : 0;
          int kk = (int)(ul / wl);
          kk = kk < cnt_r + cnt_m ? kk : cnt_r + cnt_m - 1;
          kk = kk < 0 ? 0 : kk;
          if (rpos_inl >= 0 && cnt_m == 0) {
            pick = rpos_inl + kk;  // the return run of an edge without shared neighbours: no gather
          } else if (have_slots) {
            const int4 *slot = reinterpret_cast<const int4 *>(g.wedge_slots + e_prev * 16);
            ws_a = slot[0];
            ws_b = slot[1];
            pick = kk < cnt_r ? (int)((uint32_t)ws_a.x & 0xffffu) + kk
                              : slot_entry(ws_a, ws_b, shared, kk - cnt_r,
                                           reinterpret_cast<const uint16_t *>(g.wedge_pos));
          } else {
            wraw = g.wedge_off[e_prev];
            if (kk < cnt_r) {
              pick = (int)(wraw >> N2V_WEDGE_RPOS_SHIFT) + kk;
            } else {
              const int64_t off = (int64_t)(wraw & N2V_WEDGE_OFF_MASK) + (kk - cnt_r);
              pick = w_wide ? (int)reinterpret_cast<const uint32_t *>(g.wedge_pos)[off]
                            : (int)reinterpret_cast<const uint16_t *>(g.wedge_pos)[off];
            }
          }
        }
      }
      // a uniform draw that must stay clear of the shared slots: the list of the edge
      if ((cls & 4) && !(cls & 2) && shared > 0) {
        if (have_slots) {
          const int4 *slot = reinterpret_cast<const int4 *>(g.wedge_slots + e_prev * 16);
          ws_a = slot[0];
          ws_b = slot[1];
        } else {
          wraw = g.wedge_off[e_prev];
        }
      }
    }
    int32_t x;
    int64_t e = vb + pick;  // kUnit: the edge (v -> x) itself
    n2v_hop h;
    h.col = -1;
    h.classes = 0xffffffffu;
    h.row = 0;
    if (kHops) {
      h = load_hop(g.hops + e);
      x = h.col;
    } else if (kUnit) {
      x = g.col[e];
    } else {
      const n2v_slot sl = g.slots[e];
      const double r2 = (double)u2 * (1.0 / 4294967296.0);
      x = (r2 < sl.prob) ? sl.col : sl.alias;  // slot.alias is already a vertex id
    }
    bool accept = true;
    ++trials;
    if (kClassFirst && !plain) {
      if (cls & 4) {  // drawn over the whole row: outside the layer's set -> again, inside the layer
        if (!(cls & 1)) accept = x != s;
        if (accept && !(cls & 2) && shared > 0) {
          if (have_slots) {
            bool found = false;
            slot_lower(ws_a, ws_b, shared, pick, reinterpret_cast<const uint16_t *>(g.wedge_pos), found);
            accept = !found;
          } else {
            accept = !wedge_has(g.wedge_pos, (int64_t)(wraw & N2V_WEDGE_OFF_MASK), shared, pick, w_wide);
          }
        }
      }
    } else if (!plain) {
      const uint64_t b2 = mix64(bits ^ 0xC2B2AE3D27D4EB4FULL);
      const double ua = (double)(uint32_t)(b2 >> 32) * (1.0 / 4294967296.0);
      // accept a non-return candidate x iff u < beta(x), beta = 1 (x in N(s)) or 1/q: the
      // binary search over N(s) is only needed when u falls BETWEEN those two values
      auto accept_other = [&](double u) -> bool {
        if (u < b_lo) return true;
        if (!(u < b_hi)) return false;
        bool member = false;
        if (q != 1.0 && shared != 0) {
          if (have_w && shared > 0)
            member = wedge_has(g.wedge_pos, (int64_t)(g.wedge_off[e_prev] & N2V_WEDGE_OFF_MASK), shared,
                               pick, w_wide);
          else
            member = g.pivots ? member_pivoted_lane(g.col, g.pivots, sb, m, x)
                              : member_sorted_lane(g.col + sb, m, x);
        }
        return u < (member ? 1.0 : inv_q);
      };
      if (rho >= 0.0) {  // return edge outside the envelope
        const double ub = (double)(uint32_t)b2 * (1.0 / 4294967296.0);
        if (ub < rho) {
          x = s;
          e = vb + lower_bound_lane(g.col + vb, n, s);  // the edge (v -> s): its counts are needed next
          if (kHops) h = load_hop(g.hops + e);
        } else if (x == s) {
          accept = false;  // drawn among the OTHER entries: try again
        } else {
          accept = accept_other(ua * b_hi);
        }
      } else {
        const double u = ua * beta_max;
        accept = (x == s) ? (u < inv_p) : accept_other(u);
      }
    }
    if (!accept) {
      ++trial;
      continue;
    }
    // ---- accepted: append, advance ------------------------------------------------
    emit(step + 1, x);
    e_prev = e;
    s = v;
    sb = vb;
    m = n;
    v = x;
    ++step;
    trial = 0;
    cls = -1;
    const bool finished = step == walk_length;
    bool dropped = false;
    if (!finished) {
      if (kHops) {
        vb = hop_row(h);
        n = hop_deg(h);
      } else {
        vb = g.rowptr[v];
        n = (int)(g.rowptr[v + 1] - vb);
      }
      dropped = n == 0;  // fugue.py:147: the walker vanishes at a sink
      hstep = step_bits(h0, (uint32_t)step);
      rho = -1.0;
      shared = -1;
      if (kClassFirst) {
        // (tables that come with a wedge list hold no saturated count: n2v_wedge_build)
        uint32_t ec = kHops ? h.classes : g.edge_classes[e];
        rpos_inl = -1;
        if (inl_tab && ec != 0xffffffffu && (ec & N2V_EC_INLINE) != 0u) {
          rpos_inl = (int)(ec & N2V_EC_SHARED_MASK);
          ec = ((ec >> N2V_EC_RETURN_SHIFT) & 0x7fu) << N2V_EC_RETURN_SHIFT;  // no shared neighbours
        }
        n_ret = (int)(ec >> N2V_EC_RETURN_SHIFT);
        shared = q != 1.0 ? (int)(ec & N2V_EC_SHARED_MASK) : 0;  // q == 1: shared slots ARE other slots
        const int n_oth = n - n_ret - shared;
        const int n_c1 = c1 == 1 ? n_ret : (c1 == 2 ? shared : n_oth);
        const int n_c3 = c3 == 1 ? n_ret : (c3 == 2 ? shared : n_oth);
        m_l1 = (double)n * w1;
        m_l12 = m_l1 + (double)(n - n_c1) * d2;
        m_tot = m_l12 + (double)n_c3 * d3;
      } else if (have_ec && biased && !dropped) {
        uint32_t ec = kHops ? h.classes : g.edge_classes[e];
        if (inl_tab && ec != 0xffffffffu && (ec & N2V_EC_INLINE) != 0u)
          ec = ((ec >> N2V_EC_RETURN_SHIFT) & 0x7fu) << N2V_EC_RETURN_SHIFT;  // the plain form
        const uint32_t fR = ec >> N2V_EC_RETURN_SHIFT, fM = ec & N2V_EC_SHARED_MASK;
        if (fM != N2V_EC_SHARED_MASK) shared = (int)fM;
        if (fold_return && fR != N2V_EC_RETURN_SAT) {
          const double wr = (double)fR * inv_p;
          rho = wr / (wr + (double)n * b_hi);  // candidates are drawn over all n entries
        }
      }
    }
    if (finished || dropped) {
      if (dropped)
        for (int t = step + 1; t < L1; ++t) emit(t, -1);
      valid_out[r] = finished ? 1 : 0;
      r += n_lanes;
      live = begin_walker();
    }
  }
  if (trials_out && trials) atomicAdd(trials_out, trials);
}

}  // namespace n2v

extern "C" int n2v_walk_fast_launch(const n2v_graph *g, const int32_t *start_ids,
                                    int64_t n_start, int32_t num_walks, int32_t walk_length,
                                    double p, double q, uint64_t seed, int32_t *walks_out,
                                    uint8_t *valid_out, uint32_t *status, void *stream) {
  const int64_t total = n_start * (int64_t)num_walks;
  if (total == 0) return N2V_OK;
  const bool unit = g->w == nullptr && g->w64 == nullptr;
  if (!unit && !g->slots) return N2V_EINVAL;
  const int threads = 256;
  int64_t blocks = (total + threads - 1) / threads;
  const bool hops = unit && g->hops != nullptr;
  // class-first sampling: counts, return position and shared positions of every edge at hand
  const bool cf = unit && (hops || g->edge_classes) && g->wedge_off && g->wedge_pos &&
                  !(p == 1.0 && q == 1.0);
  const bool slots = cf && hops && q < 1.0 && g->wedge_slots && g->wedge_wide != 1;
  const void *fn = slots  ? (const void *)n2v::walk_fast_kernel<true, true, true, true>
                   : cf   ? (hops ? (const void *)n2v::walk_fast_kernel<true, true, true>
                                  : (const void *)n2v::walk_fast_kernel<true, false, true>)
                   : hops ? (const void *)n2v::walk_fast_kernel<true, true, false>
                   : unit ? (const void *)n2v::walk_fast_kernel<true, false, false>
                          : (const void *)n2v::walk_fast_kernel<false, false, false>;
  const int64_t cap = n2v::resident_blocks(fn, threads, 0);
  if (blocks > cap) blocks = cap;
  // status[2..3]: 64-bit trial counter (include/n2v_hip.h)
  unsigned long long *trials = reinterpret_cast<unsigned long long *>(status + 2);
#define N2V_FAST_LAUNCH(U, H, C)                                                                  \
  hipLaunchKernelGGL((n2v::walk_fast_kernel<U, H, C>), dim3((unsigned)blocks), dim3(threads), 0, \
                     (hipStream_t)stream, *g, start_ids, n_start, num_walks, walk_length, p, q,  \
                     seed, walks_out, valid_out, status, trials)
  if (slots)
    hipLaunchKernelGGL((n2v::walk_fast_kernel<true, true, true, true>), dim3((unsigned)blocks),
                       dim3(threads), 0, (hipStream_t)stream, *g, start_ids, n_start, num_walks,
                       walk_length, p, q, seed, walks_out, valid_out, status, trials);
  else if (cf && hops)
    N2V_FAST_LAUNCH(true, true, true);
  else if (cf)
    N2V_FAST_LAUNCH(true, false, true);
  else if (hops)
    N2V_FAST_LAUNCH(true, true, false);
  else if (unit)
    N2V_FAST_LAUNCH(true, false, false);
  else
    N2V_FAST_LAUNCH(false, false, false);
#undef N2V_FAST_LAUNCH
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}

extern "C" int n2v_pivots_build(const int32_t *col, int64_t n_edges, int32_t *pivots_out,
                                void *stream) {
  if (n_edges < 0 || (n_edges > 0 && (!col || !pivots_out))) return N2V_EINVAL;
  if (n_edges == 0) return N2V_OK;
  const int64_t n_blocks = (n_edges + 31) >> 5;
  int64_t blocks = (n_blocks + 255) / 256;
  const int64_t cap = n2v::resident_blocks((const void *)n2v::pivots_build_kernel, 256, 0) * 2;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(n2v::pivots_build_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, col, n_edges, pivots_out);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}
