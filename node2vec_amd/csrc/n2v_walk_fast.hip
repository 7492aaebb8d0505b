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

__global__ __launch_bounds__(256) void walk_fast_kernel(
    n2v_graph g, const int32_t *__restrict__ start_ids, int64_t n_start, int32_t num_walks,
    int32_t walk_length, double p, double q, uint64_t seed, int32_t *__restrict__ walks_out,
    uint8_t *__restrict__ valid_out, uint32_t *__restrict__ status,
    unsigned long long *__restrict__ trials_out) {
  const int64_t total = n_start * (int64_t)num_walks;
  const int64_t n_lanes = (int64_t)gridDim.x * blockDim.x;
  const int L1 = walk_length + 1;
  const double inv_p = 1.0 / p, inv_q = 1.0 / q;
  const double beta_max = fmax(1.0, fmax(inv_p, inv_q));
  const bool biased = !(p == 1.0 && q == 1.0);

  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // current walker row
  bool live = false;
  int32_t s = -1, v = 0;
  int step = 0;
  uint32_t trial = 0;
  int64_t vb = 0, sb = 0;
  int n = 0, m = 0;
  uint64_t h0 = 0, hstep = 0;
  int32_t *out = nullptr;
  unsigned long long trials = 0;

  // (re)load a walker into this lane; returns false when the lane has none left
  auto begin_walker = [&]() -> bool {
    while (r < total) {
      const int32_t start = start_ids[r / num_walks];
      const int32_t ordinal = (int32_t)(r % num_walks) + 1;
      out = walks_out + r * L1;
      for (int t = 0; t < L1; ++t) out[t] = -1;
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
        // no out-edges (fugue.py:132), or nothing to walk: the row is just [start]
        if (ok) out[0] = start;
        valid_out[r] = ok ? 1 : 0;
        r += n_lanes;
        continue;
      }
      s = -1;
      v = start;
      step = 0;
      trial = 0;
      out[0] = start;
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
    uint64_t bits = (s < 0 || !biased) ? hstep : trial_bits(hstep, trial);
    const uint32_t u1 = (uint32_t)(bits >> 32), u2 = (uint32_t)bits;
    const int pick = pick_index(u1, n);
    const n2v_slot sl = g.slots[vb + pick];
    const double r2 = (double)u2 * (1.0 / 4294967296.0);
    const int32_t x = (r2 < sl.prob) ? sl.col : sl.alias;  // slot.alias is already a vertex id
    bool accept = true;
    ++trials;
    if (s >= 0 && biased) {
      // accept iff u < beta(x), u uniform on [0, beta_max).  beta is 1/p for the return
      // edge; otherwise it is 1 (x in N(s)) or 1/q, so the binary search over N(s) is
      // only needed when u falls BETWEEN those two values: same decisions, far fewer
      // dependent gathers (p=0.5, q=2: one trial in four).
      const uint32_t u3 = (uint32_t)(mix64(bits ^ 0xC2B2AE3D27D4EB4FULL) >> 32);
      const double u = (double)u3 * (1.0 / 4294967296.0) * beta_max;
      if (x == s) {
        accept = u < inv_p;
      } else if (q == 1.0) {
        accept = u < 1.0;
      } else {
        const double b_lo = fmin(1.0, inv_q), b_hi = fmax(1.0, inv_q);
        if (u < b_lo)
          accept = true;
        else if (!(u < b_hi))
          accept = false;
        else
          accept = u < ((g.pivots ? member_pivoted_lane(g.col, g.pivots, sb, m, x)
                                 : member_sorted_lane(g.col + sb, m, x))
                            ? 1.0
                            : inv_q);
      }
    }
    if (!accept) {
      ++trial;
      continue;
    }
    // ---- accepted: append, advance ------------------------------------------------
    out[step + 1] = x;
    s = v;
    sb = vb;
    m = n;
    v = x;
    ++step;
    trial = 0;
    bool finished = step == walk_length;
    bool dropped = false;
    if (!finished) {
      vb = g.rowptr[v];
      n = (int)(g.rowptr[v + 1] - vb);
      dropped = n == 0;  // fugue.py:147: the walker vanishes at a sink
      hstep = step_bits(h0, (uint32_t)step);
    }
    if (finished || dropped) {
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
  const int threads = 256;
  int64_t blocks = (total + threads - 1) / threads;
  if (blocks > 256 * 8) blocks = 256 * 8;
  // status[2..3]: 64-bit trial counter (include/n2v_hip.h)
  hipLaunchKernelGGL(n2v::walk_fast_kernel, dim3((unsigned)blocks), dim3(threads), 0,
                     (hipStream_t)stream, *g, start_ids, n_start, num_walks, walk_length, p, q,
                     seed, walks_out, valid_out, status,
                     reinterpret_cast<unsigned long long *>(status + 2));
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}

extern "C" int n2v_pivots_build(const int32_t *col, int64_t n_edges, int32_t *pivots_out,
                                void *stream) {
  if (n_edges < 0 || (n_edges > 0 && (!col || !pivots_out))) return N2V_EINVAL;
  if (n_edges == 0) return N2V_OK;
  const int64_t n_blocks = (n_edges + 31) >> 5;
  int64_t blocks = (n_blocks + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(n2v::pivots_build_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, col, n_edges, pivots_out);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}
