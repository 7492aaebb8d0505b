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

// Index drawn by sampling_from_alias(r1, r2) on the table that
// generate_edge_alias_tables would build.  Returns -1 on ZeroDivisionError.
__device__ __forceinline__ int exact_draw(const StepCtx &c, uint32_t u1, uint32_t u2,
                                          int lane, uint64_t *cls) {
  const int n = c.n;
  const int pick = (int)__umulhi(u1, (uint32_t)n);  // int(r1 * n), r1 = u1 / 2^32
  const double r2 = (double)u2 * (1.0 / 4294967296.0);

  // ---- pass 1: bias + sum ---------------------------------------------------
  double b_pick;
  const double total = row_sum(c, lane, cls, pick, b_pick);
  const double avg = total / (double)n;  // :172
  if (avg == 0.0) return -1;
  const double p_pick = b_pick / avg;    // :173

  // untouched underfull slot: probs[pick] never changes, alias irrelevant
  if (p_pick < 1.0 && r2 < p_pick) return pick;

  // ---- pass 2: LIFO pairing (:182-189) until slot `pick` is final ------------
  int cu = c.nch, co = c.nch;
  uint64_t um = 0, om = 0;
  double uval = 0.0, oval = 0.0;
  bool have_dem = false, have_o = false;
  double dem_r = 0.0, r = 0.0;
  int dem_idx = 0, o_idx = 0;
  double fin_prob = p_pick;
  int fin_alias = 0;
  for (;;) {
    double pu;
    int ui;
    if (have_dem) {  // the just-demoted overfull is the top of `underfull`
      pu = dem_r;
      ui = dem_idx;
      have_dem = false;
    } else {
      while (um == 0ull && cu > 0) {
        --cu;
        bool valid;
        uval = chunk_bias<true>(c, cu, lane, cls, valid) / avg;
        um = ballot64(valid && uval < 1.0);
      }
      if (um == 0ull) {  // underfull empty
        if (have_o && o_idx == pick) fin_prob = r;
        break;
      }
      int l = 63 - __clzll((long long)um);
      um &= ~(1ull << l);
      pu = readlane_f64(uval, l);
      ui = cu * 64 + l;
    }
    if (!have_o) {
      while (om == 0ull && co > 0) {
        --co;
        bool valid;
        oval = chunk_bias<true>(c, co, lane, cls, valid) / avg;
        om = ballot64(valid && !(oval < 1.0));
      }
      if (om == 0ull) {  // overfull empty: `under` stays where it was
        if (ui == pick) fin_prob = pu;
        break;
      }
      int l = 63 - __clzll((long long)om);
      om &= ~(1ull << l);
      r = readlane_f64(oval, l);
      o_idx = co * 64 + l;
      have_o = true;
    }
    if (ui == pick) {  // alias[under] = over; probs[under] is final
      fin_prob = pu;
      fin_alias = o_idx;
      break;
    }
    r = r + pu - 1.0;  // probs[over] = probs[over] + probs[under] - 1.0
    if (r < 1.0) {
      have_dem = true;
      dem_r = r;
      dem_idx = o_idx;
      have_o = false;
    }
  }
  return (r2 < fin_prob) ? pick : fin_alias;  // :95-99
}

__global__ __launch_bounds__(kWavesPerBlock * 64) void walk_exact_kernel(
    n2v_graph g, const int32_t *__restrict__ start_ids, int64_t n_start, int32_t num_walks,
    int32_t walk_length, double p, double q, uint64_t seed, int32_t *__restrict__ walks_out,
    uint8_t *__restrict__ valid_out, uint32_t *__restrict__ status) {
  __shared__ uint64_t cls_all[kWavesPerBlock][2 * kLdsChunks];
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  uint64_t *cls = cls_all[wave_in_block];
  const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t total = n_start * (int64_t)num_walks;
  const int L1 = walk_length + 1;

  StepCtx c;
  c.p = p;
  c.q = q;

  for (int64_t rr = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block; rr < total;
       rr += n_waves) {
    const int64_t r = readfirstlane_i64(rr);
    int32_t *out = walks_out + r * L1;
    for (int t = lane; t < L1; t += 64) out[t] = -1;
    const int32_t start = __builtin_amdgcn_readfirstlane(start_ids[r / num_walks]);
    const int32_t ordinal = (int32_t)(r % num_walks) + 1;  // randomwalk.py:294
    bool alive = true;
    if (start < 0 || (int64_t)start >= g.n_vertices) {
      if (lane == 0) atomicOr(status, N2V_ST_RANGE);
      alive = false;
    }
    int32_t s = -1, v = start;
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
      if (lane == 0) out[0] = start;  // path[0] after the first-step rule (:146-147)
      for (int step = 0; step < walk_length; ++step) {
        const int64_t vb = readfirstlane_i64(g.rowptr[v]);
        const int64_t ve = readfirstlane_i64(g.rowptr[v + 1]);
        const int n = (int)(ve - vb);
        if (n == 0) {  // fugue.py:147 inner join on dst: the walker vanishes
          alive = false;
          break;
        }
        c.vcol = g.col + vb;
        c.vw = g.w + vb;
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
        const uint64_t bits = step_bits(h0, (uint32_t)step);
        const int idx = exact_draw(c, (uint32_t)(bits >> 32), (uint32_t)bits, lane, cls);
        if (idx < 0) {
          if (lane == 0) atomicOr(status, N2V_ST_ZERODIV);
          alive = false;
          break;
        }
        const int32_t next = __builtin_amdgcn_readfirstlane(c.vcol[idx]);
        if (lane == 0) out[step + 1] = next;
        s = v;  // :339 src = path[-2], dst = path[-1]
        v = next;
      }
    }
    if (lane == 0) valid_out[r] = alive ? 1 : 0;
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
  const int64_t cap = 256 * 8;  // 256 CUs x 8 resident 4-wave blocks
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(n2v::walk_exact_kernel, dim3((unsigned)blocks),
                     dim3(n2v::kWavesPerBlock * 64), 0, (hipStream_t)stream, *g, start_ids,
                     n_start, num_walks, walk_length, p, q, seed, walks_out, valid_out, status);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}
