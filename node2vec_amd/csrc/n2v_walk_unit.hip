// n2v_walk_unit.hip -- K2 exact mode, specialised for UNIT edge weights (gfx950).
//
// Same contract and same bits as walk_exact_kernel (n2v_walk.hip): per step the
// index sampling_from_alias(r1, r2) returns on the table generate_edge_alias_tables
// builds (reference randomwalk.py:86-99, :157-232).  With every weight 1.0 the
// biased weight of a neighbour takes one of three values (:223-230):
//     return  1.0 / p     shared  1.0     other  1.0 / q
// and when those scale to exact integers (x * 2^20, true for the dyadic p, q of
// every BASELINE config) the whole step is integer / scalar work:
//   * the row sum of :172 is three popcounts times three constants (every partial
//     sum is exactly representable, so order does not matter);
//   * the LDS cache is the two class ballots per 64-neighbour chunk, w is never read;
//   * probs0 has three values (3 fp64 divisions per step), so the underfull /
//     overfull candidate masks of the pairing (:175-189) are scalar AND/OR of the
//     ballots, and a candidate's value is two bit tests;
//   * a uniform row (first step, or p == q == 1) has probs0 == 1.0 everywhere, no
//     underfull slot, and the draw is `pick` itself: O(1).
// Membership "x in N(s)" (:226) uses the LDS hashed-id filter + batched exact
// verification of the hits, as in the generic kernel.
#include "n2v_alias_core.h"

namespace n2v {

constexpr int kUC = 256;  // class ballots cached for the TOP 256 chunks (16384 neighbours)

struct UnitLds {
  uint64_t cls[2 * kUC];        // slot nch-1-chunk: ballot(return), ballot(shared)
  uint32_t bits[kBitWordsMax];  // hashed-id filter of N(s)
  int32_t mlist[kMaybeCap];     // filter hits waiting for verification
};

struct UnitConsts {
  double bR, bM, bO;     // 1/p, 1, 1/q
  int64_t TR, TM, TO;    // the same times 2^20 (exact integers)
};

struct UnitStep {
  const int32_t *vcol, *scol;
  int n, nch, m, iters;
  int32_t s;
  bool need_mem;
};

__device__ __forceinline__ uint64_t valid_mask(const UnitStep &c, int chunk) {
  const int rem = c.n - chunk * 64;
  return rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
}

// class ballots of one chunk: from LDS when cached, else by searching again
__device__ __forceinline__ void chunk_classes(const UnitStep &c, UnitLds &L, int chunk, int lane,
                                              uint64_t &rm, uint64_t &mm) {
  const int ci = c.nch - 1 - chunk;
  if (ci < kUC) {
    rm = L.cls[2 * ci];
    mm = L.cls[2 * ci + 1];
    return;
  }
  const int i = chunk * 64 + lane;
  const bool valid = i < c.n;
  const int32_t x = valid ? c.vcol[i] : -1;
  const bool is_ret = valid && x == c.s;
  bool is_mem = false;
  if (c.need_mem) is_mem = member_sorted(c.scol, c.m, x, c.iters) && valid && !is_ret;
  rm = ballot64(is_ret);
  mm = ballot64(is_mem);
}

__device__ __forceinline__ void verify_unit(const UnitStep &c, UnitLds &L, int count, int lane,
                                            int &nM) {
  for (int k = 0; k < count; k += 64) {
    const bool act = k + lane < count;
    const int i = act ? L.mlist[k + lane] : 0;
    const int32_t x = act ? c.vcol[i] : -1;
    const bool mem = member_sorted(c.scol, c.m, x, c.iters) && act;
    const int ci = c.nch - 1 - (i >> 6);
    if (mem && ci < kUC)
      atomicOr(reinterpret_cast<unsigned long long *>(&L.cls[2 * ci + 1]), 1ull << (i & 63));
    nM += __popcll(ballot64(mem));
  }
}

__device__ __forceinline__ int unit_draw(const UnitStep &c, const UnitConsts &K, uint32_t u1,
                                         uint32_t u2, int lane, UnitLds &L) {
  const int n = c.n;
  const int pick = (int)__umulhi(u1, (uint32_t)n);  // int(r1 * n)
  const double r2 = (double)u2 * (1.0 / 4294967296.0);

  // ---- pass 0: hashed-id filter of N(s) ----------------------------------------
  const bool use_filter = c.need_mem && c.m <= 8192 && c.m <= 8 * n + 64;
  int shift = 32;
  if (use_filter) {
    int words = 64;
    while (words < kBitWordsMax && words * 32 < 16 * c.m) words <<= 1;
    shift = 32 - (5 + (31 - __clz(words)));
    for (int wv = lane; wv < words; wv += 64) L.bits[wv] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int yb = 0; yb < c.m; yb += 256) {
      int32_t y[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = yb + u * 64 + lane;
        y[u] = j < c.m ? c.scol[j] : -1;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (yb + u * 64 + lane < c.m) {
          const uint32_t h = hash_id(y[u], shift);
          atomicOr(&L.bits[h >> 5], 1u << (h & 31));
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  // ---- pass 1: stream N(v) ids, classify, count ---------------------------------
  int nR = 0, nM = 0, mcount = 0;
  for (int chunk0 = 0; chunk0 < c.nch; chunk0 += 4) {
    int32_t xs[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = (chunk0 + u) * 64 + lane;
      xs[u] = i < n ? c.vcol[i] : -1;
    }
    bool memv[4] = {false, false, false, false};
    if (c.need_mem && !use_filter) member_sorted_x4(c.scol, c.m, xs, c.iters, memv);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int chunk = chunk0 + u;
      if (chunk >= c.nch) break;  // wave-uniform
      const int i = chunk * 64 + lane;
      const bool valid = i < n;
      const bool is_ret = valid && xs[u] == c.s;
      bool is_mem = false, maybe = false;
      if (use_filter) {
        const uint32_t h = hash_id(xs[u], shift);
        maybe = valid && !is_ret && ((L.bits[h >> 5] >> (h & 31)) & 1u);
      } else {
        is_mem = memv[u] && valid && !is_ret;
      }
      const uint64_t rm = ballot64(is_ret), mm = ballot64(is_mem);
      nR += __popcll(rm);
      nM += __popcll(mm);
      const int ci = c.nch - 1 - chunk;
      if (ci < kUC && lane == 0) {
        L.cls[2 * ci] = rm;
        L.cls[2 * ci + 1] = mm;
      }
      const uint64_t ym = ballot64(maybe);
      if (ym) {
        const int cnt = __popcll(ym);
        if (mcount + cnt > kMaybeCap) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          verify_unit(c, L, mcount, lane, nM);
          mcount = 0;
        }
        if (maybe) L.mlist[mcount + __popcll(ym & ((1ull << lane) - 1ull))] = i;
        mcount += cnt;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (mcount) verify_unit(c, L, mcount, lane, nM);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // ---- :172-173 on three values -----------------------------------------------------
  const int nO = n - nR - nM;
  const int64_t isum = (int64_t)nR * K.TR + (int64_t)nM * K.TM + (int64_t)nO * K.TO;
  const double avg = ((double)isum * (1.0 / 1048576.0)) / (double)n;
  const double vR = K.bR / avg, vM = K.bM / avg, vO = K.bO / avg;
  uint64_t prm, pmm;
  chunk_classes(c, L, pick >> 6, lane, prm, pmm);
  const bool pR = (prm >> (pick & 63)) & 1ull, pM = (pmm >> (pick & 63)) & 1ull;
  const double p_pick = pR ? vR : (pM ? vM : vO);
  if (p_pick < 1.0 && r2 < p_pick) return pick;  // untouched underfull slot
  const bool uR = vR < 1.0, uM = vM < 1.0, uO = vO < 1.0;
  const bool any_under = (nR && uR) || (nM && uM) || (nO && uO);
  const bool any_over = (nR && !uR) || (nM && !uM) || (nO && !uO);
  if (!any_under || !any_over) return (r2 < p_pick) ? pick : 0;  // the loop of :182 never runs

  // ---- pairing (:182-189), candidate masks are scalar --------------------------------
  int cu = c.nch, co = c.nch;
  uint64_t um = 0, om = 0, urm = 0, umm = 0, orm = 0, omm = 0;
  bool carry = false;
  double carry_r = 0.0;
  int carry_idx = 0;
  double fin_prob = p_pick;
  int fin_alias = 0;
  auto under_mask = [&](uint64_t rm, uint64_t mm, uint64_t vm) -> uint64_t {
    return (uR ? rm : 0ull) | (uM ? mm : 0ull) | (uO ? (vm & ~(rm | mm)) : 0ull);
  };
  for (;;) {
    while (om == 0ull && co > 0) {  // next overfull candidates
      --co;
      chunk_classes(c, L, co, lane, orm, omm);
      const uint64_t vm = valid_mask(c, co);
      om = vm & ~under_mask(orm, omm, vm);
    }
    if (om == 0ull) {
      if (carry && carry_idx == pick) fin_prob = carry_r;
      break;
    }
    const int lo = 63 - __clzll((long long)om);
    om ^= 1ull << lo;
    double r = ((orm >> lo) & 1ull) ? vR : (((omm >> lo) & 1ull) ? vM : vO);
    const int o_idx = co * 64 + lo;
    if (carry) {
      if (carry_idx == pick) {
        fin_prob = carry_r;
        fin_alias = o_idx;
        break;
      }
      r = readfirstlane_f64(r + carry_r - 1.0);
      carry = false;
      if (r < 1.0) {
        carry = true;
        carry_r = r;
        carry_idx = o_idx;
        continue;
      }
    }
    bool finished = false;
    for (;;) {
      while (um == 0ull && cu > 0) {
        --cu;
        chunk_classes(c, L, cu, lane, urm, umm);
        um = under_mask(urm, umm, valid_mask(c, cu));
      }
      if (um == 0ull) {
        if (o_idx == pick) fin_prob = r;
        finished = true;
        break;
      }
      const int l = 63 - __clzll((long long)um);
      um ^= 1ull << l;
      const double pu = ((urm >> l) & 1ull) ? vR : (((umm >> l) & 1ull) ? vM : vO);
      if (cu * 64 + l == pick) {
        fin_prob = pu;
        fin_alias = o_idx;
        finished = true;
        break;
      }
      r = readfirstlane_f64(r + pu - 1.0);  // probs[over] = probs[over] + probs[under] - 1.0
      if (r < 1.0) {
        carry = true;
        carry_r = r;
        carry_idx = o_idx;
        break;
      }
    }
    if (finished) break;
  }
  return (r2 < fin_prob) ? pick : fin_alias;
}

__global__ __launch_bounds__(kWavesPerBlock * 64) void walk_exact_unit_kernel(
    n2v_graph g, const int32_t *__restrict__ start_ids, int64_t n_start, int32_t num_walks,
    int32_t walk_length, double p, double q, UnitConsts K, uint64_t seed,
    int32_t *__restrict__ walks_out, uint8_t *__restrict__ valid_out,
    uint32_t *__restrict__ status) {
  __shared__ UnitLds lds_all[kWavesPerBlock];
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  UnitLds &L = lds_all[wave_in_block];
  const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t total = n_start * (int64_t)num_walks;
  const int L1 = walk_length + 1;
  const bool biased = !(p == 1.0 && q == 1.0);
  UnitStep c;
  c.need_mem = q != 1.0;

  for (int64_t rr = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block; rr < total;
       rr += n_waves) {
    const int64_t r = readfirstlane_i64(rr);
    int32_t *out = walks_out + r * L1;
    for (int t = lane; t < L1; t += 64) out[t] = -1;
    const int32_t start = __builtin_amdgcn_readfirstlane(start_ids[r / num_walks]);
    const int32_t ordinal = (int32_t)(r % num_walks) + 1;
    bool alive = true;
    if (start < 0 || (int64_t)start >= g.n_vertices) {
      if (lane == 0) atomicOr(status, N2V_ST_RANGE);
      alive = false;
    }
    int32_t s = -1, v = start;
    if (alive) {
      const int64_t vb = readfirstlane_i64(g.rowptr[v]);
      const int64_t ve = readfirstlane_i64(g.rowptr[v + 1]);
      alive = ve > vb;  // fugue.py:132
    }
    if (alive) {
      const uint64_t key = (uint64_t)start * (uint64_t)num_walks + (uint64_t)(ordinal - 1);
      const uint64_t h0 = walker_stream(seed, key);
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) out[0] = start;
      int64_t sb = 0;
      int m = 0;
      for (int step = 0; step < walk_length; ++step) {
        const int64_t vb = readfirstlane_i64(g.rowptr[v]);
        const int64_t ve = readfirstlane_i64(g.rowptr[v + 1]);
        const int n = (int)(ve - vb);
        if (n == 0) {  // fugue.py:147: the walker vanishes at a sink
          alive = false;
          break;
        }
        const uint64_t bits = step_bits(h0, (uint32_t)step);
        int idx;
        if (s < 0 || !biased) {
          // uniform row: probs0 == 1.0 everywhere, no underfull slot, alias unused
          idx = (int)__umulhi((uint32_t)(bits >> 32), (uint32_t)n);
        } else {
          c.vcol = g.col + vb;
          c.n = n;
          c.nch = (n + 63) >> 6;
          c.s = s;
          c.scol = g.col + sb;
          c.m = m;
          c.iters = 32 - __clz(m);
          idx = unit_draw(c, K, (uint32_t)(bits >> 32), (uint32_t)bits, lane, L);
        }
        const int32_t next = __builtin_amdgcn_readfirstlane(g.col[vb + idx]);
        if (lane == 0) out[step + 1] = next;
        s = v;  // the row of the new previous vertex is the row just walked
        sb = vb;
        m = n;
        v = next;
      }
    }
    if (lane == 0) valid_out[r] = alive ? 1 : 0;
  }
}

}  // namespace n2v

// true when x * 2^20 is an exact integer in [0, 2^31): the integer-sum argument holds
static bool scales_exactly(double x, int64_t *t_out) {
  const double t = x * 1048576.0;
  if (!(t >= 0.0) || !(t < 2147483648.0) || t != (double)(int64_t)t) return false;
  *t_out = (int64_t)t;
  return true;
}

// returns 1 when the unit-weight kernel applies (and was launched), 0 when the caller
// must use the generic kernel, < 0 on error
extern "C" int n2v_walk_exact_unit_try(const n2v_graph *g, const int32_t *start_ids,
                                       int64_t n_start, int32_t num_walks,
                                       int32_t walk_length, double p, double q, uint64_t seed,
                                       int32_t *walks_out, uint8_t *valid_out, uint32_t *status,
                                       void *stream) {
  if (g->w != nullptr) return 0;
  n2v::UnitConsts K;
  K.bR = 1.0 / p;  // the reference's weight / return_param with weight == 1.0
  K.bM = 1.0;
  K.bO = 1.0 / q;
  if (!scales_exactly(K.bR, &K.TR) || !scales_exactly(K.bM, &K.TM) || !scales_exactly(K.bO, &K.TO))
    return 0;
  if (K.TR == 0 || K.TO == 0) return 0;  // a zero class could make the row sum 0
  const int64_t total = n_start * (int64_t)num_walks;
  if (total == 0) return 1;
  int64_t blocks = (total + n2v::kWavesPerBlock - 1) / n2v::kWavesPerBlock;
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipLaunchKernelGGL(n2v::walk_exact_unit_kernel, dim3((unsigned)blocks),
                     dim3(n2v::kWavesPerBlock * 64), 0, (hipStream_t)stream, *g, start_ids,
                     n_start, num_walks, walk_length, p, q, K, seed, walks_out, valid_out,
                     status);
  if (hipGetLastError() != hipSuccess) return N2V_ELAUNCH;
  return 1;
}
