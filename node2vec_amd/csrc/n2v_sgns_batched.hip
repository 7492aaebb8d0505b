// n2v_sgns_batched.hip -- K3, the opt-in BATCHED variant (n2v_sgns_params.batched = 1).
//
// NOT gensim's sampling (the default kernel, n2v_sgns.hip, is): the `negative` draws are made
// once per centre position and shared by its <= 2 * window (centre, context) pairs, and all
// pairs of a position are trained from one snapshot of the rows (Ji et al. 2016).  A position
// then is three small dense products on the matrix cores,
//     F  = Ctx . Tgt^T          [<= 16 context rows] x [<= 16 target rows], K = dim
//     Tgt += G^T . Ctx_old      G = ((label - sigma(F)) * alpha) * multiplicities, |F| >= 6 -> 0
//     Ctx += G   . Tgt_old
// and moves 2 * 4 * dim * (2 + k) bytes of HBM per POSITION instead of per pair (~ 6 pairs).
// Normative restatement: oracle/n2v_oracle_sgns.c (its batched function); an f32 MFMA
// is bit-for-bit a k-ordered fmaf chain (v_mfma_f32_16x16x4_f32), so the deterministic mode is
// bit-identical to it.
//
// One wave64 trains one walk.  Per wave, in LDS:
//   * the CONTEXT RING: the syn0 rows of the <= 2 * window + 1 positions around the centre,
//     one physical row per distinct word (a walk revisits vertices: positions holding the same
//     word share a row, reference-counted).  A row enters once, is trained in place by every
//     centre position that has it in its window and is written back to HBM when its last
//     position leaves: each syn0 row costs one read + one write per ~11 positions;
//   * the TARGET TILE: syn1neg rows of the centre word and of the position's negatives,
//     prefetched into registers while the previous position is being trained.
// LDS image of a row tile ("planar"): plane p = d / (dim / 4) holds elements [p * dim/4,
// (p + 1) * dim/4) of every row at stride dim/4 + 4 floats; planes are a multiple of 64 floats
// apart.  Lane (row = l & 15, group = l >> 4) reads its MFMA operand stream -- row `row`, the
// elements of plane `group` -- with ds_read_b128 free of bank conflicts (9 * row mod 16 is a
// bijection; the lane groups of a b128 read mix two planes, which sit on the same banks only for
// equal rows), and the accumulator layout (4 rows x 16 consecutive columns per lane group)
// reads and writes 64 consecutive bytes per row.
#include "n2v_common.h"

namespace n2v {

constexpr int kBWaves = 2;        // waves per block (LDS: ~16 KB per wave at dim 128)
constexpr int kBExpTable = 1000;  // EXP_TABLE_SIZE

typedef float f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline uint64_t b_sentence_stream(uint64_t seed, uint64_t sentence_id) {
  return mix64(seed ^ mix64(sentence_id + 0xA0761D6478BD642FULL));
}
__host__ __device__ inline uint64_t b_draw(uint64_t hs, uint64_t idx) {
  return mix64(hs + (idx + 1ULL) * 0xE7037ED1A0B428DBULL);
}

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <int VEC>
struct BRow {
  float v[VEC];
};

// one row of `dim` = 64 * VEC floats, lane l owns elements l * VEC .. l * VEC + VEC - 1
template <int VEC>
__device__ __forceinline__ void g_load(const float *base, int lane, BRow<VEC> &r) {
  if constexpr (VEC == 1) {
    r.v[0] = base[lane];
  } else if constexpr (VEC == 2) {
    const float2 t = *reinterpret_cast<const float2 *>(base + lane * 2);
    r.v[0] = t.x;
    r.v[1] = t.y;
  } else {
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q) {
      const float4 t = *reinterpret_cast<const float4 *>(base + lane * VEC + q * 4);
      r.v[4 * q] = t.x;
      r.v[4 * q + 1] = t.y;
      r.v[4 * q + 2] = t.z;
      r.v[4 * q + 3] = t.w;
    }
  }
}
template <int VEC>
__device__ __forceinline__ void g_store(float *base, int lane, const BRow<VEC> &r) {
  if constexpr (VEC == 1) {
    base[lane] = r.v[0];
  } else if constexpr (VEC == 2) {
    *reinterpret_cast<float2 *>(base + lane * 2) = make_float2(r.v[0], r.v[1]);
  } else {
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q)
      *reinterpret_cast<float4 *>(base + lane * VEC + q * 4) =
          make_float4(r.v[4 * q], r.v[4 * q + 1], r.v[4 * q + 2], r.v[4 * q + 3]);
  }
}
// the same lane-owned elements in the planar LDS image: plane l >> 4, offset (l & 15) * VEC
template <int VEC>
__device__ __forceinline__ float *lds_vec(float *tile, int plane_stride, int row, int lane) {
  constexpr int RS = 16 * VEC + 4;
  return tile + (lane >> 4) * plane_stride + row * RS + (lane & 15) * VEC;
}
template <int VEC>
__device__ __forceinline__ void lds_put(float *tile, int plane_stride, int row, int lane,
                                        const BRow<VEC> &r) {
  float *p = lds_vec<VEC>(tile, plane_stride, row, lane);
#pragma unroll
  for (int v = 0; v < VEC; ++v) p[v] = r.v[v];
}
template <int VEC>
__device__ __forceinline__ void lds_get(float *tile, int plane_stride, int row, int lane,
                                        BRow<VEC> &r) {
  const float *p = lds_vec<VEC>(tile, plane_stride, row, lane);
#pragma unroll
  for (int v = 0; v < VEC; ++v) r.v[v] = p[v];
}

__device__ __forceinline__ int b_bisect(const uint32_t *a, int lo, int hi, uint32_t x) {
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < x)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

// VEC = dim / 64; TROWS = rows of the target tile (8: 1 + negative <= 8, two k-steps; else 16)
template <int VEC, int TROWS, int KC>  // KC: k-steps over context rows (3: 2 * window + 1 <= 12)
__global__ __launch_bounds__(kBWaves * 64) void sgns_batched_kernel(
    const int32_t *__restrict__ walks, int64_t n_walks, int32_t walk_len, float *syn0,
    float *syn1neg, const uint32_t *__restrict__ cum_table,
    const uint32_t *__restrict__ sample_int, const float *__restrict__ exp_table_g,
    n2v_sgns_params P, unsigned long long *pairs_out, int rrows) {
  constexpr int D = 64 * VEC, Q = 16 * VEC, RS = Q + 4, NCH = D / 16;
  constexpr int PT = (TROWS * RS + 63) / 64 * 64;  // plane stride of the target tile
  constexpr int KT = TROWS == 8 ? 2 : 4;           // k-steps over target rows
  const int PR = (rrows * RS + 63) / 64 * 64;      // plane stride of the context ring
  const int window = P.window, K = P.negative;
  const float alpha = P.alpha;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *exp_lds = reinterpret_cast<float *>(smem);
  const int wave_in_block = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int j16 = lane & 15, g4 = lane >> 4;
  const int per_wave_floats = 4 * PR + 4 * PT + 16 * 17;
  const int per_wave_ints = (2 * walk_len + walk_len * K + 64 + 3) & ~3;  // regions stay 16-byte aligned
  unsigned char *mine = smem + kBExpTable * sizeof(float) +
                        (size_t)wave_in_block * ((size_t)per_wave_floats + per_wave_ints) * 4;
  float *ring = reinterpret_cast<float *>(mine);
  float *tgt = ring + 4 * PR;
  float *gs = tgt + 4 * PT;
  int32_t *sent = reinterpret_cast<int32_t *>(gs + 16 * 17);
  int32_t *red = sent + walk_len;
  int32_t *negw = red + walk_len;
  int32_t *mphys = negw + walk_len * K;  // [16] M index -> physical ring row
  int32_t *mmultM = mphys + 16;          // [16] multiplicity by M index (0: unused)
  int32_t *twc = mmultM + 16;            // [16] target word by target index
  int32_t *colmult = twc + 16;           // [16] multiplicity by tile column (0: unused)
  for (int i = threadIdx.x; i < kBExpTable; i += blockDim.x) exp_lds[i] = exp_table_g[i];
  __syncthreads();

  const uint32_t domain = cum_table[P.n_vocab - 1];
  const int waves_per_block = blockDim.x >> 6;
  const int64_t n_waves = (int64_t)gridDim.x * waves_per_block;
  unsigned long long pairs = 0;
  const bool dynamic = pairs_out != nullptr && n_walks < 0xfffffff0ll;
  unsigned int *row_counter = reinterpret_cast<unsigned int *>(pairs_out + 1);
  int64_t rr = (int64_t)blockIdx.x * waves_per_block + wave_in_block;

  // target index of tile column n (-1: no target there); target t sits in column
  // 4 * (t / 2) + t % 2 when two k-steps cover the targets, else in column t
  auto tgt_of = [&](int n) { return KT == 2 ? ((n & 3) < 2 ? 2 * (n >> 2) + (n & 3) : -1) : n; };

  for (;;) {
    if (dynamic) {
      unsigned int t = 0;
      if (lane == 0) t = atomicAdd(row_counter, 1u);
      rr = (int64_t)(unsigned int)rfl((int)t);
    }
    if (rr >= n_walks) break;
    const int64_t r = readfirstlane_i64(rr);
    if (!dynamic) rr += n_waves;
    const uint64_t hs = b_sentence_stream(P.seed, (uint64_t)(P.sentence_base + r));
    // ---- sentence preparation: vocabulary filter, subsampling, reduced windows ----
    int nf = 0;
    for (int base = 0; base < walk_len; base += 64) {
      const int t = base + lane;
      int32_t tok = t < walk_len ? walks[r * walk_len + t] : -1;
      bool keep = tok >= 0 && (int64_t)tok < P.n_vocab;
      if (keep && sample_int) {
        const uint32_t rnd = (uint32_t)(b_draw(hs, 2ULL * (uint64_t)t) >> 32);
        keep = !(sample_int[tok] < rnd);
      }
      const uint64_t mask = ballot64(keep);
      const int pos = nf + __popcll(mask & ((1ull << lane) - 1ull));
      if (keep) {
        sent[pos] = tok;
        red[pos] = (int32_t)((uint32_t)(b_draw(hs, 2ULL * (uint64_t)t + 1ULL) >> 32) %
                             (uint32_t)window);
      }
      nf += __popcll(mask);
    }
    // ---- the negative draws of every position (lane-parallel; shared by the position's pairs)
    for (int q = lane; q < nf * K; q += 64) {
      const int i = q / K, d = q - i * K;
      const uint64_t idx = 2ULL * (uint64_t)walk_len +
                           ((uint64_t)i * 2ULL * (uint64_t)window) * (uint64_t)K + (uint64_t)d;
      const uint32_t x = (uint32_t)((b_draw(hs, idx) >> 16) % (uint64_t)domain);
      int blo = 0, bhi = (int)P.n_vocab;
      if (P.cum_index) {
        const uint32_t bk = x >> (31 - P.cum_index_bits);
        blo = P.cum_index[bk];
        bhi = P.cum_index[bk + 1];
      }
      negw[q] = b_bisect(cum_table, blo, bhi, x);
    }
    wave_sync();
    if (nf < 2) continue;  // a single token has no context: nothing to train

    // ---- the context ring: lane p < 16 is physical row p / position residue p ----
    int row_word = -1, row_ref = 0, pos_row = 0;

    // position j enters the window.  Returns the physical row, and in `fresh` whether the row
    // was allocated now (its syn0 row must be brought in by the caller).
    auto enter = [&](int j, bool &fresh) -> int {
      const int word = rfl(sent[j]);
      const uint64_t hit = ballot64(lane < 16 && row_ref > 0 && row_word == word);
      int row;
      if (hit) {
        row = __builtin_ctzll(hit);
        fresh = false;
        if (lane == row) ++row_ref;
      } else {
        const uint64_t freem = ballot64(lane < rrows && row_ref == 0);
        row = __builtin_ctzll(freem);  // 2 * window + 2 <= rrows: a row is always free
        fresh = true;
        if (lane == row) {
          row_word = word;
          row_ref = 1;
        }
      }
      if (lane == (j & 15)) pos_row = row;
      return row;
    };
    // position j leaves: the row goes back to HBM with its last position
    auto leave = [&](int j) {
      const int row = rl(pos_row, j & 15);
      if (lane == row) --row_ref;
      if (rl(row_ref, row) == 0) {
        BRow<VEC> t;
        lds_get<VEC>(ring, PR, row, lane, t);
        g_store<VEC>(syn0 + (int64_t)rl(row_word, row) * D, lane, t);
      }
    };

    // target list of a position: lane t < 16 holds (word, multiplicity) of target t
    auto build_targets = [&](int i, int &tw, int &tm) -> int {
      const int centre = rfl(sent[i]);
      int nt = 1;
      tw = lane == 0 ? centre : -1;
      tm = lane == 0 ? 1 : 0;
      for (int d = 0; d < K; ++d) {
        const int x = rfl(negw[i * K + d]);
        if (x == centre) continue;  // gensim: a negative equal to the positive target is skipped
        const uint64_t hit = ballot64(lane >= 1 && lane < nt && tw == x);
        if (hit) {
          if (lane == (int)__builtin_ctzll(hit)) ++tm;
        } else {
          if (lane == nt) {
            tw = x;
            tm = 1;
          }
          ++nt;
        }
      }
      return nt;
    };
    // publish the current target list: words by target index, multiplicities by tile column
    auto publish_targets = [&](int tw, int tm, int nt) {
      if (lane < 16) {
        twc[lane] = lane < nt ? tw : rl(tw, 0);
        const int t = tgt_of(lane);
        const int m = __shfl(tm, t < 0 ? 0 : t, 64);
        colmult[lane] = (t >= 0 && t < nt) ? m : 0;
      }
    };

    // ---- prologue: positions 0 .. window enter, targets of position 0 arrive ----
    for (int j = 0; j <= window && j < nf; ++j) {
      bool fresh;
      const int row = enter(j, fresh);
      if (fresh) {
        BRow<VEC> t;
        g_load<VEC>(syn0 + (int64_t)rfl(sent[j]) * D, lane, t);
        lds_put<VEC>(ring, PR, row, lane, t);
      }
    }
    int tw_c, tm_c;
    int nt_c = build_targets(0, tw_c, tm_c);
#pragma unroll
    for (int t = 0; t < TROWS; ++t)
      if (t < nt_c) {
        BRow<VEC> row;
        g_load<VEC>(syn1neg + (int64_t)rl(tw_c, t) * D, lane, row);
        lds_put<VEC>(tgt, PT, t, lane, row);
      }
    publish_targets(tw_c, tm_c, nt_c);
    wave_sync();

    for (int i = 0; i < nf; ++i) {
      const int b = rfl(red[i]);
      const int lo = max(0, i - window + b);
      const int hi = min(nf, i + window + 1 - b);
      // ---- A. requests for position i + 1: its entering context row and its target rows ----
      const bool have_next = i + 1 < nf;
      int tw_n = -1, tm_n = 0, nt_n = 0;
      uint64_t late = 0;
      BRow<VEC> nrow[TROWS];
      BRow<VEC> crow_in;
      int enter_row = -1;
      if (have_next) {
        const int jn = i + 1 + window;
        if (jn < nf) {
          bool fresh;
          const int row = enter(jn, fresh);
          if (fresh) {
            enter_row = row;
            g_load<VEC>(syn0 + (int64_t)rfl(sent[jn]) * D, lane, crow_in);
          }
        }
        nt_n = build_targets(i + 1, tw_n, tm_n);
        // a row this position is about to write must be read after the write (stays in order)
        bool mylate = false;
        for (int t = 0; t < nt_c; ++t) mylate = mylate || (tw_n == rl(tw_c, t));
        late = ballot64(mylate && lane < nt_n);
#pragma unroll
        for (int t = 0; t < TROWS; ++t)
          if (t < nt_n && !((late >> t) & 1ull))
            g_load<VEC>(syn1neg + (int64_t)rl(tw_n, t) * D, lane, nrow[t]);
      }

      // ---- B. the position's context rows: distinct physical rows, multiplicity, M index ----
      int cm = 0, rank = -1, nu = 0, npairs = 0;
      for (int j = lo; j < hi; ++j) {
        if (j == i) continue;
        ++npairs;
        const int row = rl(pos_row, j & 15);
        if (rl(cm, row) == 0) {
          if (lane == row) rank = nu;
          ++nu;
        }
        if (lane == row) ++cm;
      }
      f32x4 newt[NCH];
      float *st_dst[4];
      bool st_ok[4];
      bool trained = false;
      if (nu > 0) {
        const int row0 = (int)__builtin_ctzll(ballot64(lane < 16 && rank == 0));
        if (lane < 16) {
          mphys[lane] = row0;
          mmultM[lane] = 0;
        }
        wave_sync();
        if (lane < 16 && cm > 0) {
          const int m = KC == 3 ? 4 * (rank / 3) + rank % 3 : rank;
          mphys[m] = lane;
          mmultM[m] = cm;
        }
        wave_sync();

        // ---- F = Ctx . Tgt^T : one fmaf chain per (context row, target row) over d ----
        const int tcol = tgt_of(j16);
        const float4 *ap = reinterpret_cast<const float4 *>(ring + g4 * PR + mphys[j16] * RS);
        const float4 *bp = reinterpret_cast<const float4 *>(
            tgt + g4 * PT + ((tcol >= 0 && tcol < nt_c) ? tcol : 0) * RS);
        f32x4 f = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < Q / 4; ++q) {
          const float4 a4 = ap[q], b4 = bp[q];
          f = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, f, 0, 0, 0);
          f = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, f, 0, 0, 0);
          f = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, f, 0, 0, 0);
          f = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, f, 0, 0, 0);
        }
        // ---- G[m = 4 g + r][n = l & 15] ----
        const int cmult = colmult[j16];
        const float label = j16 == 0 ? 1.0f : 0.0f;
        float gv[4];
        int mp[4];
#pragma unroll
        for (int rI = 0; rI < 4; ++rI) {
          const int m = 4 * g4 + rI;
          const int mult = mmultM[m] * cmult;
          mp[rI] = mphys[m];
          const float fv = f[rI];
          const bool live = mult > 0 && !(fv <= -6.0f || fv >= 6.0f);
          const int e = live ? (int)((fv + 6.0f) * 83.0f) : 0;
          const float gg = ((label - exp_lds[e]) * alpha) * (float)mult;
          gv[rI] = live ? gg : 0.0f;
          gs[m * 17 + j16] = gv[rI];
        }
        wave_sync();
        float ga[KT];  // A operand of Ctx += G . Tgt: G[m = l & 15][column 4 g + s]
#pragma unroll
        for (int s = 0; s < KT; ++s) ga[s] = gs[j16 * 17 + 4 * g4 + s];

        // ---- Tgt_new = Tgt_old + G^T . Ctx_old  (kept in registers until Ctx is done) ----
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int plane = c / VEC, off = (c % VEC) * 16 + j16;
          f32x4 acc;
#pragma unroll
          for (int rI = 0; rI < 4; ++rI) {
            const int t = tgt_of(4 * g4 + rI);
            acc[rI] = (t >= 0 && t < nt_c) ? tgt[plane * PT + t * RS + off] : 0.0f;
          }
#pragma unroll
          for (int s = 0; s < 4; ++s)
            if (s < KC) {
              const float bv = ring[plane * PR + mp[s] * RS + off];
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(gv[s], bv, acc, 0, 0, 0);
            }
          newt[c] = acc;
        }
        wave_sync();
        // ---- Ctx += G . Tgt_old, in place in the ring ----
        int tk[KT];
#pragma unroll
        for (int s = 0; s < KT; ++s) {
          const int t = tgt_of(4 * g4 + s);
          tk[s] = (t >= 0 && t < nt_c) ? t : 0;
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int plane = c / VEC, off = (c % VEC) * 16 + j16;
          f32x4 acc;
#pragma unroll
          for (int rI = 0; rI < 4; ++rI) acc[rI] = ring[plane * PR + mp[rI] * RS + off];
#pragma unroll
          for (int s = 0; s < KT; ++s) {
            const float bv = tgt[plane * PT + tk[s] * RS + off];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s], bv, acc, 0, 0, 0);
          }
#pragma unroll
          for (int rI = 0; rI < 4; ++rI)
            if (mmultM[4 * g4 + rI] > 0) ring[plane * PR + mp[rI] * RS + off] = acc[rI];
        }
        // where the new target rows go: fixed now, the target list is replaced below
#pragma unroll
        for (int rI = 0; rI < 4; ++rI) {
          const int t = tgt_of(4 * g4 + rI);
          st_ok[rI] = t >= 0 && t < nt_c;
          st_dst[rI] = syn1neg + (int64_t)twc[st_ok[rI] ? t : 0] * D + j16;
        }
        trained = true;
        pairs += (unsigned long long)npairs;
        wave_sync();
      }
      // ---- C. what was requested for position i + 1 has arrived (asked for a whole position
      //         ago): it is consumed BEFORE this position's stores are issued, so that the wait
      //         for the loads never waits for a store ----
      if (have_next) {
        if (enter_row >= 0) lds_put<VEC>(ring, PR, enter_row, lane, crow_in);
#pragma unroll
        for (int t = 0; t < TROWS; ++t)
          if (t < nt_n && !((late >> t) & 1ull)) lds_put<VEC>(tgt, PT, t, lane, nrow[t]);
      }
      // ---- D. the target rows go back to HBM from the accumulator layout (64 B per row and
      //         instruction); position i - window is outside every later window ----
      if (trained) {
#pragma unroll
        for (int rI = 0; rI < 4; ++rI)
          if (st_ok[rI]) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) st_dst[rI][16 * c] = newt[c][rI];
          }
      }
      if (i - window >= 0) leave(i - window);
      if (have_next) {
        if (late) {  // rows this position has just written: read them back now, in order
#pragma unroll
          for (int t = 0; t < TROWS; ++t)
            if (t < nt_n && ((late >> t) & 1ull)) {
              g_load<VEC>(syn1neg + (int64_t)rl(tw_n, t) * D, lane, nrow[t]);
              lds_put<VEC>(tgt, PT, t, lane, nrow[t]);
            }
        }
        tw_c = tw_n;
        tm_c = tm_n;
        nt_c = nt_n;
        publish_targets(tw_c, tm_c, nt_c);
        wave_sync();
      }
    }
    // ---- epilogue: the rows still in the ring go back ----
    for (int j = max(0, nf - window); j < nf; ++j) leave(j);
    wave_sync();
  }
  if (pairs_out && lane == 0 && pairs) atomicAdd(pairs_out, pairs);
}

}  // namespace n2v

extern "C" int n2v_sgns_batched_launch(const int32_t *walks, int64_t n_walks, int32_t walk_len,
                                       float *syn0, float *syn1neg, const uint32_t *cum_table,
                                       const uint32_t *sample_int, const float *exp_table,
                                       const n2v_sgns_params *P, unsigned long long *pairs_out,
                                       void *stream) {
  using namespace n2v;
  // what the tiles hold: dim 64 / 128 / 256 (64 lanes x 1 / 2 / 4 floats), <= 15 window
  // positions + 1 in the ring (window <= 7), <= 16 target rows (negative <= 15)
  if (P->dim != 64 && P->dim != 128 && P->dim != 256) return N2V_EINVAL;
  if (2 * P->window + 2 > 16 || 1 + P->negative > 16) return N2V_EINVAL;
  const int VEC = P->dim / 64;
  const int rrows = 2 * P->window + 2 <= 12 ? 12 : 16;
  const int trows = 1 + P->negative <= 8 ? 8 : 16;
  const int kc = 2 * P->window + 1 <= 12 ? 3 : 4;
  const int RS = 16 * VEC + 4;
  const int PR = (rrows * RS + 63) / 64 * 64, PT = (trows * RS + 63) / 64 * 64;
  const size_t per_wave = ((size_t)(4 * PR + 4 * PT + 16 * 17) +
                           (size_t)((2 * walk_len + walk_len * P->negative + 64 + 3) & ~3)) * 4;
  int64_t waves = P->n_vocab / 32;
  if (waves < 1) waves = 1;
  if (waves > n_walks) waves = n_walks;
  if (P->max_waves > 0 && waves > P->max_waves) waves = P->max_waves;
  int wpb = kBWaves;
  if (P->deterministic || waves < wpb) wpb = 1;
  if (P->deterministic) waves = 1;
  const size_t lds = kBExpTable * sizeof(float) + (size_t)wpb * per_wave;
  if (lds > 160 * 1024) return N2V_EINVAL;
  int64_t blocks = (waves + wpb - 1) / wpb;
  hipStream_t st = (hipStream_t)stream;
  if (pairs_out && hipMemsetAsync(pairs_out + 1, 0, sizeof(unsigned long long), st) != hipSuccess)
    return N2V_ELAUNCH;
#define N2V_BLAUNCH(VV, TT, KK)                                                                       \
  do {                                                                                            \
    const void *fn = (const void *)sgns_batched_kernel<VV, TT, KK>;                                   \
    if (lds > 64 * 1024 &&                                                                        \
        hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
      return N2V_ELAUNCH;                                                                         \
    if (!P->deterministic) {                                                                      \
      const int64_t cap = resident_blocks(fn, wpb * 64, lds);                                     \
      if (blocks > cap) blocks = cap;                                                             \
    }                                                                                             \
    hipLaunchKernelGGL((sgns_batched_kernel<VV, TT, KK>), dim3((unsigned)blocks), dim3(wpb * 64), lds, \
                       st, walks, n_walks, walk_len, syn0, syn1neg, cum_table, sample_int,        \
                       exp_table, *P, pairs_out, rrows);                                          \
  } while (0)
#define N2V_BLAUNCH_T(VV)     \
  do {                        \
    if (trows == 8 && kc == 3)      \
      N2V_BLAUNCH(VV, 8, 3);        \
    else if (trows == 8)            \
      N2V_BLAUNCH(VV, 8, 4);        \
    else if (kc == 3)               \
      N2V_BLAUNCH(VV, 16, 3);       \
    else                            \
      N2V_BLAUNCH(VV, 16, 4);       \
  } while (0)
  switch (VEC) {
    case 1: N2V_BLAUNCH_T(1); break;
    case 2: N2V_BLAUNCH_T(2); break;
    default: N2V_BLAUNCH_T(4); break;
  }
#undef N2V_BLAUNCH_T
#undef N2V_BLAUNCH
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}
