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

// smallest stride >= n floats that is 32 mod 64
__host__ __device__ constexpr int bank_half_stride(int n) { return (n + 31) / 64 * 64 + 32; }

__device__ __forceinline__ float f4_at(const float4 &v, int k) {
  return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w));
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

// VEC = dim / 64; TROWS = rows of the target tile (8: 1 + negative <= 8, two k-steps; else 16);
// KC = k-steps over the context rows (3: 2 * window + 2 <= 12 ring rows, else 4 and 16 rows)
template <int VEC, int TROWS, int KC>
__global__ __launch_bounds__(kBWaves * 64) void sgns_batched_kernel(
    const int32_t *__restrict__ walks, int64_t n_walks, int32_t walk_len, float *syn0,
    float *syn1neg, const uint32_t *__restrict__ cum_table,
    const uint32_t *__restrict__ sample_int, const float *__restrict__ exp_table_g,
    n2v_sgns_params P, unsigned long long *pairs_out, int32_t own_negw) {
  constexpr int D = 64 * VEC, Q = 16 * VEC, RS = Q + 4, NCH = D / 16;
  constexpr int RROWS = KC == 3 ? 12 : 16;          // physical rows of the context ring
  // plane strides = 32 mod 64 floats: the four planes of a row then start on alternating bank
  // halves, which both operand shapes want (16-byte reads of 4 rows x 4 column groups; 8-byte
  // row vectors of two planes per half-wave)
  constexpr int PR = bank_half_stride((RROWS + 1) * RS);  // the ring (+ a spare row)
  constexpr int PT = bank_half_stride(TROWS * RS);      // the target tile
  constexpr int KT = TROWS == 8 ? 2 : 4;            // k-steps over target rows
  constexpr int TM_WORDS = TROWS == 8 ? 1 : 2;      // packed multiplicities: 4 bits per target
  const int window = P.window, K = P.negative;
  const int PL = K + 1 + TM_WORDS + 1;              // plan ints per position (below)
  float alpha = P.alpha;  // per launch, or per row (P.row_alpha: the rate of the row's gensim job)
  const bool hogwild = P.deterministic == 0;
  const int hub_rows = hogwild ? P.hub_rows : 0;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *exp_lds = reinterpret_cast<float *>(smem);
  const int wave_in_block = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int j16 = lane & 15, g4 = lane >> 4;
  constexpr int per_wave_floats = 4 * PR + 4 * PT + 16 * 17;
  const int per_wave_ints = (2 * walk_len + walk_len * PL + 32 + (own_negw ? walk_len * K : 0) + 3) & ~3;
  unsigned char *mine = smem + kBExpTable * sizeof(float) +
                        (size_t)wave_in_block * ((size_t)per_wave_floats + per_wave_ints) * 4;
  float *ring = reinterpret_cast<float *>(mine);
  float *tgt = ring + 4 * PR;
  float *gs = tgt + 4 * PT;
  int32_t *sent = reinterpret_cast<int32_t *>(gs + 16 * 17);
  int32_t *red = sent + walk_len;
  // the plan of a sentence, per position i: [0 .. K] the target words (centre, then the distinct
  // negatives != centre in draw order), [K + 1 ..] their multiplicities (4 bits each), then one
  // word: bits 0..4 the number of targets, bits 8.. a mask of the targets position i - 1 writes
  int32_t *plan = red + walk_len;
  int32_t *mphys = plan + walk_len * PL;  // [16] M index -> physical ring row
  int32_t *mmultM = mphys + 16;           // [16] multiplicity by M index (0: unused)
  // the raw draws are only needed until the plan is built: they borrow the two row tiles (long
  // sentences with many negatives that do not fit there get their own area)
  int32_t *negw = own_negw ? mmultM + 16 : reinterpret_cast<int32_t *>(ring);
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
  auto tm_of = [&](const int32_t *pl, int t) { return (pl[K + 1 + (t >> 3)] >> (4 * (t & 7))) & 15; };
  // per-lane constants of the accumulator layout: lane (g4, j16) holds rows 4 g4 + r, column j16
  int t_of_r[4], t_of_col = tgt_of(j16);
#pragma unroll
  for (int rI = 0; rI < 4; ++rI) t_of_r[rI] = tgt_of(4 * g4 + rI);

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
    if (P.row_alpha) alpha = P.row_alpha[r];  // gensim: the rate of this sentence's job (a scalar load)
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
    // ---- the plan: one lane per position dedupes its draws into the target list ----
    for (int i = lane; i < nf; i += 64) {
      int32_t *pl = plan + i * PL;
      const int centre = sent[i];
      pl[0] = centre;
      int nt = 1;
      int tm[TM_WORDS];
#pragma unroll
      for (int wI = 0; wI < TM_WORDS; ++wI) tm[wI] = 0;
      tm[0] = 1;  // the centre word: multiplicity 1
      for (int d = 0; d < K; ++d) {
        const int x = negw[i * K + d];
        bool first = x != centre;  // gensim: a negative equal to the positive target is skipped
        for (int e = 0; e < d; ++e) first = first && negw[i * K + e] != x;
        if (!first) continue;
        int mult = 1;
        for (int e = d + 1; e < K; ++e) mult += negw[i * K + e] == x;
        pl[nt] = x;
        if (TM_WORDS == 1 || nt < 8)
          tm[0] |= mult << (4 * (nt & 7));
        else
          tm[TM_WORDS - 1] |= mult << (4 * (nt & 7));
        ++nt;
      }
      for (int t = nt; t <= K; ++t) pl[t] = centre;  // unused slots name a real row
#pragma unroll
      for (int wI = 0; wI < TM_WORDS; ++wI) pl[K + 1 + wI] = tm[wI];
      pl[K + 1 + TM_WORDS] = nt;
    }
    wave_sync();
    // which targets of position i does position i - 1 write (they are read after that write)
    for (int i = lane; i < nf; i += 64) {
      if (i == 0) continue;
      int32_t *pl = plan + i * PL;
      const int32_t *pp = pl - PL;
      const int nt = pl[K + 1 + TM_WORDS], np = pp[K + 1 + TM_WORDS] & 31;
      int late = 0;
      for (int t = 0; t < nt; ++t) {
        const int x = pl[t];
        for (int sI = 0; sI < np; ++sI) late |= (pp[sI] == x) << t;
      }
      pl[K + 1 + TM_WORDS] = nt | (late << 8);
    }
    wave_sync();
    {  // the spare ring row: all zero (it only ever receives 0 + 0 * x); the draws lay here
      BRow<VEC> z;
#pragma unroll
      for (int v = 0; v < VEC; ++v) z.v[v] = 0.0f;
      lds_put<VEC>(ring, PR, RROWS, lane, z);
    }

    // ---- the context ring: lane p < 16 is physical row p / position residue p ----
    int row_word = -1, row_ref = 0, row_pos = 0 /* residues held by this row */, pos_row = 0;
    BRow<VEC> loaded[RROWS];  // the rows as they were read (hogwild: deltas go back atomically)

    // position j enters the window.  Returns the physical row, and in `fresh` whether the row
    // was allocated now (its syn0 row must be brought in by the caller).
    auto enter = [&](int j, int word, bool &fresh) -> int {
      const uint64_t hit = ballot64(lane < 16 && row_ref > 0 && row_word == word);
      int row;
      if (hit) {
        row = __builtin_ctzll(hit);
        fresh = false;
      } else {
        row = __builtin_ctzll(ballot64(lane < RROWS && row_ref == 0));  // 2 * window + 2 <= RROWS
        fresh = true;
      }
      if (lane == row) {
        row_word = word;
        ++row_ref;
        row_pos |= 1 << (j & 15);
      }
      if (lane == (j & 15)) pos_row = row;
      return row;
    };
    // position j leaves: the row goes back to HBM with its last position
    auto leave = [&](int j) {
      const int row = rl(pos_row, j & 15);
      if (lane == row) {
        --row_ref;
        row_pos &= ~(1 << (j & 15));
      }
      if (rl(row_ref, row) == 0) {
        BRow<VEC> t;
        lds_get<VEC>(ring, PR, row, lane, t);
        float *dst = syn0 + (int64_t)rl(row_word, row) * D;
        if (hogwild) {
          // other waves may have trained this row meanwhile: add what THIS wave learned
#pragma unroll
          for (int rI = 0; rI < RROWS; ++rI)
            if (row == rI) {
#pragma unroll
              for (int v = 0; v < VEC; ++v)
                unsafeAtomicAdd(dst + lane * VEC + v, t.v[v] - loaded[rI].v[v]);
            }
        } else {
          g_store<VEC>(dst, lane, t);
        }
      }
    };

    // ---- prologue: positions 0 .. window enter, targets of position 0 arrive ----
    for (int j = 0; j <= window && j < nf; ++j) {
      bool fresh;
      const int word = sent[j];
      const int row = enter(j, word, fresh);
      if (fresh) {
        BRow<VEC> t;
        g_load<VEC>(syn0 + (int64_t)word * D, lane, t);
        lds_put<VEC>(ring, PR, row, lane, t);
#pragma unroll
        for (int rI = 0; rI < RROWS; ++rI)
          if (row == rI) loaded[rI] = t;
      }
    }
    int nt_c = rfl(plan[K + 1 + TM_WORDS]) & 31;
#pragma unroll
    for (int t = 0; t < TROWS; ++t)
      if (t < nt_c) {
        BRow<VEC> row;
        g_load<VEC>(syn1neg + (int64_t)plan[t] * D, lane, row);
        lds_put<VEC>(tgt, PT, t, lane, row);
      }
    wave_sync();

    for (int i = 0; i < nf; ++i) {
      const int32_t *plc = plan + i * PL;
      const int b = rfl(red[i]);
      const int lo = max(0, i - window + b);
      const int hi = min(nf, i + window + 1 - b);
      // ---- A. requests for position i + 1: its entering context row and its target rows ----
      const bool have_next = i + 1 < nf;
      int nt_n = 0, late = 0;
      BRow<VEC> nrow[TROWS];
      BRow<VEC> crow_in;
      int enter_row = -1;
      if (have_next) {
        const int32_t *pln = plc + PL;
        const int jn = i + 1 + window;
        if (jn < nf) {
          bool fresh;
          const int word = sent[jn];
          const int row = enter(jn, word, fresh);
          if (fresh) {
            enter_row = row;
            g_load<VEC>(syn0 + (int64_t)word * D, lane, crow_in);
          }
        }
        const int w = rfl(pln[K + 1 + TM_WORDS]);
        nt_n = w & 31;
        late = w >> 8;  // rows position i writes: read after its stores (stays in order)
        // Straight-line: every slot of the tile is requested, slots without a target (and rows
        // that must be re-read later anyway) name the centre row again -- the same cache lines.
        // With a branch per row the compiler reuses the destination registers as address
        // temporaries and guards each request with s_waitcnt vmcnt(0): the requests then go out
        // one memory latency after the other instead of together.
        int words[TROWS];
#pragma unroll
        for (int t = 0; t < TROWS; ++t) words[t] = pln[t <= K ? t : 0];
#pragma unroll
        for (int t = 0; t < TROWS; ++t)
          g_load<VEC>(syn1neg + (int64_t)words[t] * D, lane, nrow[t]);
      }

      // ---- B. the position's context rows: the physical rows whose positions lie in the
      //         reduced window, ascending; multiplicity = how many of its positions do ----
      const int span = hi - lo;  // <= 2 * window + 1 <= 15 positions, i among them
      uint32_t jmask = ((1u << span) - 1u) << (lo & 15);
      jmask = (jmask | (jmask >> 16)) & 0xffffu;
      jmask &= ~(1u << (i & 15));
      const int npairs = __popc(jmask);
      const int cm = lane < 16 ? __popc((uint32_t)row_pos & jmask) : 0;
      const uint64_t used = ballot64(cm > 0);
      f32x4 newt[NCH];
      int64_t st_off[4];
      bool st_ok[4], st_hub[4];
      bool trained = false;
      if (used) {
        const int rank = __popcll(used & ((1ull << lane) - 1ull));
        if (lane < 16) {
          mphys[lane] = RROWS;  // M indices without a context row read the all-zero spare row
          mmultM[lane] = 0;
        }
        wave_sync();
        if (cm > 0) {
          const int m = KC == 3 ? 4 * (rank / 3) + rank % 3 : rank;
          mphys[m] = lane;
          mmultM[m] = cm;
        }
        wave_sync();

        // ---- F = Ctx . Tgt^T : one fmaf chain per (context row, target row) over d ----
        const bool col_ok = t_of_col >= 0 && t_of_col < nt_c;
        const float4 *ap = reinterpret_cast<const float4 *>(ring + g4 * PR + mphys[j16] * RS);
        const float4 *bp =
            reinterpret_cast<const float4 *>(tgt + g4 * PT + (col_ok ? t_of_col : 0) * RS);
        // four interleaved chains (k-steps s = a mod 4 go to chain a), summed pairwise at the
        // end: consecutive MFMAs never wait for each other's 40-cycle accumulator latency
        f32x4 f0 = {0.0f, 0.0f, 0.0f, 0.0f}, f1 = f0, f2 = f0, f3 = f0;
#pragma unroll
        for (int q = 0; q < Q / 4; ++q) {
          const float4 a4 = ap[q], b4 = bp[q];
          f0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, f0, 0, 0, 0);
          f1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, f1, 0, 0, 0);
          f2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, f2, 0, 0, 0);
          f3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, f3, 0, 0, 0);
        }
        f32x4 f;
#pragma unroll
        for (int rI = 0; rI < 4; ++rI) f[rI] = (f0[rI] + f1[rI]) + (f2[rI] + f3[rI]);
        // ---- G[m = 4 g + r][n = l & 15] ----
        const int cmult = col_ok ? tm_of(plc, t_of_col) : 0;
        const float label = j16 == 0 ? 1.0f : 0.0f;
        float gv[4];
        int roff[4];  // this lane's NCH-float vector of the ring row of M index 4 g4 + r
#pragma unroll
        for (int rI = 0; rI < 4; ++rI) {
          const int m = 4 * g4 + rI;
          const int mult = mmultM[m] * cmult;
          roff[rI] = (j16 >> 2) * PR + mphys[m] * RS + NCH * (j16 & 3);
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
        int toff[4];  // the same vector of the target row of tile column 4 g4 + r (row 0: none)
#pragma unroll
        for (int rI = 0; rI < 4; ++rI) {
          st_ok[rI] = t_of_r[rI] >= 0 && t_of_r[rI] < nt_c;
          toff[rI] = (j16 >> 2) * PT + (st_ok[rI] ? t_of_r[rI] : 0) * RS + NCH * (j16 & 3);
          // hub rows (hogwild, n2v_sgns_params.hub_rows): this position's contribution is ADDED
          // atomically instead of the row being overwritten -- the accumulator then starts from 0
          st_hub[rI] = st_ok[rI] && plc[st_ok[rI] ? t_of_r[rI] : 0] < hub_rows;
        }
        // In the two updates tile column j of chunk c is element d = NCH * j + c: a lane's NCH
        // columns are NCH consecutive floats of a row -- 16-byte LDS reads and writes, and the
        // new target rows leave as whole 16-byte stores (a row = 16 lanes x NCH * 4 bytes).
        // ---- Tgt_new = Tgt_old + G^T . Ctx_old  (kept in registers until Ctx is done) ----
        {
          float4 cin[4][VEC], bx[KC][VEC];
#pragma unroll
          for (int rI = 0; rI < 4; ++rI)
#pragma unroll
            for (int h = 0; h < VEC; ++h) {
              const float4 v = *reinterpret_cast<const float4 *>(tgt + toff[rI] + 4 * h);
              cin[rI][h] = (st_ok[rI] && !st_hub[rI]) ? v : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
#pragma unroll
          for (int s = 0; s < KC; ++s)
#pragma unroll
            for (int h = 0; h < VEC; ++h)
              bx[s][h] = *reinterpret_cast<const float4 *>(ring + roff[s] + 4 * h);
#pragma unroll
          for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int rI = 0; rI < 4; ++rI) newt[c][rI] = f4_at(cin[rI][c >> 2], c & 3);
#pragma unroll
          for (int s = 0; s < KC; ++s)  // k-step outer, chunk inner: independent accumulators
#pragma unroll
            for (int c = 0; c < NCH; ++c)
              newt[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv[s], f4_at(bx[s][c >> 2], c & 3),
                                                             newt[c], 0, 0, 0);
        }
        wave_sync();
        // ---- Ctx += G . Tgt_old, in place in the ring ----
        {
          float4 cin[4][VEC], bx[KT][VEC];
#pragma unroll
          for (int rI = 0; rI < 4; ++rI)
#pragma unroll
            for (int h = 0; h < VEC; ++h)
              cin[rI][h] = *reinterpret_cast<const float4 *>(ring + roff[rI] + 4 * h);
#pragma unroll
          for (int s = 0; s < KT; ++s)
#pragma unroll
            for (int h = 0; h < VEC; ++h)
              bx[s][h] = *reinterpret_cast<const float4 *>(tgt + toff[s] + 4 * h);
          f32x4 out[NCH];
#pragma unroll
          for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int rI = 0; rI < 4; ++rI) out[c][rI] = f4_at(cin[rI][c >> 2], c & 3);
#pragma unroll
          for (int s = 0; s < KT; ++s)
#pragma unroll
            for (int c = 0; c < NCH; ++c)
              out[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s], f4_at(bx[s][c >> 2], c & 3),
                                                            out[c], 0, 0, 0);
          // rows without a context (the spare row) receive what they held + 0 * x: still zero
#pragma unroll
          for (int rI = 0; rI < 4; ++rI)
#pragma unroll
            for (int h = 0; h < VEC; ++h)
              *reinterpret_cast<float4 *>(ring + roff[rI] + 4 * h) =
                  make_float4(out[4 * h][rI], out[4 * h + 1][rI], out[4 * h + 2][rI], out[4 * h + 3][rI]);
        }
        // where the new target rows go (the tile is refilled below)
#pragma unroll
        for (int rI = 0; rI < 4; ++rI)
          st_off[rI] = (int64_t)plc[st_ok[rI] ? t_of_r[rI] : 0] * D + NCH * j16;
        trained = true;
        pairs += (unsigned long long)npairs;
        wave_sync();
      }
      // ---- C. what was requested for position i + 1 has arrived (asked for a whole position
      //         ago): it is consumed BEFORE this position's stores are issued, so that the wait
      //         for the loads never waits for a store ----
      // every request of phase A is complete from here on, on EVERY path (hence outside the
      // branch): otherwise the compiler must assume that a request it did not see consumed (a slot
      // without a target, a row that is re-read below, the last position) is still in flight at
      // the top of the loop, and guards the next position's requests with waits that serialise
      // them
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), lgkmcnt / expcnt unconstrained
      if (have_next) {
        if (enter_row >= 0) {
          lds_put<VEC>(ring, PR, enter_row, lane, crow_in);
#pragma unroll
          for (int rI = 0; rI < RROWS; ++rI)
            if (enter_row == rI) loaded[rI] = crow_in;
        }
#pragma unroll
        for (int t = 0; t < TROWS; ++t)
          if (t < nt_n && !((late >> t) & 1)) lds_put<VEC>(tgt, PT, t, lane, nrow[t]);
      }
      // ---- D. the target rows go back to HBM from the accumulator layout (64 B per row and
      //         instruction); position i - window is outside every later window ----
      if (trained) {
#pragma unroll
        for (int rI = 0; rI < 4; ++rI)
          if (st_ok[rI]) {
            if (st_hub[rI]) {
#pragma unroll
              for (int c = 0; c < NCH; ++c) unsafeAtomicAdd(syn1neg + st_off[rI] + c, newt[c][rI]);
            } else {
#pragma unroll
              for (int h = 0; h < VEC; ++h)
                *reinterpret_cast<float4 *>(syn1neg + st_off[rI] + 4 * h) =
                    make_float4(newt[4 * h][rI], newt[4 * h + 1][rI], newt[4 * h + 2][rI], newt[4 * h + 3][rI]);
            }
          }
      }
      if (i - window >= 0) leave(i - window);
      if (have_next) {
        if (late) {  // rows this position has just written: read them back now, in order
          const int32_t *pln = plc + PL;
#pragma unroll
          for (int t = 0; t < TROWS; ++t)
            if (t < nt_n && ((late >> t) & 1)) {
              g_load<VEC>(syn1neg + (int64_t)pln[t] * D, lane, nrow[t]);
              lds_put<VEC>(tgt, PT, t, lane, nrow[t]);
            }
        }
        nt_c = nt_n;
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
  const int kc = 2 * P->window + 2 <= 12 ? 3 : 4;
  const int rrows = kc == 3 ? 12 : 16;
  const int trows = 1 + P->negative <= 8 ? 8 : 16;
  const int RS = 16 * VEC + 4;
  const int PR = bank_half_stride((rrows + 1) * RS), PT = bank_half_stride(trows * RS);
  const int own_negw = (size_t)walk_len * P->negative > (size_t)4 * (PR + PT);  // else they borrow the tiles
  const int PL = P->negative + 1 + (trows == 8 ? 1 : 2) + 1;
  const size_t per_wave = ((size_t)(4 * PR + 4 * PT + 16 * 17) +
                           (size_t)((2 * walk_len + walk_len * PL + 32 + (own_negw ? walk_len * P->negative : 0) + 3) & ~3)) * 4;
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
                       exp_table, *P, pairs_out, own_negw);                                          \
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
