// n2v_sgns.hip -- K3, skip-gram negative-sampling SGD for gfx950.
//
// Replaces the trainer behind the reference's call site embedding.py:126
// (gensim.models.Word2Vec(sentences=all_walks, sg=1, hs=0, negative=k)); the
// algorithm is the published word2vec / gensim-3.8 train_batch_sg +
// fast_sentence_sg_neg, restated in oracle/n2v_oracle_sgns.c (DESIGN.md "SGNS").
//
// Design: one wave64 trains one walk.  A vector of `dim` floats lives across the
// wave (lane l owns elements l*VEC .. l*VEC+VEC-1, 16-B loads where dim = 64*VEC);
// dot products are per-lane FMA chains closed by a DPP + readlane tree reduction; the arithmetic intensity is 0.66 flop/B, so the kernel is a stream of
// row gathers/scatters against HBM -- plain FMA, no MFMA.  Sentence preparation
// (vocabulary filter, subsampling, reduced windows) and the negative draws
// (bisect over the cumulative count^0.75 table) are lane-parallel: every random
// number is a pure function of (seed, sentence id, draw index).  Waves update
// syn0/syn1neg unsynchronised (hogwild), like gensim's worker threads.
#include <cstdlib>

#include "n2v_common.h"

namespace n2v {

constexpr int kSgnsWaves = 4;      // waves per block
constexpr int kExpTable = 1000;    // EXP_TABLE_SIZE
constexpr int kBuckets = 1024;     // coarse index of cum_table: bucket b covers values [b<<21, (b+1)<<21)

__host__ __device__ inline uint64_t sentence_stream(uint64_t seed, uint64_t sentence_id) {
  return mix64(seed ^ mix64(sentence_id + 0xA0761D6478BD642FULL));
}
__host__ __device__ inline uint64_t sgns_draw(uint64_t hs, uint64_t idx) {
  return mix64(hs + (idx + 1ULL) * 0xE7037ED1A0B428DBULL);
}

template <int VEC>
struct Row {
  float v[VEC];
};

// Rows are read and written with AGENT-SCOPE (`sc1`) accesses (N2V_SGNS_COHERENT, default 1; 0 = plain accesses, the
// form of rounds 1 - 5).  The XCDs' L2s are not coherent with each other and a CU's L1 is never refreshed by another
// CU's stores: with plain accesses a row trained by waves on two XCDs keeps the updates of ONE of them for as long
// as a line stays cached -- a window of micro- to milliseconds where gensim's threads on a coherent CPU race over
// nanoseconds.  Measured (round 6, profiles/r10m_sgns_coherent.log): of the rows a block of 768 sentences trains on a
// 10^7 x 128 model, 4.5 % end a whole update away from the ordered run with plain accesses, 0.95 % with these; cfg 2
// link AUC 0.8983 -> 0.9016 (hub_rows = 0) and 0.9085 -> 0.9107 (default), the rate on a 10^8 x 128 model unchanged
// (813.6 / 813.7 M pairs/s: a random 512-byte row misses every cache anyway).  Values are the same bits: the
// deterministic mode is untouched.  Rows of up to 128 floats only (4- and 8-byte accesses per lane: dim <= 128, the
// dims of BASELINE cfgs 2 - 4): the 16-byte form (buffer loads / stores with aux = sc1 through a descriptor per row)
// was built and measured too and costs 3.4 % at dim 256 and 31 % at dim 512 (profiles/r10n_sgns_coherent_rates.log),
// so wider rows keep plain accesses.
#ifndef N2V_SGNS_COHERENT
#define N2V_SGNS_COHERENT 1
#endif

__device__ __forceinline__ float row_ld1(const float *p) {
#if N2V_SGNS_COHERENT
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned int *>(p), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT));
#else
  return *p;
#endif
}
__device__ __forceinline__ void row_st1(float *p, float x) {
#if N2V_SGNS_COHERENT
  __hip_atomic_store(reinterpret_cast<unsigned int *>(p), __float_as_uint(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  *p = x;
#endif
}

template <int VEC>
__device__ __forceinline__ void load_row(const float *base, int dim, int lane, bool full,
                                         Row<VEC> &r) {
  if (full) {
    if constexpr (VEC == 1) {
      r.v[0] = row_ld1(base + lane);
    } else if constexpr (VEC == 2) {
#if N2V_SGNS_COHERENT
      const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(base + lane * 2),
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      r.v[0] = __uint_as_float((unsigned int)u);
      r.v[1] = __uint_as_float((unsigned int)(u >> 32));
#else
      float2 t = *reinterpret_cast<const float2 *>(base + lane * 2);
      r.v[0] = t.x;
      r.v[1] = t.y;
#endif
    } else {
      // (16-byte accesses stay plain: see N2V_SGNS_COHERENT)
#pragma unroll
      for (int q = 0; q < VEC / 4; ++q) {
        float4 t = *reinterpret_cast<const float4 *>(base + lane * VEC + q * 4);
        r.v[4 * q + 0] = t.x;
        r.v[4 * q + 1] = t.y;
        r.v[4 * q + 2] = t.z;
        r.v[4 * q + 3] = t.w;
      }
    }
  } else {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      int e = lane * VEC + v;
      r.v[v] = e < dim ? row_ld1(base + e) : 0.0f;
    }
  }
}

template <int VEC>
__device__ __forceinline__ void store_row(float *base, int dim, int lane, bool full,
                                          const Row<VEC> &r) {
  if (full) {
    if constexpr (VEC == 1) {
      row_st1(base + lane, r.v[0]);
    } else if constexpr (VEC == 2) {
#if N2V_SGNS_COHERENT
      const unsigned long long u = (unsigned long long)__float_as_uint(r.v[0]) |
                                   ((unsigned long long)__float_as_uint(r.v[1]) << 32);
      __hip_atomic_store(reinterpret_cast<unsigned long long *>(base + lane * 2), u, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
#else
      *reinterpret_cast<float2 *>(base + lane * 2) = make_float2(r.v[0], r.v[1]);
#endif
    } else {
#pragma unroll
      for (int q = 0; q < VEC / 4; ++q)
        *reinterpret_cast<float4 *>(base + lane * VEC + q * 4) =
            make_float4(r.v[4 * q], r.v[4 * q + 1], r.v[4 * q + 2], r.v[4 * q + 3]);
    }
  } else {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      int e = lane * VEC + v;
      if (e < dim) row_st1(base + e, r.v[v]);
    }
  }
}

// one DPP step: value of the lane selected by `kCtrl` (no LDS round trip)
template <int kCtrl>
__device__ __forceinline__ float dpp_move(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), kCtrl, 0xF, 0xF, true));
}

// Dot product across the wave.  Per-lane FMA chain over its VEC elements, then a balanced
// tree over adjacent lanes (butterfly distances 1, 2, 4, 8, 16, 32 -- the order the oracle
// restates).  Distances 1..8 are DPP modifiers on the adds (quad_perm, row_half_mirror,
// row_mirror: values are already uniform inside the mirrored groups, so mirror == xor);
// the four row sums are read with v_readlane and combined as (R0 + R1) + (R2 + R3).
// No LDS crossbar (ds_bpermute cost six dependent LDS round trips per dot), and the
// result is a scalar to the compiler, so the branches on it are scalar branches.
template <int VEC>
__device__ __forceinline__ float wave_dot(const Row<VEC> &a, const Row<VEC> &b) {
  float acc = 0.0f;
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc = __fmaf_rn(a.v[v], b.v[v], acc);
  acc = acc + dpp_move<0xB1>(acc);   // quad_perm [1,0,3,2]  : lane ^ 1
  acc = acc + dpp_move<0x4E>(acc);   // quad_perm [2,3,0,1]  : lane ^ 2
  acc = acc + dpp_move<0x141>(acc);  // row_half_mirror      : the other quad  (== lane ^ 4)
  acc = acc + dpp_move<0x140>(acc);  // row_mirror           : the other half-row (== lane ^ 8)
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 48));
  return (r0 + r1) + (r2 + r3);      // lane ^ 16, then lane ^ 32
}

__device__ __forceinline__ int bisect_left_u32(const uint32_t *a, int64_t n, uint32_t x,
                                               int iters) {
  int64_t lo = 0, hi = n;
  for (int it = 0; it < iters; ++it) {
    int64_t mid = (lo + hi) >> 1;
    uint32_t val = a[mid < n ? mid : n - 1];
    bool act = lo < hi;
    bool less = val < x;
    lo = (act && less) ? mid + 1 : lo;
    hi = (act && !less) ? mid : hi;
  }
  return (int)lo;
}

// kDepth: pairs requested ahead of the one being trained.
// kRing (round 3): the syn0 rows of the window live in an LDS ring -- one physical row per
// distinct word of the <= 2 * window + 1 positions around the centre, shared by positions that
// hold the same word, reference-counted.  A context row is then read from HBM once when its
// position enters the window and written once when its last position leaves, instead of once per
// pair (~6 pairs per position): 8 * dim * (2 + k) bytes per pair become 8 * dim * (1 + k + ~1/6).
// Values are those of training pair by pair against memory (one wave, deterministic mode: bit for
// bit the oracle); in hogwild mode a row goes back as an atomic add of what THIS wave learned, so
// that holding it for a window's worth of positions loses no other wave's training.
template <int VEC, int kDepth, int kRingRows>  // kRingRows: 0 (no ring), 12 or 16
// dim <= 128 needs 56 VGPRs: hold the full 8 waves per SIMD (the kernel is latency-bound).
// Forcing 6 waves at dim 256 was measured: it spills and is slower.  The ring variant is bounded
// by LDS (5 waves per SIMD at dim 128) and keeps the rows' loaded values in registers.
__global__ __launch_bounds__(kSgnsWaves * 64, (kRingRows ? (VEC <= 2 ? 4 : 1) : (VEC <= 2 ? 8 : 1))) void sgns_kernel(
    const int32_t *__restrict__ walks, int64_t n_walks, int32_t walk_len, float *syn0,
    float *syn1neg, const uint32_t *__restrict__ cum_table,
    const uint32_t *__restrict__ sample_int, const float *__restrict__ exp_table_g,
    n2v_sgns_params P, unsigned long long *pairs_out, int32_t sent_cap) {
  constexpr bool kRing = kRingRows > 0;
  constexpr int ring_rows = kRingRows;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *exp_lds = reinterpret_cast<float *>(smem);
  int32_t *bucket = reinterpret_cast<int32_t *>(smem + kExpTable * sizeof(float));
  const int negcap = (2 * P.window + 1) * P.negative;  // the window incl. the centre slot
  // per wave: sent[sent_cap], red[sent_cap], neg[negcap] (+ pad), then the ring rows
  const int ints_per_wave = (2 * sent_cap + negcap + 3) & ~3;
  const int per_wave = ints_per_wave + (kRing ? ring_rows * 64 * VEC : 0);  // 4-byte words
  const int wave_in_block = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  // the 1024-bucket index of cum_table is only needed (and only allocated) without the
  // caller's fine index
  const int bucket_words = P.cum_index ? 0 : (kBuckets + 1 + 3);  // + 3: 16-byte alignment
  int32_t *sent = reinterpret_cast<int32_t *>(smem + kExpTable * sizeof(float)) + bucket_words +
                  wave_in_block * per_wave;
  int32_t *red = sent + sent_cap;
  int32_t *neg = red + sent_cap;
  float *ring = reinterpret_cast<float *>(sent + ints_per_wave);  // [ring_rows][64 * VEC]
  for (int i = threadIdx.x; i < kExpTable; i += blockDim.x) exp_lds[i] = exp_table_g[i];
  const int bis_iters = 64 - __clzll((long long)P.n_vocab);
  // bucket[b] = bisect_left(cum_table, b << 21): a draw r lies in bucket r >> 21 and its
  // bisect_left is confined to [bucket[b], bucket[b+1]] -- same index, half the probes
  if (!P.cum_index)
    for (int b = threadIdx.x; b <= kBuckets; b += blockDim.x)
      bucket[b] = bisect_left_u32(cum_table, P.n_vocab, (uint32_t)b << 21, bis_iters);
  __syncthreads();

  const int dim = P.dim, window = P.window, K = P.negative;
  // hogwild only: rows [0, hub_rows) are updated by atomic adds (deterministic mode keeps the
  // reference's read-modify-write, bit for bit the oracle)
  const int hub_rows = P.deterministic ? 0 : P.hub_rows;
  const int64_t hub_span = (int64_t)hub_rows * dim;
  const bool full = dim == 64 * VEC;
  float alpha = P.alpha;  // per launch, or per row (P.row_alpha: the rate of the row's gensim job)
  const uint32_t domain = cum_table[P.n_vocab - 1];
  const int waves_per_block = blockDim.x >> 6;
  const int64_t n_waves = (int64_t)gridDim.x * waves_per_block;
  unsigned long long pairs = 0;

  // rows are taken from a shared counter (pairs_out[1], reset by the launcher) rather than by a
  // fixed stride: a few per cent of tail at dim 128-256; a single wave (deterministic mode)
  // still sees them in order
  const bool dynamic = pairs_out != nullptr && n_walks < 0xfffffff0ll;
  unsigned int *row_counter = reinterpret_cast<unsigned int *>(pairs_out + 1);
  int64_t rr = (int64_t)blockIdx.x * waves_per_block + wave_in_block;
  for (;;) {
    if (dynamic) {
      unsigned int t = 0;
      if (lane == 0) t = atomicAdd(row_counter, 1u);
      rr = (int64_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)t);
    }
    if (rr >= n_walks) break;
    const int64_t r = readfirstlane_i64(rr);
    if (!dynamic) rr += n_waves;
    const uint64_t hs = sentence_stream(P.seed, (uint64_t)(P.sentence_base + r));
    if (P.row_alpha) alpha = P.row_alpha[r];  // gensim: the rate of this sentence's job (a scalar load)
    // ---- sentence preparation (lane-parallel, order-preserving compaction) ----
    int nf = 0;
    for (int base = 0; base < walk_len; base += 64) {
      const int t = base + lane;
      int32_t tok = t < walk_len ? walks[r * walk_len + t] : -1;
      bool keep = tok >= 0 && (int64_t)tok < P.n_vocab;
      if (keep && sample_int) {
        uint32_t rnd = (uint32_t)(sgns_draw(hs, 2ULL * (uint64_t)t) >> 32);
        keep = !(sample_int[tok] < rnd);
      }
      const uint64_t mask = ballot64(keep);
      const int pos = nf + __popcll(mask & ((1ull << lane) - 1ull));
      if (keep) {
        sent[pos] = tok;
        red[pos] = (int32_t)((uint32_t)(sgns_draw(hs, 2ULL * (uint64_t)t + 1ULL) >> 32) %
                             (uint32_t)window);
      }
      nf += __popcll(mask);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- the context ring (kRing): lane p < 16 is physical row p / position residue p.  A row
    // is lane-owned in LDS too (lane l keeps elements l * VEC ..), so ring traffic needs no
    // cross-lane ordering: every lane reads back what it wrote itself.
    constexpr int kRingMax = kRing ? kRingRows : 1;
    int row_word = -1, row_ref = 0, pos_row = 0;
    Row<VEC> loaded[kRingMax];
    auto ring_row = [&](int row) { return ring + row * (64 * VEC) + lane * VEC; };
    auto ring_get = [&](int row, Row<VEC> &r) {
      const float *q = ring_row(row);
#pragma unroll
      for (int v = 0; v < VEC; ++v) r.v[v] = q[v];
    };
    auto ring_put = [&](int row, const Row<VEC> &r) {
      float *q = ring_row(row);
#pragma unroll
      for (int v = 0; v < VEC; ++v) q[v] = r.v[v];
    };
    auto ring_enter = [&](int j, int word, bool &fresh) -> int {
      const uint64_t hit = ballot64(lane < 16 && row_ref > 0 && row_word == word);
      int row;
      if (hit) {
        row = __builtin_ctzll(hit);
        fresh = false;
      } else {
        row = __builtin_ctzll(ballot64(lane < ring_rows && row_ref == 0));  // 2 w + 2 <= ring_rows
        fresh = true;
      }
      if (lane == row) {
        row_word = word;
        ++row_ref;
      }
      if (lane == (j & 15)) pos_row = row;
      return row;
    };
    auto ring_loaded = [&](int row, const Row<VEC> &r) {
#pragma unroll
      for (int q = 0; q < kRingMax; ++q)
        if (row == q) loaded[q] = r;
    };
    auto ring_leave = [&](int j) {
      const int row = __builtin_amdgcn_readlane(pos_row, j & 15);
      if (lane == row) --row_ref;
      if (__builtin_amdgcn_readlane(row_ref, row) == 0) {
        Row<VEC> t;
        ring_get(row, t);
        float *dst = syn0 + (int64_t)__builtin_amdgcn_readlane(row_word, row) * dim;
        if (P.deterministic) {
          store_row<VEC>(dst, dim, lane, full, t);
        } else {
#pragma unroll
          for (int q = 0; q < kRingMax; ++q)
            if (row == q) {
#pragma unroll
              for (int v = 0; v < VEC; ++v)
                unsafeAtomicAdd(dst + lane * VEC + v, t.v[v] - loaded[q].v[v]);
            }
        }
      }
    };
    if (kRing) {
      for (int j = 0; j <= window && j < nf; ++j) {
        bool fresh;
        const int word = sent[j];
        const int row = ring_enter(j, word, fresh);
        if (fresh) {
          Row<VEC> t;
          load_row<VEC>(syn0 + (int64_t)word * dim, dim, lane, full, t);
          ring_put(row, t);
          ring_loaded(row, t);
        }
      }
    }

    for (int i = 0; i < nf; ++i) {
      // the row of position i + 1 + window is requested now and lands after this position's pairs
      Row<VEC> ring_in;
      int ring_in_row = -1;
      if (kRing && i + 1 + window < nf) {
        bool fresh;
        const int word = sent[i + 1 + window];
        const int row = ring_enter(i + 1 + window, word, fresh);
        if (fresh) {
          ring_in_row = row;
          load_row<VEC>(syn0 + (int64_t)word * dim, dim, lane, full, ring_in);
        }
      }
      // LDS loads at a uniform address are lane-varying to the compiler: mark them scalar
      const int32_t centre = __builtin_amdgcn_readfirstlane(sent[i]);
      const int b = __builtin_amdgcn_readfirstlane(red[i]);
      const int lo = max(0, i - window + b);
      const int hi = min(nf, i + window + 1 - b);
      // ---- negative draws for every pair of this position, lane-parallel ----
      const int nslots = (hi - lo) * K;
      for (int q = lane; q < nslots; q += 64) {
        const int jj = q / K, d = q - jj * K;
        const int j = lo + jj;
        const int rel = j - i + window - (j > i ? 1 : 0);
        const uint64_t idx = 2ULL * (uint64_t)walk_len +
                             ((uint64_t)i * 2ULL * (uint64_t)window + (uint64_t)rel) *
                                 (uint64_t)K + (uint64_t)d;
        const uint32_t x = (uint32_t)((sgns_draw(hs, idx) >> 16) % (uint64_t)domain);
        // bisect_left(cum_table, x) is confined to the bucket of x: [index[b], index[b + 1]] with
        // b = x >> shift -- through the caller's fine index in HBM (n2v_cum_index_build: with
        // 10^8 words a draw costs one index sector + one table sector instead of ~17 dependent
        // probes), else through the 1024-bucket index in LDS.  Same result either way.
        int blo, bhi;
        if (P.cum_index) {
          const uint32_t bk = x >> (31 - P.cum_index_bits);
          blo = P.cum_index[bk];
          bhi = P.cum_index[bk + 1];
        } else {
          blo = bucket[x >> 21];
          bhi = bucket[(x >> 21) + 1];
        }
        while (blo < bhi) {  // bisect_left inside the bucket
          const int mid = (blo + bhi) >> 1;
          if (cum_table[mid] < x)
            blo = mid + 1;
          else
            bhi = mid;
        }
        neg[q] = (j == i) ? centre : blo;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();

      // The centre row syn1neg[centre] is the label-1 target of EVERY pair of this
      // position and no negative may equal it (such a draw is skipped), so it lives in
      // registers for the whole position and is stored once: same values as updating
      // it in memory pair by pair.
      float *pc = syn1neg + (int64_t)centre * dim;
      Row<VEC> crow;
      load_row<VEC>(pc, dim, lane, full, crow);
      const bool hub_centre = hub_rows > 0 && centre < hub_rows;
      Row<VEC> crow0 = crow;  // as loaded (hub rows go back as an atomic add of the difference)
      constexpr int KP = VEC <= 4 ? 5 : (VEC == 8 ? 3 : 1);  // negative rows in flight
      struct PairBuf {
        Row<VEC> row1;
        Row<VEC> rows[KP];
        int32_t tg[KP];
        int j;
        int rowi;          // kRing: the ring row of the context word
        bool late_row1;    // the pair trained just before writes this row: load it afterwards
        bool late[KP];
      };
      // start every row load of pair (i, j): the context row and the first KP negatives
      // Start the row loads of pair (i, j): the context row and the first KP negatives.
      // `jprev` >= 0 names the pair that will be trained between this request and the use
      // of the rows (lookahead): a row that pair writes -- its context row, or any of its
      // K negatives -- must not be read early, so it is marked `late` and loaded when the
      // pair is processed.  With that, lookahead order == sequential order, bit for bit.
      constexpr bool kAhead = kDepth > 0;
      auto issue_negs = [&](int j, int jprev, int jprev2, PairBuf &B) {
        const int32_t *ng = neg + (j - lo) * K;
#pragma unroll
        for (int e = 0; e < KP; ++e) {
          // targets are wave-uniform; an LDS load is lane-varying to the compiler
          B.tg[e] = (e < K) ? __builtin_amdgcn_readfirstlane(ng[e]) : centre;  // centre = "skip"
          B.late[e] = false;
        }
        for (int which = 0; which < 2; ++which) {
          const int jp = which == 0 ? jprev : jprev2;
          if (jp < 0) continue;
          const int32_t *ngp = neg + (jp - lo) * K;
          for (int d = 0; d < K; ++d) {
            const int32_t tp = __builtin_amdgcn_readfirstlane(ngp[d]);
#pragma unroll
            for (int e = 0; e < KP; ++e) B.late[e] = B.late[e] || B.tg[e] == tp;
          }
        }
#pragma unroll
        for (int e = 0; e < KP; ++e)
          if (B.tg[e] != centre && !B.late[e])
            load_row<VEC>(syn1neg + (int64_t)B.tg[e] * dim, dim, lane, full, B.rows[e]);
      };
      auto issue = [&](int j, int jprev, int jprev2, PairBuf &B) {
        B.j = j;
        const int32_t cj = __builtin_amdgcn_readfirstlane(sent[j]);
        B.late_row1 = (jprev >= 0 && cj == __builtin_amdgcn_readfirstlane(sent[jprev])) ||
                      (jprev2 >= 0 && cj == __builtin_amdgcn_readfirstlane(sent[jprev2]));
        B.rowi = 0;
        if (kRing) {
          B.rowi = __builtin_amdgcn_readlane(pos_row, j & 15);
          if (!B.late_row1) ring_get(B.rowi, B.row1);
        } else if (!B.late_row1) {
          load_row<VEC>(syn0 + (int64_t)cj * dim, dim, lane, full, B.row1);
        }
        if (kAhead) issue_negs(j, jprev, jprev2, B);  // strict order: after the centre word
      };
      // one negative target: f, sigma, the two FMAs, store (word2vec's inner body, label 0)
      auto train_negative = [&](Row<VEC> &row1, Row<VEC> &work, Row<VEC> &row2, float *p2) {
        const float f = wave_dot<VEC>(row1, row2);
        if (f <= -6.0f || f >= 6.0f) return;
        const float g = (0.0f - exp_lds[(int)((f + 6.0f) * 83.0f)]) * alpha;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          work.v[v] = __fmaf_rn(g, row2.v[v], work.v[v]);
          row2.v[v] = __fmaf_rn(g, row1.v[v], row2.v[v]);
        }
        if (hub_rows > 0 && (p2 - syn1neg) < hub_span) {
          // a hub row (the vocabulary is in descending count order): hundreds of waves hold it at
          // any moment and a plain store would overwrite what they learned -- add this wave's
          // contribution instead
#pragma unroll
          for (int v = 0; v < VEC; ++v)
            if (lane * VEC + v < dim) unsafeAtomicAdd(p2 + lane * VEC + v, g * row1.v[v]);
        } else {
          store_row<VEC>(p2, dim, lane, full, row2);
        }
      };
      auto process = [&](PairBuf &A) {
        const int j = A.j;
        float *p1 = syn0 + (int64_t)__builtin_amdgcn_readfirstlane(sent[j]) * dim;
        if (A.late_row1) {
          if (kRing)
            ring_get(A.rowi, A.row1);
          else
            load_row<VEC>(p1, dim, lane, full, A.row1);
        }
        Row<VEC> work;
#pragma unroll
        for (int v = 0; v < VEC; ++v) work.v[v] = 0.0f;
        {  // d == 0: the centre word, label 1
          const float f = wave_dot<VEC>(A.row1, crow);
          if (!(f <= -6.0f || f >= 6.0f)) {
            const float g = (1.0f - exp_lds[(int)((f + 6.0f) * 83.0f)]) * alpha;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
              work.v[v] = __fmaf_rn(g, crow.v[v], work.v[v]);
              crow.v[v] = __fmaf_rn(g, A.row1.v[v], crow.v[v]);
            }
          }
        }
        if (!kAhead) issue_negs(j, -1, -1, A);
#pragma unroll
        for (int e = 0; e < KP; ++e) {  // first group: rows already in flight
          if (A.tg[e] == centre) continue;  // drawn the centre word, or past K
          float *p2 = syn1neg + (int64_t)A.tg[e] * dim;
          bool dup = A.late[e];  // written by the previous pair, or earlier in this group:
#pragma unroll                   // read the update back
          for (int e2 = 0; e2 < e; ++e2) dup = dup || A.tg[e2] == A.tg[e];
          if (dup) load_row<VEC>(p2, dim, lane, full, A.rows[e]);
          train_negative(A.row1, work, A.rows[e], p2);
        }
        const int32_t *ng = neg + (j - lo) * K;
        for (int d0 = KP; d0 < K; d0 += KP) {  // further groups (negative > KP)
          int32_t tg[KP];
          Row<VEC> rows[KP];
#pragma unroll
          for (int e = 0; e < KP; ++e) {
            tg[e] = (d0 + e < K) ? __builtin_amdgcn_readfirstlane(ng[d0 + e]) : centre;
            if (tg[e] != centre) load_row<VEC>(syn1neg + (int64_t)tg[e] * dim, dim, lane, full, rows[e]);
          }
#pragma unroll
          for (int e = 0; e < KP; ++e) {
            if (tg[e] == centre) continue;
            float *p2 = syn1neg + (int64_t)tg[e] * dim;
            bool dup = false;
#pragma unroll
            for (int e2 = 0; e2 < e; ++e2) dup = dup || tg[e2] == tg[e];
            if (dup) load_row<VEC>(p2, dim, lane, full, rows[e]);
            train_negative(A.row1, work, rows[e], p2);
          }
        }
        if (!kRing && hub_rows > 0 && (p1 - syn0) < hub_span) {
#pragma unroll
          for (int v = 0; v < VEC; ++v)
            if (lane * VEC + v < dim) unsafeAtomicAdd(p1 + lane * VEC + v, work.v[v]);
        } else {
#pragma unroll
          for (int v = 0; v < VEC; ++v) A.row1.v[v] = A.row1.v[v] + work.v[v];
          if (kRing)
            ring_put(A.rowi, A.row1);
          else
            store_row<VEC>(p1, dim, lane, full, A.row1);
        }
        ++pairs;
      };
      int j = lo + (lo == i ? 1 : 0);
      if (kAhead) {
        // The kernel is latency-bound (waves spend ~87 % of their cycles waiting on
        // memory), so the next pair's rows are requested before this pair is trained:
        // two pairs' worth of loads in flight per wave.  Rows the pair in between writes
        // are excluded from the early request (see issue), so the result is identical to
        // training strictly in order -- deterministic mode uses the same path.
        auto next_pair = [&](int jj) { return jj + 1 == i ? jj + 2 : jj + 1; };
        PairBuf bufA, bufB;
        if (j < hi) issue(j, -1, -1, bufA);
        while (j < hi) {
          const int jn = next_pair(j);
          if (jn < hi) issue(jn, j, -1, bufB);
          process(bufA);
          bufA = bufB;
          j = jn;
        }
      } else {
        PairBuf buf;
        while (j < hi) {
          issue(j, -1, -1, buf);
          process(buf);
          ++j;
          if (j == i) ++j;
        }
      }
      if (hub_centre) {
#pragma unroll
        for (int v = 0; v < VEC; ++v)
          if (lane * VEC + v < dim) unsafeAtomicAdd(pc + lane * VEC + v, crow.v[v] - crow0.v[v]);
      } else {
        store_row<VEC>(pc, dim, lane, full, crow);
      }
      if (kRing) {
        if (ring_in_row >= 0) {
          ring_put(ring_in_row, ring_in);
          ring_loaded(ring_in_row, ring_in);
        }
        if (i - window >= 0) ring_leave(i - window);
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (kRing)
      for (int j = max(0, nf - window); j < nf; ++j) ring_leave(j);
  }
  if (pairs_out && lane == 0 && pairs) atomicAdd(pairs_out, pairs);
}

}  // namespace n2v

namespace n2v {
// index[b] = bisect_left(cum_table, b << (31 - bits)), b = 0 .. 2^bits (2^bits + 1 entries)
__global__ __launch_bounds__(256) void cum_index_kernel(const uint32_t *__restrict__ cum_table,
                                                        int64_t n_vocab, int bits,
                                                        int32_t *__restrict__ index) {
  const int64_t n = (1ll << bits) + 1;
  const int iters = 64 - __clzll((long long)n_vocab);
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n;
       b += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t x = (uint64_t)b << (31 - bits);
    index[b] = x > 0x7fffffffull ? (int32_t)n_vocab
                                 : bisect_left_u32(cum_table, n_vocab, (uint32_t)x, iters);
  }
}
}  // namespace n2v

extern "C" int n2v_cum_index_build(const uint32_t *cum_table, int64_t n_vocab, int32_t bits,
                                   int32_t *index_out, void *stream) {
  if (!cum_table || !index_out || n_vocab < 1 || n_vocab >= (1ll << 31) || bits < 1 || bits > 30)
    return N2V_EINVAL;
  const int64_t n = (1ll << bits) + 1;
  int64_t blocks = (n + 255) / 256;
  const int64_t cap = n2v::resident_blocks((const void *)n2v::cum_index_kernel, 256, 0) * 2;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(n2v::cum_index_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, cum_table, n_vocab, bits, index_out);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}

// the opt-in batched variant (n2v_sgns_batched.hip)
extern "C" int n2v_sgns_batched_launch(const int32_t *walks, int64_t n_walks, int32_t walk_len,
                                       float *syn0, float *syn1neg, const uint32_t *cum_table,
                                       const uint32_t *sample_int, const float *exp_table,
                                       const n2v_sgns_params *P, unsigned long long *pairs_out,
                                       void *stream);

// n2v_sgns_train, or (dry_waves != NULL) only its launch geometry: the waves it would keep in flight
static int sgns_train_impl(const int32_t *walks, int64_t n_walks, int32_t walk_len,
                           float *syn0, float *syn1neg, const uint32_t *cum_table,
                           const uint32_t *sample_int, const float *exp_table,
                           const n2v_sgns_params *P, unsigned long long *pairs_out,
                           void *stream, int64_t *dry_waves) {
  if (!P) return N2V_EINVAL;
  if (!dry_waves && (!walks || !syn0 || !syn1neg || !cum_table || !exp_table)) return N2V_EINVAL;
  if (n_walks < 0 || walk_len < 1 || walk_len > N2V_SGNS_MAX_SENTENCE) return N2V_EINVAL;
  if (P->n_vocab < 1 || P->dim < 1 || P->dim > 1024 || P->window < 1 || P->window > 32 ||
      P->negative < 1 || P->negative > 32)
    return N2V_EINVAL;
  if (P->batched != 0 && P->batched != 1) return N2V_EINVAL;
  if (dry_waves) *dry_waves = 0;
  if (n_walks == 0) return N2V_OK;
  if (P->batched) {
    if (dry_waves) return N2V_OK;  // (the batched trainer runs plain stores: nobody asks)
    return n2v_sgns_batched_launch(walks, n_walks, walk_len, syn0, syn1neg, cum_table,
                                   sample_int, exp_table, P, pairs_out, stream);
  }
  using namespace n2v;
  int V = 1;
  while (64 * V < P->dim) V *= 2;
  // the window cache (kRing): rows of 64 * V floats, 2 * window + 2 of them (12 or 16), when the
  // dimension fills the wave exactly and the ring fits beside the other per-wave buffers
  const int ring_rows = 2 * P->window + 2 <= 12 ? 12 : 16;
  const bool ring_fits = P->dim == 64 * V && V <= 2 && 2 * P->window + 2 <= 16;
  if (P->window_cache != 0 && P->window_cache != 1) return N2V_EINVAL;
  if (P->window_cache == 1 && !ring_fits) return N2V_EINVAL;
  if (P->hub_rows < 0) return N2V_EINVAL;
  const bool use_ring = P->window_cache == 1;
  const int sent_cap = (walk_len + 3) & ~3;
  const int ints_per_wave = (2 * sent_cap + (2 * P->window + 1) * P->negative + 3) & ~3;
  const size_t lds = kExpTable * sizeof(float) + (P->cum_index ? 0 : (kBuckets + 1 + 3) * sizeof(int32_t)) +
                     (size_t)kSgnsWaves * ((size_t)ints_per_wave + (use_ring ? (size_t)ring_rows * 64 * V : 0)) * 4;
  // Hogwild concurrency is scaled to the model: unsynchronised waves are harmless
  // while collisions on a row are rare (gensim runs <= 16 threads); on a tiny
  // vocabulary thousands of racing waves would overwrite each other's updates.
  // One wave per 32 vocabulary rows, up to the whole chip (8192 waves >= 256 K rows).
  int64_t waves = P->n_vocab / 32;
  if (waves < 1) waves = 1;
  if (waves > n_walks) waves = n_walks;
  if (P->max_waves > 0 && waves > P->max_waves) waves = P->max_waves;
  int64_t blocks = (waves + kSgnsWaves - 1) / kSgnsWaves;
  if (P->cum_index && (P->cum_index_bits < 1 || P->cum_index_bits > 30)) return N2V_EINVAL;
#ifdef N2V_SGNS_TUNE
  if (const char *e = getenv("N2V_SGNS_BLOCKS_PER_CU")) { int64_t cap = 256 * (int64_t)atoi(e); if (blocks > cap) blocks = cap; }
#endif
  dim3 block(kSgnsWaves * 64);
  if (waves < kSgnsWaves) block = dim3((unsigned)waves * 64);
  if (P->deterministic) {
    blocks = 1;
    block = dim3(64);
  }
  hipStream_t st = (hipStream_t)stream;
  // pairs_out[1] is the kernel's row counter: start it at zero on the same stream
  if (!dry_waves && pairs_out && hipMemsetAsync(pairs_out + 1, 0, sizeof(unsigned long long), st) != hipSuccess)
    return N2V_ELAUNCH;
  // lookahead depth: 1 pair for dim <= 512, none above (registers).  Depth 2 was measured
  // in rounds 2 and 3 (ring variant) and lost every time.
#ifndef N2V_SGNS_DEPTH
#define N2V_SGNS_DEPTH(VV) ((VV) <= 8 ? 1 : 0)
#endif
#define N2V_LAUNCH_R(VV, RR)                                                                  \
  do {                                                                                       \
    constexpr int kD = N2V_SGNS_DEPTH(VV);                   \
    const void *fn = (const void *)sgns_kernel<VV, kD, RR>;                                   \
    if (lds > 64 * 1024 &&                                                                   \
        hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
      return N2V_ELAUNCH;                                                                    \
    if (!P->deterministic) {                                                                 \
      const int64_t cap = resident_blocks(fn, (int)block.x, lds);                            \
      if (blocks > cap) blocks = cap;                                                        \
    }                                                                                        \
    if (dry_waves) {                                                                         \
      *dry_waves = blocks * (int64_t)(block.x / 64);                                         \
      break;                                                                                 \
    }                                                                                        \
    hipLaunchKernelGGL((sgns_kernel<VV, kD, RR>), dim3((unsigned)blocks), block, lds, st,     \
                       walks, n_walks, walk_len, syn0, syn1neg, cum_table, sample_int,        \
                       exp_table, *P, pairs_out, sent_cap);                                   \
  } while (0)
#define N2V_LAUNCH_RING(VV)         \
  do {                              \
    if (!use_ring)                  \
      N2V_LAUNCH_R(VV, 0);          \
    else if (ring_rows == 12)       \
      N2V_LAUNCH_R(VV, 12);         \
    else                            \
      N2V_LAUNCH_R(VV, 16);         \
  } while (0)
  switch (V) {
    case 1: N2V_LAUNCH_RING(1); break;
    case 2: N2V_LAUNCH_RING(2); break;
    case 4: N2V_LAUNCH_R(4, 0); break;
    case 8: N2V_LAUNCH_R(8, 0); break;
    default: N2V_LAUNCH_R(16, 0); break;
  }
#undef N2V_LAUNCH_RING
#undef N2V_LAUNCH_R
  if (dry_waves) return N2V_OK;
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}

extern "C" int n2v_sgns_train(const int32_t *walks, int64_t n_walks, int32_t walk_len,
                              float *syn0, float *syn1neg, const uint32_t *cum_table,
                              const uint32_t *sample_int, const float *exp_table,
                              const n2v_sgns_params *P, unsigned long long *pairs_out,
                              void *stream) {
  return sgns_train_impl(walks, n_walks, walk_len, syn0, syn1neg, cum_table, sample_int, exp_table, P,
                         pairs_out, stream, nullptr);
}

extern "C" int64_t n2v_sgns_hogwild_waves(const n2v_sgns_params *P, int64_t n_walks, int32_t walk_len) {
  int64_t waves = 0;
  const int rc = sgns_train_impl(nullptr, n_walks, walk_len, nullptr, nullptr, nullptr, nullptr, nullptr, P,
                                 nullptr, nullptr, &waves);
  return rc == N2V_OK ? waves : (int64_t)rc;
}

namespace n2v {
// gensim's rate of the job of each row (word2vec.py _job_producer / _get_next_alpha): the same
// double operations in the same order, then the cast to fp32
__global__ __launch_bounds__(256) void job_alpha_kernel(int job_rows, int epoch, int epochs, int64_t row0,
                                                       int64_t rows, double alpha0, double alpha_min,
                                                       int64_t n, float *__restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t job = (row0 + i) / (int64_t)job_rows;
    const double pushed = (double)(job * (int64_t)job_rows);
    const double epoch_progress = 1.0 * pushed / (double)rows;
    const double progress = ((double)epoch + epoch_progress) / (double)epochs;
    const double next_alpha = alpha0 - (alpha0 - alpha_min) * progress;
    out[i] = (float)(alpha_min > next_alpha ? alpha_min : next_alpha);  // max(end_alpha, next_alpha)
  }
}
}  // namespace n2v

extern "C" int n2v_sgns_job_alpha(int32_t job_rows, int32_t epoch, int32_t epochs, int64_t row0,
                                  int64_t rows, double alpha0, double alpha_min, int64_t n, float *out,
                                  void *stream) {
  if (job_rows < 1 || epoch < 0 || epochs < 1 || row0 < 0 || rows < 1 || n < 0) return N2V_EINVAL;
  if (n == 0) return N2V_OK;
  if (!out) return N2V_EINVAL;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(n2v::job_alpha_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (int)job_rows, (int)epoch, (int)epochs, row0, rows, alpha0, alpha_min, n, out);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}
