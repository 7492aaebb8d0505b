/*
 * n2v_oracle_sgns.c -- single-thread CPU restatement of skip-gram negative sampling.
 *
 * TEST INFRASTRUCTURE ONLY (see n2v_oracle.h).
 *
 * PARITY UNPINNED.  The reference does not contain this arithmetic: its
 * embedding.py:126 forwards to gensim.models.Word2Vec (gensim ~= 3.8.2,
 * requirements.txt:27), which is neither vendored under /root/reference nor
 * installed here, and the reference's tests (tests/test_embedding.py:34-84) assert
 * only types and shapes.  This file restates the published word2vec / gensim-3.8
 * `train_batch_sg` + `fast_sentence_sg_neg` algorithm (DESIGN.md "SGNS"):
 *   - tokens outside the vocabulary and subsampled tokens are removed before
 *     windowing; keep iff sample_int[w] >= random_int32;
 *   - per centre position a reduced window b in [0, window) is drawn; contexts are
 *     j in [i - window + b, i + window - b], j != i;
 *   - the INPUT row is syn0[context j]; targets are syn1neg[centre i] (label 1)
 *     and `negative` draws from the cumulative count^0.75 table by
 *     bisect_left(cum_table, (r >> 16) % cum_table[-1]) (label 0); a negative equal
 *     to the centre word is skipped, not redrawn;
 *   - f = dot; |f| >= MAX_EXP(6) skips the target; sigma from the 1000-entry
 *     EXP_TABLE with index (int)((f + 6) * 83)  [EXP_TABLE_SIZE / MAX_EXP / 2 in
 *     integer arithmetic, as in word2vec.c and gensim's word2vec_inner.pyx];
 *   - g = (label - sigma) * alpha; work += g * syn1neg[t]; syn1neg[t] += g * syn0[j];
 *     after all targets syn0[j] += work.  All fp32.
 * Deviations, deliberate and documented: random draws come from the build's
 * counter-based stream (gensim's 48-bit LCG and numpy RandomState depend on its
 * thread/job batching and cannot be replayed), and the dot product is summed in
 * the order the wave64 kernel uses (BLAS sdot leaves the order unspecified), which
 * makes the deterministic GPU mode bit-identical to this file.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "n2v_oracle.h"

static inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

static inline uint64_t sentence_stream(uint64_t seed, uint64_t sentence_id) {
  return mix64(seed ^ mix64(sentence_id + 0xA0761D6478BD642FULL));
}

static inline uint64_t draw(uint64_t hs, uint64_t idx) {
  return mix64(hs + (idx + 1ULL) * 0xE7037ED1A0B428DBULL);
}

static int vec_width(int dim) {
  int v = 1;
  while (64 * v < dim) v *= 2;
  return v;
}

/* dot product in wave64 order: lane l owns elements l*V .. l*V+V-1 (fma chain),
 * then an xor-butterfly over lane distances 1,2,4,8,16,32 (a balanced tree over
 * adjacent lanes: what the kernel's DPP quad/row steps + row-sum combine compute) */
static float wave_dot(const float *a, const float *b, int dim, int V) {
  float p[64], t[64];
  for (int l = 0; l < 64; ++l) {
    float acc = 0.0f;
    for (int v = 0; v < V; ++v) {
      int e = l * V + v;
      if (e < dim) acc = fmaf(a[e], b[e], acc);
    }
    p[l] = acc;
  }
  for (int off = 1; off < 64; off <<= 1) {
    for (int l = 0; l < 64; ++l) t[l] = p[l] + p[l ^ off];
    memcpy(p, t, sizeof(p));
  }
  return p[0];
}

static int64_t bisect_left_u32(const uint32_t *a, int64_t n, uint32_t x) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (a[mid] < x)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

/* Trains rows [0, n_walks) in order, single thread.  Same contract as
 * n2v_sgns_train (include/n2v_hip.h) on host pointers.  Returns pairs trained. */
int64_t n2v_oracle_sgns_train(const int32_t *walks, int64_t n_walks, int32_t walk_len,
                              float *syn0, float *syn1neg, const uint32_t *cum_table,
                              const uint32_t *sample_int, const float *exp_table,
                              int64_t n_vocab, int64_t sentence_base, uint64_t seed,
                              int32_t dim, int32_t window, int32_t negative, float alpha) {
  if (walk_len > 256 || dim < 1 || dim > 1024 || window < 1 || negative < 1) return -1;
  const int V = vec_width(dim);
  int32_t sent[256];
  uint32_t red[256];
  float *work = (float *)malloc(sizeof(float) * (size_t)dim);
  int64_t pairs = 0;
  const uint32_t domain = cum_table[n_vocab - 1];
  for (int64_t r = 0; r < n_walks; ++r) {
    const uint64_t hs = sentence_stream(seed, (uint64_t)(sentence_base + r));
    /* sentence preparation: drop OOV + subsampled tokens, draw reduced windows */
    int nf = 0;
    for (int t = 0; t < walk_len; ++t) {
      int32_t tok = walks[r * walk_len + t];
      if (tok < 0 || tok >= n_vocab) continue;
      uint32_t rnd = (uint32_t)(draw(hs, 2ULL * (uint64_t)t) >> 32);
      if (sample_int && sample_int[tok] < rnd) continue;
      sent[nf] = tok;
      red[nf] = (uint32_t)(draw(hs, 2ULL * (uint64_t)t + 1ULL) >> 32) % (uint32_t)window;
      ++nf;
    }
    for (int i = 0; i < nf; ++i) {
      const int32_t centre = sent[i];
      int lo = i - window + (int)red[i];
      if (lo < 0) lo = 0;
      int hi = i + window + 1 - (int)red[i];
      if (hi > nf) hi = nf;
      for (int j = lo; j < hi; ++j) {
        if (j == i) continue;
        const int rel = j - i + window - (j > i ? 1 : 0);
        float *row1 = syn0 + (int64_t)sent[j] * dim;
        memset(work, 0, sizeof(float) * (size_t)dim);
        for (int d = 0; d <= negative; ++d) {
          int64_t target;
          float label;
          if (d == 0) {
            target = centre;
            label = 1.0f;
          } else {
            uint64_t idx = 2ULL * (uint64_t)walk_len +
                           ((uint64_t)i * 2ULL * (uint64_t)window + (uint64_t)rel) *
                               (uint64_t)negative + (uint64_t)(d - 1);
            uint32_t rr = (uint32_t)((draw(hs, idx) >> 16) % (uint64_t)domain);
            target = bisect_left_u32(cum_table, n_vocab, rr);
            if (target == centre) continue;
            label = 0.0f;
          }
          float *row2 = syn1neg + target * dim;
          float f = wave_dot(row1, row2, dim, V);
          if (f <= -6.0f || f >= 6.0f) continue;
          float s = exp_table[(int)((f + 6.0f) * 83.0f)];
          float g = (label - s) * alpha;
          for (int e = 0; e < dim; ++e) {
            float r2 = row2[e];
            work[e] = fmaf(g, r2, work[e]);
            row2[e] = fmaf(g, row1[e], r2);
          }
        }
        for (int e = 0; e < dim; ++e) row1[e] = row1[e] + work[e];
        ++pairs;
      }
    }
  }
  free(work);
  return pairs;
}

/* ------------------------------------------------------------------------------------------
 * The opt-in BATCHED variant (n2v_sgns_params.batched = 1; node2vec_amd/csrc/n2v_sgns_batched.hip).
 * NOT gensim's sampling: the `negative` draws are made once per centre position and shared by
 * its <= 2 * window (centre, context) pairs, and all pairs of a position are trained from one
 * snapshot of the rows (Ji et al., "Parallelizing Word2Vec in Shared and Distributed Memory",
 * 2016: the position becomes one small dense product).  Per position i with contexts J:
 *   context rows  = the distinct words of J (in the order of the kernel's LDS ring rows, which
 *                   the code below simulates), multiplicity mu_c
 *   target rows   = centre (label 1), then the distinct negative words != centre in draw
 *                   order, multiplicity mu_t (a repeated draw counts twice)
 *   F[c][t]       = syn0[c] . syn1neg[t]
 *   G[c][t]       = 0 if |F| >= 6 else ((label_t - sigma(F)) * alpha) * (mu_c * mu_t)
 *   syn1neg[t]   += sum_c G[c][t] * syn0_old[c];   syn0[c] += sum_t G[c][t] * syn1neg_old[t]
 * The summation orders below are those of the kernel's v_mfma_f32_16x16x4_f32 chains (an f32
 * MFMA is bit-for-bit a k-ordered fmaf chain), so the deterministic GPU mode is bit-identical.
 * Negative draws of position i use the indices the per-pair mode gives relative position 0. */
int64_t n2v_oracle_sgns_train_batched(const int32_t *walks, int64_t n_walks, int32_t walk_len,
                                      float *syn0, float *syn1neg, const uint32_t *cum_table,
                                      const uint32_t *sample_int, const float *exp_table,
                                      int64_t n_vocab, int64_t sentence_base, uint64_t seed,
                                      int32_t dim, int32_t window, int32_t negative, float alpha) {
  if (walk_len > 256 || dim < 4 || dim > 1024 || (dim & 3) || window < 1 || 2 * window + 1 > 16 ||
      negative < 1 || negative > 15)
    return -1;
  const int Q = dim / 4;
  const int KC = (2 * window + 1 <= 12) ? 3 : 4; /* k-steps over context rows */
  const int KT = (1 + negative <= 8) ? 2 : 4;    /* k-steps over target rows */
  int32_t sent[256];
  uint32_t red[256];
  float *cold = (float *)malloc(sizeof(float) * (size_t)dim * 16);
  float *told = (float *)malloc(sizeof(float) * (size_t)dim * 16);
  int64_t pairs = 0;
  const uint32_t domain = cum_table[n_vocab - 1];
  for (int64_t r = 0; r < n_walks; ++r) {
    const uint64_t hs = sentence_stream(seed, (uint64_t)(sentence_base + r));
    int nf = 0;
    for (int t = 0; t < walk_len; ++t) {
      int32_t tok = walks[r * walk_len + t];
      if (tok < 0 || tok >= n_vocab) continue;
      uint32_t rnd = (uint32_t)(draw(hs, 2ULL * (uint64_t)t) >> 32);
      if (sample_int && sample_int[tok] < rnd) continue;
      sent[nf] = tok;
      red[nf] = (uint32_t)(draw(hs, 2ULL * (uint64_t)t + 1ULL) >> 32) % (uint32_t)window;
      ++nf;
    }
    /* The kernel keeps the syn0 rows of the window in an LDS ring of `rrows` physical rows (one
     * per distinct word, shared by the positions that hold it, lowest free row first) and sums
     * over the context rows in ascending PHYSICAL row order: the ring is simulated here so that
     * the fmaf chains have the kernel's order.  Events: positions 0 .. window enter; then per
     * centre i: position i + 1 + window enters (the kernel prefetches it), i is trained,
     * position i - window leaves. */
    if (nf < 2) continue;
    const int rrows = (2 * window + 2 <= 12) ? 12 : 16;
    int32_t row_word[16];
    int row_ref[16], pos_row[256];
    for (int k = 0; k < 16; ++k) row_ref[k] = 0;
#define RING_ENTER(J)                                                        \
  do {                                                                       \
    int k_ = 0;                                                              \
    while (k_ < 16 && !(row_ref[k_] > 0 && row_word[k_] == sent[(J)])) ++k_; \
    if (k_ == 16) {                                                          \
      k_ = 0;                                                                \
      while (k_ < rrows && row_ref[k_] != 0) ++k_;                           \
      row_word[k_] = sent[(J)];                                              \
    }                                                                        \
    ++row_ref[k_];                                                           \
    pos_row[(J)] = k_;                                                       \
  } while (0)
    for (int j = 0; j <= window && j < nf; ++j) RING_ENTER(j);
    for (int i = 0; i < nf; ++i) {
      const int32_t centre = sent[i];
      int lo = i - window + (int)red[i];
      if (lo < 0) lo = 0;
      int hi = i + window + 1 - (int)red[i];
      if (hi > nf) hi = nf;
      if (i + 1 < nf && i + 1 + window < nf) RING_ENTER(i + 1 + window);
      int32_t uword[16], tword[16];
      int umult[16], tmult[16], nu = 0, nt = 0, npairs = 0;
      int cmr[16];
      for (int k = 0; k < 16; ++k) cmr[k] = 0;
      for (int j = lo; j < hi; ++j) {
        if (j == i) continue;
        ++npairs;
        ++cmr[pos_row[j]];
      }
      for (int k = 0; k < 16; ++k)
        if (cmr[k] > 0) {
          uword[nu] = row_word[k];
          umult[nu++] = cmr[k];
        }
      if (i - window >= 0) --row_ref[pos_row[i - window]]; /* leaves after this position; its
                                                              row is not reused before then */
      if (nu == 0) continue;
      tword[0] = centre;
      tmult[0] = 1;
      nt = 1;
      for (int d = 0; d < negative; ++d) {
        uint64_t idx = 2ULL * (uint64_t)walk_len +
                       ((uint64_t)i * 2ULL * (uint64_t)window) * (uint64_t)negative + (uint64_t)d;
        uint32_t rr = (uint32_t)((draw(hs, idx) >> 16) % (uint64_t)domain);
        int32_t target = (int32_t)bisect_left_u32(cum_table, n_vocab, rr);
        if (target == centre) continue;
        int k = 1;
        while (k < nt && tword[k] != target) ++k;
        if (k < nt) {
          ++tmult[k];
        } else {
          tword[nt] = target;
          tmult[nt++] = 1;
        }
      }
      float G[16][16];
      for (int u = 0; u < nu; ++u) memcpy(cold + (size_t)u * dim, syn0 + (int64_t)uword[u] * dim, sizeof(float) * (size_t)dim);
      for (int t = 0; t < nt; ++t) memcpy(told + (size_t)t * dim, syn1neg + (int64_t)tword[t] * dim, sizeof(float) * (size_t)dim);
      for (int u = 0; u < nu; ++u)
        for (int t = 0; t < nt; ++t) {
          const float *a = cold + (size_t)u * dim, *b = told + (size_t)t * dim;
          /* four interleaved chains (k-steps s = c mod 4 go to chain c), summed pairwise */
          float ch[4] = {0.0f, 0.0f, 0.0f, 0.0f};
          for (int s = 0; s < Q; ++s)
            for (int g = 0; g < 4; ++g) ch[s & 3] = fmaf(a[g * Q + s], b[g * Q + s], ch[s & 3]);
          float acc = (ch[0] + ch[1]) + (ch[2] + ch[3]);
          float gg = 0.0f;
          if (!(acc <= -6.0f || acc >= 6.0f)) {
            const float label = t == 0 ? 1.0f : 0.0f;
            gg = ((label - exp_table[(int)((acc + 6.0f) * 83.0f)]) * alpha) * (float)(umult[u] * tmult[t]);
          }
          G[u][t] = gg;
        }
      for (int t = 0; t < nt; ++t) { /* syn1neg[t] from the OLD context rows */
        float *row = syn1neg + (int64_t)tword[t] * dim;
        for (int e = 0; e < dim; ++e) {
          float acc = told[(size_t)t * dim + e];
          for (int s = 0; s < KC; ++s)
            for (int g = 0; g < 4; ++g) {
              const int u = KC == 3 ? 3 * g + s : 4 * g + s;
              if (u < nu) acc = fmaf(G[u][t], cold[(size_t)u * dim + e], acc);
            }
          row[e] = acc;
        }
      }
      for (int u = 0; u < nu; ++u) { /* syn0[c] from the OLD target rows */
        float *row = syn0 + (int64_t)uword[u] * dim;
        for (int e = 0; e < dim; ++e) {
          float acc = cold[(size_t)u * dim + e];
          for (int s = 0; s < KT; ++s)
            for (int g = 0; g < 4; ++g) {
              const int t = KT == 2 ? 2 * g + s : 4 * g + s;
              if (t < nt) acc = fmaf(G[u][t], told[(size_t)t * dim + e], acc);
            }
          row[e] = acc;
        }
      }
      pairs += npairs;
    }
#undef RING_ENTER
  }
  free(cold);
  free(told);
  return pairs;
}
