/*
 * n2v_oracle.c -- CPU restatement of the reference's walk path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY (see n2v_oracle.h).  Parity: pinned against
 * tests/golden/ JSON files, generated from the reference's node2vec/randomwalk.py.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: every fp64 operation
 * below must round exactly like CPython's float arithmetic does).
 */
#include "n2v_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* randomwalk.py:157-190 generate_alias_tables                               */
/* ------------------------------------------------------------------------ */
static int alias_tables_ws(const double *node_weights, int64_t n,
                           int32_t *alias, double *probs, int32_t *underfull,
                           int32_t *overfull) {
  if (n <= 0) return N2V_ORACLE_EZERODIV; /* sum([]) / 0, randomwalk.py:172 */
  /* :171 alias = [0]*n ; :172 avg = sum(w)/n (left-to-right fp64 sum) */
  double total = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    alias[i] = 0;
    total = total + node_weights[i];
  }
  double avg_weight = total / (double)n;
  if (avg_weight == 0.0) return N2V_ORACLE_EZERODIV; /* x / 0.0, :173 */
  /* :173 probs = [x / avg for x in w] */
  for (int64_t i = 0; i < n; ++i) probs[i] = node_weights[i] / avg_weight;

  /* :175-180 ascending split into two stacks */
  int64_t nu = 0, no = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (probs[i] < 1.0)
      underfull[nu++] = (int32_t)i;
    else
      overfull[no++] = (int32_t)i;
  }
  /* :182-189 LIFO pairing; leftovers keep alias 0 and their current probs */
  while (nu > 0 && no > 0) {
    int32_t under = underfull[--nu];
    int32_t over = overfull[--no];
    alias[under] = over;
    probs[over] = probs[over] + probs[under] - 1.0;
    if (probs[over] < 1.0)
      underfull[nu++] = over;
    else
      overfull[no++] = over;
  }
  return N2V_ORACLE_OK;
}

int n2v_oracle_alias_tables(const double *node_weights, int64_t n,
                            int32_t *alias, double *probs) {
  if (n <= 0) return N2V_ORACLE_EZERODIV;
  int32_t *ws = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)n);
  if (!ws) return N2V_ORACLE_ENOMEM;
  int rc = alias_tables_ws(node_weights, n, alias, probs, ws, ws + n);
  free(ws);
  return rc;
}

/* weight of edge e as the fp64 the reference computes with (randomwalk.py:20 keeps Python
 * floats): fp64 storage, else fp32 storage widened, else 1.0 (indexer.py:20-21) */
static inline double csr_weight(const n2v_oracle_csr *g, int64_t e) {
  if (g->w64) return g->w64[e];
  return g->w ? (double)g->w[e] : 1.0;
}

/* `x in src_nbs_id` (randomwalk.py:226) on the sorted neighbour list */
static int contains_sorted(const int32_t *a, int64_t n, int32_t x) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    int64_t mid = lo + ((hi - lo) >> 1);
    if (a[mid] < x)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo < n && a[lo] == x;
}

/* randomwalk.py:219-231: the p/q bias of one neighbour list */
static void edge_bias(int64_t src_id, const int32_t *src_nbs, int64_t n_src_nbs,
                      const int32_t *dst_ids, const double *dst_w, int64_t n,
                      double return_param, double inout_param, double *out) {
  for (int64_t i = 0; i < n; ++i) {
    int32_t dst_neighbor_id = dst_ids[i];
    double weight = dst_w[i];
    double unnorm_prob;
    if ((int64_t)dst_neighbor_id == src_id) /* :223 go back to the src */
      unnorm_prob = weight / return_param;
    else if (contains_sorted(src_nbs, n_src_nbs, dst_neighbor_id)) /* :226 */
      unnorm_prob = weight;
    else /* :229 a brand new vertex */
      unnorm_prob = weight / inout_param;
    out[i] = unnorm_prob;
  }
}

/* ------------------------------------------------------------------------ */
/* randomwalk.py:193-232 generate_edge_alias_tables                          */
/* ------------------------------------------------------------------------ */
int n2v_oracle_edge_alias_tables(int64_t src_id, const int32_t *src_nbs,
                                 int64_t n_src_nbs, const int32_t *dst_ids,
                                 const double *dst_w, int64_t n_ids,
                                 int64_t n_w, double return_param,
                                 double inout_param, int32_t *alias,
                                 double *probs) {
  if (n_ids != n_w) return N2V_ORACLE_EINVAL; /* :212-213 */
  if (return_param == 0 || inout_param == 0) return N2V_ORACLE_EINVAL; /* :214 */
  int64_t n = n_ids;
  if (n <= 0) return N2V_ORACLE_EZERODIV;
  double *biased = (double *)malloc(sizeof(double) * (size_t)n);
  if (!biased) return N2V_ORACLE_ENOMEM;
  edge_bias(src_id, src_nbs, n_src_nbs, dst_ids, dst_w, n, return_param,
            inout_param, biased);
  int rc = n2v_oracle_alias_tables(biased, n, alias, probs); /* :232 */
  free(biased);
  return rc;
}

/* randomwalk.py:86-99 */
int64_t n2v_oracle_sampling_from_alias(const int32_t *alias,
                                       const double *probs, int64_t n,
                                       double first_random,
                                       double second_random) {
  int64_t pick = (int64_t)(first_random * (double)n); /* int(r1 * len) */
  if (second_random < probs[pick]) return pick;
  return alias[pick];
}

/* randomwalk.py:70-84 */
int64_t n2v_oracle_sampling_from_alias_wiki(const int32_t *alias,
                                            const double *probs, int64_t n,
                                            double first_random) {
  int64_t pick = (int64_t)((double)n * first_random);
  double y = (double)n * first_random - (double)pick;
  if (y < probs[pick]) return pick;
  return alias[pick];
}

/* randomwalk.py:123-153 */
int n2v_oracle_path_append(int64_t *path, int64_t *len,
                           const int32_t *dst_neighbors, const int32_t *alias,
                           const double *probs, int64_t n, double first_random,
                           int use_second, double second_random) {
  int64_t next_index =
      use_second ? n2v_oracle_sampling_from_alias(alias, probs, n, first_random,
                                                  second_random)
                 : n2v_oracle_sampling_from_alias_wiki(alias, probs, n,
                                                       first_random);
  int64_t next_vertex = dst_neighbors[next_index];
  if (*len == 2 && path[0] < 0) { /* :146-147 first step */
    path[0] = path[1];
    path[1] = next_vertex;
  } else { /* :150 */
    path[*len] = next_vertex;
    *len += 1;
  }
  return N2V_ORACLE_OK;
}

/* ------------------------------------------------------------------------ */
/* Counter-based uniform stream (the build's own; DESIGN.md "RNG")           */
/* ------------------------------------------------------------------------ */
static inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

void n2v_oracle_uniform_bits(uint64_t seed, uint64_t walk_key, uint32_t step,
                             uint32_t *u1, uint32_t *u2) {
  uint64_t h0 = mix64(seed ^ mix64(walk_key + 0x9E3779B97F4A7C15ULL));
  uint64_t h = mix64(h0 + ((uint64_t)step + 1ULL) * 0xD1B54A32D192ED03ULL);
  *u1 = (uint32_t)(h >> 32);
  *u2 = (uint32_t)h;
}

/* ------------------------------------------------------------------------ */
/* fugue.py:130-155 random_walk, one walker                                  */
/* ------------------------------------------------------------------------ */
typedef struct {
  double *wd;     /* widened weights of the current neighbour list */
  double *biased; /* edge_bias output */
  double *probs;
  int32_t *alias;
  int32_t *stk; /* 2 * cap */
  int64_t cap;
} walk_ws;

static int ws_reserve(walk_ws *ws, int64_t n) {
  if (n <= ws->cap) return 0;
  int64_t cap = ws->cap ? ws->cap : 64;
  while (cap < n) cap *= 2;
  free(ws->wd);
  free(ws->biased);
  free(ws->probs);
  free(ws->alias);
  free(ws->stk);
  ws->wd = (double *)malloc(sizeof(double) * (size_t)cap);
  ws->biased = (double *)malloc(sizeof(double) * (size_t)cap);
  ws->probs = (double *)malloc(sizeof(double) * (size_t)cap);
  ws->alias = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
  ws->stk = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)cap);
  ws->cap = cap;
  return (ws->wd && ws->biased && ws->probs && ws->alias && ws->stk) ? 0 : -1;
}

static void ws_free(walk_ws *ws) {
  free(ws->wd);
  free(ws->biased);
  free(ws->probs);
  free(ws->alias);
  free(ws->stk);
  memset(ws, 0, sizeof(*ws));
}

/* one walker = initiate_random_walk row (randomwalk.py:291-296) followed by
 * walk_length calls of the next_step_random_walk body (:316-339). */
static int walk_one(const n2v_oracle_csr *g, int32_t start, int32_t ordinal,
                    int32_t num_walks, int32_t walk_length, double p, double q,
                    uint64_t seed, walk_ws *ws, int32_t *out, uint8_t *valid) {
  const int64_t L1 = (int64_t)walk_length + 1;
  *valid = 0;
  for (int64_t t = 0; t < L1; ++t) out[t] = -1;
  if (start < 0 || start >= g->n_vertices) return N2V_ORACLE_EINVAL;
  /* fugue.py:132 walk_start = df_adj[["id"]]: only vertices with out-edges */
  if (g->rowptr[start + 1] == g->rowptr[start]) return N2V_ORACLE_OK;

  /* randomwalk.py:295 {"src": -i, "dst": v, "path": [-i, v]} */
  int64_t src = -(int64_t)ordinal;
  int64_t dst = start;
  int64_t len = 2; /* logical len(path) */
  int64_t path0 = src;
  uint64_t walk_key = (uint64_t)start * (uint64_t)num_walks + (uint64_t)(ordinal - 1);
  out[0] = start; /* becomes path[0] after the first-step rule */
  int64_t n_out = 1;

  for (int32_t step = 0; step < walk_length; ++step) {
    int64_t vb = g->rowptr[dst], n = g->rowptr[dst + 1] - vb;
    /* fugue.py:147 inner_join(df_dst): no adjacency row => walker vanishes */
    if (n == 0) return N2V_ORACLE_OK;
    if (ws_reserve(ws, n)) return N2V_ORACLE_ENOMEM;
    const int32_t *dst_ids = g->col + vb;
    for (int64_t i = 0; i < n; ++i) ws->wd[i] = csr_weight(g, vb + i);
    int rc;
    if (src < 0) { /* randomwalk.py:320-321 */
      rc = alias_tables_ws(ws->wd, n, ws->alias, ws->probs, ws->stk, ws->stk + n);
    } else { /* :322-332 */
      if (p == 0 || q == 0) return N2V_ORACLE_EINVAL;
      int64_t sb = g->rowptr[src], m = g->rowptr[src + 1] - sb;
      edge_bias(src, g->col + sb, m, dst_ids, ws->wd, n, p, q, ws->biased);
      rc = alias_tables_ws(ws->biased, n, ws->alias, ws->probs, ws->stk, ws->stk + n);
    }
    if (rc != N2V_ORACLE_OK) return rc;
    /* :336-337 two uniforms; r = u / 2^32 is exact in fp64 */
    uint32_t u1, u2;
    n2v_oracle_uniform_bits(seed, walk_key, (uint32_t)step, &u1, &u2);
    double r1 = (double)u1 * (1.0 / 4294967296.0);
    double r2 = (double)u2 * (1.0 / 4294967296.0);
    int64_t idx = n2v_oracle_sampling_from_alias(ws->alias, ws->probs, n, r1, r2);
    int64_t next_vertex = dst_ids[idx];
    /* :146-153 first-step rule: [-i, v] -> [v, next]; else append */
    if (len == 2 && path0 < 0) {
      path0 = dst;
      out[0] = (int32_t)dst;
      out[1] = (int32_t)next_vertex;
      n_out = 2;
    } else {
      out[n_out++] = (int32_t)next_vertex;
      len += 1;
    }
    /* :339 src = path[-2], dst = path[-1] */
    src = dst;
    dst = next_vertex;
  }
  (void)n_out;
  *valid = 1;
  return N2V_ORACLE_OK;
}

int n2v_oracle_random_walk(const n2v_oracle_csr *g, const int32_t *start_ids,
                           int64_t n_start, int32_t num_walks,
                           int32_t walk_length, double return_param,
                           double inout_param, uint64_t seed,
                           int32_t *walks_out, uint8_t *valid_out,
                           int32_t n_threads) {
  if (!g || !start_ids || num_walks < 0 || walk_length < 0) return N2V_ORACLE_EINVAL;
  const int64_t L1 = (int64_t)walk_length + 1;
  const int64_t total = n_start * (int64_t)num_walks;
  int status = N2V_ORACLE_OK;
  if (n_threads < 1) n_threads = 1;
#pragma omp parallel num_threads(n_threads)
  {
    walk_ws ws;
    memset(&ws, 0, sizeof(ws));
#pragma omp for schedule(dynamic, 4)
    for (int64_t r = 0; r < total; ++r) {
      int32_t start = start_ids[r / num_walks];
      int32_t ordinal = (int32_t)(r % num_walks) + 1; /* randomwalk.py:294 */
      int rc = walk_one(g, start, ordinal, num_walks, walk_length, return_param,
                        inout_param, seed, &ws, walks_out + r * L1, valid_out + r);
      if (rc != N2V_ORACLE_OK) {
#pragma omp critical
        status = rc;
      }
    }
    ws_free(&ws);
  }
  return status;
}

/* ------------------------------------------------------------------------ */
/* Exact transition probabilities (chi-square target)                        */
/* ------------------------------------------------------------------------ */
int n2v_oracle_transition_probs(const n2v_oracle_csr *g, int64_t s, int64_t v,
                                double return_param, double inout_param,
                                double *prob_out) {
  int64_t vb = g->rowptr[v], n = g->rowptr[v + 1] - vb;
  if (n <= 0) return N2V_ORACLE_EZERODIV;
  double *wd = (double *)malloc(sizeof(double) * (size_t)n);
  if (!wd) return N2V_ORACLE_ENOMEM;
  for (int64_t i = 0; i < n; ++i) wd[i] = csr_weight(g, vb + i);
  if (s >= 0) {
    int64_t sb = g->rowptr[s], m = g->rowptr[s + 1] - sb;
    edge_bias(s, g->col + sb, m, g->col + vb, wd, n, return_param, inout_param,
              prob_out);
  } else {
    memcpy(prob_out, wd, sizeof(double) * (size_t)n);
  }
  double total = 0.0;
  for (int64_t i = 0; i < n; ++i) total += prob_out[i];
  free(wd);
  if (total == 0.0) return N2V_ORACLE_EZERODIV;
  for (int64_t i = 0; i < n; ++i) prob_out[i] /= total;
  return N2V_ORACLE_OK;
}

/* ------------------------------------------------------------------------ */
/* randomwalk.py:238-262 trim_hotspot_vertices, on CSR rows                  */
/* ------------------------------------------------------------------------ */
/* The reference samples with pandas DataFrame.sample(n=cap, random_state=seed)
 * (numpy RandomState; cannot be replayed).  What it pins (tests/test_randomwalk.py:
 * 194-224, tests/test_fugue.py:23-28) is: exactly `cap` edges survive for a row
 * above the cap, they are a subset of the row, other rows and all weights are
 * untouched.  This restates the build's selection sampling (Knuth Algorithm S) on
 * the counter-based stream so that the HIP kernel can be compared bit for bit. */
static inline uint64_t mulhi64(uint64_t a, uint64_t b) {
  return (uint64_t)(((unsigned __int128)a * (unsigned __int128)b) >> 64);
}

int n2v_oracle_trim_mark(const int64_t *rowptr, int64_t n_rows, int64_t max_out_degree,
                         uint64_t seed, uint8_t *keep_out) {
  if (max_out_degree <= 0) max_out_degree = 100000; /* constants.py:6, randomwalk.py:252 */
  for (int64_t row = 0; row < n_rows; ++row) {
    int64_t b = rowptr[row], d = rowptr[row + 1] - b;
    if (d <= max_out_degree) { /* :254 */
      for (int64_t i = 0; i < d; ++i) keep_out[b + i] = 1;
      continue;
    }
    uint64_t h = mix64(seed ^ mix64((uint64_t)row + 0x2545F4914F6CDD1DULL));
    int64_t need = max_out_degree;
    for (int64_t i = 0; i < d; ++i) {
      uint64_t u = mix64(h + ((uint64_t)i + 1ULL) * 0x9FB21C651E98DF25ULL);
      int take = (int64_t)mulhi64(u, (uint64_t)(d - i)) < need;
      keep_out[b + i] = (uint8_t)take;
      need -= take;
    }
  }
  return N2V_ORACLE_OK;
}

/* ------------------------------------------------------------------------ */
/* Class counts of the table of randomwalk.py:219-231 on a unit-weight graph */
/* ------------------------------------------------------------------------ */
/* For every edge e = (s -> v): how many neighbours x of v take the branch x == s (:223)
 * and how many the branch x in src_nbs_id (:226), evaluated exactly as the reference does
 * (walk N(v), test each id).  Checker for n2v_edge_classes_build; same packing
 * (bits 0..23 shared, 24..31 return, saturating). */
int n2v_oracle_edge_classes(const n2v_oracle_csr *g, uint32_t *classes_out) {
  for (int64_t s = 0; s < g->n_vertices; ++s) {
    const int64_t sb = g->rowptr[s], ds = g->rowptr[s + 1] - sb;
    for (int64_t e = sb; e < sb + ds; ++e) {
      const int64_t v = g->col[e];
      const int64_t vb = g->rowptr[v], dv = g->rowptr[v + 1] - vb;
      uint32_t n_ret = 0, n_sh = 0;
      for (int64_t j = 0; j < dv; ++j) {
        const int32_t x = g->col[vb + j];
        if ((int64_t)x == s)
          ++n_ret;
        else if (contains_sorted(g->col + sb, ds, x))
          ++n_sh;
      }
      if (n_ret > 0xffu) n_ret = 0xffu;
      if (n_sh > 0xffffffu) n_sh = 0xffffffu;
      classes_out[e] = (n_ret << 24) | n_sh;
    }
  }
  return N2V_ORACLE_OK;
}
