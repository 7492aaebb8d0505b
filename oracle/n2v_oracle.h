/*
 * n2v_oracle.h -- CPU restatement of the reference's node2vec walk path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product path (node2vec_amd/ + libn2v_hip.so) never
 * links, imports or falls back to it.
 *
 * Parity status: PINNED for the walk half.  Every function below is checked
 * against golden vectors produced by importing the reference's
 * node2vec/randomwalk.py (tests/golden/gen_golden.py, tests/golden/ JSON files) and
 * against the known-answer values of the reference's tests/test_randomwalk.py.
 * The SGNS half lives in n2v_oracle_sgns.c and is "parity unpinned" (its
 * arithmetic is third-party gensim, absent from /root/reference).
 *
 * All citations are file:line relative to /root/reference/.
 */
#ifndef N2V_ORACLE_H
#define N2V_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes */
#define N2V_ORACLE_OK 0
#define N2V_ORACLE_EINVAL (-1)   /* reference raises ValueError            */
#define N2V_ORACLE_EZERODIV (-2) /* reference raises ZeroDivisionError     */
#define N2V_ORACLE_ENOMEM (-3)

/* CSR view of the reference's df_adj (fugue.py:130): one row per source vertex,
 * neighbours sorted by dst id ascending (presort="dst"), multi-edges kept.
 * Weights: fp64 (`w64`, what the reference's Python floats are) or fp32 storage (`w`,
 * widened to fp64 before any arithmetic); both NULL = every weight 1.0. */
typedef struct {
  int64_t n_vertices;
  const int64_t *rowptr; /* [n_vertices + 1] */
  const int32_t *col;    /* [n_edges] */
  const float *w;        /* [n_edges] or NULL */
  const double *w64;     /* [n_edges] or NULL */
} n2v_oracle_csr;

/* randomwalk.py:157-190 generate_alias_tables */
int n2v_oracle_alias_tables(const double *node_weights, int64_t n,
                            int32_t *alias, double *probs);

/* randomwalk.py:193-232 generate_edge_alias_tables.  src_nbs must be sorted
 * ascending (it stands for the Python set of randomwalk.py:318). */
int n2v_oracle_edge_alias_tables(int64_t src_id, const int32_t *src_nbs,
                                 int64_t n_src_nbs, const int32_t *dst_ids,
                                 const double *dst_w, int64_t n_ids,
                                 int64_t n_w, double return_param,
                                 double inout_param, int32_t *alias,
                                 double *probs);

/* randomwalk.py:86-99 AliasProb.sampling_from_alias (two uniforms) */
int64_t n2v_oracle_sampling_from_alias(const int32_t *alias,
                                       const double *probs, int64_t n,
                                       double first_random,
                                       double second_random);

/* randomwalk.py:70-84 AliasProb.sampling_from_alias_wiki (one uniform) */
int64_t n2v_oracle_sampling_from_alias_wiki(const int32_t *alias,
                                            const double *probs, int64_t n,
                                            double first_random);

/* randomwalk.py:123-153 RandomPath.append.  path has room for *len + 1
 * entries.  use_second != 0 selects the two-uniform sampler. */
int n2v_oracle_path_append(int64_t *path, int64_t *len,
                           const int32_t *dst_neighbors, const int32_t *alias,
                           const double *probs, int64_t n, double first_random,
                           int use_second, double second_random);

/* The build's counter-based uniform stream (DESIGN.md "RNG").  Replaces the
 * two random.random() calls of randomwalk.py:336-337; replayed into the
 * reference by tests/golden/gen_golden.py. */
void n2v_oracle_uniform_bits(uint64_t seed, uint64_t walk_key, uint32_t step,
                             uint32_t *u1, uint32_t *u2);

/* fugue.py:130-155 random_walk + randomwalk.py:279-349, on a CSR graph.
 * walks_out: [n_start * num_walks, walk_length + 1] int32, row r =
 * start index r / num_walks, ordinal r % num_walks + 1.
 * valid_out[r] = 0 where the reference emits no row (start vertex without
 * out-edges, fugue.py:132; walker dropped at a sink, fugue.py:147).
 * n_threads > 1 runs walkers on OpenMP threads (results are identical: each
 * walker owns its own counter-based stream). */
int n2v_oracle_random_walk(const n2v_oracle_csr *g, const int32_t *start_ids,
                           int64_t n_start, int32_t num_walks,
                           int32_t walk_length, double return_param,
                           double inout_param, uint64_t seed,
                           int32_t *walks_out, uint8_t *valid_out,
                           int32_t n_threads);

/* Exact one-step transition distribution pi(x | s, v) per the bias rule of
 * randomwalk.py:220-231 (prob_out[i] for neighbour i of v; s < 0 = first
 * step).  Used as the chi-square target for the rejection ("fast") sampler. */
int n2v_oracle_transition_probs(const n2v_oracle_csr *g, int64_t s, int64_t v,
                                double return_param, double inout_param,
                                double *prob_out);

/* randomwalk.py:238-262 trim_hotspot_vertices on CSR rows: keep_out[e] = 1 for
 * surviving edges (exactly max_out_degree per row above the cap). */
int n2v_oracle_trim_mark(const int64_t *rowptr, int64_t n_rows, int64_t max_out_degree,
                         uint64_t seed, uint8_t *keep_out);

/* Per-edge class counts of the table of randomwalk.py:219-231 on a unit-weight graph
 * (checker for n2v_edge_classes_build of include/n2v_hip.h, same packing). */
int n2v_oracle_edge_classes(const n2v_oracle_csr *g, uint32_t *classes_out);

/* n2v_oracle_sgns.c (PARITY UNPINNED, see its header) */
int64_t n2v_oracle_sgns_train(const int32_t *walks, int64_t n_walks, int32_t walk_len,
                              float *syn0, float *syn1neg, const uint32_t *cum_table,
                              const uint32_t *sample_int, const float *exp_table,
                              int64_t n_vocab, int64_t sentence_base, uint64_t seed,
                              int32_t dim, int32_t window, int32_t negative, float alpha);
/* the opt-in batched variant (negatives shared by the pairs of a centre position) */
int64_t n2v_oracle_sgns_train_batched(const int32_t *walks, int64_t n_walks, int32_t walk_len,
                              float *syn0, float *syn1neg, const uint32_t *cum_table,
                              const uint32_t *sample_int, const float *exp_table,
                              int64_t n_vocab, int64_t sentence_base, uint64_t seed,
                              int32_t dim, int32_t window, int32_t negative, float alpha);

#ifdef __cplusplus
}
#endif
#endif
