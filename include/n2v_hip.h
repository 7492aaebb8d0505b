/*
 * n2v_hip.h -- C ABI of libn2v_hip.so, the MI355X (gfx950) node2vec hot path.
 *
 * The reference (graph-embedding/node2vec, node2vec-fugue 0.3.5) is pure Python
 * and has no FFI of its own; the hot path sits behind Python interfaces
 * (SURVEY.md 8b).  This header is the boundary a maintainer of the reference
 * binds to replace that path (ctypes stub: INTEGRATION.md).  Each entry point
 * cites the reference code it replaces, file:line relative to the reference.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (hipMalloc'd / torch CUDA storage) unless
 *    its name ends in _host; the caller owns all buffers;
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls
 *    only enqueue work, they never synchronise and never allocate;
 *  - return value: N2V_OK or a negative N2V_E* code for argument / launch
 *    errors detected on the host; data-dependent errors (the reference's
 *    ZeroDivisionError) are OR-ed into the device word status[0] as N2V_ST_*
 *    bits, to be read by the caller after it synchronises.  `status` points to
 *    FOUR uint32 words, zeroed by the caller: [0] status bits, [1] scratch (the exact walk
 *    kernels hand out walkers through it; the library resets it on the stream before a launch),
 *    [2..3] a 64-bit counter that N2V_WALK_FAST increments by its number of
 *    trials (draws, accepted or not; for the algorithmic-bytes accounting of DESIGN.md);
 *  - no global state: re-entrant, any number of streams / devices.
 */
#ifndef N2V_HIP_H
#define N2V_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define N2V_ABI_VERSION 15

#define N2V_OK 0
#define N2V_EINVAL (-1)  /* maps to ValueError (randomwalk.py:212-217)      */
#define N2V_ELAUNCH (-2) /* HIP launch / runtime error -> RuntimeError      */
#define N2V_ENOGPU (-3)

/* bits of the device status word */
#define N2V_ST_ZERODIV 1u /* sum(weights) == 0: ZeroDivisionError, randomwalk.py:172-173 */
#define N2V_ST_RANGE 2u   /* a start id outside [0, n_vertices)                          */
#define N2V_ST_OVERFLOW 4u /* n2v_partition_forward: a mailbox or the word pool was too small */

/* walk sampler modes */
#define N2V_WALK_EXACT 0 /* per-step biased alias rebuild, bit-identical to the reference */
#define N2V_WALK_FAST 1  /* same transition distribution, not the same draws: on a unit-weight graph
                          * with edge_classes + wedge_off / wedge_pos a step is ONE draw from the
                          * layers of its table (no rejection at q >= 1 with p <= q); otherwise
                          * candidates from the first-order table are rejected by beta / beta_max */

/* One entry of a first-order Walker alias table, packed so that a draw is ONE
 * 16-byte access: the neighbour id, the neighbour id BEHIND the alias index of
 * generate_alias_tables (row[alias[i]], randomwalk.py:140-141 looks it up the same
 * way) and the pseudo-probability (randomwalk.py:157-190). */
typedef struct n2v_slot {
  int32_t col;
  int32_t alias; /* vertex id, not an index */
  double prob;
} n2v_slot;

/* One entry of the hop table (optional, unit-weight graphs; n2v_hops_build): everything a
 * walker needs once it has chosen edge e = (v -> x), in ONE 16-byte access -- the neighbour
 * id, the class counts of that edge (edge_classes[e], needed at the walker's next step) and
 * the row of x: rowptr[x] in the low 40 bits, the out-degree of x in the high 24.  The
 * reference fetches the same facts by two joins per step (fugue.py:146-148); on the GPU each
 * would be a dependent random gather, and random 64-byte sectors per step are what bounds the
 * walk kernels (DESIGN.md "K2").  Costs 16 bytes per edge of HBM. */
typedef struct n2v_hop {
  int32_t col;
  uint32_t classes; /* layout of edge_classes[]; all ones = unknown (the kernels classify) */
  uint64_t row;     /* rowptr[col] | (uint64_t)degree(col) << 40 */
} n2v_hop;
#define N2V_HOP_DEG_SHIFT 40
#define N2V_HOP_ROW_MASK 0xffffffffffull

/* The reference's adjacency DataFrame df_adj (fugue.py:130, randomwalk.py:266-275)
 * as CSR in HBM: one row per vertex id, neighbours sorted by dst ascending,
 * multi-edges kept.  Weights: the reference carries Python floats (randomwalk.py:20,
 * indexer.py:24), so `w64` holds them as fp64; `w` is the 4-byte form for weights that are
 * exactly representable in fp32 (same bits after widening); at most one of the two is set,
 * both NULL = every weight is 1.0 (what index_graph_* produces, indexer.py:20-21) and the
 * walk kernels never read weights.
 * `slots` holds the per-row first-order alias tables written by n2v_alias_build,
 * CSR-aligned (N2V_WALK_FAST on weighted graphs; exact walks with p == q == 1 on weighted
 * graphs).  `pivots` is an optional search index over `col` written by n2v_pivots_build
 * (NULL = plain binary search): the test "x in N_out(src)" of randomwalk.py:226 then
 * touches 2-3 cache lines instead of log2(degree).  `edge_classes` (unit-weight graphs,
 * optional) is written by n2v_edge_classes_build: per edge e = (s -> v) the class counts
 * of the table generate_edge_alias_tables builds at (s, v) -- see there.  `hops` (unit-weight
 * graphs, optional) is written by n2v_hops_build; with it a step of the p == q == 1, the
 * class-count and the rejection kernels is one gather instead of two or three. */
typedef struct n2v_graph {
  int64_t n_vertices;
  int64_t n_edges;
  const int64_t *rowptr; /* [n_vertices + 1] */
  const int32_t *col;    /* [n_edges] */
  const float *w;        /* [n_edges] fp32 storage, or NULL */
  const double *w64;     /* [n_edges] fp64 storage, or NULL */
  const n2v_slot *slots; /* [n_edges] or NULL */
  const int32_t *pivots; /* [(n_edges + 31) / 32]: col[min(32 j + 31, n_edges - 1)], or NULL */
  const uint32_t *edge_classes; /* [n_edges] or NULL */
  const struct n2v_hop *hops;   /* [n_edges] or NULL (unit-weight graphs) */
  const uint64_t *wedge_off;    /* [n_edges] or NULL: see n2v_wedge_build */
  const void *wedge_pos;        /* uint16 / uint32 positions, or NULL */
  int32_t wedge_wide;           /* 0: wedge_pos is uint16 (every degree < 65536), 1: uint32,
                                   T >= 2 (mixed; 65536 in production): the lists of the edges INTO a row
                                   of T entries or more are uint32 and lie behind the uint16 lists of all
                                   other edges -- see n2v_wedge_build */
  int32_t reserved;             /* 0 (diagnostics: bit 0 set = do not use the all-tables kernel,
                                   bit 1 set = do not use wedge_slots) */
  const uint64_t *hops8;        /* the 8-byte hop table (n2v_hops8_build) or NULL */
  int32_t hop8_col_bits;        /* field widths of a hops8 entry, see n2v_hops8_build */
  int32_t hop8_row_bits;
  const int64_t *hop8_rowptr;   /* [n_vertices + 1] row starts of the hops8 table when its rows are
                                   padded (hop8_align_shift > 0), else NULL (= rowptr) */
  int32_t hop8_align_shift;     /* rows of the hops8 table start at multiples of 2^shift entries */
  int32_t reserved2;            /* bit 0 (N2V_HOPS_INLINE_RPOS): the class words of `hops` carry inline
                                   return positions, see n2v_hops_build; bit 1 (N2V_SLOTS_FOLDED): see
                                   n2v_wedge_slots_fold; else 0 */
  const uint16_t *wedge_slots;  /* [n_edges][16] or NULL: see n2v_wedge_slots_build */
  /* The degree-ranked form of a unit-weight graph (p == q == 1 walks; n2v_rank_hops_build), or all
   * NULL / 0.  Vertices are numbered again by descending degree (`rank`), rows are laid out in rank
   * order, so the row of a rank follows from the DEGREE CLASS the rank falls in and an entry of the
   * table is the 4-byte rank of the neighbour alone: a step is one 4-byte gather. */
  const uint32_t *rank_hops;        /* [n_edges]: rank of every neighbour; rows in rank order, the entries
                                       of a row in the CSR's order (so pick = int(r1 * n) is unchanged) */
  const int32_t *rank_of;           /* [n_vertices]: vertex id -> rank */
  const int32_t *rank_vertex;       /* [n_vertices]: rank -> vertex id */
  const uint64_t *rank_head;        /* [rank_head_n]: row offset | degree << 40 of ranks below rank_head_n
                                       (the few top vertices whose degrees are all different) */
  const uint32_t *rank_class_first; /* [rank_classes]: first rank of every degree class of the ranks from
                                       rank_head_n on, ascending, [0] == rank_head_n; then n_vertices in
                                       the entry after the last class and in the padding */
  const uint32_t *rank_class_off;   /* [rank_classes]: offset in rank_hops of the row of that first rank
                                       (n_edges after the last class; the form needs n_edges < 2^32).
                                       The degree of class c is (off[c+1] - off[c]) / (first[c+1] - first[c]) */
  int32_t rank_head_n;              /* 0 .. 2^22 */
  int32_t rank_classes;             /* a power of two, 2 .. 8192, > the number of classes (the walk kernel
                                       keeps both arrays in LDS: 8 bytes per class) */
  int32_t rank_emit;                /* what the ranked kernel writes to walks_out: 0 = vertex ids (one more
                                       gather per token through rank_vertex), 1 = ranks (the caller
                                       composes rank_vertex into its own per-token lookup, as
                                       n2v_corpus_index's index_of) */
  int32_t reserved3;                /* 0 */
  /* Row sums of the tables of the steps into LONG rows, for ONE (p, q) whose 1/p or 1/q is not dyadic
   * (n2v_edge_row_sums_build; ABI 15), or NULL / 0.  The walk kernels ignore the table unless row_sums_p / _q are
   * the p, q of the walk. */
  const double *row_sums;           /* [n_edges]; valid for the edges into rows of >= row_sums_from entries */
  double row_sums_p;
  double row_sums_q;
  int32_t row_sums_from;
  int32_t reserved4;                /* 0 */
} n2v_graph;

/* edge_classes[e] for e = (s -> v): bits 0..23 = number of entries x of N(v) with
 * x in N_out(s) and x != s ("shared", randomwalk.py:226-227), bits 24..31 = number of
 * entries of N(v) equal to s ("return", :223-224); a field that does not fit is stored
 * saturated (all ones) and the kernels then classify the row themselves. */
#define N2V_EC_SHARED_MASK 0x00ffffffu
#define N2V_EC_RETURN_SHIFT 24
#define N2V_EC_RETURN_SAT 0xffu
/* Inside a hop table built with N2V_HOPS_INLINE_RPOS (never in edge_classes[]): an edge WITHOUT
 * shared neighbours stores  N2V_EC_INLINE | return count (7 bits) << 24 | return position (24
 * bits)  -- the one thing a step that runs the pairing loop still needs of such an edge, so that it
 * does not fetch the edge's wedge slot for it (half of the edges of cfg 4, all of cfg 5). */
#define N2V_EC_INLINE 0x80000000u
#define N2V_HOPS_INLINE_RPOS 1
/* bit 1 of reserved2 (ABI 15): the wedge slots of the edges into rows of wedge_wide .. wedge_wide + 65536 entries are
 * FOLDED slots (n2v_wedge_slots_fold) -- the exact slots kernel then steps those rows like every other */
#define N2V_SLOTS_FOLDED 2

int n2v_abi_version(void);
const char *n2v_status_string(int code);

/* Number of GPUs visible to the library (no GPU initialisation side effects
 * beyond hipGetDeviceCount). */
int n2v_device_count(void);

/* K1 -- first-order alias tables for every row, exactly
 * generate_alias_tables(weights of the row) (randomwalk.py:157-190): same LIFO
 * pairing order, fp64 arithmetic, leftovers keep alias 0 (= the row's first
 * neighbour).  Rows of degree 0 are skipped; a row whose weights sum to 0 sets
 * N2V_ST_ZERODIV.  slots_out is CSR-aligned ([n_edges]); slot.alias holds the
 * neighbour id the alias index points to, slot.col a copy of col. */
int n2v_alias_build(const n2v_graph *g, n2v_slot *slots_out, uint32_t *status,
                    void *stream);

/* Per-edge class counts for exact walks on a unit-weight graph (g->w == g->w64 == NULL).
 * The table generate_edge_alias_tables builds for a step (s -> v) (randomwalk.py:219-231)
 * is a function of the EDGE (s, v) only, and with unit weights its row sum (:172) and the
 * value probs[i] of every slot before the pairing (:173) follow from two counts: how many
 * neighbours of v are s itself and how many are out-neighbours of s.  The reference
 * recomputes them (a set intersection) at every step; this pass computes them once per
 * edge -- one wave per source row, the shorter list searched in the longer -- and
 * N2V_WALK_EXACT then decides most steps from {rowptr[v], classes[e], col[pick], one
 * membership test}, bit-identical to the per-step rebuild, and only rebuilds the class
 * layout of a row when the pairing loop (:182-189) actually has to run for slot `pick`.
 * classes_out: [n_edges] uint32, layout above.  status: the four words described at the
 * top (word [1] is the kernel's batch counter). */
int n2v_edge_classes_build(const n2v_graph *g, uint32_t *classes_out, uint32_t *status,
                           void *stream);

/* Hop table of a unit-weight graph (struct n2v_hop above): hops_out[e] = {col[e],
 * g->edge_classes ? g->edge_classes[e] : all ones, rowptr[col[e]] | degree(col[e]) << 40}.
 * With g->wedge_off set and g->reserved2 & N2V_HOPS_INLINE_RPOS the class word of an edge whose
 * shared count is 0 is written in the N2V_EC_INLINE form (every return count must be below 128,
 * else N2V_ST_RANGE); such a table serves the wedge-slots kernel of N2V_WALK_EXACT and
 * N2V_WALK_FAST, and n2v_walk returns N2V_EINVAL when it would have to hand it to another exact
 * kernel -- build the plain form for those (the table takes milliseconds).
 * N2V_EINVAL when the graph has weights or 2^40 edges or more; a row of 2^24 entries or more
 * sets N2V_ST_RANGE in status[0] (read after synchronising): the table must then be discarded
 * (walk without it). */
int n2v_hops_build(const n2v_graph *g, struct n2v_hop *hops_out, uint32_t *status, void *stream);

/* The 8-byte hop table of a unit-weight graph for exact walks with p == q == 1 (no class counts
 * needed).  Row v of the table starts at entry T[v] = hop8_rowptr ? hop8_rowptr[v] : rowptr[v] (a
 * multiple of 2^align_shift) and holds, for the k-th neighbour x of v,
 *   x | (T[x] >> align_shift) << col_bits | code << (col_bits + row_bits),
 * code = min(degree(x), 2^(64 - col_bits - row_bits) - 1); the top code is an escape: the degree is
 * then read from rowptr (the few high-degree rows: cache-resident).  The chip delivers ~50 G random
 * 8-byte gathers per second against ~40 G 16-byte gathers over the table of n2v_hops_build
 * (profiles/r3v_probe_gathers.log), and a walk step of this kernel IS one such gather.  Padding
 * the rows (align_shift 3) frees three bits of the row field for the degree code: at cfg 4
 * (27-bit ids) 9 bits instead of 7, so that the escape rows' rowptr entries fit the L2.
 * Requires n_vertices <= 2^col_bits, (table entries >> align_shift) < 2^row_bits and
 * col_bits + row_bits <= 62 (N2V_EINVAL otherwise).  hops8_out: [T[n_vertices]] uint64. */
int n2v_hops8_build(const n2v_graph *g, int32_t col_bits, int32_t row_bits, int32_t align_shift,
                    const int64_t *hop8_rowptr, uint64_t *hops8_out, void *stream);

/* Shared-position lists ("wedge table") of a graph (weights play no part).  For edge e = (s -> v) the list
 *   wedge_pos[off .. off + n_shared)   off = wedge_off[e] & (2^40 - 1), n_shared = low 24 bits
 *                                      of edge_classes[e]
 * holds, ascending, the positions j with N(v)[j] in N_out(s) and N(v)[j] != s -- WHICH slots of
 * the table generate_edge_alias_tables builds at (s, v) carry the unchanged weight
 * (randomwalk.py:226-227) -- and wedge_off[e] >> 40 is the position of the first N(v)[j] == s
 * (the weight / p slots, :223-224; meaningful when the return count of edge_classes[e] is not
 * 0).  With them N2V_WALK_EXACT needs neither a search over N(s) nor a pass over N(v) to lay
 * the table of a step out: the membership test of the drawn slot is a search in its edge's
 * short list, and the steps that must run the pairing loop (:182-189) read the class of every
 * slot off the list.  One entry per (edge, common neighbour) pair: six per triangle.
 *   list_off      [n_edges] exclusive prefix sum of the shared counts (the caller's cumsum over
 *                 edge_classes & N2V_EC_SHARED_MASK); may be the same buffer as wedge_off_out
 *   wedge_pos_out [sum of the counts] uint16 when wide == 0 (every out-degree < 65536), uint32
 *                 when wide == 1.  wide == T >= 2 (T <= 65536; "mixed"): the list of an edge into a
 *                 row of fewer than T entries is uint16, offsets in uint16 units; the list of an
 *                 edge into a row of T entries or more ("wide row": its positions need more than 16
 *                 bits) is uint32, offsets in uint32 units from the same base, and the caller lays
 *                 those lists behind all the uint16 ones (list_off says where, per edge).  A walker
 *                 knows the degree of the row it stands on, so it knows the width of the list it
 *                 reads: a few hubs (the reference caps rows at 100 000, constants.py:6) cost the
 *                 other rows nothing.  The same value goes into n2v_graph.wedge_wide.
 * g->edge_classes must be set.  A list whose length disagrees with its count sets
 * N2V_ST_RANGE in status[0] (the tables must then be discarded). */
#define N2V_WEDGE_RPOS_SHIFT 40
#define N2V_WEDGE_OFF_MASK 0xffffffffffull
int n2v_wedge_build(const n2v_graph *g, const uint64_t *list_off, uint64_t *wedge_off_out,
                    void *wedge_pos_out, int32_t wide, uint32_t *status, void *stream);

/* Wedge slots: the wedge table laid out so that a biased exact step finds the list of the edge it
 * came along with ONE gather at a place it knows a step ahead (slot e = edge index), requested
 * together with the hop entry, instead of wedge_off[e] and then the list behind that offset (two
 * dependent gathers; random 64-byte sectors per step are what bounds the walk kernels).  32 bytes
 * = 16 halfwords per edge:
 *   [0] return position (wedge_off[e] >> 40)   [1] number of list entries below the return position
 *   n_shared <= 14:  [2 .. 2 + n_shared) the list itself
 *   n_shared  > 14:  [4 .. 8) the list's offset in wedge_pos as 64 bits, [8 .. 16) eight pivots
 *                    list[((k + 1) * n_shared) / 9], k = 0 .. 7 (the search enters the right ninth)
 * 16-bit positions only (wedge_wide == 0, or mixed: the slots of the edges into wide rows are left
 * zero and never read -- those steps go through wedge_off); g->edge_classes, g->wedge_off and g->wedge_pos must be
 * set.  At cfg 4 half of the steps need a list and four fifths of those lists are short.
 * slots_out: [n_edges * 16] uint16. */
int n2v_wedge_slots_build(const n2v_graph *g, uint16_t *slots_out, void *stream);

/* Folded lists and slots for the edges into WIDE rows of a mixed wedge table (ABI 15).  The reference trims rows at
 * 100 000 (constants.py:6), so the rows of its hubs hold positions that need 17 bits.  The mixed table keeps the lists
 * of the edges into such rows as uint32 without a slot, and a walker standing there took its step through a second,
 * 32-bit instance of the step (wedge_off, then the list): a wave pays for every code path ONE of its lanes takes, and
 * on cfg 4 trimmed at 100 000 nearly every wave-step has such a lane.  A FOLDED list stores a position below
 * T = wedge_wide as it is and one from T on minus T, in the same order -- both parts ascend, entry k is
 * list[k] + (k >= nlow ? T : 0) with nlow the number of entries below T, and a lower bound is one search in one part --
 * so it is a uint16 list like every other (rows of up to T + 65536 entries), gets a slot like every other, and the
 * exact slots kernel steps such a row with the same instructions as the rest:
 *   n_shared <= 14:  [0] return position (folded)  [1] below | nlow << 4 | upper << 8 (upper: the return position is
 *                    >= T)  [2 .. 2 + n_shared) the folded list
 *   n_shared  > 14:  [0] return position (folded)  [1] below, [2] nlow (low 16 bits)  [3] upper | (below >> 16) << 4 |
 *                    (nlow >> 16) << 8  [4 .. 8) offset of the folded copy in wedge_pos (2-byte units)  [8 .. 16) eight
 *                    pivots of the folded copy, as in n2v_wedge_slots_build
 * g: a graph with a mixed table (wedge_wide = T >= 2), edge_classes, wedge_off, wedge_pos set; fold_off [n_edges]:
 * where the folded copy of edge e's list goes (2-byte units from wedge_pos; read for the edges into rows of T ..
 * T + 65536 entries with more than 14 shared neighbours only); wedge_pos_rw == g->wedge_pos (the caller's buffer, with
 * room behind the uint32 lists); slots [n_edges * 16] as n2v_wedge_slots_build left them (the slots of those edges
 * are rewritten).  The uint32 lists stay: every other kernel reads them.  Set N2V_SLOTS_FOLDED in reserved2 to have
 * the slots kernel use the result; without the bit the same walks come out of the 32-bit instance. */
int n2v_wedge_slots_fold(const n2v_graph *g, const uint64_t *fold_off, void *wedge_pos_rw, uint16_t *slots,
                         void *stream);

/* Row sums for values that are not dyadic (ABI 15).  generate_alias_tables divides every weight of the step's table
 * by sum(node_weights) / n (randomwalk.py:172-173), a sum rounded at every addition in slot order.  With dyadic 1/p,
 * 1/q it is an integer combination of the edge's class counts; otherwise a step whose decision is closer to its
 * threshold than the rounding of that sum (1 % of the steps on long rows) has to add the row up in the reference's
 * order, run by run between the shared positions: O(list) dependent operations by ONE lane while its wave waits
 * (cfg 4 trimmed at 100 000, (3, 0.7): 102 of 149 ms per launch, profiles/r12c_*).  The sum depends on the EDGE walked
 * and on (p, q) only: this pass computes it once for every edge into a row of >= min_row entries -- the same routine,
 * hence the same bits -- and a walk at that (p, q) reads it with one gather.
 *   edges [k]: the edges to compute (the caller lists those into rows of >= min_row entries, longest lists first so
 *   that the lanes of a wave finish together); sums_out [n_edges] (entries of other edges are not written).
 * g: unit weights, edge_classes, wedge_off, wedge_pos set.  Put sums_out, p, q, min_row into n2v_graph.row_sums*. */
int n2v_edge_row_sums_build(const n2v_graph *g, double p, double q, const int64_t *edges, int64_t k,
                            double *sums_out, void *stream);

/* Search index for N2V_WALK_FAST: the last id of every aligned block of 32 entries of
 * `col` (one 128-byte line).  Inside a sorted row the block ends ascend, so a
 * membership query is a binary search over (degree / 32) pivots followed by one over a
 * single line.  pivots_out: [(n_edges + 31) / 32] int32. */
int n2v_pivots_build(const int32_t *col, int64_t n_edges, int32_t *pivots_out, void *stream);

/* K2 -- the whole of fugue.random_walk's loop (fugue.py:137-153) on device:
 * initiate_random_walk (randomwalk.py:279-296), walk_length x
 * next_step_random_walk (randomwalk.py:300-339) and to_path (:343-349).
 *
 *   start_ids   [n_start]   candidate start vertices (walk_start, fugue.py:132-134)
 *   walks_out   [n_start * num_walks, walk_length + 1] int32, row r = start
 *               r / num_walks, ordinal r % num_walks + 1 (randomwalk.py:294)
 *   valid_out   [n_start * num_walks] 1 where the reference emits a row; 0 for
 *               a start vertex without out-edges (fugue.py:132) and for walkers
 *               that reach a vertex without out-edges before the last step
 *               (inner join, fugue.py:147)
 *
 * The two uniforms of randomwalk.py:336-337 come from the counter-based stream
 * of DESIGN.md "RNG", keyed by (seed, start vertex, ordinal, step): results do
 * not depend on launch geometry or on how start_ids are sharded over GPUs.
 * mode N2V_WALK_EXACT reproduces generate_edge_alias_tables + sampling_from_alias
 * bit for bit; N2V_WALK_FAST draws from the same distribution by rejection.
 * With return_param == inout_param == 1 (the reference's defaults) and g->slots set,
 * exact mode reads the K1 tables instead of rebuilding them: same bits, one gather per
 * step. */
int n2v_walk(const n2v_graph *g, const int32_t *start_ids, int64_t n_start,
             int32_t num_walks, int32_t walk_length, double return_param,
             double inout_param, uint64_t seed, int32_t mode,
             int32_t *walks_out, uint8_t *valid_out, uint32_t *status,
             void *stream);

/* ONE STEP of exact walks on a WEIGHTED graph (g->w or g->w64), one lane per walker
 * (csrc/n2v_walk_wlanes.hip) -- the step-synchronous form of n2v_walk for the graphs whose tables have
 * no closed form: every value of the table generate_edge_alias_tables builds (randomwalk.py:193-232)
 * is different, so a step is O(row) with two serial parts, the left-to-right row sum (:172) and the
 * pairing loop (:182-189).  n2v_walk gives a walker a wave and runs those on one lane of it; here 64
 * walkers share a wave and their serial chains run side by side -- which pays when the lanes of a
 * wave stand on rows of similar length, hence `order`.  The caller keeps the state and loops:
 *     walks  [n_rows, walk_length + 1] int32, row r = start r / num_walks, ordinal r % num_walks + 1;
 *            before step 0: walks[r][0] = the start vertex, or -1 (and valid[r] = 0) for a start
 *            vertex out of range or without out-edges (fugue.py:132); everything else -1
 *     valid  [n_rows] 1 while the walker walks; the step clears it when the walker reaches a vertex
 *            without out-edges before the last step (fugue.py:147) or its row sums to 0
 *            (N2V_ST_ZERODIV)
 *     edge_state [n_rows] the edge (index into col) every walker walked last; the step writes it
 *     order  [n_rows] the rows in the order lanes take them -- sorted by the out-degree of
 *            walks[r][step], descending, vanished walkers last (the ascending sort of
 *            n2v_walk_weighted_keys' keys) -- or NULL (row order).  PRECONDITION when scratch and
 *            row_sums are given: the order IS sorted that way.  The wave-per-walker kernels stop at the
 *            first row at or below the cut between the two kernels and at the first vanished walker
 *            (everything behind it is the lane kernel's, or nobody's), so a long row BEHIND a short one
 *            in an unsorted order would not be stepped and no status bit would say so.  The library
 *            cannot check a device-side permutation without a pass and a host read; it does not.
 *   for step = 0 .. walk_length - 1:  n2v_walk_weighted_step(..., step, ...)
 * and the result is n2v_walk's, bit for bit (same uniform stream, keyed by start vertex, ordinal and
 * step).  For return_param or inout_param != 1 the steps after the first read the classes of the
 * slots from g->edge_classes, g->wedge_off and g->wedge_pos (n2v_edge_classes_build, n2v_wedge_build:
 * they depend on the ids alone and are built for a weighted graph as for a unit one).
 *     scratch  int64 [2 (n_rows + 2)] or NULL, row_sums fp64 [n_vertices + 2] or NULL: the sum of the stored
 *            weights of every row, in any order (all weights finite and >= 0), then row_sums[n_vertices] =
 *            a power of two that divides every stored weight (fp32 weights: 2^(e - 24) of the smallest
 *            one, w = m 2^e; 0 = none known: then the margins are the general ones) and
 *            row_sums[n_vertices + 1] = the largest stored weight.  With an order AND both of
 *            these no walker's pairing loop is replayed: the one slot the draw asks for is DECIDED from
 *            sums over the row -- by a lane per walker on rows below 768 slots, by a wave per walker on the
 *            rows above -- (the k-th
 *            overfull slot is demoted where the running sum of the underfull slots' deficits passes that
 *            of the overfull slots' excess; the row sum itself comes from row_sums and the shared and
 *            return slots: no pass), every comparison with a margin that covers the roundings of the
 *            reference's loop (16 n^2 2^-52 and up); a walker whose draw some comparison cannot decide
 *            by that margin gets a second chance in a launch of its own -- the row sum in the
 *            reference's order, hence the margins of an exact sum -- and, if still undecided, is stepped by
 *            the exact wave-per-walker kernel, all in the same call (after it scratch[0] = how many had the
 *            second chance, scratch[n_rows + 2] = how many the exact kernel stepped; their rows follow each
 *            count).  Same bits as without them.
 *     hubs   NULL or the summaries of the long rows (n2v_weighted_hubs, above): the wave kernel then takes
 *            the sums of a step over such a row from them instead of a pass over the row.  Same bits. */
/* Summaries of the HUB rows of a weighted graph for n2v_walk_weighted_step (optional): for every row of at least
 * min_slots slots, cut into blocks of 256 slots in row order, the weights of each block SORTED ascending (the last
 * block padded with +inf) and their prefix sums -- with them the sums of a step over such a row (the deficits and
 * excesses of its slots at the walker's own threshold) are one binary search per block instead of a pass over the
 * row; the slots that are not "other" (shared positions, return run) are corrected from the wedge list.
 *   block0  int32 [n_vertices]: the first block of the row in sorted / prefix, -1 = the row has no summary
 *   sorted  [n_blocks][256] weights as stored (fp32 beside n2v_graph.w, fp64 beside w64)
 *   prefix  fp64 [n_blocks][257]: prefix[b][k] = the sum of the k smallest weights of block b */
typedef struct n2v_weighted_hubs {
  const int32_t *block0;
  const void *sorted;
  const double *prefix;
  int32_t min_slots;
  int32_t lane_cut;   /* 0 = chosen by the library from the number of walkers; > 0: rows of at least this many slots
                       * are decided by a wave per walker, shorter ones by a lane per walker (tuning, tests).  A struct
                       * with block0 == NULL carries this field alone. */
} n2v_weighted_hubs;

/* The sort keys of n2v_walk_weighted_step's `order`, one pass: keys[r] = rank_of[walks[r][step]] (rank_of: the
 * place of every vertex in the order of descending out-degree, ties by id) for a walker that walks, 0x7fffffff
 * for one that has vanished or never started -- sorting them ascending gives the order the step wants. */
int n2v_walk_weighted_keys(const int32_t *walks, const uint8_t *valid, const int32_t *rank_of,
                           int64_t n_vertices, int64_t n_rows, int32_t step, int32_t walk_length,
                           int32_t *keys, void *stream);
int n2v_walk_weighted_step(const n2v_graph *g, const int32_t *start_ids, int32_t num_walks,
                           const int64_t *order, int64_t n_rows, int32_t step, int32_t walk_length,
                           double return_param, double inout_param, uint64_t seed,
                           int64_t *edge_state, int32_t *walks, uint8_t *valid, uint32_t *status,
                           int64_t *scratch, const double *row_sums, const n2v_weighted_hubs *hubs,
                           void *stream);

/* n2v_walk with a workspace lent by the caller (the library never allocates).  Exact biased walks
 * on a unit-weight graph that carries the hop and wedge tables, dyadic return_param / inout_param,
 * then run as passes over 32-byte walker records kept in the workspace: launches that hold the
 * quick exits and closed forms of the pairing loop (randomwalk.py:182-189) only, each followed by
 * a launch that replays the steps whose closed form declined -- same walks, bit for bit, as
 * n2v_walk (csrc/n2v_walk_wedge2.hip; measured SLOWER than the one-launch kernel on every BASELINE
 * graph).  That variant is a BUILD OPTION since round 5 (`make WEDGE2=1`): the default
 * library reports 0 bytes from n2v_walk_workspace_bytes and n2v_walk_ws is n2v_walk whatever it is lent.
 * (With the option: 4 main/replay rounds before the finishing launch, N2V_WEDGE2_ROUNDS overrides.)  n2v_walk_workspace_bytes says how many bytes the
 * call can use (0: this graph / mode / (p, q) has no use for one); workspace == NULL, or fewer
 * bytes than that, is n2v_walk.  The workspace must be 16-byte aligned; its contents mean nothing
 * before or after the call, and it may be reused by the next call on the same stream. */
int64_t n2v_walk_workspace_bytes(const n2v_graph *g, int64_t n_start, int32_t num_walks,
                                 int32_t walk_length, double return_param, double inout_param,
                                 int32_t mode);
int n2v_walk_ws(const n2v_graph *g, const int32_t *start_ids, int64_t n_start,
                int32_t num_walks, int32_t walk_length, double return_param,
                double inout_param, uint64_t seed, int32_t mode,
                int32_t *walks_out, uint8_t *valid_out, uint32_t *status,
                void *workspace, int64_t workspace_bytes, void *stream);

/* The reference's transformer-level protocol on MATERIALISED tables, one partition (a batch of
 * walker rows) per call.  n2v_walk never builds the table of a step; next_step_random_walk
 * (randomwalk.py:300-339) and the reference's known-answer tests (tests/test_randomwalk.py:
 * 131-189, 268-306) are stated on it, with the two uniforms supplied by the caller
 * (random.random(), :336-337).  Rows are CSR-packed: row r holds the neighbour list of the
 * walker's current vertex, ids[rowptr[r] .. rowptr[r+1]) sorted by id (fugue.py:130).
 *
 *   n2v_edge_bias    w_out[e] = the unnormalised probability generate_edge_alias_tables gives
 *                    entry e (randomwalk.py:219-231): weight / return_param if ids[e] ==
 *                    src_id[r], weight if ids[e] is in the row's src_nbs list (sorted; the
 *                    caller's set src_nbs_id, :318), weight / inout_param otherwise; rows with
 *                    src_id[r] < 0 (first step, :319-320) and src_id == NULL keep the weight.
 *                    w / w64: fp32 or fp64 weights (both NULL = 1.0).  N2V_EINVAL when
 *                    return_param or inout_param is 0 (ValueError, :214-217).
 *   n2v_alias_build  on {rowptr, ids, w64 = w_out} then IS generate_alias_tables(biased row)
 *                    (:232, :157-190).
 *   n2v_alias_draw   sampling_from_alias(r1, r2) (:86-99) followed by the neighbour lookup of
 *                    RandomPath.append (:140-144): vertex_out[r] = the id behind the drawn
 *                    index.  r2 == NULL selects sampling_from_alias_wiki(r1) (:70-84).  A row
 *                    without neighbours yields -1; r1 outside [0, 1) sets N2V_ST_RANGE (the
 *                    reference raises IndexError). */
int n2v_edge_bias(const int64_t *rowptr, const int32_t *ids, const float *w, const double *w64,
                  const int32_t *src_id, const int64_t *src_rowptr, const int32_t *src_nbs,
                  int64_t n_rows, int64_t nnz, double return_param, double inout_param,
                  double *w_out, void *stream);
int n2v_alias_draw(const int64_t *rowptr, const n2v_slot *slots, int64_t n_rows,
                   const double *r1, const double *r2, int32_t *vertex_out, uint32_t *status,
                   void *stream);

/* The two uniforms n2v_walk draws for walker `key[i]` (= start vertex * num_walks + ordinal - 1)
 * at step `step[i]` (0-based), as the fp64 numbers u / 2^32 the reference's random.random()
 * stands for (randomwalk.py:336-337).  With them n2v_edge_bias + n2v_alias_build +
 * n2v_alias_draw reproduce one step of n2v_walk's exact mode bit for bit: the step function of
 * graph-partitioned walking (node2vec_amd/partitioned.py), where a walker's two rows live on
 * different GPUs and the step is taken where the current vertex is stored. */
int n2v_walk_uniforms(uint64_t seed, const int64_t *key, const int32_t *step, int64_t n,
                      double *r1_out, double *r2_out, void *stream);

/* a9 -- trim_hotspot_vertices (randomwalk.py:238-262): rows with more than
 * max_out_degree edges keep a uniform sample without replacement of exactly
 * max_out_degree of them (max_out_degree <= 0 means 100000, constants.py:6).
 * keep_out[e] = 1 for surviving edges; weights are untouched. */
int n2v_trim_mark(const int64_t *rowptr, int64_t n_rows, int64_t max_out_degree,
                  uint64_t seed, uint8_t *keep_out, void *stream);

/* K3 -- one pass of skip-gram negative-sampling SGD over a block of walks: the
 * arithmetic behind gensim.models.Word2Vec(sentences, sg=1, hs=0, negative=k) at
 * the reference's call site embedding.py:126 (algorithm: DESIGN.md "SGNS"; the
 * trainer itself is third-party gensim 3.8, absent from the reference tree).
 *
 *   walks       [n_walks, walk_len] int32 vocabulary indices, < 0 = token outside
 *               the vocabulary (dropped before windowing, as gensim does)
 *   syn0        [n_vocab, dim] fp32 input vectors  (model.wv.vectors), in place
 *   syn1neg     [n_vocab, dim] fp32 output vectors (model.trainables.syn1neg)
 *   cum_table   [n_vocab] uint32 cumulative count^0.75, scaled to 2^31 - 1
 *   sample_int  [n_vocab] uint32 keep-thresholds of frequent-word subsampling,
 *               or NULL when `sample` == 0
 *   exp_table   [1000] fp32 sigmoid table over [-6, 6) (word2vec EXP_TABLE)
 *   pairs_out   TWO device uint64 (or NULL): [0] += number of (centre, context) pairs
 *               trained; [1] scratch, the kernel hands out rows through it (reset by the
 *               library on the stream before the launch)
 *
 * params->batched = 1 selects the opt-in BATCHED trainer (dim 64 / 128 / 256, window <= 7,
 * negative <= 15, else N2V_EINVAL): the `negative` draws are made once per centre position and
 * shared by its <= 2 * window pairs, and the pairs of a position are trained from one snapshot of
 * the rows -- NOT gensim's sampling (Ji et al. 2016).  A position then is three small dense
 * products on v_mfma_f32_16x16x4_f32 and moves 8 * dim * (2 + negative) bytes per position
 * instead of per pair.  Its deterministic mode is bit-identical to its own CPU restatement
 * (oracle/n2v_oracle_sgns.c, the batched function).
 *
 * One wave trains one walk (sentence); walks are spread over the grid hogwild
 * (unsynchronised updates, as gensim's worker threads).  deterministic != 0 runs
 * every walk in order on a single wave: bit-identical to oracle/n2v_oracle_sgns.c.
 * All randomness is counter-based on (seed, sentence_base + row, draw index), so
 * results do not depend on launch geometry in deterministic mode. */
typedef struct n2v_sgns_params {
  int64_t n_vocab;
  int64_t sentence_base; /* global index of walks[0] (RNG key, multi-GPU shards) */
  uint64_t seed;
  int32_t dim;      /* 1 .. 1024 */
  int32_t window;   /* 1 .. 32  (the reference allows 5 .. 30, embedding.py:110) */
  int32_t negative; /* 1 .. 32 */
  float alpha;      /* learning rate of this pass (host applies the linear decay) */
  int32_t deterministic;
  int32_t cum_index_bits;   /* log2 of the number of buckets of cum_index (1 .. 30) */
  const int32_t *cum_index; /* device, [2^bits + 1] from n2v_cum_index_build, or NULL */
  int32_t max_waves; /* 0: hogwild concurrency = one wave per 32 vocabulary rows, up to the whole
                        chip; > 0: at most this many waves train concurrently */
  int32_t batched;   /* 0: gensim's sampling -- k negatives drawn per (centre, context) pair;
                        1 (opt-in, NOT gensim's sampling): the k negatives are drawn once per
                        centre position and shared by its <= 2 * window pairs, which turns a
                        position into one small dense product (see n2v_sgns_train) */
  int32_t window_cache; /* 1: the default kernel keeps the syn0 rows of the window in LDS (read and
                           written once per position instead of once per pair; same values, bit
                           for bit in deterministic mode; dim 64 / 128 and window <= 7, else
                           N2V_EINVAL).  Measured slower than 0 on MI355X (the ring costs 3 of 8
                           waves per SIMD and the kernel is bound by rows in flight): off by default */
  int32_t hub_rows;     /* hogwild mode only: rows [0, hub_rows) of syn0 / syn1neg (the most frequent
                           words: the vocabulary is in descending count order) are updated by
                           atomic adds of each wave's contribution instead of read-modify-write
                           stores, so that concurrent waves do not overwrite each other on hubs.
                           0 = off (gensim's unsynchronised updates everywhere).  The batched
                           trainer always returns its context rows as atomic deltas; hub_rows
                           adds the same for its target rows (8 adds per lane and row: measured
                           -34 % at 4096 on cfg 3, link AUC 0.886 -> 0.899 on cfg 2) */
  const float *row_alpha; /* [n_walks] or NULL: the learning rate of every row of the launch (gensim
                             lowers it per job of sentences: n2v_sgns_job_alpha writes such an
                             array); NULL = `alpha` for every row */
} n2v_sgns_params;

#define N2V_SGNS_MAX_SENTENCE 256 /* longer walks: split rows on the host */

/* Optional index over cum_table for the negative draws of K3: index_out[b] =
 * bisect_left(cum_table, b << (31 - bits)) for b = 0 .. 2^bits.  A draw r then needs
 * bisect_left only inside [index[r >> (31 - bits)], index[(r >> (31 - bits)) + 1]]: the same
 * word as gensim's bisect over the whole table, in one or two memory sectors instead of
 * log2(n_vocab) dependent probes (at 10^8 words the probes were a fifth of the kernel's read
 * sectors).  index_out: [2^bits + 1] int32. */
int n2v_cum_index_build(const uint32_t *cum_table, int64_t n_vocab, int32_t bits,
                        int32_t *index_out, void *stream);

/* gensim's learning-rate schedule, per row of a launch.  gensim.models.Word2Vec (embedding.py:126)
 * lowers the rate per JOB -- a batch of consecutive sentences of at most batch_words raw words
 * (constants.py:58: 1000) -- to  max(end, start - (start - end) * (epoch + pushed / total) / epochs),
 * pushed = sentences queued before the job (word2vec.py _job_producer, _get_next_alpha; Python
 * doubles, then cast to fp32).  out[i] = that rate for sentence row0 + i of epoch `epoch`, whose job
 * is (row0 + i) / job_rows (job_rows = max(1, batch_words / sentence length) for the equal-length
 * sentences of a walk corpus): the same double operations in the same order.  Pass `out` as
 * n2v_sgns_params.row_alpha.  out: [n] fp32. */
int n2v_sgns_job_alpha(int32_t job_rows, int32_t epoch, int32_t epochs, int64_t row0, int64_t rows,
                       double alpha0, double alpha_min, int64_t n, float *out, void *stream);

/* How many waves n2v_sgns_train(P, n_walks, walk_len) keeps in flight on this device (its hogwild
 * concurrency rule, max_waves and the occupancy of the kernel instance it picks, all applied) --
 * nothing is launched.  The host chooses n2v_sgns_params.hub_rows from it (node2vec_amd/sgns.py
 * auto_hub_rows: a row held by more than one wave at a time on average gets atomic adds).  0 for the
 * batched trainer; a negative status for parameters n2v_sgns_train would refuse. */
int64_t n2v_sgns_hogwild_waves(const n2v_sgns_params *P, int64_t n_walks, int32_t walk_len);

int n2v_sgns_train(const int32_t *walks, int64_t n_walks, int32_t walk_len,
                   float *syn0, float *syn1neg, const uint32_t *cum_table,
                   const uint32_t *sample_int, const float *exp_table,
                   const n2v_sgns_params *params_host,
                   unsigned long long *pairs_out, void *stream);

/* The exchange step of the multi-GPU SGNS path.  The reference trains on
 * one machine: gensim's worker threads share syn0 / syn1neg (embedding.py:126).  Sharded
 * over GPUs, every rank trains its own walks on a full replica and the replicas are
 * averaged every few launches by an all-reduce (RCCL) between these two elementwise passes:
 *
 *   n2v_delta_pack   before_out[i] = cur[i] (snapshot), and the rank's contribution
 *                    wire_out[i] = cur[i]                        N2V_WIRE_F32 (fp32 [n])
 *                                = bf16(cur[i] - ref_bf16[i])    N2V_WIRE_BF16 (bf16 [n])
 *   (caller: all-reduce SUM of wire over the ranks)
 *   n2v_delta_apply  mean = wire_sum[i] / world                       (F32)
 *                         = ref_bf16[i] + wire_sum[i] / world         (BF16; ref := bf16(mean))
 *                    cur[i] += mean - before[i]
 *
 * so that updates made to cur while the collective was in flight survive.  With
 * before_out / before == NULL no snapshot is taken and apply stores cur[i] = mean exactly
 * (a blocking exchange: every rank ends with the same bits).  ref_bf16 is a
 * bf16 reference copy shared by all ranks (n2v_delta_ref_init writes bf16(cur) once, on
 * identical replicas); the F32 form needs no reference at all. */
#define N2V_WIRE_F32 0
#define N2V_WIRE_BF16 1
int n2v_delta_ref_init(const float *cur, int64_t n, uint16_t *ref_bf16_out, void *stream);
int n2v_delta_pack(const float *cur, const uint16_t *ref_bf16, int64_t n, float *before_out,
                   void *wire_out, int32_t wire_dtype, void *stream);
int n2v_delta_apply(float *cur, uint16_t *ref_bf16, const float *before, const void *wire_sum,
                    int32_t wire_dtype, int32_t world, int64_t n, void *stream);
/* The sum itself, when the caller moves bytes instead of calling an all-reduce (so that the result
 * does not depend on the collective library's reduction order or accumulator type): `parts` holds
 * `world` contributions of m elements each, rank after rank (what an all-to-all of the ranks' wire
 * buffers delivers for this rank's shard); out[i] = parts[0][i] + parts[1][i] + ... in fp32, in
 * rank order, stored in the wire type (bf16: ONE rounding of the fp32 sum).  An all-gather of the
 * shards then gives every rank the same wire_sum for n2v_delta_apply. */
int n2v_delta_reduce(const void *parts, int32_t wire_dtype, int32_t world, int64_t m, void *out,
                     void *stream);

/* The two elementwise passes between K2 and K3 when the corpus is streamed batch by batch
 * (node2vec_amd/pipeline.py; the reference materialises the walks as strings and lets gensim's
 * build_vocab count them, embedding.py:125-126):
 *   n2v_corpus_count  counts[v] += occurrences of v in the rows with valid[row] != 0 (valid NULL =
 *                     every row); tokens outside [0, n_vertices) are ignored.  counts: uint64.
 *   n2v_corpus_index  idx_out[r][t] = index_of[walks[r][t]] for tokens of valid rows, else -1
 *                     (the int32 vocabulary indices n2v_sgns_train takes). */
int n2v_corpus_count(const int32_t *walks, const uint8_t *valid, int64_t n_rows, int32_t len,
                     int64_t n_vertices, unsigned long long *counts, void *stream);
int n2v_corpus_index(const int32_t *walks, const uint8_t *valid, const int32_t *index_of,
                     int64_t n_rows, int32_t len, int64_t n_vertices, int32_t *idx_out, void *stream);

/* Graph-partitioned walking (node2vec_amd/partitioned.py): one step of the k
 * walkers resident on ONE part of a vertex-range partition -- the reference's per-step join
 * (fugue.py:146-149: walker row x adjacency row of its current vertex, the row of its previous
 * vertex carried along) followed by next_step_random_walk (randomwalk.py:300-339), in one launch,
 * one wave per walker, no table materialised.
 *   rowptr / col / w / w64   the part's rows [lo, lo + n_local) (rowptr rebased to 0; ids global;
 *                            at most one of w, w64; both NULL = unit weights)
 *   head      int64 [k][head_cols] (head_cols >= 4): (output row, RNG key = start * num_walks +
 *             ordinal - 1, s << 32 | v, step[, classes]); s = -1 (high word all ones) on the
 *             first step (:320-321)
 *   src_ptr / src_ids        what travelled with the walkers, packed (int64 [k + 1] offsets):
 *     src_kind N2V_SRC_ROWS    the rows N(s), sorted ids; read only when q != 1 and s >= 0 (may
 *                              be NULL otherwise)
 *     src_kind N2V_SRC_WEDGES  (unit weights, head_cols >= 5) the WEDGE LIST of the edge (s -> v)
 *                              walked last -- the positions in N(v) of the neighbours v shares
 *                              with s, ascending, 32-bit -- cut from the wedge table of the rank
 *                              that stores that edge (n2v_wedge_build on its part), with
 *                              head[.][4] = edge_classes[e] | return position << 32: the class of
 *                              every slot of the step's table is then known by position and
 *                              neither N(s) nor a pass over N(v) is needed.  With q == 1 the
 *                              lists are empty (only the return run matters), with p == q == 1
 *                              nothing is read beyond the first four header words
 *   A header whose output row (head[i][0]) is NEGATIVE is an EMPTY SLOT of a capacity-bounded
 *   mailbox: n2v_partition_step draws nothing for it and n2v_partition_forward neither logs nor
 *   forwards it, so a mailbox of fixed capacity can be stepped whole, without its count on the host.
 *   next_out  int32 [k]: the vertex drawn (-1 when status reports an error for that walker)
 *   edge_out  int64 [k] or NULL: the index (into the part's col) of the edge drawn
 *   status    uint32 [4] as for n2v_walk; [1] is used as the walker counter.
 * Draws are those of n2v_walk's exact mode (same stream keyed by (seed, key, step)), so walks are
 * bit-identical to n2v_walk over the unpartitioned graph wherever the walker happens to be.
 *   n2v_gather_rows    out[out_ptr[j] ..) = ids[ptr[rows[j]] .. ptr[rows[j] + 1]) for j < k: packs
 *                      the rows that leave with migrating walkers (out_ptr = their prefix sums)
 *   n2v_gather_wedges  the same for wedge lists: out[out_ptr[j] ..) = the list of edge edges[j]
 *                      widened to 32 bits, and head[j][4] = edge_classes | return position << 32 */
#define N2V_SRC_ROWS 0
#define N2V_SRC_WEDGES 1
#define N2V_SRC_WEDGES_AT 2 /* as N2V_SRC_WEDGES, but src_ptr is int64 [k]: where the list of walker i
                               STARTS in src_ids (its length is the shared count in head[i][4]) -- the
                               form n2v_partition_forward leaves the lists in */
int n2v_partition_step(const int64_t *rowptr, const int32_t *col, const float *w, const double *w64,
                       int64_t lo, int64_t n_local, const int64_t *head, int32_t head_cols,
                       const int64_t *src_ptr, const int32_t *src_ids, int32_t src_kind, int64_t k,
                       double p, double q, uint64_t seed, int32_t *next_out, int64_t *edge_out,
                       uint32_t *status, void *stream);
/* The elementwise half of the routing that follows a step: for walker i the path record
 * log_out[i] = (output row, step + 1, next) -- or (row, -1, -1) when next < 0: the vertex it stood
 * on has no out-edges, it vanished on arrival (fugue.py:147) --, the header it travels on with,
 * dest_out[i] = the part that owns `next` (bounds: first vertex of every part, ascending) or
 * n_parts when the walk is complete or the walker vanished, len_out[i] / src_out[i] = how many
 * words travel with it and where they come from (carry 0: none; N2V_SRC_ROWS + 1: the row of its
 * vertex, src = local row; N2V_SRC_WEDGES + 1: the wedge list of edge[i], src = that edge;
 * N2V_SRC_WEDGES + 2: src = that edge but no words -- q == 1, only the counts and the return
 * position of the edge travel, in the header).  The caller groups the walkers by a stable sort
 * of dest_out, or calls n2v_partition_group. */
int n2v_partition_route(const int64_t *head_in, int32_t head_cols, const int32_t *next,
                        const int64_t *edge, int64_t k, int32_t walk_length, const int64_t *bounds,
                        int32_t n_parts, int32_t carry, const int64_t *rowptr, int64_t lo,
                        const uint32_t *edge_classes, int64_t *log_out, int64_t *head_out,
                        int32_t *dest_out, int64_t *len_out, int64_t *src_out, void *stream);
/* Groups what n2v_partition_route wrote by destination, keeping the order inside a destination
 * (a stable counting sort: per-block counts, one scan, one scatter): head_out / len_out / src_out
 * = the rows of head / len / src ordered by dest (0 .. n_parts, n_parts = not forwarded, last);
 * cuts_out int64 [n_parts + 1]: where destination d starts (cuts_out[n_parts] = the number of
 * forwarded walkers).  n_parts <= 64.  work: int64 scratch of ((k + 255) / 256 + 1) * (n_parts + 1)
 * words. */
int n2v_partition_group(const int32_t *dest, const int64_t *head, int32_t head_cols,
                        const int64_t *len, const int64_t *src, int64_t k, int32_t n_parts,
                        int64_t *work, int64_t *head_out, int64_t *len_out, int64_t *src_out,
                        int64_t *cuts_out, void *stream);
/* Route, group and gather in ONE launch, no size known to the host (round 4).  For walker i: the
 * path record -- log_out[i] as n2v_partition_route writes it, or, when log_out is NULL, straight
 * into walks_out[row][step + 1] (valid_out[row] = 0 for a walker that vanished): the single-process
 * form, where the emitting rank's arrays are at hand --; then, if the walk goes on, the walker is
 * APPENDED to the mailbox of the part that owns `next`:
 *   box_head  int64 [n_parts][cap][head_cols]   headers (head[.][4] = edge_classes | return position
 *                                               << 32 of the edge drawn when carry != 0)
 *   box_off   int64 [n_parts][cap]              where its list starts in the destination's pool
 *   box_words int32 [n_parts][wcap]             the wedge lists (carry N2V_SRC_WEDGES + 1): one pool per
 *                                               destination, so that what goes to one part is
 *                                               contiguous (an exchange sends pool d to rank d)
 *   box_count uint64 [2 n_parts]                walkers appended per destination, then words used per
 *                                               destination.  The caller zeroes it before the first
 *                                               launch of a step; several source parts may append to
 *                                               the same boxes (one atomic add per block and
 *                                               destination).
 * A full mailbox or pool sets N2V_ST_OVERFLOW in status[0] (the step must be repeated with larger
 * ones; nothing is written out of bounds).  carry: 0 (headers only: p == q == 1), N2V_SRC_WEDGES + 1
 * (the list of edge[i]), N2V_SRC_WEDGES + 2 (q == 1: counts and return position only); rows do not
 * travel this way.  The next n2v_partition_step takes mailbox d as head = box_head[d] / src_ptr =
 * box_off[d] / src_ids = box_words[d] with src_kind N2V_SRC_WEDGES_AT.  Replaces n2v_partition_route +
 * n2v_partition_group + a prefix sum + n2v_gather_wedges and the host read between them; the order
 * of the walkers in a mailbox is not defined, the walks are (the RNG is keyed by walker and step). */
int n2v_partition_forward(const int64_t *head_in, int32_t head_cols, const int32_t *next,
                          const int64_t *edge, int64_t k, int32_t walk_length, const int64_t *bounds,
                          int32_t n_parts, int32_t carry, const uint32_t *edge_classes,
                          const uint64_t *wedge_off, const void *wedge_pos, int32_t wide,
                          int64_t *box_head, int64_t *box_off, int32_t *box_words,
                          unsigned long long *box_count, int64_t cap, int64_t wcap, int64_t *log_out,
                          int32_t *walks_out, uint8_t *valid_out, uint32_t *status, void *stream);
/* n2v_partition_forward into mailboxes with a capacity PER DESTINATION, laid out back to back (what an
 * all-to-all with fixed split sizes sends as it lies): box_starts int64 [2 * n_parts + 2] -- mailbox d
 * is box_head / box_off [box_starts[d], box_starts[d + 1]), its word pool box_words
 * [box_starts[n_parts + 1 + d], box_starts[n_parts + 2 + d]); box_off holds the list start INSIDE the
 * pool of its destination.  The path records go to log_out.  With empty slots (see n2v_partition_step)
 * this is the step of a rank of a multi-GPU walk that reads nothing on the host: capacities fixed
 * once, overflow reported in the status word (N2V_ST_OVERFLOW) and looked at after the last step.
 * Replaces the shuffle of fugue.py:146-149. */
int n2v_partition_forward_boxes(const int64_t *head_in, int32_t head_cols, const int32_t *next,
                                const int64_t *edge, int64_t k, int32_t walk_length,
                                const int64_t *bounds, int32_t n_parts, int32_t carry,
                                const uint32_t *edge_classes, const uint64_t *wedge_off,
                                const void *wedge_pos, int32_t wide, int64_t *box_head, int64_t *box_off,
                                int32_t *box_words, unsigned long long *box_count,
                                const int64_t *box_starts, int64_t *log_out, uint32_t *status,
                                void *stream);
int n2v_gather_rows(const int64_t *ptr, const int32_t *ids, const int64_t *rows,
                    const int64_t *out_ptr, int64_t k, int32_t *out, void *stream);
int n2v_gather_wedges(const uint32_t *edge_classes, const uint64_t *wedge_off, const void *wedge_pos,
                      int32_t wide, const int64_t *edges, const int64_t *out_ptr, int64_t k,
                      int32_t *out, int64_t *head, int32_t head_cols, void *stream);

/* Fills the entry table of the degree-ranked form (n2v_graph.rank_hops) of a unit-weight graph.
 * The caller ranks the vertices -- any STABLE sort by descending degree: rank_vertex[r] = the vertex
 * of rank r, rank_of its inverse -- and passes rank_rowptr[r] = the sum of the degrees of ranks
 * below r ([n_vertices + 1], int64).  out[rank_rowptr[r] + k] = rank_of[col[rowptr[v] + k]] for
 * v = rank_vertex[r].  The head and class tables of n2v_graph are a few thousand words the caller
 * derives from the sorted degrees (node2vec_amd/graph.py build_ranked shows how).  Replaces nothing
 * in the reference (its adjacency rows are Python lists, fugue.py:130); it is the layout that lets
 * a p == q == 1 step (randomwalk.py:86-99 with probs == 1.0) be one 4-byte gather.
 * out: [g->n_edges] uint32. */
int n2v_rank_hops_build(const n2v_graph *g, const int32_t *rank_of, const int32_t *rank_vertex,
                        const int64_t *rank_rowptr, uint32_t *out, void *stream);

/* Measurement aid (bench.py; nothing on the product path calls it): the rate this device
 * sustains for the access shapes of K2 and K3 on the CALLER's buffer, so that the ceilings the
 * kernels are compared with are observed on the box the bench runs on.  One launch; the caller
 * times it (HIP events on `stream`).
 *   mode 0  independent random reads (a hop-table gather), 4 in flight per lane; element width
 *           = row_bytes (16, 8 or 4; 0 = 16)
 *   mode 1  one dependent chain of such reads per lane (a walker)
 *   mode 2  random rows of row_bytes (512 | 1024 | 2048) read by one wave each (a syn0 row)
 *   mode 3  the same rows read, modified and written back (a trained row)
 *   mode 4  one dependent chain of random 4-byte reads per lane with a binary search over an LDS
 *           table of row_bytes degree classes (a power of two, 64 .. 8192) between them: a walker on
 *           a graph numbered by descending degree whose entries are the neighbour id alone
 *   mode 5  the shape of a biased exact step: a dependent chain of hop entries of row_bytes (16 | 8)
 *           bytes, and with 49 % of the steps an independent 32-byte read of a wedge slot; the buffer
 *           is cut into E hop entries followed by E slots (E = buffer_bytes / (row_bytes + 32)); six
 *           waves per SIMD as the walk kernel runs
 * iters: accesses per lane (modes 0, 1) / rows per wave (modes 2, 3), a multiple of 4.
 * *accesses_host (host pointer, optional) receives the number of accesses the launch makes.
 * buffer: 16-byte aligned device memory, overwritten in mode 3.  sink: one device word. */
int n2v_mem_probe(void *buffer, int64_t buffer_bytes, int32_t mode, int32_t iters,
                  int32_t row_bytes, int64_t *accesses_host, uint32_t *sink, void *stream);

#ifdef __cplusplus
}
#endif
#endif
