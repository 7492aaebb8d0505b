// n2v_walk_uniform.hip -- K2 exact mode on a unit-weight graph with p == q == 1 (the
// reference's defaults, constants.py:22,26; BASELINE cfg 4).
//
// With every weight 1.0 and p == q == 1 the table generate_edge_alias_tables builds at a
// step (reference randomwalk.py:193-232) has probs == [1.0] * n exactly (the row sum of
// :172 is the integer n, n / n == 1.0, every x / 1.0 == 1.0), no slot is underfull, the loop
// of :182 never runs, and sampling_from_alias(r1, r2) (:86-99) returns pick = int(r1 * n)
// whatever r2 is.  A step is therefore two dependent gathers -- the row pointer pair of the
// current vertex and col[row + pick] -- and the kernel is bound by the chip's rate of random
// 64-byte sector reads (scripts/micro/gather_ceiling.hip: ~25-27 G such steps/s), so what it
// can save is requests:
//   * one LANE per walker, 8 waves per SIMD, nothing else in flight;
//   * the path is NOT stored word by word (a 4-byte store into a 324-byte-pitch row costs a
//     32-byte write request each, as many requests as the reads): every lane keeps the 16
//     words of the 64-byte sector of walks_out it is currently filling in registers
//     (selected by a v_cndmask chain, no scratch) and stores the sector whole when it is
//     complete -- four aligned 16-byte stores per 16 steps.
//   * with the hop table (n2v_hops_build, 16 bytes per edge: neighbour id + its row pointer
//     and degree) the two dependent gathers of a step become ONE: the entry that names the next
//     vertex also says where its row starts and how long it is.  The chip sustains ~50 G random
//     64-byte sector reads per second whatever the kernel does (profiles/r02_gather_ceiling.log),
//     so halving the sectors per step is the only lever left; it costs 16 B/edge of HBM
//     (12 GB at cfg 4 of 288 GB).
//   * round 4, the degree-ranked form (n2v_graph.rank_hops): the chip serves random 4-byte reads of
//     a 3 GB table at 52 G/s against 40 G/s for 16-byte reads of a 12 GB one (n2v_mem_probe modes
//     1 / 4, profiles/r4x_probe_classes.log), and an entry can be the neighbour's id alone once
//     vertices are numbered by descending degree: rows lie in rank order, so the row of a rank is
//     offset(class) + (rank - first(class)) * degree(class), the class found by a 13-step search of
//     an LDS table (the probe shows the search is free: 51.6 - 53.2 G/s).  The few top ranks whose
//     degrees are all different are looked up in a small cached table instead.
// Same uniform stream as every other kernel (step_bits, n2v_common.h): bit-identical walks.
#include "n2v_common.h"

namespace n2v {

// row start and degree of rank x (degree-ranked form): the head table for the top ranks, else the
// degree class of x by a fixed-depth search of the LDS table (first[0] == head_n <= x; the entry
// after the last class and the padding hold n_vertices / n_edges, never <= x).  8 bytes per class:
// its degree is (offset of the next class - its own) / (its number of ranks).
__device__ __forceinline__ void rank_row(const n2v_graph &g, const uint32_t *first, const uint32_t *off,
                                         uint32_t x, int64_t &vb, int &n) {
  if (x < (uint32_t)g.rank_head_n) {
    const uint64_t e = g.rank_head[x];
    vb = (int64_t)(e & N2V_HOP_ROW_MASK);
    n = (int)(e >> N2V_HOP_DEG_SHIFT);
    return;
  }
  int c = 0;
  for (int half = g.rank_classes >> 1; half > 0; half >>= 1)
    if (first[c + half] <= x) c += half;
  const uint32_t f0 = first[c], o0 = off[c];
  const uint32_t d = (off[c + 1] - o0) / (first[c + 1] - f0);
  n = (int)d;
  vb = (int64_t)(o0 + (x - f0) * d);  // < n_edges < 2^32
}

// kHops: 0 = CSR arrays (two gathers per step), 1 = the 16-byte hop table, 2 = the 8-byte hop
// table (round 3: the chip serves 8-byte gathers over a table half the size a quarter faster),
// 3 = the degree-ranked 4-byte table (blocks of 1024 threads, the class table in LDS: up to 8191
// classes in 64 KB, two blocks per CU)
template <int kHops, int kThreads>
__global__ __launch_bounds__(kThreads, 8) void walk_uniform_kernel(
    n2v_graph g, const int32_t *__restrict__ start_ids, int64_t n_start, int32_t num_walks,
    int32_t walk_length, uint64_t seed, int32_t *__restrict__ walks_out,
    uint8_t *__restrict__ valid_out, uint32_t *__restrict__ status) {
  const int lane = threadIdx.x & 63;
  const int64_t total = n_start * (int64_t)num_walks;
  const int L1 = walk_length + 1;
  // whole sectors need a 64-byte aligned output base (torch / hipMalloc give >= 256)
  const bool base_aligned = (reinterpret_cast<uintptr_t>(walks_out) & 63u) == 0;
  extern __shared__ uint32_t rank_lds[];
  const uint32_t *cls_first = rank_lds;
  const uint32_t *cls_where = rank_lds + (kHops == 3 ? g.rank_classes : 0);
  if (kHops == 3) {
    for (int c = threadIdx.x; c < g.rank_classes; c += kThreads) {
      rank_lds[c] = g.rank_class_first[c];
      rank_lds[g.rank_classes + c] = g.rank_class_off[c];
    }
    __syncthreads();
  }
  const bool emit_rank = kHops == 3 && g.rank_emit != 0;
  for (;;) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(&status[1], 64u);
    const int64_t base = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    if (base >= total) break;
    const int64_t r = base + lane;
    const bool have = r < total;
    int32_t v = -1;
    uint64_t h0 = 0;
    bool alive = have;
    if (have) {
      v = start_ids[r / num_walks];
      const int32_t ordinal = (int32_t)(r % num_walks) + 1;
      h0 = walker_stream(seed, (uint64_t)v * (uint64_t)num_walks + (uint64_t)(ordinal - 1));
      if (v < 0 || (int64_t)v >= g.n_vertices) {
        atomicOr(status, N2V_ST_RANGE);
        alive = false;
      }
    }
    int64_t vb = 0;
    int n = 0;
    int32_t v_emit = v;
    if (kHops == 3) {
      if (alive) {
        const int32_t rk = g.rank_of[v];
        if (emit_rank) v_emit = rk;
        rank_row(g, cls_first, cls_where, (uint32_t)rk, vb, n);
        alive = n > 0;  // fugue.py:132
      }
    } else if (alive) {
      vb = g.rowptr[v];
      n = (int)(g.rowptr[v + 1] - vb);
      alive = n > 0;  // fugue.py:132
      if (kHops == 2 && g.hop8_rowptr) vb = g.hop8_rowptr[v];  // the padded table's own row start
    }
    bool walking = alive;
    // absolute word index of path position 0; word a lives in buf[a & 15]
    const int64_t w0 = r * (int64_t)L1;
    int32_t buf[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) buf[k] = -1;
    int lo = (int)(w0 & 15);  // first word of the current sector that belongs to this row
    auto put = [&](int64_t a, int32_t x) {
      const int k = (int)(a & 15);
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) buf[kk] = (k == kk) ? x : buf[kk];
    };
    auto flush = [&](int64_t a) {  // words [sector(a) + lo, a] are complete: store them
      const int k = (int)(a & 15);
      int32_t *sec = walks_out + (a & ~(int64_t)15);
#if defined(N2V_ABLATE_UNIFORM) && N2V_ABLATE_UNIFORM == 1  // timing only: no path stores (placement diagnosis)
      if (buf[k] == 0x7fffffff) sec[k] = 0;
#else
      if (lo == 0 && k == 15 && base_aligned) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          reinterpret_cast<int4 *>(sec)[u] =
              make_int4(buf[4 * u], buf[4 * u + 1], buf[4 * u + 2], buf[4 * u + 3]);
      } else {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
          if (kk >= lo && kk <= k) sec[kk] = buf[kk];
      }
#endif
      lo = 0;
    };
    if (have) {
      put(w0, alive ? v_emit : -1);
      if ((w0 & 15) == 15 || walk_length == 0) flush(w0);
    }
    for (int step = 0; step < walk_length; ++step) {
      if (ballot64(have) == 0ull) break;
      int32_t x = -1;
      if (walking) {
        const uint64_t bits = step_bits(h0, (uint32_t)step);
        const int pick = pick_index((uint32_t)(bits >> 32), n);  // int(r1 * n); r2 is irrelevant
        if (kHops == 3) {
#if defined(N2V_ABLATE_UNIFORM) && N2V_ABLATE_UNIFORM == 2  // timing only: no table read (placement diagnosis)
          const uint32_t xr = (uint32_t)((bits >> 7) % (uint64_t)g.n_vertices);
#else
          const uint32_t xr = g.rank_hops[vb + pick];
#endif
          x = emit_rank ? (int32_t)xr : g.rank_vertex[xr];
          if (step + 1 < walk_length) rank_row(g, cls_first, cls_where, xr, vb, n);
        } else if (kHops == 2) {
          const uint64_t h = g.hops8[vb + pick];
          const int cb = g.hop8_col_bits, rb = g.hop8_row_bits;
          x = (int32_t)(h & ((1ull << cb) - 1ull));
          vb = (int64_t)(((h >> cb) & ((1ull << rb) - 1ull)) << g.hop8_align_shift);
          const uint64_t code = h >> (cb + rb), esc = (1ull << (64 - cb - rb)) - 1ull;
          n = (int)code;
          // a high-degree row: its degree is read from rowptr (few such rows: cached)
          if (code == esc) n = (int)(g.rowptr[x + 1] - g.rowptr[x]);
        } else if (kHops == 1) {
          const n2v_hop h = load_hop(g.hops + vb + pick);
          x = h.col;
          vb = hop_row(h);
          n = hop_deg(h);
        } else {
          x = g.col[vb + pick];
          if (step + 1 < walk_length) {
            vb = g.rowptr[x];
            n = (int)(g.rowptr[x + 1] - vb);
          }
        }
        // fugue.py:147: the walker vanishes at a sink, the rest of its row is -1
        if (step + 1 < walk_length && n == 0) {
          walking = false;
          alive = false;
        }
      }
      if (have) {
        const int64_t a = w0 + step + 1;
        put(a, x);
        if ((a & 15) == 15 || step + 1 == walk_length) flush(a);
      }
    }
    if (have) valid_out[r] = alive ? 1 : 0;
  }
}

}  // namespace n2v

// returns 1 when the kernel applies (and was launched), 0 when it does not, < 0 on error
extern "C" int n2v_walk_uniform_try(const n2v_graph *g, const int32_t *start_ids, int64_t n_start,
                                    int32_t num_walks, int32_t walk_length, double p, double q,
                                    uint64_t seed, int32_t *walks_out, uint8_t *valid_out,
                                    uint32_t *status, void *stream) {
  if (g->w != nullptr || g->w64 != nullptr || p != 1.0 || q != 1.0) return 0;
  const int64_t total = n_start * (int64_t)num_walks;
  if (total >= 0xffffff00ll) return 0;
  if (total == 0) return 1;
  // status[1] is the kernel's walker counter: start it at zero on the same stream
  if (hipMemsetAsync(status + 1, 0, sizeof(uint32_t), (hipStream_t)stream) != hipSuccess)
    return N2V_ELAUNCH;
  int64_t blocks = (total + 255) / 256;
  const int form = g->rank_hops ? 3 : g->hops8 ? 2 : (g->hops ? 1 : 0);
  if (form == 3) {
    const int P = g->rank_classes;
    if (P < 2 || P > 8192 || (P & (P - 1)) || !g->rank_of || !g->rank_class_first || !g->rank_class_off ||
        g->n_edges >= (1ll << 32) ||
        g->rank_head_n < 0 || g->rank_head_n > (1 << 22) || (g->rank_head_n > 0 && !g->rank_head) ||
        (g->rank_emit == 0 && !g->rank_vertex) || (g->rank_emit & ~1))
      return N2V_EINVAL;
    const size_t lds = (size_t)P * 8;
    blocks = (total + 1023) / 1024;
    const int64_t cap3 = n2v::resident_blocks((const void *)n2v::walk_uniform_kernel<3, 1024>, 1024, lds);
    if (blocks > cap3) blocks = cap3;
    hipLaunchKernelGGL((n2v::walk_uniform_kernel<3, 1024>), dim3((unsigned)blocks), dim3(1024), lds,
                       (hipStream_t)stream, *g, start_ids, n_start, num_walks, walk_length, seed, walks_out,
                       valid_out, status);
    if (hipGetLastError() != hipSuccess) return N2V_ELAUNCH;
    return 1;
  }
  if (form == 2 && (g->hop8_col_bits < 1 || g->hop8_row_bits < 1 ||
                    g->hop8_col_bits + g->hop8_row_bits > 62 || g->hop8_align_shift < 0 ||
                    g->hop8_align_shift > 6 || (g->hop8_align_shift > 0 && !g->hop8_rowptr)))
    return N2V_EINVAL;
  const void *fn = form == 2   ? (const void *)n2v::walk_uniform_kernel<2, 256>
                   : form == 1 ? (const void *)n2v::walk_uniform_kernel<1, 256>
                               : (const void *)n2v::walk_uniform_kernel<0, 256>;
  const int64_t cap = n2v::resident_blocks(fn, 256, 0);
  if (blocks > cap) blocks = cap;
#define N2V_UNIFORM_LAUNCH(F)                                                                     \
  hipLaunchKernelGGL((n2v::walk_uniform_kernel<F, 256>), dim3((unsigned)blocks), dim3(256), 0,   \
                     (hipStream_t)stream, *g, start_ids, n_start, num_walks, walk_length, seed,  \
                     walks_out, valid_out, status)
  if (form == 2)
    N2V_UNIFORM_LAUNCH(2);
  else if (form == 1)
    N2V_UNIFORM_LAUNCH(1);
  else
    N2V_UNIFORM_LAUNCH(0);
#undef N2V_UNIFORM_LAUNCH
  if (hipGetLastError() != hipSuccess) return N2V_ELAUNCH;
  return 1;
}
