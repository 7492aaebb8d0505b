// n2v_partition.hip -- the entry points of graph-partitioned walking (node2vec_amd/partitioned.py,
// SURVEY.md 8f-4): the reference's per-step joins (fugue.py:146-149) for a graph whose rows are
// split over ranks by vertex range.
//   n2v_partition_step   one step of the walkers resident on one part.  Dispatches to
//                          - partition_step_wedge_kernel (n2v_walk_wedge.hip): unit weights, the
//                            wedge list of the edge walked last travelled with the walker (or
//                            p == q == 1): one LANE per walker, closed-form pairing;
//                          - partition_step_unit_kernel (n2v_walk_unit.hip): unit weights, the row
//                            of the previous vertex travelled: one wave per walker;
//                          - partition_step_kernel (n2v_walk.hip): any weights, one wave per walker.
//   n2v_partition_route  what happens to every walker after its step: path record, next header,
//                        destination, words to carry (elementwise)
//   n2v_partition_group  the walkers grouped by destination: a stable counting sort (per-block
//                        counts, one scan, one scatter)
//   n2v_gather_rows / n2v_gather_wedges  pack what travels with the migrating walkers
//   n2v_partition_forward  route + group + gather in ONE launch and without a size known to the
//                        host: every walker is appended to the mailbox of its destination
//                        (wave-aggregated atomics), its wedge list copied behind it
#include "n2v_common.h"

extern "C" int n2v_partition_step_generic_launch(const int64_t *rowptr, const int32_t *col,
                                                 const float *w, const double *w64, int64_t lo,
                                                 int64_t n_local, const int64_t *head,
                                                 int32_t head_cols, const int64_t *src_ptr,
                                                 const int32_t *src_ids, int64_t k, double p,
                                                 double q, uint64_t seed, int32_t *next_out,
                                                 int64_t *edge_out, uint32_t *status,
                                                 void *stream);  // n2v_walk.hip
extern "C" int n2v_partition_step_unit_try(const int64_t *rowptr, const int32_t *col, int64_t lo,
                                           int64_t n_local, const int64_t *head, int32_t head_cols,
                                           const int64_t *src_ptr, const int32_t *src_ids,
                                           int32_t wedge_lists, int64_t k, double p, double q,
                                           uint64_t seed, int32_t *next_out, int64_t *edge_out,
                                           uint32_t *status, void *stream);  // n2v_walk_unit.hip

extern "C" int n2v_partition_step(const int64_t *rowptr, const int32_t *col, const float *w,
                                  const double *w64, int64_t lo, int64_t n_local,
                                  const int64_t *head, int32_t head_cols, const int64_t *src_ptr,
                                  const int32_t *src_ids, int32_t src_kind, int64_t k, double p,
                                  double q, uint64_t seed, int32_t *next_out, int64_t *edge_out,
                                  uint32_t *status, void *stream) {
  if (k < 0 || n_local < 0 || k >= 0xfffffff0ll || (w && w64)) return N2V_EINVAL;
  if (p == 0.0 || q == 0.0) return N2V_EINVAL;  // randomwalk.py:209-212 (ValueError upstream)
  if (src_kind != N2V_SRC_ROWS && src_kind != N2V_SRC_WEDGES && src_kind != N2V_SRC_WEDGES_AT)
    return N2V_EINVAL;
  if (head_cols < 4) return N2V_EINVAL;  // (wedge lists with p or q != 1: 5, checked below)
  if (src_kind != N2V_SRC_ROWS && (w || w64)) return N2V_EINVAL;  // unit-weight parts only
  if (k == 0) return N2V_OK;
  if (!rowptr || !head || !next_out || !status) return N2V_EINVAL;  // (col: NULL for a part without edges)
  if (q != 1.0 && !src_ptr) return N2V_EINVAL;
  if (hipMemsetAsync(status + 1, 0, sizeof(uint32_t), (hipStream_t)stream) != hipSuccess)
    return N2V_ELAUNCH;
  if (!w && !w64) {  // unit weights: the closed forms of n2v_walk_unit.hip
    const int rc = n2v_partition_step_unit_try(rowptr, col, lo, n_local, head, head_cols, src_ptr,
                                               src_ids, src_kind == N2V_SRC_ROWS ? 0 : src_kind, k, p,
                                               q, seed, next_out, edge_out, status, stream);
    if (rc != 0) return rc < 0 ? rc : N2V_OK;
  }
  if (src_kind != N2V_SRC_ROWS) return N2V_EINVAL;  // (p, q) outside the unit kernels' range
  return n2v_partition_step_generic_launch(rowptr, col, w, w64, lo, n_local, head, head_cols, src_ptr,
                                           src_ids, k, p, q, seed, next_out, edge_out, status, stream);
}

namespace n2v {

// rows[j] of a CSR copied back to back: out[out_ptr[j] ..] = ids[ptr[rows[j]] .. ptr[rows[j] + 1])
// (the rows that leave with migrating walkers); one wave per row
__global__ __launch_bounds__(256) void gather_rows_kernel(const int64_t *__restrict__ ptr,
                                                          const int32_t *__restrict__ ids,
                                                          const int64_t *__restrict__ rows,
                                                          const int64_t *__restrict__ out_ptr,
                                                          int64_t k, int32_t *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  for (int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); j < k; j += n_waves) {
    const int64_t r = rows[j];
    const int64_t b = ptr[r], o = out_ptr[j];
    const int64_t len = out_ptr[j + 1] - o;
    for (int64_t t = lane; t < len; t += 64) out[o + t] = ids[b + t];
  }
}

}  // namespace n2v

// the wedge lists of `edges` (local edge indices of one part) copied back to back as 32-bit
// positions, and the fifth header word of the walkers that carry them
__global__ __launch_bounds__(256) void n2v_gather_wedges_kernel(
    const uint32_t *__restrict__ edge_classes, const uint64_t *__restrict__ wedge_off,
    const void *__restrict__ wedge_pos, int wide, const int64_t *__restrict__ edges,
    const int64_t *__restrict__ out_ptr, int64_t k, int32_t *__restrict__ out,
    int64_t *__restrict__ head, int head_cols) {
  const int lane = threadIdx.x & 63;
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  for (int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); j < k; j += n_waves) {
    const int64_t e = edges[j];
    const uint64_t raw = wedge_off[e];
    const int64_t b = (int64_t)(raw & N2V_WEDGE_OFF_MASK), o = out_ptr[j];
    const int64_t len = out_ptr[j + 1] - o;
    if (lane == 0)
      head[j * head_cols + 4] = (int64_t)((uint64_t)edge_classes[e] | (raw >> N2V_WEDGE_RPOS_SHIFT) << 32);
    for (int64_t t = lane; t < len; t += 64)
      out[o + t] = wide ? (int32_t)reinterpret_cast<const uint32_t *>(wedge_pos)[b + t]
                        : (int32_t)reinterpret_cast<const uint16_t *>(wedge_pos)[b + t];
  }
}

extern "C" int n2v_gather_rows(const int64_t *ptr, const int32_t *ids, const int64_t *rows,
                               const int64_t *out_ptr, int64_t k, int32_t *out, void *stream) {
  if (k < 0) return N2V_EINVAL;
  if (k == 0) return N2V_OK;
  if (!ptr || !ids || !rows || !out_ptr || !out) return N2V_EINVAL;
  int64_t blocks = (k + 3) / 4;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(n2v::gather_rows_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, ptr, ids, rows, out_ptr, k, out);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}

// what happens to every walker after its step (partitioned walking): the path record, the header
// it travels on with, the rank it goes to and how many words go with it -- the elementwise half of
// the routing; the grouping by destination is a sort of dest_out by the caller
__global__ __launch_bounds__(256) void n2v_partition_route_kernel(
    const int64_t *__restrict__ head_in, int head_cols, const int32_t *__restrict__ next,
    const int64_t *__restrict__ edge, int64_t k, int walk_length, const int64_t *__restrict__ bounds,
    int n_parts, int carry, const int64_t *__restrict__ rowptr, int64_t lo,
    const uint32_t *__restrict__ edge_classes, int64_t *__restrict__ log_out,
    int64_t *__restrict__ head_out, int32_t *__restrict__ dest_out, int64_t *__restrict__ len_out,
    int64_t *__restrict__ src_out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < k;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t *hd = head_in + i * head_cols;
    const int64_t row = hd[0], key = hd[1], v = (int64_t)(uint32_t)hd[2], step = (int64_t)(uint32_t)hd[3];
    const int32_t nx = next[i];
    int32_t dest = n_parts;  // not forwarded
    int64_t len = 0, src = 0;
    if (nx < 0) {
      // the current vertex has no out-edges: the walker vanished on arrival (fugue.py:147)
      log_out[3 * i] = row;
      log_out[3 * i + 1] = -1;
      log_out[3 * i + 2] = -1;
    } else {
      log_out[3 * i] = row;
      log_out[3 * i + 1] = step + 1;
      log_out[3 * i + 2] = nx;
      if (step + 1 < walk_length) {
        int a = 0, b = n_parts;  // last part whose first vertex is <= nx
        while (b - a > 1) {
          const int mid = (a + b) >> 1;
          if (bounds[mid] <= (int64_t)nx)
            a = mid;
          else
            b = mid;
        }
        dest = a;
        if (carry == N2V_SRC_WEDGES + 1) {
          src = edge[i];
          len = (int64_t)(edge_classes[src] & N2V_EC_SHARED_MASK);
        } else if (carry == N2V_SRC_WEDGES + 2) {  // q == 1: counts and return position only
          src = edge[i];
        } else if (carry == N2V_SRC_ROWS + 1) {
          src = v - lo;
          len = rowptr[src + 1] - rowptr[src];
        }
      }
    }
    int64_t *ho = head_out + i * head_cols;
    ho[0] = row;
    ho[1] = key;
    ho[2] = (v << 32) | (int64_t)(uint32_t)nx;
    ho[3] = step + 1;
    for (int c = 4; c < head_cols; ++c) ho[c] = 0;
    dest_out[i] = dest;
    len_out[i] = len;
    src_out[i] = src;
  }
}

extern "C" int n2v_partition_route(const int64_t *head_in, int32_t head_cols, const int32_t *next,
                                   const int64_t *edge, int64_t k, int32_t walk_length,
                                   const int64_t *bounds, int32_t n_parts, int32_t carry,
                                   const int64_t *rowptr, int64_t lo, const uint32_t *edge_classes,
                                   int64_t *log_out, int64_t *head_out, int32_t *dest_out,
                                   int64_t *len_out, int64_t *src_out, void *stream) {
  if (k < 0 || head_cols < 4 || n_parts < 1 || carry < 0 || carry > 3) return N2V_EINVAL;
  if (k == 0) return N2V_OK;
  if (!head_in || !next || !bounds || !log_out || !head_out || !dest_out || !len_out || !src_out)
    return N2V_EINVAL;
  if (carry == N2V_SRC_ROWS + 1 && !rowptr) return N2V_EINVAL;
  if (carry >= N2V_SRC_WEDGES + 1 && !edge) return N2V_EINVAL;  // (edge_classes: NULL for a part without edges)
  int64_t blocks = (k + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(n2v_partition_route_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, head_in, (int)head_cols, next, edge, k, (int)walk_length,
                     bounds, (int)n_parts, (int)carry, rowptr, lo, edge_classes, log_out, head_out,
                     dest_out, len_out, src_out);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}

// ---- grouping by destination: a stable counting sort in three launches ---------------------------
// (a block owns 256 consecutive walkers; destinations 0 .. n_parts, n_parts = "not forwarded")
constexpr int kGroupMaxParts = 64;

__global__ __launch_bounds__(256) void n2v_group_count_kernel(const int32_t *__restrict__ dest, int64_t k,
                                                              int p1, int64_t *__restrict__ work) {
  __shared__ int cnt[kGroupMaxParts + 1];
  for (int d = threadIdx.x; d < p1; d += 256) cnt[d] = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < k) atomicAdd(&cnt[dest[i]], 1);
  __syncthreads();
  for (int d = threadIdx.x; d < p1; d += 256) work[(int64_t)blockIdx.x * p1 + d] = cnt[d];
}

// work[b][d] := walkers of destination d in the blocks before b; work[nb][d] := where destination d
// starts in the grouped order; cuts[d] = the same (cuts[n_parts] = forwarded walkers)
__global__ __launch_bounds__(128) void n2v_group_scan_kernel(int64_t nb, int p1, int64_t *__restrict__ work,
                                                             int64_t *__restrict__ cuts) {
  __shared__ int64_t total[kGroupMaxParts + 1];
  const int d = threadIdx.x;
  if (d < p1) {
    int64_t run = 0;
    for (int64_t b = 0; b < nb; ++b) {
      const int64_t c = work[b * p1 + d];
      work[b * p1 + d] = run;
      run += c;
    }
    total[d] = run;
  }
  __syncthreads();
  if (d == 0) {
    int64_t run = 0;
    for (int t = 0; t < p1; ++t) {
      work[nb * p1 + t] = run;
      cuts[t] = run;
      run += total[t];
    }
  }
}

__global__ __launch_bounds__(256) void n2v_group_scatter_kernel(
    const int32_t *__restrict__ dest, const int64_t *__restrict__ head, int head_cols,
    const int64_t *__restrict__ len, const int64_t *__restrict__ src, int64_t k, int p1, int64_t nb,
    const int64_t *__restrict__ work, int64_t *__restrict__ head_out, int64_t *__restrict__ len_out,
    int64_t *__restrict__ src_out) {
  __shared__ int wave_cnt[4][kGroupMaxParts + 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int d = i < k ? dest[i] : -1;
  int rank = 0;
  for (int t = 0; t < p1; ++t) {  // stable rank among the walkers of the same destination
    const unsigned long long m = __ballot(d == t);
    if (d == t) rank = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wv][t] = __popcll(m);
  }
  __syncthreads();
  if (d >= 0) {
    for (int w2 = 0; w2 < wv; ++w2) rank += wave_cnt[w2][d];
    const int64_t pos = work[nb * p1 + d] + work[(int64_t)blockIdx.x * p1 + d] + rank;
    for (int c = 0; c < head_cols; ++c) head_out[pos * head_cols + c] = head[i * head_cols + c];
    len_out[pos] = len[i];
    src_out[pos] = src[i];
  }
}

extern "C" int n2v_partition_group(const int32_t *dest, const int64_t *head, int32_t head_cols,
                                   const int64_t *len, const int64_t *src, int64_t k, int32_t n_parts,
                                   int64_t *work, int64_t *head_out, int64_t *len_out,
                                   int64_t *src_out, int64_t *cuts_out, void *stream) {
  if (k < 0 || head_cols < 1 || n_parts < 1 || n_parts > kGroupMaxParts) return N2V_EINVAL;
  if (!cuts_out) return N2V_EINVAL;
  const int p1 = n_parts + 1;
  if (k == 0) {
    N2V_HIP_CHECK(hipMemsetAsync(cuts_out, 0, sizeof(int64_t) * p1, (hipStream_t)stream));
    return N2V_OK;
  }
  if (!dest || !head || !len || !src || !work || !head_out || !len_out || !src_out) return N2V_EINVAL;
  const int64_t nb = (k + 255) / 256;
  if (nb >= (1ll << 31)) return N2V_EINVAL;
  hipLaunchKernelGGL(n2v_group_count_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, dest,
                     k, p1, work);
  hipLaunchKernelGGL(n2v_group_scan_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, nb, p1, work,
                     cuts_out);
  hipLaunchKernelGGL(n2v_group_scatter_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, dest,
                     head, (int)head_cols, len, src, k, p1, nb, work, head_out, len_out, src_out);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}

extern "C" int n2v_gather_wedges(const uint32_t *edge_classes, const uint64_t *wedge_off,
                                 const void *wedge_pos, int32_t wide, const int64_t *edges,
                                 const int64_t *out_ptr, int64_t k, int32_t *out, int64_t *head,
                                 int32_t head_cols, void *stream) {
  if (k < 0 || head_cols < 5) return N2V_EINVAL;
  if (k == 0) return N2V_OK;
  if (!edge_classes || !wedge_off || !wedge_pos || !edges || !out_ptr || !out || !head)
    return N2V_EINVAL;
  int64_t blocks = (k + 3) / 4;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(n2v_gather_wedges_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, edge_classes, wedge_off, wedge_pos, (int)wide, edges, out_ptr,
                     k, out, head, (int)head_cols);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}

// ---- route + group + gather in one launch (round 4) ------------------------------------------------
// One LANE per walker.  What n2v_partition_route computes per walker, then instead of a sort by
// destination and a prefix sum over the list lengths (three launches, a scan, a host read of the
// cuts, a gather): the lanes of a wave that go to the same part reserve consecutive places in that
// part's mailbox with ONE atomic add, the wave reserves the words of its lists with one more, and
// every lane writes its header and copies its list (lists of 32 words or more by the whole wave).
// The order inside a mailbox is whatever the atomics gave -- a walk depends on its key and step
// alone (counter-based RNG), never on its place in a batch.
constexpr int kFwdThreads = 1024;  // a block reserves mailbox places ONCE per destination and pass
constexpr int kFwdMaxParts = 256;

__global__ __launch_bounds__(kFwdThreads) void n2v_partition_forward_kernel(
    const int64_t *__restrict__ head_in, int head_cols, const int32_t *__restrict__ next,
    const int64_t *__restrict__ edge, int64_t k, int walk_length, const int64_t *__restrict__ bounds,
    int n_parts, int carry, const uint32_t *__restrict__ edge_classes,
    const uint64_t *__restrict__ wedge_off, const void *__restrict__ wedge_pos, int wide,
    int64_t *__restrict__ box_head, int64_t *__restrict__ box_off, int32_t *__restrict__ box_words,
    unsigned long long *__restrict__ box_count, int64_t cap, int64_t wcap,
    const int64_t *__restrict__ box_starts, int64_t *__restrict__ log_out,
    int32_t *__restrict__ walks_out, uint8_t *__restrict__ valid_out, uint32_t *__restrict__ status) {
  // Same-address atomics serialise in L2 (~0.1 us each): one per wave and destination made this
  // kernel take 0.85 ms for 6 x 10^5 walkers.  So the block counts first -- walkers and words per
  // wave and destination in LDS --, ONE thread per destination reserves the block's places and
  // words in that part's mailbox, and every wave takes its share of them.
  __shared__ uint32_t cnt[kFwdThreads / 64][kFwdMaxParts];   // [wave][dest] walkers
  __shared__ uint32_t cntw[kFwdThreads / 64][kFwdMaxParts];  // [wave][dest] words
  __shared__ unsigned long long base[kFwdMaxParts], basew[kFwdMaxParts];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t k_up = (k + kFwdThreads - 1) / kFwdThreads * kFwdThreads;  // whole blocks stay in the loop
  for (int64_t i = (int64_t)blockIdx.x * kFwdThreads + threadIdx.x; i < k_up;
       i += (int64_t)gridDim.x * kFwdThreads) {
    // (a header whose output row is negative is an empty slot of a capacity-bounded mailbox: nothing is
    // logged or forwarded for it)
    const bool have = i < k && head_in[i * head_cols] >= 0;
    int64_t row = 0, key = 0, v = 0, step = 0, src = 0;
    int32_t nx = -1;
    int dest = -1;  // -1: not forwarded
    uint32_t len = 0;
    uint64_t extra = 0;
    for (int d = lane; d < n_parts; d += 64) {
      cnt[wv][d] = 0;
      cntw[wv][d] = 0;
    }
    if (have) {
      const int64_t *hd = head_in + i * head_cols;
      row = hd[0], key = hd[1], v = (int64_t)(uint32_t)hd[2], step = (int64_t)(uint32_t)hd[3];
      nx = next[i];
      if (log_out) {  // (row, step + 1, next), or (row, -1, -1): vanished on arrival (fugue.py:147)
        log_out[3 * i] = row;
        log_out[3 * i + 1] = nx < 0 ? -1 : step + 1;
        log_out[3 * i + 2] = nx < 0 ? -1 : (int64_t)nx;
      } else if (nx < 0) {
        valid_out[row] = 0;
      } else {
        walks_out[row * (int64_t)(walk_length + 1) + step + 1] = nx;
      }
      if (nx >= 0 && step + 1 < walk_length) {
        int a = 0, b = n_parts;  // last part whose first vertex is <= nx
        while (b - a > 1) {
          const int mid = (a + b) >> 1;
          if (bounds[mid] <= (int64_t)nx)
            a = mid;
          else
            b = mid;
        }
        dest = a;
        if (carry >= N2V_SRC_WEDGES + 1 && (!edge_classes || !wedge_off)) {
          atomicOr(status, N2V_ST_RANGE);  // a part without edges (NULL tables) forwards nobody
          dest = -1;
        } else if (carry >= N2V_SRC_WEDGES + 1) {
          src = edge[i];
          const uint32_t ec = edge_classes[src];
          const uint64_t raw = wedge_off[src];
          extra = (uint64_t)ec | (raw >> N2V_WEDGE_RPOS_SHIFT) << 32;
          if (carry == N2V_SRC_WEDGES + 1) len = ec & N2V_EC_SHARED_MASK;
          src = (int64_t)(raw & N2V_WEDGE_OFF_MASK);  // from here on: where the list starts
        }
      }
    }
    // inside the wave, per destination: the rank of every lane among the lanes that go there and
    // the words of the lanes before it (a masked scan)
    int rank = 0;
    uint32_t wrank = 0;
    uint64_t todo = n2v::ballot64(dest >= 0);
    while (todo) {
      const int d = __shfl(dest, __builtin_ctzll(todo), 64);
      const uint64_t m = n2v::ballot64(dest == d);
      const uint32_t x = dest == d ? len : 0u;
      uint32_t incl = x;
      if (carry == N2V_SRC_WEDGES + 1)  // (only then do words travel)
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t t = __shfl_up(incl, o, 64);
          if (lane >= o) incl += t;
        }
      if (dest == d) {
        rank = __popcll(m & ((1ull << lane) - 1ull));
        wrank = incl - x;
      }
      if (lane == 63) {
        cnt[wv][d] = (uint32_t)__popcll(m);
        cntw[wv][d] = incl;
      }
      todo &= ~m;
    }
    __syncthreads();
    // one thread per destination: the block's reservation of places and of words in that part's
    // mailbox; cnt / cntw[w][d] become the offset of wave w inside it
    for (int d = threadIdx.x; d < n_parts; d += kFwdThreads) {
      uint32_t run = 0, runw = 0;
      for (int w2 = 0; w2 < kFwdThreads / 64; ++w2) {
        const uint32_t c = cnt[w2][d], cw = cntw[w2][d];
        cnt[w2][d] = run;
        cntw[w2][d] = runw;
        run += c;
        runw += cw;
      }
      base[d] = run ? atomicAdd(&box_count[d], (unsigned long long)run) : 0ull;
      basew[d] = runw ? atomicAdd(&box_count[n_parts + d], (unsigned long long)runw) : 0ull;
    }
    __syncthreads();
    int64_t pos = -1, woff = 0;
    if (dest >= 0) {
      pos = (int64_t)base[dest] + cnt[wv][dest] + rank;
      woff = (int64_t)basew[dest] + cntw[wv][dest] + (int64_t)wrank;  // inside the pool of `dest`
    }
    __syncthreads();  // (cnt is cleared at the top of the next pass)
    // where the mailbox and the word pool of `dest` start and how much they hold: equal shares of the
    // arrays, or (box_starts) ragged ones -- a capacity per destination
    int64_t hstart = 0, hcap = 0, wstart = 0, wcap_d = 0;
    if (dest >= 0) {
      if (box_starts) {
        hstart = box_starts[dest], hcap = box_starts[dest + 1] - hstart;
        wstart = box_starts[n_parts + 1 + dest], wcap_d = box_starts[n_parts + 2 + dest] - wstart;
      } else {
        hstart = (int64_t)dest * cap, hcap = cap, wstart = (int64_t)dest * wcap, wcap_d = wcap;
      }
    }
    bool fits = dest >= 0 && pos < hcap && woff + (int64_t)len <= wcap_d;
    if (dest >= 0 && !fits) atomicOr(status, N2V_ST_OVERFLOW);
    if (fits) {
      int64_t *ho = box_head + (hstart + pos) * head_cols;
      ho[0] = row;
      ho[1] = key;
      ho[2] = (v << 32) | (int64_t)(uint32_t)nx;
      ho[3] = step + 1;
      if (head_cols > 4) ho[4] = (int64_t)extra;
      for (int c = 5; c < head_cols; ++c) ho[c] = 0;
      box_off[hstart + pos] = woff;
    }
    if (!fits) len = 0;
    const int64_t wabs = wstart + woff;  // in box_words
    // lists: short ones lane by lane, long ones by the whole wave
    uint64_t big = n2v::ballot64(len >= 32u);
    while (big) {
      const int l = __builtin_ctzll(big);
      big &= big - 1;
      const int64_t b = __shfl(src, l, 64), o = __shfl(wabs, l, 64);
      const int n = (int)__shfl(len, l, 64);
      for (int t = lane; t < n; t += 64)
        box_words[o + t] = wide ? (int32_t)reinterpret_cast<const uint32_t *>(wedge_pos)[b + t]
                                : (int32_t)reinterpret_cast<const uint16_t *>(wedge_pos)[b + t];
    }
    if (len < 32u)
      for (uint32_t t = 0; t < len; ++t)
        box_words[wabs + t] = wide ? (int32_t)reinterpret_cast<const uint32_t *>(wedge_pos)[src + t]
                                   : (int32_t)reinterpret_cast<const uint16_t *>(wedge_pos)[src + t];
  }
}

static int partition_forward(const int64_t *head_in, int32_t head_cols, const int32_t *next,
                             const int64_t *edge, int64_t k, int32_t walk_length,
                             const int64_t *bounds, int32_t n_parts, int32_t carry,
                             const uint32_t *edge_classes, const uint64_t *wedge_off,
                             const void *wedge_pos, int32_t wide, int64_t *box_head,
                             int64_t *box_off, int32_t *box_words, unsigned long long *box_count,
                             int64_t cap, int64_t wcap, const int64_t *box_starts, int64_t *log_out,
                             int32_t *walks_out, uint8_t *valid_out, uint32_t *status, void *stream) {
  if (k < 0 || head_cols < 4 || n_parts < 1 || n_parts > kFwdMaxParts || cap < 0 || wcap < 0 ||
      walk_length < 0)
    return N2V_EINVAL;
  if (carry != 0 && carry != N2V_SRC_WEDGES + 1 && carry != N2V_SRC_WEDGES + 2) return N2V_EINVAL;
  if (carry != 0 && head_cols < 5) return N2V_EINVAL;
  if (k == 0) return N2V_OK;
  if (!head_in || !next || !bounds || !box_head || !box_off || !box_count || !status) return N2V_EINVAL;
  if (!log_out && (!walks_out || !valid_out)) return N2V_EINVAL;
  if (carry != 0 && !edge) return N2V_EINVAL;  // (edge_classes, wedge_off: NULL for a part without edges)
  if (carry == N2V_SRC_WEDGES + 1 && !box_words) return N2V_EINVAL;
  int64_t blocks = (k + kFwdThreads - 1) / kFwdThreads;
  const int64_t cap_blocks = n2v::resident_blocks((const void *)n2v_partition_forward_kernel, kFwdThreads, 0);
  if (blocks > cap_blocks) blocks = cap_blocks;
  hipLaunchKernelGGL(n2v_partition_forward_kernel, dim3((unsigned)blocks), dim3(kFwdThreads), 0, (hipStream_t)stream,
                     head_in, (int)head_cols, next, edge, k, (int)walk_length, bounds, (int)n_parts, (int)carry,
                     edge_classes, wedge_off, wedge_pos, (int)wide, box_head, box_off, box_words, box_count, cap,
                     wcap, box_starts, log_out, walks_out, valid_out, status);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}

extern "C" int n2v_partition_forward(const int64_t *head_in, int32_t head_cols, const int32_t *next,
                                     const int64_t *edge, int64_t k, int32_t walk_length,
                                     const int64_t *bounds, int32_t n_parts, int32_t carry,
                                     const uint32_t *edge_classes, const uint64_t *wedge_off,
                                     const void *wedge_pos, int32_t wide, int64_t *box_head,
                                     int64_t *box_off, int32_t *box_words, unsigned long long *box_count,
                                     int64_t cap, int64_t wcap, int64_t *log_out, int32_t *walks_out,
                                     uint8_t *valid_out, uint32_t *status, void *stream) {
  return partition_forward(head_in, head_cols, next, edge, k, walk_length, bounds, n_parts, carry, edge_classes,
                           wedge_off, wedge_pos, wide, box_head, box_off, box_words, box_count, cap, wcap, nullptr,
                           log_out, walks_out, valid_out, status, stream);
}

extern "C" int n2v_partition_forward_boxes(const int64_t *head_in, int32_t head_cols, const int32_t *next,
                                           const int64_t *edge, int64_t k, int32_t walk_length,
                                           const int64_t *bounds, int32_t n_parts, int32_t carry,
                                           const uint32_t *edge_classes, const uint64_t *wedge_off,
                                           const void *wedge_pos, int32_t wide, int64_t *box_head,
                                           int64_t *box_off, int32_t *box_words,
                                           unsigned long long *box_count, const int64_t *box_starts,
                                           int64_t *log_out, uint32_t *status, void *stream) {
  if (!box_starts || !log_out) return N2V_EINVAL;
  return partition_forward(head_in, head_cols, next, edge, k, walk_length, bounds, n_parts, carry, edge_classes,
                           wedge_off, wedge_pos, wide, box_head, box_off, box_words, box_count, 0, 0, box_starts,
                           log_out, nullptr, nullptr, status, stream);
}
