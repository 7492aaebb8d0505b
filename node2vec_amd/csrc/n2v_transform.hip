// n2v_transform.hip -- the reference's transformer-level functions, one partition per launch.
//
// n2v_walk never materialises the per-step table of generate_edge_alias_tables.  The
// reference's own row-level protocol (next_step_random_walk, randomwalk.py:300-339: rows of
// {src, path, src_neighbors, dst_neighbors} in, one draw with two EXPLICIT uniforms out) and
// its known-answer tests (tests/test_randomwalk.py:131-189, :268-306) are stated on the
// materialised table, so this file provides that form for a batch of rows:
//
//   n2v_edge_bias   the biased weights of randomwalk.py:219-231 for every row (w/p, w, w/q)
//   (n2v_alias_build on the biased rows = generate_alias_tables, randomwalk.py:157-190)
//   n2v_alias_draw  sampling_from_alias / sampling_from_alias_wiki (:70-99) + the neighbour
//                   lookup of RandomPath.append (:140-144) with caller-supplied fp64 uniforms
//
// fp64 throughout, one IEEE operation per reference operation (-ffp-contract=off).
#include "n2v_common.h"

namespace n2v {

// one thread per neighbour entry; the row of an entry is found by binary search over rowptr
__global__ __launch_bounds__(256) void edge_bias_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ ids,
    const float *__restrict__ w, const double *__restrict__ w64,
    const int32_t *__restrict__ src_id, const int64_t *__restrict__ src_rowptr,
    const int32_t *__restrict__ src_nbs, int64_t n_rows, int64_t nnz, double p, double q,
    double *__restrict__ w_out) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz;
       e += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = 0, hi = n_rows;  // last row with rowptr[row] <= e
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (rowptr[mid] <= e)
        lo = mid;
      else
        hi = mid;
    }
    const int64_t row = lo;
    double weight = 1.0;  // plain branches: never a select of two loads of different widths
    if (w64)
      weight = w64[e];
    else if (w)
      weight = (double)w[e];
    const int32_t s = src_id ? src_id[row] : -1;
    double b = weight;  // first step (src < 0): generate_alias_tables(dst weights), :319-320
    if (s >= 0) {
      const int32_t x = ids[e];
      if (x == s) {
        b = weight / p;  // :223-224
      } else {
        const int64_t sb = src_rowptr[row];
        const int m = (int)(src_rowptr[row + 1] - sb);
        if (!member_sorted_lane(src_nbs + sb, m, x)) b = weight / q;  // :229-230 (else :226-227)
      }
    }
    w_out[e] = b;
  }
}

__global__ __launch_bounds__(256) void alias_draw_kernel(
    const int64_t *__restrict__ rowptr, const n2v_slot *__restrict__ slots, int64_t n_rows,
    const double *__restrict__ r1, const double *__restrict__ r2,
    int32_t *__restrict__ vertex_out, uint32_t *__restrict__ status) {
  for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < n_rows;
       row += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = rowptr[row];
    const int64_t n = rowptr[row + 1] - b;
    if (n <= 0) {
      vertex_out[row] = -1;
      continue;
    }
    const double first = r1[row];
    const double scaled = first * (double)n;  // int(first_random * len(alias)), :95 / :79
    int64_t pick = (int64_t)scaled;
    if (pick < 0 || pick >= n) {  // the reference raises IndexError for r1 outside [0, 1)
      atomicOr(status, N2V_ST_RANGE);
      vertex_out[row] = -1;
      continue;
    }
    const n2v_slot sl = slots[b + pick];
    // two uniforms: second_random < probs[pick] (:96); one uniform (wiki): y = n * r - pick (:80-81)
    const double y = r2 ? r2[row] : scaled - (double)pick;
    vertex_out[row] = (y < sl.prob) ? sl.col : sl.alias;
  }
}

// the two uniforms n2v_walk uses at `step` of the walker with stream key `key` (DESIGN.md "RNG"):
// r = u / 2^32, exact in fp64, so a draw from a materialised table with them is the draw of
// n2v_walk's exact mode (pick = int(r1 * n) is the same integer, r2 < prob the same comparison)
__global__ __launch_bounds__(256) void walk_uniforms_kernel(uint64_t seed,
                                                            const int64_t *__restrict__ key,
                                                            const int32_t *__restrict__ step,
                                                            int64_t n, double *__restrict__ r1,
                                                            double *__restrict__ r2) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t bits = step_bits(walker_stream(seed, (uint64_t)key[i]), (uint32_t)step[i]);
    r1[i] = (double)(uint32_t)(bits >> 32) * (1.0 / 4294967296.0);
    r2[i] = (double)(uint32_t)bits * (1.0 / 4294967296.0);
  }
}

}  // namespace n2v

extern "C" int n2v_walk_uniforms(uint64_t seed, const int64_t *key, const int32_t *step, int64_t n,
                                 double *r1_out, double *r2_out, void *stream) {
  if (n < 0) return N2V_EINVAL;
  if (n == 0) return N2V_OK;
  if (!key || !step || !r1_out || !r2_out) return N2V_EINVAL;
  int64_t blocks = (n + 255) / 256;
  const int64_t cap = n2v::resident_blocks((const void *)n2v::walk_uniforms_kernel, 256, 0) * 2;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(n2v::walk_uniforms_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, seed, key, step, n, r1_out, r2_out);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}

extern "C" int n2v_edge_bias(const int64_t *rowptr, const int32_t *ids, const float *w,
                             const double *w64, const int32_t *src_id,
                             const int64_t *src_rowptr, const int32_t *src_nbs, int64_t n_rows,
                             int64_t nnz, double return_param, double inout_param,
                             double *w_out, void *stream) {
  if (n_rows < 0 || nnz < 0) return N2V_EINVAL;
  // generate_edge_alias_tables raises ValueError on p == 0 or q == 0 (randomwalk.py:214-217)
  if (return_param == 0.0 || inout_param == 0.0) return N2V_EINVAL;
  if (n_rows == 0 || nnz == 0) return N2V_OK;
  if (!rowptr || !ids || !w_out || (w && w64)) return N2V_EINVAL;
  if (src_id && (!src_rowptr || !src_nbs)) return N2V_EINVAL;
  int64_t blocks = (nnz + 255) / 256;
  const int64_t cap = n2v::resident_blocks((const void *)n2v::edge_bias_kernel, 256, 0) * 2;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(n2v::edge_bias_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, rowptr, ids, w, w64, src_id, src_rowptr, src_nbs, n_rows,
                     nnz, return_param, inout_param, w_out);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}

extern "C" int n2v_alias_draw(const int64_t *rowptr, const n2v_slot *slots, int64_t n_rows,
                              const double *r1, const double *r2, int32_t *vertex_out,
                              uint32_t *status, void *stream) {
  if (n_rows < 0) return N2V_EINVAL;
  if (n_rows == 0) return N2V_OK;
  if (!rowptr || !slots || !r1 || !vertex_out || !status) return N2V_EINVAL;
  int64_t blocks = (n_rows + 255) / 256;
  const int64_t cap = n2v::resident_blocks((const void *)n2v::alias_draw_kernel, 256, 0) * 2;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(n2v::alias_draw_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, rowptr, slots, n_rows, r1, r2, vertex_out, status);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}
