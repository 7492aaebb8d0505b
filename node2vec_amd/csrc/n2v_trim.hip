// n2v_trim.hip -- hotspot trimming, the device form of trim_hotspot_vertices
// (reference randomwalk.py:238-262): a source vertex with more than `cap` out-edges
// keeps a uniform sample WITHOUT replacement of exactly `cap` of them; weights are
// untouched (no renormalisation).  The reference draws with pandas
// DataFrame.sample (numpy RandomState), which cannot be replayed; this kernel uses
// selection sampling (Knuth's Algorithm S) on the build's counter-based stream:
// edge i of a row with d edges is kept with probability (cap - kept) / (d - i).
// One lane per row; only rows above the cap do any work (setup path, O(E_hub)).
#include "n2v_common.h"

namespace n2v {

__global__ void trim_mark_kernel(const int64_t *__restrict__ rowptr, int64_t n_rows,
                                 int64_t cap, uint64_t seed, uint8_t *__restrict__ keep) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n_rows) return;
  const int64_t b = rowptr[row], d = rowptr[row + 1] - b;
  if (d <= cap) return;  // keep[] was preset to 1 by the caller
  const uint64_t h = mix64(seed ^ mix64((uint64_t)row + 0x2545F4914F6CDD1DULL));
  int64_t need = cap;
  for (int64_t i = 0; i < d; ++i) {
    const uint64_t u = mix64(h + ((uint64_t)i + 1ULL) * 0x9FB21C651E98DF25ULL);
    // floor(u / 2^64 * (d - i)) is uniform on [0, d - i)
    const bool take = (int64_t)__umul64hi(u, (uint64_t)(d - i)) < need;
    keep[b + i] = take ? 1 : 0;
    need -= take ? 1 : 0;
  }
}

}  // namespace n2v

extern "C" int n2v_trim_mark(const int64_t *rowptr, int64_t n_rows, int64_t max_out_degree,
                             uint64_t seed, uint8_t *keep_out, void *stream) {
  if (!rowptr || !keep_out || n_rows < 0) return N2V_EINVAL;
  if (max_out_degree <= 0) max_out_degree = 100000;  // constants.py:6
  if (n_rows == 0) return N2V_OK;
  const int threads = 64;
  const int64_t blocks = (n_rows + threads - 1) / threads;
  hipLaunchKernelGGL(n2v::trim_mark_kernel, dim3((unsigned)blocks), dim3(threads), 0,
                     (hipStream_t)stream, rowptr, n_rows, max_out_degree, seed, keep_out);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}
