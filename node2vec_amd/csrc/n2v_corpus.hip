// n2v_corpus.hip -- the two elementwise passes between K2 and K3 when the corpus is streamed
// (node2vec_amd/pipeline.py): counting the tokens of a batch of walks into the vocabulary's
// count vector (what gensim's build_vocab does over the sentences, embedding.py:126) and mapping
// vertex ids to vocabulary indices.  At BASELINE cfg 4 a batch is 8.5 x 10^8 tokens; done with
// framework ops (widen to int64, clamp, index_add, gather, where) the passes cost more than the
// walk kernel itself.
#include "n2v_common.h"

namespace n2v {

__global__ __launch_bounds__(256) void corpus_count_kernel(const int32_t *__restrict__ walks,
                                                          const uint8_t *__restrict__ valid,
                                                          int64_t n_rows, int32_t len,
                                                          int64_t n_vertices,
                                                          unsigned long long *__restrict__ counts) {
  const int64_t total = n_rows * (int64_t)len;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int32_t tok = walks[t];
    if (tok < 0 || (int64_t)tok >= n_vertices) continue;
    // (a batch is < 2^31 tokens: 32-bit division; a walker that vanished at a sink emits no row)
    const int64_t row = total < (1ll << 31) ? (int64_t)((uint32_t)t / (uint32_t)len) : t / len;
    if (valid && !valid[row]) continue;
    atomicAdd(counts + tok, 1ull);
  }
}

__global__ __launch_bounds__(256) void corpus_index_kernel(const int32_t *__restrict__ walks,
                                                          const uint8_t *__restrict__ valid,
                                                          const int32_t *__restrict__ index_of,
                                                          int64_t n_rows, int32_t len,
                                                          int64_t n_vertices,
                                                          int32_t *__restrict__ idx_out) {
  const int64_t total = n_rows * (int64_t)len;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int32_t tok = walks[t];
    int32_t out = -1;
    const int64_t row = total < (1ll << 31) ? (int64_t)((uint32_t)t / (uint32_t)len) : t / len;
    if (tok >= 0 && (int64_t)tok < n_vertices && (!valid || valid[row])) out = index_of[tok];
    idx_out[t] = out;
  }
}

}  // namespace n2v

extern "C" int n2v_corpus_count(const int32_t *walks, const uint8_t *valid, int64_t n_rows,
                                int32_t len, int64_t n_vertices, unsigned long long *counts,
                                void *stream) {
  if (n_rows < 0 || len < 1 || n_vertices < 0 || (n_rows > 0 && (!walks || !counts))) return N2V_EINVAL;
  if (n_rows == 0) return N2V_OK;
  const int64_t total = n_rows * (int64_t)len;
  int64_t blocks = (total + 255) / 256;
  const int64_t cap = n2v::resident_blocks((const void *)n2v::corpus_count_kernel, 256, 0) * 4;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(n2v::corpus_count_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, walks, valid, n_rows, len, n_vertices, counts);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}

extern "C" int n2v_corpus_index(const int32_t *walks, const uint8_t *valid, const int32_t *index_of,
                                int64_t n_rows, int32_t len, int64_t n_vertices, int32_t *idx_out,
                                void *stream) {
  if (n_rows < 0 || len < 1 || n_vertices < 0 || (n_rows > 0 && (!walks || !index_of || !idx_out)))
    return N2V_EINVAL;
  if (n_rows == 0) return N2V_OK;
  const int64_t total = n_rows * (int64_t)len;
  int64_t blocks = (total + 255) / 256;
  const int64_t cap = n2v::resident_blocks((const void *)n2v::corpus_index_kernel, 256, 0) * 4;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(n2v::corpus_index_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, walks, valid, index_of, n_rows, len, n_vertices, idx_out);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}
