// gather_ceiling.hip -- what the chip sustains for the access shapes of the walk kernels:
// random 4-byte / 16-byte reads (independent, and as a dependent chain per lane like a walker),
// random 4-byte writes vs 64-byte aligned chunk writes (the path store), by table size.
// Diagnostic only (scripts/): build with hipcc --offload-arch=gfx950 -O3, run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__host__ __device__ inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

__global__ void fill(uint32_t *t, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    t[i] = (uint32_t)(mix64(i * 0x9E3779B97F4A7C15ULL) % n);
}

// independent random 4-byte reads, 8 in flight per lane
__global__ __launch_bounds__(256) void rd4_indep(const uint32_t *__restrict__ t, uint64_t n, int iters, uint32_t *out) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (int k = 0; k < iters; k += 8) {
    uint32_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = t[mix64(g * 0x100000001B3ULL + k + u) % n];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// dependent chain per lane: idx = t[idx] (one load in flight per lane, like a walker)
__global__ __launch_bounds__(256) void rd4_chain(const uint32_t *__restrict__ t, uint64_t n, int iters, uint32_t *out) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t idx = (uint32_t)(mix64(g) % n);
  for (int k = 0; k < iters; ++k) idx = t[idx];
  if (idx == 0xffffffffu) out[0] = idx;
}

// two dependent reads per step (rowptr-like 16-byte pair, then a 4-byte element), chain per lane
__global__ __launch_bounds__(256) void rd_walklike(const uint4 *__restrict__ a, const uint32_t *__restrict__ t, uint64_t n16, uint64_t n, int iters, uint32_t *out) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t idx = (uint32_t)(mix64(g) % n);
  for (int k = 0; k < iters; ++k) {
    const uint4 r = a[idx % n16];
    idx = t[(r.x ^ idx) % n];
  }
  if (idx == 0xffffffffu) out[0] = idx;
}

// independent random 16-byte reads
__global__ __launch_bounds__(256) void rd16_indep(const uint4 *__restrict__ t, uint64_t n16, int iters, uint32_t *out) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (int k = 0; k < iters; k += 4) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = t[mix64(g * 0x100000001B3ULL + k + u) % n16];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u].x + v[u].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// random 4-byte writes: lane writes word k of its own 324-byte row (the path store of a lane-per-walker kernel)
__global__ __launch_bounds__(256) void wr4_rows(uint32_t *t, uint64_t rows, int iters) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t row = mix64(g) % rows;
  for (int k = 0; k < iters; ++k) t[row * 81 + k] = (uint32_t)k;
}

// the same bytes, 16 words at a time (64-byte chunks aligned to 64 B)
__global__ __launch_bounds__(256) void wr64_rows(uint32_t *t, uint64_t rows, int iters) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t row = mix64(g) % rows;
  for (int k = 0; k + 16 <= iters; k += 16) {
    uint4 *p = reinterpret_cast<uint4 *>(t + row * 96 + k);  // 384-byte pitch: 64-byte aligned chunks
#pragma unroll
    for (int u = 0; u < 4; ++u) p[u] = make_uint4(k, k + 1, k + 2, k + 3);
  }
}

template <typename F>
static double timeit(F launch, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) launch();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e-3 / reps;
}

int main(int argc, char **argv) {
  const int blocks = 256 * 8, threads = 256, iters = 256;
  const uint64_t lanes = (uint64_t)blocks * threads;
  uint32_t *out;
  CK(hipMalloc(&out, 64));
  const double sizes_gb[] = {0.0625, 1.0, 4.0, 16.0};
  for (double gb : sizes_gb) {
    const uint64_t n = (uint64_t)(gb * 1024.0 * 1024.0 * 1024.0 / 4.0);
    uint32_t *t;
    CK(hipMalloc(&t, n * 4));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, t, n);
    CK(hipDeviceSynchronize());
    const double acc = (double)lanes * iters;
    double s;
    s = timeit([&] { hipLaunchKernelGGL(rd4_indep, dim3(blocks), dim3(threads), 0, 0, t, n, iters, out); }, 3);
    printf("table %7.3f GB  rd4 independent : %7.2f G reads/s\n", gb, acc / s / 1e9);
    s = timeit([&] { hipLaunchKernelGGL(rd4_chain, dim3(blocks), dim3(threads), 0, 0, t, n, iters, out); }, 3);
    printf("table %7.3f GB  rd4 chain/lane  : %7.2f G reads/s\n", gb, acc / s / 1e9);
    s = timeit([&] { hipLaunchKernelGGL(rd_walklike, dim3(blocks), dim3(threads), 0, 0, (const uint4 *)t, t, n / 4, n, iters, out); }, 3);
    printf("table %7.3f GB  walk-like (16B then 4B, dependent): %7.2f G steps/s\n", gb, acc / s / 1e9);
    s = timeit([&] { hipLaunchKernelGGL(rd16_indep, dim3(blocks), dim3(threads), 0, 0, (const uint4 *)t, n / 4, iters, out); }, 3);
    printf("table %7.3f GB  rd16 independent: %7.2f G reads/s\n", gb, acc / s / 1e9);
    const uint64_t rows = n / 96;
    s = timeit([&] { hipLaunchKernelGGL(wr4_rows, dim3(blocks), dim3(threads), 0, 0, t, rows, 80); }, 3);
    printf("table %7.3f GB  wr4 row-strided : %7.2f G words/s\n", gb, (double)lanes * 80 / s / 1e9);
    s = timeit([&] { hipLaunchKernelGGL(wr64_rows, dim3(blocks), dim3(threads), 0, 0, t, rows, 80); }, 3);
    printf("table %7.3f GB  wr64 chunks     : %7.2f G words/s\n", gb, (double)lanes * 80 / s / 1e9);
    fflush(stdout);
    CK(hipFree(t));
  }
  return 0;
}
