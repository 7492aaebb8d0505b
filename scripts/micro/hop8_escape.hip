// hop8_escape.hip -- would an 8-byte hop entry + a cached second lookup beat the 16-byte entry?
// steps/s of a chain {8-byte gather over BIG_GB} + {a dependent 16-byte gather over a small table
// for X % of the steps}.  Diagnostic only (scripts/): hipcc --offload-arch=gfx950 -O3, run on the
// GPU box.  Result of round 3: profiles/r3v_probe_hop8*.log (48 G promised at a 1 MB side table;
// the real escape rows are hot lines shared by every CU and cost more: DESIGN.md section 5).
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

__global__ __launch_bounds__(256, 8) void chain(const uint2 *__restrict__ t, uint64_t n_el,
                                                const uint4 *__restrict__ small, uint64_t n_small,
                                                int iters, uint32_t escape_share, uint32_t *sink) {
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t idx = mix64(gid) % n_el;
  uint32_t acc = 0;
  for (int k = 0; k < iters; ++k) {
    const uint2 r = t[idx];
    uint64_t h = mix64(((uint64_t)r.x << 32 | r.y) ^ (gid + (uint64_t)k * 0x9E3779B97F4A7C15ULL));
    if ((uint32_t)h % 100u < escape_share) {
      const uint4 e = small[(h >> 32) % n_small];
      h ^= e.x + e.w;
    }
    acc += (uint32_t)h;
    idx = h % n_el;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char **argv) {
  const double big_gb = argc > 1 ? atof(argv[1]) : 6.0;
  const uint64_t big_bytes = (uint64_t)(big_gb * (1ull << 30)) / 8 * 8;
  void *big, *small;
  uint32_t *sink;
  CK(hipMalloc(&big, big_bytes));
  CK(hipMalloc(&small, 64ull << 20));
  CK(hipMalloc(&sink, 16));
  CK(hipMemset(big, 0, big_bytes));
  CK(hipMemset(small, 0, 64ull << 20));
  int cus = 256;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  const int blocks = cus * 8, iters = 256;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  const int small_mbs[] = {1, 2, 4}, shares[] = {0, 40, 53, 60};
  for (int small_mb : small_mbs)
    for (int share : shares) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(chain, dim3(blocks), dim3(256), 0, 0, (const uint2 *)big, big_bytes / 8,
                           (const uint4 *)small, ((uint64_t)small_mb << 20) / 16, iters, (uint32_t)share, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
      }
      printf("small table %3d MB, %3d %% of the steps look it up: %6.2f G steps/s\n", small_mb, share,
             (double)blocks * 256 * iters / (best * 1e-3) / 1e9);
    }
  return 0;
}
