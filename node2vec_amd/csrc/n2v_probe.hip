// n2v_probe.hip -- n2v_mem_probe: what the memory system of THIS device sustains for the access
// shapes of the two hot kernels, measured on the caller's own buffer (bench.py times it with HIP
// events and reports it beside the kernels' rates; nothing on the product path calls it).
//
//   mode 0  independent random 16-byte reads, 4 in flight per lane -- a hop-table gather (K2)
//   mode 1  one dependent chain of random 16-byte reads per lane     -- a walker (K2)
//   mode 2  random rows of `row_bytes` read by one wave, 8 bytes per lane at 512 bytes, four rows
//           in flight                                              -- a syn0 / syn1neg row (K3)
//   mode 3  the same rows read, changed and written back           -- a trained row (K3)
//   mode 4  one dependent chain of random 4-byte reads per lane with a binary search over an LDS
//           table of `row_bytes` degree classes between them        -- a walker on a graph whose
//           vertices are numbered by descending degree (entry = the id alone; row start and degree
//           follow from the id's class)
//
//   mode 5  the shape of a BIASED exact step (n2v_walk_wedge.hip): one dependent chain per lane of hop
//           entries (`row_bytes` = 16: today's table; 8: the {rank, classes} entry a degree-ranked
//           table would have), and with every step, for 49 % of the steps (cfg 4's share of edges
//           with shared neighbours), an independent 32-byte read of the wedge slot of the edge walked
//           last; buffer = [E hop entries | E slots]; 6 waves per SIMD like the kernel (LDS-limited)
//
// Addresses come from the counter-based mixer of the walk RNG: uniform over the buffer, so with
// a buffer much larger than the 256 MB Infinity Cache every access is a miss.
#include "n2v_common.h"

namespace n2v {

template <typename T>  // uint4 (16-byte), uint2 (8-byte) or uint32_t (4-byte) elements
__global__ __launch_bounds__(256) void probe_gather_kernel(const T *__restrict__ t, uint64_t n_el,
                                                           int iters, int dependent,
                                                           uint32_t *sink) {
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  auto first = [](const T &v) -> uint32_t {
    if constexpr (sizeof(T) == 4) return v; else return v.x;
  };
  auto last = [](const T &v) -> uint32_t {
    if constexpr (sizeof(T) == 4) return v; else if constexpr (sizeof(T) == 8) return v.y; else return v.w;
  };
  if (dependent) {
    uint64_t idx = mix64(gid) % n_el;
    for (int k = 0; k < iters; ++k) {
      const T r = t[idx];
      acc += last(r);
      idx = mix64(((uint64_t)first(r) << 32 | last(r)) ^ (gid + (uint64_t)k * 0x9E3779B97F4A7C15ULL)) % n_el;
    }
  } else {
    for (int k = 0; k < iters; k += 4) {
      T v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = t[mix64(gid * 0x100000001B3ULL + (uint64_t)(k + u)) % n_el];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc += first(v[u]) + last(v[u]);
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;  // keeps the loads alive
}

// mode 4: what a step costs when the 4-byte entry names the next vertex and its row is found
// through the class table (first id of the class, row offset of that id, degree of the class)
__global__ __launch_bounds__(1024) void probe_class_chain_kernel(const uint32_t *__restrict__ t, uint64_t n_el,
                                                                 int classes, int iters, uint32_t *sink) {
  extern __shared__ uint32_t lds_probe[];
  uint32_t *first = lds_probe;                                          // [classes] ascending ids
  uint64_t *where = reinterpret_cast<uint64_t *>(lds_probe + classes);  // [classes] offset | degree << 40
  const uint64_t n_v = n_el / 8 > 2 ? n_el / 8 : 2;
  for (int c = threadIdx.x; c < classes; c += blockDim.x) {
    // class c holds the ids [n_v (c / classes)^3, ...): one vertex per class at the head, millions at
    // the tail, degrees falling from `classes` to 1
    const double f = (double)c / (double)classes;
    first[c] = (uint32_t)((double)n_v * f * f * f);
    where[c] = ((uint64_t)((double)n_el * f * f) & ((1ull << 40) - 1)) | (uint64_t)(classes - c) << 40;
  }
  __syncthreads();
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  uint64_t idx = mix64(gid) % n_el;
  for (int k = 0; k < iters; ++k) {
    const uint32_t r = t[idx];
    acc += r;
    const uint64_t bits = mix64((uint64_t)r ^ (gid + (uint64_t)k * 0x9E3779B97F4A7C15ULL));
    const uint32_t v = (uint32_t)(((bits >> 32) * n_v) >> 32);  // stands for the id the entry names
    int c = 0;
    for (int half = classes >> 1; half > 0; half >>= 1)  // largest c with first[c] <= v
      if (first[c + half] <= v) c += half;
    const uint64_t wv = where[c];
    const uint32_t deg = (uint32_t)(wv >> 40);
    const uint64_t row = (wv & ((1ull << 40) - 1)) + (uint64_t)(v - first[c]) * deg;
    idx = (row + (uint32_t)(((uint64_t)(uint32_t)bits * deg) >> 32)) % n_el;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// mode 5: hop chain + slot reads, the go / no-go of 8-byte hop entries for the biased kernels
template <typename T>
__global__ __launch_bounds__(256) void probe_biased_step_kernel(const T *__restrict__ hops, const uint4 *__restrict__ slots,
                                                                uint64_t n_el, int iters, uint32_t *sink) {
  extern __shared__ uint32_t lds_hold[];  // 24 KB per block: six blocks per CU, as the walk kernel has
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (iters < 0) lds_hold[threadIdx.x] = (uint32_t)gid;
  uint32_t acc = 0;
  uint64_t idx = mix64(gid) % n_el, prev = mix64(gid + 1) % n_el;
  for (int k = 0; k < iters; ++k) {
    const uint64_t bits = mix64(idx ^ (gid + (uint64_t)k * 0x9E3779B97F4A7C15ULL));
    const bool need_slot = (uint32_t)(bits & 0xff) < 125u;  // 49 %
    uint4 sa = make_uint4(0, 0, 0, 0), sb = make_uint4(0, 0, 0, 0);
    if (need_slot) {
      sa = slots[2 * prev];
      sb = slots[2 * prev + 1];
    }
    const T r = hops[idx];
    uint32_t lo, hi;
    if constexpr (sizeof(T) == 8) { lo = r.x; hi = r.y; } else { lo = r.x; hi = r.w; }
    acc += hi + sa.x + sb.w;
    prev = idx;
    idx = mix64(((uint64_t)lo << 32 | hi) ^ (uint64_t)sa.y ^ bits) % n_el;
  }
  if (acc == 0x12345678u) sink[0] = acc + lds_hold[0];
}

template <bool kWrite>
__global__ __launch_bounds__(256) void probe_rows_kernel(float *t, uint64_t n_rows, int row_floats,
                                                         int iters, uint32_t *sink) {
  const int lane = threadIdx.x & 63;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int per_lane = row_floats / 64;  // 2 at 512-byte rows (float2 per lane), 4 at 1024
  float acc = 0.0f;
  for (int k = 0; k < iters; k += 4) {  // four rows in flight per wave
    float *rp[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      rp[u] = t + (mix64(wave * 0x100000001B3ULL + (uint64_t)(k + u)) % n_rows) * (uint64_t)row_floats;
    if (per_lane == 2) {
      float2 a[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = reinterpret_cast<const float2 *>(rp[u])[lane];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc += a[u].x;
        if (kWrite) reinterpret_cast<float2 *>(rp[u])[lane] = make_float2(a[u].x + 1.0f, a[u].y);
      }
    } else {
      for (int q = 0; q < per_lane / 4; ++q) {
        float4 a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = reinterpret_cast<const float4 *>(rp[u])[lane * (per_lane / 4) + q];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc += a[u].x;
          if (kWrite) {
            a[u].x += 1.0f;
            reinterpret_cast<float4 *>(rp[u])[lane * (per_lane / 4) + q] = a[u];
          }
        }
      }
    }
  }
  if (acc == 1234.5f) sink[0] = 1;
}

}  // namespace n2v

extern "C" int n2v_mem_probe(void *buffer, int64_t buffer_bytes, int32_t mode, int32_t iters,
                             int32_t row_bytes, int64_t *accesses_host, uint32_t *sink,
                             void *stream) {
  if (!buffer || !sink || buffer_bytes < 4096 || iters < 4 || (iters & 3) || mode < 0 || mode > 5)
    return N2V_EINVAL;
  if ((reinterpret_cast<uintptr_t>(buffer) & 15u) != 0) return N2V_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int threads = 256;
  if (mode == 5) {
    if (row_bytes != 8 && row_bytes != 16) return N2V_EINVAL;
    const uint64_t n_el = (uint64_t)(buffer_bytes / (row_bytes + 32)) & ~(uint64_t)1;  // (slots 32-byte aligned)
    if (n_el < 2) return N2V_EINVAL;
    const size_t lds = 24 * 1024;
    const void *fn = row_bytes == 8 ? (const void *)n2v::probe_biased_step_kernel<uint2>
                                    : (const void *)n2v::probe_biased_step_kernel<uint4>;
    const int64_t blocks = n2v::resident_blocks(fn, threads, lds);
    if (accesses_host) *accesses_host = blocks * threads * (int64_t)iters;
    const uint4 *slots = reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(buffer) + n_el * row_bytes);
    if (row_bytes == 8)
      hipLaunchKernelGGL(n2v::probe_biased_step_kernel<uint2>, dim3((unsigned)blocks), dim3(threads), lds, st,
                         (const uint2 *)buffer, slots, n_el, iters, sink);
    else
      hipLaunchKernelGGL(n2v::probe_biased_step_kernel<uint4>, dim3((unsigned)blocks), dim3(threads), lds, st,
                         (const uint4 *)buffer, slots, n_el, iters, sink);
    N2V_HIP_CHECK(hipGetLastError());
    return N2V_OK;
  }
  if (mode == 4) {
    const int classes = row_bytes;  // a power of two
    if (classes < 64 || classes > 8192 || (classes & (classes - 1))) return N2V_EINVAL;
    const size_t lds = (size_t)classes * 12;
    const int64_t blocks = n2v::resident_blocks((const void *)n2v::probe_class_chain_kernel, 1024, lds);
    if (accesses_host) *accesses_host = blocks * 1024 * (int64_t)iters;
    hipLaunchKernelGGL(n2v::probe_class_chain_kernel, dim3((unsigned)blocks), dim3(1024), lds, st,
                       (const uint32_t *)buffer, (uint64_t)(buffer_bytes / 4), classes, iters, sink);
    N2V_HIP_CHECK(hipGetLastError());
    return N2V_OK;
  }
  if (mode <= 1) {
    // row_bytes selects the element width of the gathers here: 0 / 16 = 16-byte, 8, 4
    const int width = row_bytes == 0 ? 16 : row_bytes;
    if (width != 16 && width != 8 && width != 4) return N2V_EINVAL;
    const void *fn = width == 16  ? (const void *)n2v::probe_gather_kernel<uint4>
                     : width == 8 ? (const void *)n2v::probe_gather_kernel<uint2>
                                  : (const void *)n2v::probe_gather_kernel<uint32_t>;
    const int64_t blocks = n2v::resident_blocks(fn, threads, 0);
    if (accesses_host) *accesses_host = blocks * threads * (int64_t)iters;
    const uint64_t n_el = (uint64_t)(buffer_bytes / width);
    if (width == 16)
      hipLaunchKernelGGL(n2v::probe_gather_kernel<uint4>, dim3((unsigned)blocks), dim3(threads), 0, st,
                         (const uint4 *)buffer, n_el, iters, mode, sink);
    else if (width == 8)
      hipLaunchKernelGGL(n2v::probe_gather_kernel<uint2>, dim3((unsigned)blocks), dim3(threads), 0, st,
                         (const uint2 *)buffer, n_el, iters, mode, sink);
    else
      hipLaunchKernelGGL(n2v::probe_gather_kernel<uint32_t>, dim3((unsigned)blocks), dim3(threads), 0,
                         st, (const uint32_t *)buffer, n_el, iters, mode, sink);
  } else {
    if (row_bytes != 512 && row_bytes != 1024 && row_bytes != 2048) return N2V_EINVAL;
    const uint64_t n_rows = (uint64_t)(buffer_bytes / row_bytes);
    if (n_rows < 2) return N2V_EINVAL;
    const void *fn = mode == 2 ? (const void *)n2v::probe_rows_kernel<false>
                               : (const void *)n2v::probe_rows_kernel<true>;
    const int64_t blocks = n2v::resident_blocks(fn, threads, 0);
    if (accesses_host) *accesses_host = blocks * (threads / 64) * (int64_t)iters;
    if (mode == 2)
      hipLaunchKernelGGL(n2v::probe_rows_kernel<false>, dim3((unsigned)blocks), dim3(threads), 0, st,
                         (float *)buffer, n_rows, row_bytes / 4, iters, sink);
    else
      hipLaunchKernelGGL(n2v::probe_rows_kernel<true>, dim3((unsigned)blocks), dim3(threads), 0, st,
                         (float *)buffer, n_rows, row_bytes / 4, iters, sink);
  }
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}
