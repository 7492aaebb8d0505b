// n2v_sync.hip -- the two elementwise passes of the multi-GPU SGNS exchange step.
//
// The reference trains on one machine (gensim worker threads share syn0 / syn1neg,
// embedding.py:126).  Sharded over GPUs, every rank trains its own walks on a full replica
// and the replicas are averaged every few launches: between two
// collectives a rank only needs
//   pack   wire = what this rank contributes to the sum, `before` = a snapshot of the rows
//   apply  rows += mean - before   (updates made while the collective was in flight stay)
// `wire` is either the fp32 rows themselves (parameter averaging: no reference copy at all)
// or bf16(row - ref) against a bf16 reference every rank shares (half the bytes on xGMI;
// the mean of the replicas equals ref + mean(row - ref) for ANY shared ref, and the
// differences are small, so bf16 costs ~2^-9 of the update, not of the weight).
// Both kernels stream: 16 B per lane, grid-stride, HBM-bound.
#include <hip/hip_bf16.h>

#include "n2v_common.h"

namespace n2v {

__device__ __forceinline__ float bf16_bits_to_float(uint16_t b) {
  return __uint_as_float((uint32_t)b << 16);
}
__device__ __forceinline__ uint16_t float_to_bf16_bits(float x) {
  return __bfloat16_as_ushort(__float2bfloat16(x));  // round to nearest even, NaN stays NaN
}

template <bool kBf16>
__global__ __launch_bounds__(256) void delta_pack_kernel(const float *__restrict__ cur,
                                                        const uint16_t *__restrict__ ref,
                                                        int64_t n4, int64_t n,
                                                        float *__restrict__ before,
                                                        void *__restrict__ wire) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 c = reinterpret_cast<const float4 *>(cur)[i];
    if (before) reinterpret_cast<float4 *>(before)[i] = c;
    if constexpr (kBf16) {
      const ushort4 r = reinterpret_cast<const ushort4 *>(ref)[i];
      ushort4 o;
      o.x = float_to_bf16_bits(c.x - bf16_bits_to_float(r.x));
      o.y = float_to_bf16_bits(c.y - bf16_bits_to_float(r.y));
      o.z = float_to_bf16_bits(c.z - bf16_bits_to_float(r.z));
      o.w = float_to_bf16_bits(c.w - bf16_bits_to_float(r.w));
      reinterpret_cast<ushort4 *>(wire)[i] = o;
    } else {
      reinterpret_cast<float4 *>(wire)[i] = c;
    }
  }
  // tail (n not a multiple of 4)
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float c = cur[i];
    if (before) before[i] = c;
    if constexpr (kBf16)
      reinterpret_cast<uint16_t *>(wire)[i] = float_to_bf16_bits(c - bf16_bits_to_float(ref[i]));
    else
      reinterpret_cast<float *>(wire)[i] = c;
  }
}

template <bool kBf16>
__global__ __launch_bounds__(256) void delta_apply_kernel(float *__restrict__ cur,
                                                         uint16_t *__restrict__ ref,
                                                         const float *__restrict__ before,
                                                         const void *__restrict__ wire_sum,
                                                         float inv_world, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float mean;
    if constexpr (kBf16) {
      mean = bf16_bits_to_float(ref[i]) +
             bf16_bits_to_float(reinterpret_cast<const uint16_t *>(wire_sum)[i]) * inv_world;
      ref[i] = float_to_bf16_bits(mean);
    } else {
      mean = reinterpret_cast<const float *>(wire_sum)[i] * inv_world;
    }
    cur[i] = before ? cur[i] + (mean - before[i]) : mean;
  }
}

// out[i] = sum over ranks of parts[r * m + i], accumulated in fp32 in rank order
template <bool kBf16>
__global__ __launch_bounds__(256) void delta_reduce_kernel(const void *__restrict__ parts, int world,
                                                          int64_t m, void *__restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
    float acc;
    if constexpr (kBf16) {
      const uint16_t *p = reinterpret_cast<const uint16_t *>(parts);
      acc = bf16_bits_to_float(p[i]);
      for (int r = 1; r < world; ++r) acc = acc + bf16_bits_to_float(p[(int64_t)r * m + i]);
      reinterpret_cast<uint16_t *>(out)[i] = float_to_bf16_bits(acc);
    } else {
      const float *p = reinterpret_cast<const float *>(parts);
      acc = p[i];
      for (int r = 1; r < world; ++r) acc = acc + p[(int64_t)r * m + i];
      reinterpret_cast<float *>(out)[i] = acc;
    }
  }
}

__global__ __launch_bounds__(256) void ref_init_kernel(const float *__restrict__ cur, int64_t n,
                                                      uint16_t *__restrict__ ref) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    ref[i] = float_to_bf16_bits(cur[i]);
}

static int64_t stream_blocks(int64_t items, const void *kernel) {
  int64_t blocks = (items + 255) / 256;
  const int64_t cap = resident_blocks(kernel, 256, 0) * 4;
  if (blocks > cap) blocks = cap;
  return blocks < 1 ? 1 : blocks;
}

}  // namespace n2v

extern "C" int n2v_delta_ref_init(const float *cur, int64_t n, uint16_t *ref_bf16_out,
                                  void *stream) {
  if (n < 0 || (n > 0 && (!cur || !ref_bf16_out))) return N2V_EINVAL;
  if (n == 0) return N2V_OK;
  const int64_t blocks = n2v::stream_blocks(n, (const void *)n2v::ref_init_kernel);
  hipLaunchKernelGGL(n2v::ref_init_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, cur, n, ref_bf16_out);
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}

extern "C" int n2v_delta_pack(const float *cur, const uint16_t *ref_bf16, int64_t n,
                              float *before_out, void *wire_out, int32_t wire_dtype,
                              void *stream) {
  if (n < 0 || (wire_dtype != N2V_WIRE_F32 && wire_dtype != N2V_WIRE_BF16)) return N2V_EINVAL;
  if (n == 0) return N2V_OK;
  if (!cur || !wire_out || (wire_dtype == N2V_WIRE_BF16 && !ref_bf16)) return N2V_EINVAL;
  // 16-byte accesses need 16-byte aligned bases (8 for the bf16 side)
  const bool aligned = ((uintptr_t)cur % 16 == 0) && ((uintptr_t)before_out % 16 == 0) &&
                       ((uintptr_t)wire_out % 16 == 0) && ((uintptr_t)ref_bf16 % 8 == 0);
  const int64_t n4 = aligned ? n / 4 : 0;
  hipStream_t st = (hipStream_t)stream;
  if (wire_dtype == N2V_WIRE_BF16) {
    const int64_t blocks = n2v::stream_blocks(n4 + 1, (const void *)n2v::delta_pack_kernel<true>);
    hipLaunchKernelGGL(n2v::delta_pack_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, cur,
                       ref_bf16, n4, n, before_out, wire_out);
  } else {
    const int64_t blocks = n2v::stream_blocks(n4 + 1, (const void *)n2v::delta_pack_kernel<false>);
    hipLaunchKernelGGL(n2v::delta_pack_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, cur,
                       ref_bf16, n4, n, before_out, wire_out);
  }
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}

extern "C" int n2v_delta_apply(float *cur, uint16_t *ref_bf16, const float *before,
                               const void *wire_sum, int32_t wire_dtype, int32_t world,
                               int64_t n, void *stream) {
  if (n < 0 || world < 1 || (wire_dtype != N2V_WIRE_F32 && wire_dtype != N2V_WIRE_BF16))
    return N2V_EINVAL;
  if (n == 0) return N2V_OK;
  if (!cur || !wire_sum || (wire_dtype == N2V_WIRE_BF16 && !ref_bf16)) return N2V_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const float inv = 1.0f / (float)world;
  if (wire_dtype == N2V_WIRE_BF16) {
    const int64_t blocks = n2v::stream_blocks(n, (const void *)n2v::delta_apply_kernel<true>);
    hipLaunchKernelGGL(n2v::delta_apply_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, cur,
                       ref_bf16, before, wire_sum, inv, n);
  } else {
    const int64_t blocks = n2v::stream_blocks(n, (const void *)n2v::delta_apply_kernel<false>);
    hipLaunchKernelGGL(n2v::delta_apply_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st,
                       cur, ref_bf16, before, wire_sum, inv, n);
  }
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}

extern "C" int n2v_delta_reduce(const void *parts, int32_t wire_dtype, int32_t world, int64_t m,
                                void *out, void *stream) {
  if (m < 0 || world < 1 || (wire_dtype != N2V_WIRE_F32 && wire_dtype != N2V_WIRE_BF16))
    return N2V_EINVAL;
  if (m == 0) return N2V_OK;
  if (!parts || !out) return N2V_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (wire_dtype == N2V_WIRE_BF16) {
    const int64_t blocks = n2v::stream_blocks(m, (const void *)n2v::delta_reduce_kernel<true>);
    hipLaunchKernelGGL(n2v::delta_reduce_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st,
                       parts, (int)world, m, out);
  } else {
    const int64_t blocks = n2v::stream_blocks(m, (const void *)n2v::delta_reduce_kernel<false>);
    hipLaunchKernelGGL(n2v::delta_reduce_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st,
                       parts, (int)world, m, out);
  }
  N2V_HIP_CHECK(hipGetLastError());
  return N2V_OK;
}
