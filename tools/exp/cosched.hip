// EXPERIMENT (round 5): do matrix-core-bound and HBM-bound workgroups of ONE launch run side by side on a CU?  Part A: workgroups that only issue
// fp32 MFMAs (register operands; `lds_a` bytes of LDS and ~128 VGPRs reserved like a GEMM tile kernel); part B: workgroups that stream a buffer
// (read + write, float4).  A's blocks come first in the grid (resident from the start), B's fill what is left.
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void mfma_part(float* out, int iters) {
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    a += 1e-6f;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__device__ __forceinline__ void stream_part(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4, int blk, int nblk) {
  for (size_t i = (size_t)blk * blockDim.x + threadIdx.x; i < n4; i += (size_t)nblk * blockDim.x) {
    float4 v = src[i];
    v.x += 1.f;
    dst[i] = v;
  }
}

__global__ __launch_bounds__(256, 2) void cosched_kernel(float* out, int iters, int a_blocks, const float4* src, float4* dst, size_t n4, int b_blocks) {
  extern __shared__ float lds[];
  if ((int)blockIdx.x < a_blocks) {
    if (threadIdx.x == 0) lds[0] = 1.f;      // keep the allocation
    mfma_part(out, iters);
  } else {
    stream_part(src, dst, n4, blockIdx.x - a_blocks, b_blocks);
  }
}

// stand-alone streaming kernel with its own (small) resource footprint
__global__ __launch_bounds__(256) void stream_kernel(const float4* src, float4* dst, size_t n4) { stream_part(src, dst, n4, blockIdx.x, gridDim.x); }

extern "C" int exp_cosched(float* out, int iters, int a_blocks, const float* src, float* dst, size_t n, int b_blocks, int lds_bytes, void* s) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cosched_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(cosched_kernel, dim3(a_blocks + b_blocks), dim3(256), lds_bytes, (hipStream_t)s, out, iters, a_blocks, (const float4*)src, (float4*)dst, n / 4,
                     b_blocks);
  return (int)hipGetLastError();
}
extern "C" int exp_stream(const float* src, float* dst, size_t n, int blocks, void* s) {
  hipLaunchKernelGGL(stream_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const float4*)src, (float4*)dst, n / 4);
  return (int)hipGetLastError();
}
