// Shared helpers for the gfx950 kernels of libspgan_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "spgan_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SPGAN_CHECK_ARG(cond) \
  do {                        \
    if (!(cond)) return SPGAN_EINVAL; \
  } while (0)

// Launch epilogue: report launch-time errors as a status, never synchronise.
static inline int spgan_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SPGAN_OK : (int)e;
}

__device__ __forceinline__ float lrelu_f(float v, float s) { return v > 0.f ? v : v * s; }
__device__ __forceinline__ float lrelu_mask(float z, float s) { return z > 0.f ? 1.f : s; }

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// wave64 all-lane sum
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
