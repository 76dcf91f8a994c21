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

// More than 64 KB of dynamic LDS must be opted into once per kernel AND per device (hipFuncSetAttribute acts on the current device's
// copy of the function).  One instance per launch site (`static LdsOptIn opt;`): a bit per device ordinal, set with a relaxed atomic --
// the attribute call is idempotent, so two host threads racing on the first launch only repeat it.  A process that drives several GPUs
// (one host thread per device, or hipSetDevice between launches) gets the attribute on each of them.
struct LdsOptIn {
  unsigned long long done = 0;      // devices 0..63; any ordinal beyond that sets the attribute on every launch
  void ensure(const void* kernel, int bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = (dev >= 0 && dev < 64) ? (1ull << dev) : 0ull;
    if (bit && (__atomic_load_n(&done, __ATOMIC_RELAXED) & bit)) return;
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (bit) __atomic_fetch_or(&done, bit, __ATOMIC_RELAXED);
  }
};

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
