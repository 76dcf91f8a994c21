// Self-attention over the points of one shape (`Attention`, Generation/modules.py:534-558; Generator.py:116-117,191-192,
// the --attn variant): beta = softmax_j(theta_i . phi_j), o_i = sum_j beta_ij g_j, y = gamma * W_o o + x.
// The four projections and the three N x N contractions per shape run on the MFMA kernels of gemm.hip; this file holds the
// HBM-bound rest: the row softmax of the [N,N] score matrices (forward / backward, in place) and the gated residual.
#include "common.hpp"

namespace {

constexpr int SM_THREADS = 256;

__device__ __forceinline__ float block_max(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// One 256-thread workgroup per row; the row (<= 16 KB) is read from HBM once, the later passes hit L1/L2.
// VPT float4 per thread are kept in registers when the row fits (cols <= 256*4*VPT), otherwise the row is re-read.
template <int VPT>
__global__ __launch_bounds__(SM_THREADS) void softmax_rows_kernel(float* __restrict__ S, int cols) {
  __shared__ float red[4];
  float* row = S + (size_t)blockIdx.x * cols;
  const int nv = cols >> 2;
  f32x4 v[VPT];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = threadIdx.x + i * SM_THREADS;
    if (c < nv) {
      v[i] = *reinterpret_cast<const f32x4*>(row + 4 * c);
      m = fmaxf(fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)), m);
    }
  }
  m = block_max(m, red);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = threadIdx.x + i * SM_THREADS;
    if (c < nv) {
      v[i].x = expf(v[i].x - m); v[i].y = expf(v[i].y - m); v[i].z = expf(v[i].z - m); v[i].w = expf(v[i].w - m);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  s = block_sum(s, red);
  const float inv = 1.f / s;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = threadIdx.x + i * SM_THREADS;
    if (c < nv) {
      v[i].x *= inv; v[i].y *= inv; v[i].z *= inv; v[i].w *= inv;
      *reinterpret_cast<f32x4*>(row + 4 * c) = v[i];
    }
  }
}

// Any width (scalar, strided): three passes over the row.
__global__ __launch_bounds__(SM_THREADS) void softmax_rows_any_kernel(float* __restrict__ S, int cols) {
  __shared__ float red[4];
  float* row = S + (size_t)blockIdx.x * cols;
  float m = -INFINITY;
  for (int c = threadIdx.x; c < cols; c += SM_THREADS) m = fmaxf(m, row[c]);
  m = block_max(m, red);
  float s = 0.f;
  for (int c = threadIdx.x; c < cols; c += SM_THREADS) s += expf(row[c] - m);
  s = block_sum(s, red);
  const float inv = 1.f / s;
  for (int c = threadIdx.x; c < cols; c += SM_THREADS) row[c] = expf(row[c] - m) * inv;
}

// dS = P o (dP - sum_j dP_j P_j), in place on dP.
template <int VPT>
__global__ __launch_bounds__(SM_THREADS) void softmax_rows_bwd_kernel(const float* __restrict__ P, float* __restrict__ dP, int cols) {
  __shared__ float red[4];
  const float* prow = P + (size_t)blockIdx.x * cols;
  float* drow = dP + (size_t)blockIdx.x * cols;
  const int nv = cols >> 2;
  f32x4 p[VPT], d[VPT];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = threadIdx.x + i * SM_THREADS;
    if (c < nv) {
      p[i] = *reinterpret_cast<const f32x4*>(prow + 4 * c);
      d[i] = *reinterpret_cast<const f32x4*>(drow + 4 * c);
      s += (p[i].x * d[i].x + p[i].y * d[i].y) + (p[i].z * d[i].z + p[i].w * d[i].w);
    }
  }
  s = block_sum(s, red);
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = threadIdx.x + i * SM_THREADS;
    if (c < nv) {
      d[i].x = p[i].x * (d[i].x - s); d[i].y = p[i].y * (d[i].y - s); d[i].z = p[i].z * (d[i].z - s); d[i].w = p[i].w * (d[i].w - s);
      *reinterpret_cast<f32x4*>(drow + 4 * c) = d[i];
    }
  }
}

__global__ __launch_bounds__(SM_THREADS) void softmax_rows_bwd_any_kernel(const float* __restrict__ P, float* __restrict__ dP, int cols) {
  __shared__ float red[4];
  const float* prow = P + (size_t)blockIdx.x * cols;
  float* drow = dP + (size_t)blockIdx.x * cols;
  float s = 0.f;
  for (int c = threadIdx.x; c < cols; c += SM_THREADS) s += prow[c] * drow[c];
  s = block_sum(s, red);
  for (int c = threadIdx.x; c < cols; c += SM_THREADS) drow[c] = prow[c] * (drow[c] - s);
}

// y = gamma[0] * o + x   (gamma is a device scalar: the learnable gate, modules.py:546,558)
__global__ void scale_residual_kernel(const float* __restrict__ o, const float* __restrict__ x, const float* __restrict__ gamma,
                                      float* __restrict__ y, size_t n4) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n4) return;
  const float g = gamma[0];
  const f32x4 a = reinterpret_cast<const f32x4*>(o)[t], b = reinterpret_cast<const f32x4*>(x)[t];
  f32x4 r;
  r.x = g * a.x + b.x; r.y = g * a.y + b.y; r.z = g * a.z + b.z; r.w = g * a.w + b.w;
  reinterpret_cast<f32x4*>(y)[t] = r;
}

// d_o = gamma * dy;  partial[block] = sum dy*o over the block's elements (fixed order -> deterministic dgamma)
constexpr int SR_ITEMS = 8;     // float4 per thread
__global__ __launch_bounds__(SM_THREADS) void scale_residual_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ o,
                                                                         const float* __restrict__ gamma, float* __restrict__ d_o,
                                                                         float* __restrict__ partial, size_t n4) {
  __shared__ float red[4];
  const float g = gamma[0];
  float s = 0.f;
  const size_t base = (size_t)blockIdx.x * SM_THREADS * SR_ITEMS;
#pragma unroll
  for (int i = 0; i < SR_ITEMS; ++i) {
    const size_t t = base + (size_t)i * SM_THREADS + threadIdx.x;
    if (t < n4) {
      const f32x4 a = reinterpret_cast<const f32x4*>(dy)[t], b = reinterpret_cast<const f32x4*>(o)[t];
      s += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
      f32x4 r;
      r.x = g * a.x; r.y = g * a.y; r.z = g * a.z; r.w = g * a.w;
      reinterpret_cast<f32x4*>(d_o)[t] = r;
    }
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(SM_THREADS) void sum_partials_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
  __shared__ double redd[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += SM_THREADS) s += (double)partial[i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) redd[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)((redd[0] + redd[1]) + (redd[2] + redd[3]));
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int spgan_softmax_rows(float* S, long rows, int cols, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(S && rows >= 0 && cols > 0 && rows < (1L << 31));
  if (rows == 0) return SPGAN_OK;
  hipStream_t s = (hipStream_t)s_;
  const bool vec = (cols % 4 == 0) && al16(S);
  if (vec && cols <= SM_THREADS * 4 * 2) hipLaunchKernelGGL(softmax_rows_kernel<2>, dim3((unsigned)rows), dim3(SM_THREADS), 0, s, S, cols);
  else if (vec && cols <= SM_THREADS * 4 * 4) hipLaunchKernelGGL(softmax_rows_kernel<4>, dim3((unsigned)rows), dim3(SM_THREADS), 0, s, S, cols);
  else hipLaunchKernelGGL(softmax_rows_any_kernel, dim3((unsigned)rows), dim3(SM_THREADS), 0, s, S, cols);
  return spgan_launch_status();
}

extern "C" int spgan_softmax_rows_bwd(const float* P, float* dP, long rows, int cols, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(P && dP && rows >= 0 && cols > 0 && rows < (1L << 31));
  if (rows == 0) return SPGAN_OK;
  hipStream_t s = (hipStream_t)s_;
  const bool vec = (cols % 4 == 0) && al16(P) && al16(dP);
  if (vec && cols <= SM_THREADS * 4 * 2) hipLaunchKernelGGL(softmax_rows_bwd_kernel<2>, dim3((unsigned)rows), dim3(SM_THREADS), 0, s, P, dP, cols);
  else if (vec && cols <= SM_THREADS * 4 * 4) hipLaunchKernelGGL(softmax_rows_bwd_kernel<4>, dim3((unsigned)rows), dim3(SM_THREADS), 0, s, P, dP, cols);
  else hipLaunchKernelGGL(softmax_rows_bwd_any_kernel, dim3((unsigned)rows), dim3(SM_THREADS), 0, s, P, dP, cols);
  return spgan_launch_status();
}

extern "C" int spgan_scale_residual(const float* o, const float* x, const float* gamma, float* y, size_t n, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(o && x && gamma && y && n % 4 == 0 && al16(o) && al16(x) && al16(y));
  if (n == 0) return SPGAN_OK;
  const size_t n4 = n / 4;
  hipLaunchKernelGGL(scale_residual_kernel, dim3(cdiv((long)n4, 256)), dim3(256), 0, (hipStream_t)s_, o, x, gamma, y, n4);
  return spgan_launch_status();
}

extern "C" size_t spgan_scale_residual_bwd_ws_bytes(size_t n) {
  return (size_t)cdiv((long)(n / 4), SM_THREADS * SR_ITEMS) * sizeof(float);
}

extern "C" int spgan_scale_residual_bwd(const float* dy, const float* o, const float* gamma, float* d_o, float* dgamma, void* ws,
                                        size_t ws_bytes, size_t n, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(dy && o && gamma && d_o && dgamma && n % 4 == 0 && n > 0 && al16(dy) && al16(o) && al16(d_o));
  SPGAN_CHECK_ARG(ws && ws_bytes >= spgan_scale_residual_bwd_ws_bytes(n));
  const size_t n4 = n / 4;
  const int blocks = cdiv((long)n4, SM_THREADS * SR_ITEMS);
  hipStream_t s = (hipStream_t)s_;
  hipLaunchKernelGGL(scale_residual_bwd_kernel, dim3(blocks), dim3(SM_THREADS), 0, s, dy, o, gamma, d_o, (float*)ws, n4);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(SM_THREADS), 0, s, (const float*)ws, blocks, dgamma);
  return spgan_launch_status();
}
