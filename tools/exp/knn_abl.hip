// EXPERIMENT: where does knn_f32 spend its time?  ABL 0 full, 1 no insertion, 2 no dot product
#include "common.hpp"
namespace {
template <int KP, int CP, int ABL>
__global__ __launch_bounds__(256) void knn_k(const float* __restrict__ x, int N, int C, int k, int32_t* __restrict__ idx) {
  constexpr int TC = 64;
  __shared__ __attribute__((aligned(16))) float cand[TC * CP];
  __shared__ float cnorm[TC];
  const int b = blockIdx.y;
  const int q = blockIdx.x * 256 + threadIdx.x;
  const float* xb = x + (size_t)b * N * C;
  const bool qok = q < N;
  float xq[CP];
#pragma unroll
  for (int c = 0; c < CP; ++c) xq[c] = (qok && c < C) ? xb[(size_t)q * C + c] : 0.f;
  float qn = 0.f;
#pragma unroll
  for (int c = 0; c < CP; ++c) qn = fmaf(xq[c], xq[c], qn);
  float bd[KP]; int bi[KP];
#pragma unroll
  for (int t = 0; t < KP; ++t) { bd[t] = INFINITY; bi[t] = 0x7fffffff; }
  for (int c0 = 0; c0 < N; c0 += TC) {
    const int nc = min(TC, N - c0);
    __syncthreads();
    for (int e = threadIdx.x; e < TC * CP; e += 256) { const int j = e / CP, c = e % CP; cand[e] = (j < nc && c < C) ? xb[(size_t)(c0 + j) * C + c] : 0.f; }
    __syncthreads();
    if (threadIdx.x < TC) { float sq = 0.f;
#pragma unroll
      for (int c = 0; c < CP; ++c) sq = fmaf(cand[threadIdx.x * CP + c], cand[threadIdx.x * CP + c], sq);
      cnorm[threadIdx.x] = sq; }
    __syncthreads();
    if (!qok) continue;
    for (int j = 0; j < nc; ++j) {
      float dot = 0.f;
      if (ABL != 2) {
        const float4* v = reinterpret_cast<const float4*>(cand + j * CP);
        float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
        for (int c = 0; c < CP / 4; ++c) { const float4 a = v[c]; d0 = fmaf(xq[4*c], a.x, d0); d1 = fmaf(xq[4*c+1], a.y, d1); d2 = fmaf(xq[4*c+2], a.z, d2); d3 = fmaf(xq[4*c+3], a.w, d3); }
        dot = (d0 + d1) + (d2 + d3);
      } else dot = xq[j & (CP - 1)];
      const float d = (-2.f * dot + qn) + cnorm[j];
      if (ABL == 1) { if (d < bd[0]) { bd[0] = d; bi[0] = c0 + j; } }
      else if (d < bd[KP - 1]) {
        bd[KP - 1] = d; bi[KP - 1] = c0 + j;
#pragma unroll
        for (int t = KP - 1; t > 0; --t) if (bd[t] < bd[t - 1]) { const float td = bd[t]; bd[t] = bd[t-1]; bd[t-1] = td; const int ti = bi[t]; bi[t] = bi[t-1]; bi[t-1] = ti; }
      }
    }
  }
  if (qok) { int32_t* o = idx + ((size_t)b * N + q) * k;
#pragma unroll
    for (int t = 1; t < KP; ++t) if (t <= k) o[t - 1] = b * N + bi[t]; }
}
}
extern "C" int exp_knn(const float* x, int B, int N, int C, int k, int abl, int32_t* idx, void* s_) {
  dim3 g(cdiv(N, 256), B), b(256); hipStream_t s = (hipStream_t)s_;
  if (abl == 0) hipLaunchKernelGGL((knn_k<11, 64, 0>), g, b, 0, s, x, N, C, k, idx);
  if (abl == 1) hipLaunchKernelGGL((knn_k<11, 64, 1>), g, b, 0, s, x, N, C, k, idx);
  if (abl == 2) hipLaunchKernelGGL((knn_k<11, 64, 2>), g, b, 0, s, x, N, C, k, idx);
  return (int)hipGetLastError();
}
