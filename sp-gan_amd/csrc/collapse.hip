// spgan_wt_diag_w: the two weight-only operands of the collapsed backward of the layer in front of the max-pool
// (Generation/Discriminator.py:77-81,104; DESIGN.md "collapsed backward of the 256->1024 layer") in ONE launch:
//   G[i, j]  = sum_c W[c, i] * alpha[c] * W[c, j]            W^T diag(alpha) W              [K, K]
//   cvec[j]  = sum_c (alpha[c]*b[c] + beta[c]) * W[c, j]     (alpha*b + beta) . W            [K]     (optional)
// for W [C, K] (C = 1024 output channels, K = 256 inputs).  Before: rowscale_outer + gemm_tn + splitk_reduce + a small gemm_nt -- four
// launches of ~5-10 us for 0.13 GFLOP, issued in every one of the step's four collapsed backward passes and twice in the double backward.
// One workgroup per 32 x 32 tile of G; its four waves split the C-reduction in contiguous quarters (v_mfma_f32_32x32x2_f32, operands
// straight from global memory / L2: W is 1 MB and every workgroup reads 64 of its columns), their partial tiles are summed in wave order
// through LDS -- deterministic.  One extra row of workgroups forms cvec (32 columns each).
#include "common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void wt_diag_w_kernel(const float* __restrict__ W, int ldw, int C, int K, const float* __restrict__ alpha,
                                                        const float* __restrict__ beta, const float* __restrict__ bias, float* __restrict__ G,
                                                        int ldg, float* __restrict__ cvec) {
  __shared__ float red[3][32 * 33];
  __shared__ float cred[4][2][32];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const int per = C / 4;                 // C % 256 == 0 (host): every wave reduces a multiple of 64 channels
  const int cb = w * per;
  if (blockIdx.y == gridDim.y - 1 && cvec != nullptr) {
    // the extra row of workgroups: cvec for columns j0 .. j0+31; lane (l31, lh) of wave w sums the channels cb + lh, cb + lh + 2, ...
    float cv = 0.f;
    const float* wb = W + (size_t)(cb + lh) * ldw + j0 + l31;
    for (int k0 = 0; k0 < per / 2; k0 += 16) {
      float al[16], bi[16], be[16], b[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int c = cb + 2 * (k0 + u) + lh;
        al[u] = alpha[c]; bi[u] = bias[c]; be[u] = beta[c];
        b[u] = wb[(size_t)2 * (k0 + u) * ldw];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) cv = fmaf(fmaf(al[u], bi[u], be[u]), b[u], cv);
    }
    cred[w][lh][l31] = cv;
    __syncthreads();
    if (tid < 32) {
      float v = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) v += cred[u][0][tid] + cred[u][1][tid];
      cvec[j0 + tid] = v;
    }
    return;
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* wa = W + (size_t)(cb + lh) * ldw + i0 + l31;
  const float* wb = W + (size_t)(cb + lh) * ldw + j0 + l31;
  // 32 k-steps per round: their 96 loads (L2 hits: W is 1 MB and shared by all workgroups) are issued together -- the kernel is
  // latency-bound, not bandwidth- or MFMA-bound (per % 64 == 0, host)
  constexpr int RU = 32;
  for (int k0 = 0; k0 < per / 2; k0 += RU) {
    float al[RU], a[RU], b[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      al[u] = alpha[cb + 2 * (k0 + u) + lh];
      a[u] = wa[(size_t)2 * (k0 + u) * ldw];
      b[u] = wb[(size_t)2 * (k0 + u) * ldw];
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u] * al[u], b[u], acc, 0, 0, 0);
  }
  // partial tiles of waves 1..3 -> LDS; wave 0 adds them in wave order and stores
  if (w > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[w - 1][((r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + l31] = acc[r];
  }
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      float v = acc[r];
#pragma unroll
      for (int u = 0; u < 3; ++u) v += red[u][row * 33 + l31];
      G[(size_t)(i0 + row) * ldg + j0 + l31] = v;
    }
  }
}

}  // namespace

extern "C" int spgan_wt_diag_w(const float* W, int ldw, int C, int K, const float* alpha, const float* beta, const float* bias, float* G, int ldg,
                               float* cvec, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(W && alpha && G && C > 0 && K > 0 && C % 256 == 0 && K % 32 == 0 && ldw >= K && ldg >= K);
  SPGAN_CHECK_ARG(!cvec || (beta && bias));
  hipLaunchKernelGGL(wt_diag_w_kernel, dim3(K / 32, K / 32 + (cvec ? 1 : 0)), dim3(256), 0, (hipStream_t)s_, W, ldw, C, K, alpha, beta, bias, G, ldg, cvec);
  return spgan_launch_status();
}
