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
#include "sparse_rows.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WDW_LDS_FLOATS = 3 * 32 * 33 + 4 * 2 * 32;   // partial tiles of waves 1..3 + the cvec partials

// workgroup (bx, by) of a (K/32) x (K/32 [+ 1 with cvec]) grid; lds: WDW_LDS_FLOATS floats
__device__ __forceinline__ void wt_diag_w_body(int bx, int by, int ny, float* lds, const float* __restrict__ W, int ldw, int C, int K,
                                               const float* __restrict__ alpha, const float* __restrict__ beta, const float* __restrict__ bias,
                                               float* __restrict__ G, int ldg, float* __restrict__ cvec) {
  float (*red)[32 * 33] = reinterpret_cast<float (*)[32 * 33]>(lds);
  float (*cred)[2][32] = reinterpret_cast<float (*)[2][32]>(lds + 3 * 32 * 33);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int i0 = by * 32, j0 = bx * 32;
  const int per = C / 4;                 // C % 256 == 0 (host): every wave reduces a multiple of 64 channels
  const int cb = w * per;
  if (by == ny - 1 && cvec != nullptr) {
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

__global__ __launch_bounds__(256) void wt_diag_w_kernel(const float* __restrict__ W, int ldw, int C, int K, const float* __restrict__ alpha,
                                                        const float* __restrict__ beta, const float* __restrict__ bias, float* __restrict__ G,
                                                        int ldg, float* __restrict__ cvec) {
  __shared__ float lds[WDW_LDS_FLOATS];
  wt_diag_w_body(blockIdx.x, blockIdx.y, gridDim.y, lds, W, ldw, C, K, alpha, beta, bias, G, ldg, cvec);
}

// spgan_collapse_prep: everything the collapsed backward needs before its big launch, in ONE launch -- one or two W^T diag(alpha) W problems
// (72 or 64 workgroups each, latency-bound on the cold weight matrix: 29 us alone) and the sparse-row product S.W (2048 workgroups that
// stream 67 MB out: 33 us alone).  The weight workgroups come first in the grid, so they are resident from the start and finish under the
// streaming ones.  Both parts run the device functions of their stand-alone kernels: bit-identical results.
__global__ __launch_bounds__(256) void collapse_prep_kernel(const spgan_collapse_prep_args a, int wgs_per_prob0, int wgs_per_prob1, int chunks, int RB) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  int id = blockIdx.x;
  const int kx = a.K / 32;
  if (id < wgs_per_prob0) {
    wt_diag_w_body(id % kx, id / kx, wgs_per_prob0 / kx, dyn, a.W, a.ldw, a.C, a.K, a.alpha[0], a.beta[0], a.bias[0], a.G[0], a.ldg, a.cvec[0]);
    return;
  }
  id -= wgs_per_prob0;
  if (id < wgs_per_prob1) {
    wt_diag_w_body(id % kx, id / kx, wgs_per_prob1 / kx, dyn, a.W, a.ldw, a.C, a.K, a.alpha[1], a.beta[1], a.bias[1], a.G[1], a.ldg, a.cvec[1]);
    return;
  }
  id -= wgs_per_prob1;
  sparse_rows_nt_body(id % chunks, id / chunks, reinterpret_cast<unsigned*>(dyn), a.sp_val, a.sp_arg, a.rows, a.C, a.W, a.ldw, a.K, a.E, a.lde, RB);
}

}  // namespace

extern "C" int spgan_collapse_prep(const spgan_collapse_prep_args* a, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(a && a->W && a->C > 0 && a->K > 0 && a->C % 256 == 0 && a->K % 32 == 0 && a->ldw >= a->K && a->ldg >= a->K && a->C <= 8192);
  SPGAN_CHECK_ARG(a->nprob >= 1 && a->nprob <= 2 && a->sp_val && a->sp_arg && a->E && a->B > 0 && a->rows > 0 && a->lde >= a->K);
  int wgs[2] = {0, 0};
  for (int p = 0; p < a->nprob; ++p) {
    SPGAN_CHECK_ARG(a->alpha[p] && a->G[p] && (!a->cvec[p] || (a->beta[p] && a->bias[p])));
    wgs[p] = (a->K / 32) * (a->K / 32 + (a->cvec[p] ? 1 : 0));
  }
  const int RB = sparse_rows_nt_rb(a->rows, a->C);
  const int chunks = cdiv(a->rows, RB);
  size_t lds = sparse_rows_nt_lds(RB, a->C);
  if (lds < WDW_LDS_FLOATS * sizeof(float)) lds = WDW_LDS_FLOATS * sizeof(float);
  hipLaunchKernelGGL(collapse_prep_kernel, dim3(wgs[0] + wgs[1] + chunks * a->B), dim3(256), lds, (hipStream_t)s_, *a, wgs[0], wgs[1], chunks, RB);
  return spgan_launch_status();
}

extern "C" int spgan_wt_diag_w(const float* W, int ldw, int C, int K, const float* alpha, const float* beta, const float* bias, float* G, int ldg,
                               float* cvec, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(W && alpha && G && C > 0 && K > 0 && C % 256 == 0 && K % 32 == 0 && ldw >= K && ldg >= K);
  SPGAN_CHECK_ARG(!cvec || (beta && bias));
  hipLaunchKernelGGL(wt_diag_w_kernel, dim3(K / 32, K / 32 + (cvec ? 1 : 0)), dim3(256), 0, (hipStream_t)s_, W, ldw, C, K, alpha, beta, bias, G, ldg, cvec);
  return spgan_launch_status();
}
