// EXPERIMENT: gemm_nt v2 -- 32x32x2 MFMA, LDS double buffer with one barrier per k-tile.
#include "common.hpp"
typedef float f32x16 __attribute__((ext_vector_type(16)));
namespace {
constexpr int BM = 128, BN = 128, BK = 32, LDT = BK + 2;   // 34: row*34 mod 64 distinct even banks for 32 rows

template <int DB, int WPS>
__global__ __launch_bounds__(256, WPS) void gemm_v2_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, float* __restrict__ Y, int ldy,
                                                      int M, int N, int K, const float* __restrict__ bias) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                       // [DB+1][BM*LDT]
  float* Bs = smem + (DB + 1) * BM * LDT;
  const int tilesN = (N + BN - 1) / BN, tilesM = (M + BM - 1) / BM;
  const int id = blockIdx.x, xcd = id & 7, t = id >> 3;
  const int tn = t % tilesN, tm = xcd + 8 * (t / tilesN);
  if (tm >= tilesM) return;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 ra[4], rb[4];
  const int lrow = tid >> 3, lc4 = (tid & 7) * 4;
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + lrow + 32 * i, k = k0 + lc4;
      ra[i] = (m < M && k < K) ? *reinterpret_cast<const float4*>(A + (size_t)m * lda + k) : make_float4(0, 0, 0, 0);
      const int n = n0 + lrow + 32 * i;
      rb[i] = (n < N && k < K) ? *reinterpret_cast<const float4*>(W + (size_t)n * ldw + k) : make_float4(0, 0, 0, 0);
    }
  };
  auto sstore = [&](int buf) {
    float* a = As + buf * BM * LDT; float* b = Bs + buf * BN * LDT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float* pa = &a[(lrow + 32 * i) * LDT + lc4];
      *reinterpret_cast<float2*>(pa) = make_float2(ra[i].x, ra[i].y);
      *reinterpret_cast<float2*>(pa + 2) = make_float2(ra[i].z, ra[i].w);
      float* pb = &b[(lrow + 32 * i) * LDT + lc4];
      *reinterpret_cast<float2*>(pb) = make_float2(rb[i].x, rb[i].y);
      *reinterpret_cast<float2*>(pb + 2) = make_float2(rb[i].z, rb[i].w);
    }
  };
  auto compute = [&](int buf) {
    const float* a = As + buf * BM * LDT; const float* b = Bs + buf * BN * LDT;
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      float2 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const float2*>(&a[(wm * 64 + i * 32 + l31) * LDT + kk * 4 + 2 * lh]);
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const float2*>(&b[(wn * 64 + j * 32 + l31) * LDT + kk * 4 + 2 * lh]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
    }
  };
  const int nk = (K + BK - 1) / BK;
  gload(0); sstore(0); __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload((kt + 1) * BK);
    if (DB) {
      compute(kt & 1);
      if (kt + 1 < nk) sstore((kt + 1) & 1);
      __syncthreads();
    } else {
      compute(0);
      __syncthreads();
      if (kt + 1 < nk) { sstore(0); __syncthreads(); }
    }
  }
  // C/D of 32x32x2: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l31;
      const float bb = (col < N && bias) ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < M && col < N) Y[(size_t)row * ldy + col] = acc[i][j][r] + bb;
      }
    }
}
}  // namespace
extern "C" int exp_gemm_v2(const float* A, int lda, const float* W, int ldw, float* Y, int ldy, int M, int N, int K, const float* bias, int variant, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  const int tm8 = cdiv(cdiv(M, BM), 8) * 8;
  dim3 g(tm8 * cdiv(N, BN)), b(256);
  const size_t sh1 = (size_t)(BM + BN) * LDT * 4, sh2 = 2 * sh1;
  static bool init = false;
  if (!init) {
    hipFuncSetAttribute((const void*)gemm_v2_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh2);
    hipFuncSetAttribute((const void*)gemm_v2_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh2);
    init = true;
  }
  switch (variant) {
    case 0: hipLaunchKernelGGL((gemm_v2_kernel<0, 2>), g, b, sh1, s, A, lda, W, ldw, Y, ldy, M, N, K, bias); break;
    case 1: hipLaunchKernelGGL((gemm_v2_kernel<1, 2>), g, b, sh2, s, A, lda, W, ldw, Y, ldy, M, N, K, bias); break;
    case 2: hipLaunchKernelGGL((gemm_v2_kernel<0, 3>), g, b, sh1, s, A, lda, W, ldw, Y, ldy, M, N, K, bias); break;
    case 3: hipLaunchKernelGGL((gemm_v2_kernel<1, 1>), g, b, sh2, s, A, lda, W, ldw, Y, ldy, M, N, K, bias); break;
  }
  return (int)hipGetLastError();
}
