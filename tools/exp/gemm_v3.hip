// EXPERIMENT: macro-tile / wave-tile geometry sweep for the plain NT product (32x32x2 fp32 MFMA, LDS double buffer).
// Assumes M % BM == 0, N % BN == 0, K % BK == 0, 16-byte aligned rows.
#include "common.hpp"
typedef float f32x16 __attribute__((ext_vector_type(16)));
namespace {

template <int THREADS, int WGM, int WGN, int TI, int TJ, int BK, int WPS, int PRO, int EPI>
__global__ __launch_bounds__(THREADS, WPS) void gemm_v3_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, float* __restrict__ Y,
                                                              int ldy, int M, int N, int K, const float* __restrict__ bias, const float* __restrict__ psc_,
                                                              const float* __restrict__ psh_, float* __restrict__ stats, float* __restrict__ pool_val,
                                                              int* __restrict__ pool_arg) {
  constexpr int BM = WGM * TI * 32, BN = WGN * TJ * 32, LDT = BK + 2;
  constexpr int F4R = BK / 4;                 // float4 per tile row
  constexpr int RPP = THREADS / F4R;          // rows staged per pass
  constexpr int AS = BM / RPP, BS = BN / RPP;  // staging slots per thread
  static_assert(WGM * WGN * 64 == THREADS, "waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                  // [2][BM*LDT]
  float* Bs = smem + 2 * BM * LDT;   // [2][BN*LDT]
  const int tilesN = N / BN, tilesM = M / BM;
  const int id = blockIdx.x, xcd = id & 7, t = id >> 3;
  const int tn = t % tilesN, tm = xcd + 8 * (t / tilesN);
  if (tm >= tilesM) return;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WGN, wn = wave % WGN;
  const int l31 = lane & 31, lh = lane >> 5;
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 ra[AS], rb[BS];
  float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = make_float4(0.f, 0.f, 0.f, 0.f);
  const int lrow = tid / F4R, lc4 = (tid % F4R) * 4;
  const float* pa = A + (size_t)(m0 + lrow) * lda + lc4;
  const float* pw = W + (size_t)(n0 + lrow) * ldw + lc4;
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < AS; ++i) ra[i] = *reinterpret_cast<const float4*>(pa + (size_t)(RPP * i) * lda + k0);
#pragma unroll
    for (int i = 0; i < BS; ++i) rb[i] = *reinterpret_cast<const float4*>(pw + (size_t)(RPP * i) * ldw + k0);
    if (PRO) {
      psc = *reinterpret_cast<const float4*>(psc_ + k0 + lc4);
      psh = *reinterpret_cast<const float4*>(psh_ + k0 + lc4);
    }
  };
  auto sstore = [&](int buf) {
    float* a = As + buf * BM * LDT; float* b = Bs + buf * BN * LDT;
    if (PRO) {
#pragma unroll
      for (int i = 0; i < AS; ++i) {
        ra[i].x = lrelu_f(fmaf(ra[i].x, psc.x, psh.x), 0.01f);
        ra[i].y = lrelu_f(fmaf(ra[i].y, psc.y, psh.y), 0.01f);
        ra[i].z = lrelu_f(fmaf(ra[i].z, psc.z, psh.z), 0.01f);
        ra[i].w = lrelu_f(fmaf(ra[i].w, psc.w, psh.w), 0.01f);
      }
    }
#pragma unroll
    for (int i = 0; i < AS; ++i) {
      float* q = &a[(lrow + RPP * i) * LDT + lc4];
      *reinterpret_cast<float2*>(q) = make_float2(ra[i].x, ra[i].y);
      *reinterpret_cast<float2*>(q + 2) = make_float2(ra[i].z, ra[i].w);
    }
#pragma unroll
    for (int i = 0; i < BS; ++i) {
      float* q = &b[(lrow + RPP * i) * LDT + lc4];
      *reinterpret_cast<float2*>(q) = make_float2(rb[i].x, rb[i].y);
      *reinterpret_cast<float2*>(q + 2) = make_float2(rb[i].z, rb[i].w);
    }
  };
  auto compute = [&](int buf, int kk0, int kk1) {
    const float* a = As + buf * BM * LDT + (wm * TI * 32 + l31) * LDT + 2 * lh;
    const float* b = Bs + buf * BN * LDT + (wn * TJ * 32 + l31) * LDT + 2 * lh;
#pragma unroll
    for (int kk = kk0; kk < kk1; ++kk) {
      float2 af[TI], bf[TJ];
#pragma unroll
      for (int i = 0; i < TI; ++i) af[i] = *reinterpret_cast<const float2*>(a + i * 32 * LDT + kk * 4);
#pragma unroll
      for (int j = 0; j < TJ; ++j) bf[j] = *reinterpret_cast<const float2*>(b + j * 32 * LDT + kk * 4);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
    }
  };
  constexpr int KK = BK / 4;
  const int nk = K / BK;
  gload(0); sstore(0); __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload((kt + 1) * BK);
    compute(kt & 1, 0, KK / 2);
    if (kt + 1 < nk) sstore((kt + 1) & 1);
    compute(kt & 1, KK / 2, KK);
    __syncthreads();
  }
  if (EPI == 0) {
  float* yb = Y + (size_t)(m0 + wm * TI * 32 + 4 * lh) * ldy + n0 + wn * TJ * 32 + l31;
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const float bb = bias ? bias[n0 + wn * TJ * 32 + j * 32 + l31] : 0.f;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) yb[(size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldy + j * 32] = acc[i][j][r] + bb;
  }
  } else {
    // statistics + pooling records per 128-row group (= one M-wave when TI == 4, two when TI == 2), nothing stored
    constexpr float NL = (float)(16 * TI);
    constexpr int RG = TI * 32;               // rows of one wave
    float* xs = As;  // [6][WGM][BN]
    const int rbase = m0 + wm * RG + 4 * lh;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const float bb = bias ? bias[n0 + wn * TJ * 32 + j * 32 + l31] : 0.f;
      float sl = 0.f;
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][j][r] += bb; sl += acc[i][j][r]; }
      const float mean = sl * (1.f / NL);
      float m2 = 0.f, vx = -INFINITY, vn = INFINITY;
      int ax = 0x7fffffff, an = 0x7fffffff;
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[i][j][r];
          const float d = v - mean;
          m2 = fmaf(d, d, m2);
          const int row = rbase + i * 32 + (r & 3) + 8 * (r >> 2);
          if (v > vx) { vx = v; ax = row; }
          if (v < vn) { vn = v; an = row; }
        }
      const float so = __shfl_xor(sl, 32), m2o = __shfl_xor(m2, 32);
      const float dl = (so - sl) * (1.f / NL);
      const float S = sl + so, M2 = (m2 + m2o) + dl * dl * (0.5f * NL);
      const float ovx = __shfl_xor(vx, 32), ovn = __shfl_xor(vn, 32);
      const int oax = __shfl_xor(ax, 32), oan = __shfl_xor(an, 32);
      if (ovx > vx || (ovx == vx && oax < ax)) { vx = ovx; ax = oax; }
      if (ovn < vn || (ovn == vn && oan < an)) { vn = ovn; an = oan; }
      if (lh == 0) {
        const int c = wm * BN + (wn * TJ + j) * 32 + l31;
        xs[c] = S; xs[WGM * BN + c] = M2; xs[2 * WGM * BN + c] = vx; xs[3 * WGM * BN + c] = __int_as_float(ax);
        xs[4 * WGM * BN + c] = vn; xs[5 * WGM * BN + c] = __int_as_float(an);
      }
    }
    __syncthreads();
    constexpr int WPG = 128 / RG;      // waves per 128-row record group (TI=2: 2, TI=4: 1)
    if ((wm % WPG) == 0 && lh == 0) {
      const int rec = (m0 + wm * RG) / 128;
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int c0 = (wn * TJ + j) * 32 + l31, col = n0 + c0;
        int c = wm * BN + c0;
        float S = xs[c], M2 = xs[WGM * BN + c], n = 2.f * NL;
        float vx = xs[2 * WGM * BN + c], vn = xs[4 * WGM * BN + c];
        int ax = __float_as_int(xs[3 * WGM * BN + c]), an = __float_as_int(xs[5 * WGM * BN + c]);
#pragma unroll
        for (int w = 1; w < WPG; ++w) {
          c += BN;
          const float Sb = xs[c], nb = 2.f * NL;
          const float dl = Sb / nb - S / n;
          M2 = (M2 + xs[WGM * BN + c]) + dl * dl * (n * nb / (n + nb));
          S += Sb; n += nb;
          if (xs[2 * WGM * BN + c] > vx) { vx = xs[2 * WGM * BN + c]; ax = __float_as_int(xs[3 * WGM * BN + c]); }
          if (xs[4 * WGM * BN + c] < vn) { vn = xs[4 * WGM * BN + c]; an = __float_as_int(xs[5 * WGM * BN + c]); }
        }
        const size_t o = ((size_t)rec * N + col) * 2;
        stats[o] = S; stats[o + 1] = M2;
        pool_val[o] = vx; pool_val[o + 1] = vn;
        pool_arg[o] = ax; pool_arg[o + 1] = an;
      }
    }
  }
}

struct Ex { const float* psc; const float* psh; float* stats; float* pool_val; int* pool_arg; };
template <int THREADS, int WGM, int WGN, int TI, int TJ, int BK, int WPS, int PRO, int EPI>
int launch2(const float* A, int lda, const float* W, int ldw, float* Y, int ldy, int M, int N, int K, const float* bias, hipStream_t s, const Ex& e) {
  constexpr int BM = WGM * TI * 32, BN = WGN * TJ * 32, LDT = BK + 2;
  if (M % BM || N % BN || K % BK) return -1;
  const size_t sh = (size_t)2 * (BM + BN) * LDT * 4;
  auto kern = gemm_v3_kernel<THREADS, WGM, WGN, TI, TJ, BK, WPS, PRO, EPI>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
  const int tm8 = cdiv(M / BM, 8) * 8;
  hipLaunchKernelGGL(kern, dim3(tm8 * (N / BN)), dim3(THREADS), sh, s, A, lda, W, ldw, Y, ldy, M, N, K, bias, e.psc, e.psh, e.stats, e.pool_val, e.pool_arg);
  return (int)hipGetLastError();
}
template <int THREADS, int WGM, int WGN, int TI, int TJ, int BK, int WPS>
int launch(const float* A, int lda, const float* W, int ldw, float* Y, int ldy, int M, int N, int K, const float* bias, hipStream_t s, const Ex& e, int mode) {
  switch (mode) {
    case 0: return launch2<THREADS, WGM, WGN, TI, TJ, BK, WPS, 0, 0>(A, lda, W, ldw, Y, ldy, M, N, K, bias, s, e);
    case 1: return launch2<THREADS, WGM, WGN, TI, TJ, BK, WPS, 1, 0>(A, lda, W, ldw, Y, ldy, M, N, K, bias, s, e);
    case 2: return launch2<THREADS, WGM, WGN, TI, TJ, BK, WPS, 0, 1>(A, lda, W, ldw, Y, ldy, M, N, K, bias, s, e);
    case 3: return launch2<THREADS, WGM, WGN, TI, TJ, BK, WPS, 1, 1>(A, lda, W, ldw, Y, ldy, M, N, K, bias, s, e);
  }
  return -3;
}
}  // namespace

extern "C" int exp_gemm_v3(const float* A, int lda, const float* W, int ldw, float* Y, int ldy, int M, int N, int K, const float* bias, int variant, void* s_,
                           int mode, const float* psc, const float* psh, float* stats, float* pool_val, int* pool_arg) {
  hipStream_t s = (hipStream_t)s_;
  const Ex e{psc, psh, stats, pool_val, pool_arg};
  switch (variant) {
    case 0: return launch<256, 2, 2, 2, 2, 32, 3>(A, lda, W, ldw, Y, ldy, M, N, K, bias, s, e, mode);   // 128x128 (today's geometry)
    case 1: return launch<512, 4, 2, 2, 4, 32, 2>(A, lda, W, ldw, Y, ldy, M, N, K, bias, s, e, mode);   // 256x256, 8 waves of 64x128, BK 32
    case 2: return launch<512, 4, 2, 2, 4, 16, 2>(A, lda, W, ldw, Y, ldy, M, N, K, bias, s, e, mode);   // 256x256, 8 waves of 64x128, BK 16
    case 3: return launch<256, 2, 2, 2, 4, 16, 2>(A, lda, W, ldw, Y, ldy, M, N, K, bias, s, e, mode);   // 128x256, 4 waves of 64x128, BK 16, 2 WG/CU
    case 4: return launch<256, 2, 2, 4, 2, 16, 2>(A, lda, W, ldw, Y, ldy, M, N, K, bias, s, e, mode);   // 256x128, 4 waves of 128x64
    case 5: return launch<256, 2, 2, 4, 4, 16, 1>(A, lda, W, ldw, Y, ldy, M, N, K, bias, s, e, mode);   // 256x256, 4 waves of 128x128 (256 accumulators)
    case 6: return launch<256, 2, 2, 4, 4, 32, 1>(A, lda, W, ldw, Y, ldy, M, N, K, bias, s, e, mode);   // same, BK 32 (139 KB)
    case 7: return launch<512, 2, 4, 4, 2, 32, 2>(A, lda, W, ldw, Y, ldy, M, N, K, bias, s, e, mode);   // 256x256, 8 waves of 128x64
    case 8: return launch<256, 2, 2, 2, 4, 32, 1>(A, lda, W, ldw, Y, ldy, M, N, K, bias, s, e, mode);   // 128x256, BK 32 (104 KB), 1 WG/CU
    case 9: return launch<512, 4, 2, 2, 2, 32, 2>(A, lda, W, ldw, Y, ldy, M, N, K, bias, s, e, mode);   // 256x128, 8 waves of 64x64
  }
  return -2;
}
