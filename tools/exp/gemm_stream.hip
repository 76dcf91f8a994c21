// EXPERIMENT (round 5): a streaming NT product for short K (K = 64 / 128): Y[M,N] = A[M,K] . W[N,K]^T + bias.
// The 128 x 128 / 256 x 256 tile kernels run these shapes (M = 65,536, N = 128..320, K = 64..128: a handful of k-tiles per output tile, one round of tiles)
// at 45-57 % of the fp32 matrix peak: load, MFMA and store phases of a workgroup's single tile barely overlap.  Here a workgroup OWNS a block
// of 128 output columns -- its W block [128, K] stays in LDS for the whole launch -- and walks a contiguous run of 64-row tiles of A:
// double-buffered A tiles, the next tile's global loads in flight under the current tile's MFMAs, the finished tile's stores issued while
// the next one is computed.  8 waves (2 x 4), one 32 x 32 MFMA tile per wave and M-tile.
#include "common.hpp"
typedef float f32x16 __attribute__((ext_vector_type(16)));
namespace {
#define ROFF(r) (((r) & 3) + 8 * ((r) >> 2))

template <int K, int DBL>
__global__ __launch_bounds__(512, 2) void gemm_stream_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, float* __restrict__ Y,
                                                            int ldy, int M, int N, const float* __restrict__ bias, int tiles_per_wg, int wgs_per_cb) {
  constexpr int LDK = K + 2;
  constexpr int F4R = K / 4;            // float4 per row
  constexpr int RPP = 512 / F4R;        // rows per staging pass
  constexpr int AS = 64 / RPP;          // A slots per thread
  constexpr int WS = 128 / RPP;         // W slots per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                     // [128][LDK]
  float* As = smem + 128 * LDK;         // [2][64][LDK]
  const int ncb = N / 128;
  const int per_xcd = gridDim.x >> 3;
  const int L = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);      // an XCD works on consecutive row runs and all their column blocks
  if (L >= wgs_per_cb * ncb) return;
  const int wg = L / ncb, cb = L - wg * ncb;
  const int tilesM = M / 64;
  const int t0 = wg * tiles_per_wg, t1 = min(tilesM, t0 + tiles_per_wg);
  if (t0 >= t1) return;
  const int n0 = cb * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
  const int l31 = lane & 31, lh = lane >> 5;
  const int lrow = tid / F4R, lc4 = (tid % F4R) * 4;
  // W block -> LDS
  {
    float4 rw[WS];
#pragma unroll
    for (int i = 0; i < WS; ++i) rw[i] = *reinterpret_cast<const float4*>(W + (size_t)(n0 + lrow + RPP * i) * ldw + lc4);
#pragma unroll
    for (int i = 0; i < WS; ++i) {
      float* d = Ws + (lrow + RPP * i) * LDK + lc4;
      *reinterpret_cast<float2*>(d) = make_float2(rw[i].x, rw[i].y);
      *reinterpret_cast<float2*>(d + 2) = make_float2(rw[i].z, rw[i].w);
    }
  }
  float4 ra[AS];
  auto gload = [&](int t) {
    const float* pa = A + ((size_t)t * 64 + lrow) * lda + lc4;
#pragma unroll
    for (int i = 0; i < AS; ++i) ra[i] = *reinterpret_cast<const float4*>(pa + (size_t)(RPP * i) * lda);
  };
  auto sstore = [&](int buf) {
    float* a = As + buf * 64 * LDK + lrow * LDK + lc4;
#pragma unroll
    for (int i = 0; i < AS; ++i) {
      *reinterpret_cast<float2*>(a + RPP * i * LDK) = make_float2(ra[i].x, ra[i].y);
      *reinterpret_cast<float2*>(a + RPP * i * LDK + 2) = make_float2(ra[i].z, ra[i].w);
    }
  };
  const float bv = bias ? bias[n0 + 32 * wn + l31] : 0.f;
  const float* bfrag = Ws + (32 * wn + l31) * LDK + 2 * lh;
  auto compute = [&](int buf, f32x16& acc) {
    const float* a = As + buf * 64 * LDK + (32 * wm + l31) * LDK + 2 * lh;
#pragma unroll
    for (int kk = 0; kk < K / 4; ++kk) {
      const float2 af = *reinterpret_cast<const float2*>(a + 4 * kk);
      const float2 bf = *reinterpret_cast<const float2*>(bfrag + 4 * kk);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc, 0, 0, 0);
    }
  };
  auto ystore = [&](int t, const f32x16& acc) {
    float* yb = Y + ((size_t)t * 64 + 32 * wm + 4 * lh) * ldy + n0 + 32 * wn + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) yb[(size_t)ROFF(r) * ldy] = acc[r] + bv;
  };
  gload(t0);
  sstore(0);
  __syncthreads();
  if (DBL) {
    // two accumulator sets: tile t's stores are issued after tile t+1's MFMAs were issued
    f32x16 acc0, acc1;
    for (int t = t0; t < t1; t += 2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
      if (t + 1 < t1) gload(t + 1);
      compute(0, acc0);
      if (t + 1 < t1) sstore(1);
      if (t > t0) ystore(t - 1, acc1);
      __syncthreads();
      if (t + 1 >= t1) { ystore(t, acc0); break; }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
      if (t + 2 < t1) gload(t + 2);
      compute(1, acc1);
      if (t + 2 < t1) sstore(0);
      ystore(t, acc0);
      __syncthreads();
      if (t + 2 >= t1) { ystore(t + 1, acc1); break; }
    }
  } else {
    for (int t = t0; t < t1; ++t) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      if (t + 1 < t1) gload(t + 1);
      compute((t - t0) & 1, acc);
      if (t + 1 < t1) sstore((t - t0 + 1) & 1);
      ystore(t, acc);
      __syncthreads();
    }
  }
}

template <int K, int DBL>
int launch(const float* A, int lda, const float* W, int ldw, float* Y, int ldy, int M, int N, const float* bias, int slots, hipStream_t s) {
  const size_t lds = (size_t)(128 + 2 * 64) * (K + 2) * sizeof(float);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_stream_kernel<K, DBL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int ncb = N / 128, tilesM = M / 64;
  int wgs_per_cb = slots / ncb;
  if (wgs_per_cb < 1) wgs_per_cb = 1;
  const int tpw = (tilesM + wgs_per_cb - 1) / wgs_per_cb;
  wgs_per_cb = (tilesM + tpw - 1) / tpw;
  const int grid = ((wgs_per_cb * ncb + 7) / 8) * 8;
  hipLaunchKernelGGL((gemm_stream_kernel<K, DBL>), dim3(grid), dim3(512), lds, s, A, lda, W, ldw, Y, ldy, M, N, bias, tpw, wgs_per_cb);
  return (int)hipGetLastError();
}
}  // namespace

extern "C" int exp_gemm_stream(const float* A, int lda, const float* W, int ldw, float* Y, int ldy, int M, int N, int K, const float* bias, int dbl, int slots,
                               void* s) {
  if (M % 64 || N % 128 || (K != 64 && K != 128)) return -22;
  hipStream_t st = (hipStream_t)s;
  if (K == 128) return dbl ? launch<128, 1>(A, lda, W, ldw, Y, ldy, M, N, bias, slots, st) : launch<128, 0>(A, lda, W, ldw, Y, ldy, M, N, bias, slots, st);
  return dbl ? launch<64, 1>(A, lda, W, ldw, Y, ldy, M, N, bias, slots, st) : launch<64, 0>(A, lda, W, ldw, Y, ldy, M, N, bias, slots, st);
}
