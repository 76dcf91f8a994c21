// Shared-MLP contractions on the CDNA4 matrix cores (v_mfma_f32_16x16x4_f32: f32 in, f32
// accumulate, bit-exact fmaf chains).  Two kernels:
//
//   gemm_nt  Y[M,N] = epi( pro(A)[M,K] . W[N,K]^T )   forward 1x1 convs / linears and input-gradients
//   gemm_tn  C[Na,Nb] = sum_m A[m,Na]^T . pro(B)[m,Nb] weight-gradients (deterministic split over m)
//
// Tiling (wave64, 4 waves / workgroup):
//   gemm_nt: 128 x (32*TN) output tile, BK=32.  Each wave owns 64 x (16*TN) as 4 x TN MFMA tiles.
//            LDS tiles are row-major [row][BK+4]: (BK+4)/4 is odd, so the ds_read_b64 fragment reads
//            (16 rows x 2 k-pairs per half-wave) touch all 64 banks exactly once, and rows stay 16-byte
//            aligned for ds_write_b128 staging.  One lane's b64 read supplies the k-operands of TWO
//            consecutive MFMAs (any permutation of k inside a tile is legal as long as A and B agree).
//   gemm_tn: both operands are staged [m][cols+16] exactly as they lie in memory (no transpose); a
//            fragment read is lanes-along-columns b32, leading dimension == 16 (mod 32) keeps the two
//            m-rows of a half-wave on disjoint banks.
//   Workgroup -> tile map is XCD-aware: the 8 XCDs each have a private L2 and block b lands on XCD b%8,
//   so all N-tiles of one M-tile are given ids with the same (b % 8): the A rows they share are fetched
//   into one L2 only.
//
// Fused prologues/epilogues (see spgan_hip.h): BatchNorm-apply + LeakyReLU on the operand load, the
// EdgeBlock per-edge difference gather, bias / per-shape bias / activation, per-tile column statistics
// for the following train-mode BatchNorm, and the LeakyReLU/BatchNorm backward masks with their
// column sums.
#include "common.hpp"

namespace {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDT = BK + 4;  // 36 floats: 16B-aligned rows, (LDT/4) odd -> conflict-free b64 fragment reads

struct Tile {
  int tm, tn;
};

// XCD-aware bijection from the 1-D block id to (tileM, tileN); ids with tileM >= tilesM are idle.
__device__ __forceinline__ Tile map_tile(int tilesN) {
  const int id = blockIdx.x;
  const int xcd = id & 7;
  const int t = id >> 3;
  Tile r;
  r.tn = t % tilesN;
  r.tm = xcd + 8 * (t / tilesN);
  return r;
}

__device__ __forceinline__ float4 ld4(const float* p, bool vec, int k, int K) {
  // p points at element k of a row; K is the row's logical length.
  if (vec && k + 3 < K) return *reinterpret_cast<const float4*>(p);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (k < K) v.x = p[0];
  if (k + 1 < K) v.y = p[1];
  if (k + 2 < K) v.z = p[2];
  if (k + 3 < K) v.w = p[3];
  return v;
}

__device__ __forceinline__ float4 affine_lrelu4(float4 v, float4 sc, float4 sh, float slope) {
  v.x = lrelu_f(fmaf(v.x, sc.x, sh.x), slope);
  v.y = lrelu_f(fmaf(v.y, sc.y, sh.y), slope);
  v.z = lrelu_f(fmaf(v.z, sc.z, sh.z), slope);
  v.w = lrelu_f(fmaf(v.w, sc.w, sh.w), slope);
  return v;
}

// Zero the k >= K tail of a transformed operand (affine of a padded zero is not zero).
__device__ __forceinline__ float4 mask_tail(float4 v, int k, int K) {
  if (k >= K) v.x = 0.f;
  if (k + 1 >= K) v.y = 0.f;
  if (k + 2 >= K) v.z = 0.f;
  if (k + 3 >= K) v.w = 0.f;
  return v;
}

template <int AMODE>
__device__ __forceinline__ float4 load_a(const spgan_gemm_nt_args& p, int m, int k, bool vecA) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (m >= p.M || k >= p.K) return v;
  if (AMODE == SPGAN_A_PLAIN) {
    return ld4(p.A + (size_t)m * p.lda + k, vecA, k, p.K);
  } else if (AMODE == SPGAN_A_AFFINE_LRELU) {
    v = ld4(p.A + (size_t)m * p.lda + k, vecA, k, p.K);
    float4 sc = ld4(p.p_scale + k, false, k, p.K);
    float4 sh = ld4(p.p_shift + k, false, k, p.K);
    return mask_tail(affine_lrelu4(v, sc, sh, p.p_slope), k, p.K);
  } else {  // SPGAN_A_EDGE
    const int i = m / p.e_k;
    const int j = p.e_idx[m];
    float4 vj = ld4(p.A + (size_t)j * p.lda + k, vecA, k, p.K);
    float4 vi = ld4(p.A + (size_t)i * p.lda + k, vecA, k, p.K);
    float4 eb = ld4(p.e_bias + k, false, k, p.K);
    float4 sc = ld4(p.p_scale + k, false, k, p.K);
    float4 sh = ld4(p.p_shift + k, false, k, p.K);
    v.x = (vj.x - vi.x) + eb.x;
    v.y = (vj.y - vi.y) + eb.y;
    v.z = (vj.z - vi.z) + eb.z;
    v.w = (vj.w - vi.w) + eb.w;
    return mask_tail(affine_lrelu4(v, sc, sh, p.p_slope), k, p.K);
  }
}

// Sum per-lane column partials over the 4 row groups of a wave (lanes l, l^16, l^32, l^48) and
// over the two M-waves of the workgroup.  `red` is [2][BN] floats of LDS.
template <int TN>
__device__ __forceinline__ void col_reduce(float (&part)[TN], float* red, int wm, int wn, int lane) {
  constexpr int BN = 32 * TN;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    float v = part[tn];
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (lane < 16) red[wm * BN + wn * TN * 16 + tn * 16 + lane] = v;
  }
  __syncthreads();
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int c = wn * TN * 16 + tn * 16 + (lane & 15);
    part[tn] = red[c] + red[BN + c];
  }
  __syncthreads();
}

template <int AMODE, int EPI, int TN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const spgan_gemm_nt_args p) {
  constexpr int BN = 32 * TN;
  __shared__ __attribute__((aligned(16))) float smem[BM * LDT + BN * LDT + 2 * BN];
  float* As = smem;
  float* Bs = smem + BM * LDT;
  float* red = Bs + BN * LDT;

  const int tilesN = (p.N + BN - 1) / BN;
  const int tilesM = (p.M + BM - 1) / BM;
  const Tile t = map_tile(tilesN);
  if (t.tm >= tilesM) return;
  const int m0 = t.tm * BM, n0 = t.tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lg = lane >> 4;

  const bool vecA = ((p.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.A) & 15) == 0);
  const bool vecW = ((p.ldw & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.W) & 15) == 0);

  f32x4 acc[4][TN];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 ra[4], rb[TN];
  const int lrow = tid >> 3, lc4 = (tid & 7) * 4;  // staging slot: row (tid/8 + 32*i), k offset 4*(tid%8)

  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = load_a<AMODE>(p, m0 + lrow + 32 * i, k0 + lc4, vecA);
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int n = n0 + lrow + 32 * i, k = k0 + lc4;
      rb[i] = (n < p.N && k < p.K) ? ld4(p.W + (size_t)n * p.ldw + k, vecW, k, p.K) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&As[(lrow + 32 * i) * LDT + lc4]) = ra[i];
#pragma unroll
    for (int i = 0; i < TN; ++i) *reinterpret_cast<float4*>(&Bs[(lrow + 32 * i) * LDT + lc4]) = rb[i];
  };

  const int nk = (p.K + BK - 1) / BK;
  gload(0);
  sstore();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload((kt + 1) * BK);  // next tile's HBM loads fly under this tile's MFMAs
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      float2 af[4], bf[TN];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        af[i] = *reinterpret_cast<const float2*>(&As[(wm * 64 + i * 16 + l15) * LDT + kk * 8 + 2 * lg]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bf[j] = *reinterpret_cast<const float2*>(&Bs[(wn * TN * 16 + j * 16 + l15) * LDT + kk * 8 + 2 * lg]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (kt + 1 < nk) {
      sstore();
      __syncthreads();
    }
  }

  // ---------------------------------------------------------------- epilogue
  // C/D layout of v_mfma_f32_16x16x4_f32: col = lane&15, row = 4*(lane>>4) + reg.
  const int rbase = m0 + wm * 64 + 4 * lg;
  const int cbase = n0 + wn * TN * 16 + l15;
  const int rows_valid = min(BM, p.M - m0);

  if (EPI == SPGAN_EPI_LINEAR) {
    float csum[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = cbase + j * 16;
      const bool cok = col < p.N;
      const float b = (cok && p.bias) ? p.bias[col] : 0.f;
      csum[j] = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rbase + i * 16 + r;
          float v = acc[i][j][r] + b;
          if (p.rowbias && cok && row < p.M) v += p.rowbias[(size_t)(row / p.rows_per_group) * p.ld_rowbias + col];
          acc[i][j][r] = v;  // keep the pre-activation value for the statistics pass
          if (row < p.M) csum[j] += v;
          if (cok && row < p.M) {
            float o = v;
            if (p.act == SPGAN_ACT_LRELU) o = lrelu_f(v, p.act_slope);
            else if (p.act == SPGAN_ACT_TANH) o = tanhf(v);
            p.Y[(size_t)row * p.ldy + col] = o;
          }
        }
    }
    if (p.stats) {
      // per-tile (sum, centred M2): combined later with Chan's formula -> no E[x^2]-E[x]^2 cancellation
      col_reduce<TN>(csum, red, wm, wn, lane);
      float m2[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const float mean = csum[j] / (float)rows_valid;
        m2[j] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = rbase + i * 16 + r;
            const float d = acc[i][j][r] - mean;
            if (row < p.M) m2[j] = fmaf(d, d, m2[j]);
          }
      }
      col_reduce<TN>(m2, red, wm, wn, lane);
      if (wm == 0 && lg == 0) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = cbase + j * 16;
          if (col < p.N) {
            float* o = p.stats + ((size_t)t.tm * p.N + col) * 2;
            o[0] = csum[j];
            o[1] = m2[j];
          }
        }
      }
    }
  } else if (EPI == SPGAN_EPI_MASK_OUT) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = cbase + j * 16;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rbase + i * 16 + r;
          if (col < p.N && row < p.M) {
            const float ref = p.ref[(size_t)row * p.ld_ref + col];
            p.Y[(size_t)row * p.ldy + col] = acc[i][j][r] * lrelu_mask(ref, p.b_slope);
          }
        }
    }
  } else {  // BNBWD / EDGE_BNBWD
    float s0[TN], s1[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = cbase + j * 16;
      const bool cok = col < p.N;
      const float sc = cok ? p.b_scale[col] : 0.f, sh = cok ? p.b_shift[col] : 0.f;
      const float mu = cok ? p.b_mean[col] : 0.f, inv = cok ? p.b_invstd[col] : 0.f;
      const float eb = (EPI == SPGAN_EPI_EDGE_BNBWD && cok) ? p.e_bias2[col] : 0.f;
      s0[j] = 0.f;
      s1[j] = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rbase + i * 16 + r;
          if (cok && row < p.M) {
            float y;
            if (EPI == SPGAN_EPI_EDGE_BNBWD) {
              const int pi = row / p.e_k, pj = p.e_idx[row];
              y = (p.ref[(size_t)pj * p.ld_ref + col] - p.ref[(size_t)pi * p.ld_ref + col]) + eb;
            } else {
              y = p.ref[(size_t)row * p.ld_ref + col];
            }
            const float z = fmaf(y, sc, sh);
            const float g = acc[i][j][r] * lrelu_mask(z, p.b_slope);
            const float xh = (y - mu) * inv;
            p.Y[(size_t)row * p.ldy + col] = g;
            s0[j] += g;
            s1[j] = fmaf(g, xh, s1[j]);
          }
        }
    }
    if (p.stats) {
      col_reduce<TN>(s0, red, wm, wn, lane);
      col_reduce<TN>(s1, red, wm, wn, lane);
      if (wm == 0 && lg == 0) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = cbase + j * 16;
          if (col < p.N) {
            float* o = p.stats + ((size_t)t.tm * p.N + col) * 2;
            o[0] = s0[j];
            o[1] = s1[j];
          }
        }
      }
    }
  }
}

template <int AMODE, int EPI>
int launch_nt(const spgan_gemm_nt_args& a, hipStream_t s) {
  const int tilesM = cdiv(a.M, BM);
  const int tm8 = cdiv(tilesM, 8) * 8;
  if (a.N > 64) {
    hipLaunchKernelGGL((gemm_nt_kernel<AMODE, EPI, 4>), dim3(tm8 * cdiv(a.N, 128)), dim3(256), 0, s, a);
  } else if (a.N > 32) {
    hipLaunchKernelGGL((gemm_nt_kernel<AMODE, EPI, 2>), dim3(tm8 * cdiv(a.N, 64)), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL((gemm_nt_kernel<AMODE, EPI, 1>), dim3(tm8 * cdiv(a.N, 32)), dim3(256), 0, s, a);
  }
  return spgan_launch_status();
}

// ------------------------------------------------------------------------------------------ gemm_tn
constexpr int TKM = 16;         // m-rows per staging step
constexpr int TA = 128;         // output rows (columns of A) per workgroup

template <int BMODE>
__device__ __forceinline__ float4 load_b_tn(const spgan_gemm_tn_args& p, int m, int c, bool vecB) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (m >= p.M || c >= p.Nb) return v;
  if (BMODE == SPGAN_A_PLAIN) {
    return ld4(p.B + (size_t)m * p.ldb + c, vecB, c, p.Nb);
  } else if (BMODE == SPGAN_A_AFFINE_LRELU) {
    v = ld4(p.B + (size_t)m * p.ldb + c, vecB, c, p.Nb);
    float4 sc = ld4(p.p_scale + c, false, c, p.Nb);
    float4 sh = ld4(p.p_shift + c, false, c, p.Nb);
    return mask_tail(affine_lrelu4(v, sc, sh, p.p_slope), c, p.Nb);
  } else {
    const int i = m / p.e_k;
    const int j = p.e_idx[m];
    float4 vj = ld4(p.B + (size_t)j * p.ldb + c, vecB, c, p.Nb);
    float4 vi = ld4(p.B + (size_t)i * p.ldb + c, vecB, c, p.Nb);
    float4 eb = ld4(p.e_bias + c, false, c, p.Nb);
    float4 sc = ld4(p.p_scale + c, false, c, p.Nb);
    float4 sh = ld4(p.p_shift + c, false, c, p.Nb);
    v.x = (vj.x - vi.x) + eb.x;
    v.y = (vj.y - vi.y) + eb.y;
    v.z = (vj.z - vi.z) + eb.z;
    v.w = (vj.w - vi.w) + eb.w;
    return mask_tail(affine_lrelu4(v, sc, sh, p.p_slope), c, p.Nb);
  }
}

// grid: (tilesA * tilesB, splits).  Each workgroup reduces `rows_per_split` m-rows into one
// TA x (32*TN) partial tile written to ws[split][Na][Nb].
template <int BMODE, int TN>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const spgan_gemm_tn_args p, int rows_per_split) {
  constexpr int TB = 32 * TN;
  constexpr int LDA_ = TA + 16, LDB_ = TB + 16;
  __shared__ __attribute__((aligned(16))) float smem[TKM * LDA_ + TKM * LDB_];
  float* As = smem;
  float* Bs = smem + TKM * LDA_;

  const int tilesB = (p.Nb + TB - 1) / TB;
  const int ta = blockIdx.x / tilesB, tb = blockIdx.x % tilesB;
  const int a0 = ta * TA, b0 = tb * TB;
  const int split = blockIdx.y;
  const int mbeg = split * rows_per_split;
  const int mend = min(p.M, mbeg + rows_per_split);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;  // wave owns A-cols [wm*64, +64), B-cols [wn*16*TN, +16*TN)
  const int l15 = lane & 15, lg = lane >> 4;
  const bool vecA = ((p.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.A) & 15) == 0);
  const bool vecB = ((p.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.B) & 15) == 0);

  f32x4 acc[4][TN];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // staging: A tile 16 x 128 floats = 512 float4 -> 2 per thread; B tile 16 x TB -> (TB/4*16)/256 per thread
  constexpr int BSLOTS = (TKM * TB / 4 + 255) / 256;
  float4 ra[2], rb[BSLOTS];
  auto gload = [&](int mb) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int s = tid + 256 * i, r = s >> 5, c = (s & 31) * 4;
      const int m = mb + r, col = a0 + c;
      ra[i] = (m < mend && col < p.Na) ? ld4(p.A + (size_t)m * p.lda + col, vecA, col, p.Na) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < BSLOTS; ++i) {
      const int s = tid + 256 * i;
      const int r = s / (TB / 4), c = (s % (TB / 4)) * 4;
      const int m = mb + r;
      rb[i] = (r < TKM && m < mend) ? load_b_tn<BMODE>(p, m, b0 + c, vecB) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int s = tid + 256 * i, r = s >> 5, c = (s & 31) * 4;
      *reinterpret_cast<float4*>(&As[r * LDA_ + c]) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < BSLOTS; ++i) {
      const int s = tid + 256 * i;
      const int r = s / (TB / 4), c = (s % (TB / 4)) * 4;
      if (r < TKM) *reinterpret_cast<float4*>(&Bs[r * LDB_ + c]) = rb[i];
    }
  };

  if (mbeg < mend) {
    gload(mbeg);
    sstore();
    __syncthreads();
    for (int mb = mbeg; mb < mend; mb += TKM) {
      const bool more = mb + TKM < mend;
      if (more) gload(mb + TKM);
#pragma unroll
      for (int kq = 0; kq < TKM / 4; ++kq) {
        float af[4], bf[TN];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = As[(kq * 4 + lg) * LDA_ + wm * 64 + i * 16 + l15];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = Bs[(kq * 4 + lg) * LDB_ + wn * TN * 16 + j * 16 + l15];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
      if (more) {
        sstore();
        __syncthreads();
      }
    }
  }
  // partial tile -> ws[split][Na][Nb]   (D: row = A-col index, col = B-col index)
  float* out = p.ws + (size_t)split * p.Na * p.Nb;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = a0 + wm * 64 + i * 16 + 4 * lg + r;
        const int col = b0 + wn * TN * 16 + j * 16 + l15;
        if (row < p.Na && col < p.Nb) out[(size_t)row * p.Nb + col] = acc[i][j][r];
      }
}

// Fixed-order sum over the split partials: 64 consecutive outputs x 4 split-slices per workgroup.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int Na, int Nb, float* __restrict__ C,
                                                            int ldc, float beta) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const size_t stride = (size_t)Na * Nb;
  float s0 = 0.f, s1 = 0.f;
  if (i < Na * Nb) {
    int k = sl;
    for (; k + 4 < splits; k += 8) {
      s0 += ws[(size_t)k * stride + i];
      s1 += ws[(size_t)(k + 4) * stride + i];
    }
    if (k < splits) s0 += ws[(size_t)k * stride + i];
  }
  red[sl][lane] = s0 + s1;
  __syncthreads();
  if (sl == 0 && i < Na * Nb) {
    const float s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    const int r = i / Nb, c = i % Nb;
    float* o = C + (size_t)r * ldc + c;
    *o = (beta == 0.f) ? s : fmaf(beta, *o, s);
  }
}

// Split choice: about two workgroups per CU in total, at least 256 m-rows each.
inline void tn_plan(int M, int Na, int Nb, int* splits, int* rows) {
  const int TB = Nb > 64 ? 128 : (Nb > 32 ? 64 : 32);
  const int tiles = cdiv(Na, TA) * cdiv(Nb, TB);
  int want = cdiv(512, tiles);
  int r = cdiv(M, want);
  if (r < 256) r = 256;
  r = cdiv(r, TKM) * TKM;
  *rows = r;
  *splits = cdiv(M, r);
}

template <int BMODE>
int launch_tn(const spgan_gemm_tn_args& a, hipStream_t s) {
  int splits, rows;
  tn_plan(a.M, a.Na, a.Nb, &splits, &rows);
  if (a.Nb > 64) {
    hipLaunchKernelGGL((gemm_tn_kernel<BMODE, 4>), dim3(cdiv(a.Na, TA) * cdiv(a.Nb, 128), splits), dim3(256), 0, s, a, rows);
  } else if (a.Nb > 32) {
    hipLaunchKernelGGL((gemm_tn_kernel<BMODE, 2>), dim3(cdiv(a.Na, TA) * cdiv(a.Nb, 64), splits), dim3(256), 0, s, a, rows);
  } else {
    hipLaunchKernelGGL((gemm_tn_kernel<BMODE, 1>), dim3(cdiv(a.Na, TA) * cdiv(a.Nb, 32), splits), dim3(256), 0, s, a, rows);
  }
  const int n = a.Na * a.Nb;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(n, 64)), dim3(256), 0, s, a.ws, splits, a.Na, a.Nb, a.C, a.ldc, a.beta);
  return spgan_launch_status();
}

}  // namespace

extern "C" int spgan_gemm_nt(const spgan_gemm_nt_args* a, spgan_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  SPGAN_CHECK_ARG(a && a->A && a->W && a->Y && a->M > 0 && a->N > 0 && a->K > 0);
  SPGAN_CHECK_ARG(a->lda >= (a->a_mode == SPGAN_A_EDGE ? a->K : a->K) && a->ldw >= a->K && a->ldy >= a->N);
  if (a->a_mode != SPGAN_A_PLAIN) SPGAN_CHECK_ARG(a->p_scale && a->p_shift);
  if (a->a_mode == SPGAN_A_EDGE) SPGAN_CHECK_ARG(a->e_idx && a->e_bias && a->e_k > 0);
  if (a->rowbias) SPGAN_CHECK_ARG(a->rows_per_group > 0 && a->ld_rowbias >= a->N);
  switch (a->epi_mode) {
    case SPGAN_EPI_LINEAR:
      if (a->a_mode == SPGAN_A_PLAIN) return launch_nt<SPGAN_A_PLAIN, SPGAN_EPI_LINEAR>(*a, s);
      if (a->a_mode == SPGAN_A_AFFINE_LRELU) return launch_nt<SPGAN_A_AFFINE_LRELU, SPGAN_EPI_LINEAR>(*a, s);
      if (a->a_mode == SPGAN_A_EDGE) return launch_nt<SPGAN_A_EDGE, SPGAN_EPI_LINEAR>(*a, s);
      return SPGAN_EINVAL;
    case SPGAN_EPI_MASK_OUT:
      SPGAN_CHECK_ARG(a->a_mode == SPGAN_A_PLAIN && a->ref && a->ld_ref >= a->N);
      return launch_nt<SPGAN_A_PLAIN, SPGAN_EPI_MASK_OUT>(*a, s);
    case SPGAN_EPI_BNBWD:
      SPGAN_CHECK_ARG(a->a_mode == SPGAN_A_PLAIN && a->ref && a->ld_ref >= a->N && a->b_scale && a->b_shift && a->b_mean && a->b_invstd);
      return launch_nt<SPGAN_A_PLAIN, SPGAN_EPI_BNBWD>(*a, s);
    case SPGAN_EPI_EDGE_BNBWD:
      SPGAN_CHECK_ARG(a->a_mode == SPGAN_A_PLAIN && a->ref && a->ld_ref >= a->N && a->b_scale && a->b_shift && a->b_mean && a->b_invstd &&
                      a->e_idx && a->e_k > 0 && a->e_bias2);
      return launch_nt<SPGAN_A_PLAIN, SPGAN_EPI_EDGE_BNBWD>(*a, s);
    default:
      return SPGAN_EINVAL;
  }
}

extern "C" size_t spgan_gemm_tn_ws_bytes(int M, int Na, int Nb) {
  if (M <= 0 || Na <= 0 || Nb <= 0) return 0;
  int splits, rows;
  tn_plan(M, Na, Nb, &splits, &rows);
  return (size_t)splits * Na * Nb * sizeof(float);
}

extern "C" int spgan_gemm_tn(const spgan_gemm_tn_args* a, spgan_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  SPGAN_CHECK_ARG(a && a->A && a->B && a->C && a->ws && a->M > 0 && a->Na > 0 && a->Nb > 0);
  SPGAN_CHECK_ARG(a->lda >= a->Na && a->ldb >= a->Nb && a->ldc >= a->Nb);
  SPGAN_CHECK_ARG(a->ws_bytes >= spgan_gemm_tn_ws_bytes(a->M, a->Na, a->Nb));
  if (a->b_mode != SPGAN_A_PLAIN) SPGAN_CHECK_ARG(a->p_scale && a->p_shift);
  switch (a->b_mode) {
    case SPGAN_A_PLAIN: return launch_tn<SPGAN_A_PLAIN>(*a, s);
    case SPGAN_A_AFFINE_LRELU: return launch_tn<SPGAN_A_AFFINE_LRELU>(*a, s);
    case SPGAN_A_EDGE:
      SPGAN_CHECK_ARG(a->e_idx && a->e_bias && a->e_k > 0);
      return launch_tn<SPGAN_A_EDGE>(*a, s);
    default: return SPGAN_EINVAL;
  }
}
