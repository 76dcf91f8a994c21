// EXPERIMENT (not part of the product): ablation variants of gemm_nt to locate its bottleneck.
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

template <int AMODE, int EPI, int TN, int ABL>
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
    if (ABL != 1) { if (kt + 1 < nk) gload((kt + 1) * BK); }
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
        for (int j = 0; j < TN; ++j) { if (ABL != 3) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0); else acc[i][j][0] += af[i].x * bf[j].x; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) { if (ABL != 3) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0); else acc[i][j][1] += af[i].y * bf[j].y; }
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
          if (cok && row < p.M && (ABL != 2 || v == 123456.f)) {
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


}  // namespace
extern "C" int exp_gemm(const spgan_gemm_nt_args* a, int abl, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  const int tm8 = cdiv(cdiv(a->M, BM), 8) * 8;
  dim3 g(tm8 * cdiv(a->N, 128)), b(256);
  switch (abl) {
    case 0: hipLaunchKernelGGL((gemm_nt_kernel<0, 0, 4, 0>), g, b, 0, s, *a); break;
    case 1: hipLaunchKernelGGL((gemm_nt_kernel<0, 0, 4, 1>), g, b, 0, s, *a); break;
    case 2: hipLaunchKernelGGL((gemm_nt_kernel<0, 0, 4, 2>), g, b, 0, s, *a); break;
    case 3: hipLaunchKernelGGL((gemm_nt_kernel<0, 0, 4, 3>), g, b, 0, s, *a); break;
  }
  return (int)hipGetLastError();
}
