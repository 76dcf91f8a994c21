// spgan_gemm_dual: the backward of one 1x1-conv layer behind a train-mode BatchNorm + LeakyReLU as ONE launch -- the input-gradient
// product AND the weight-gradient product from ONE staging of the incoming gradient tile (round-3 review item 1; the layers:
// Generation/Generator.py:56-63 conv_w.3 of an EdgeBlock, Generation/Discriminator.py:55-65 mlps.3):
//
//   dy[m, :]   = p*g[m, :] + q*y[m, :] + r                  the layer's output gradient, the lazy BatchNorm-backward operand (or dense)
//   a[m, :]    = lrelu(sc*pre[m, :] + sh)                    the layer's INPUT: the previous layer's BatchNorm + LeakyReLU applied on load;
//                                                            pre[m] = B[m] (plain) or B[idx[m]] - B[m / k] + e_bias (per-edge operand)
//   dW[NA,NB]  = sum_m dy[m, :]^T a[m, :]                    weight gradient (split over workgroups, partials summed in fixed order)
//   G[m, :]    = (dy[m, :] . W) * lrelu'(sc*pre[m, :] + sh)  gradient w.r.t. the previous layer's BatchNorm output, masked
//   stats      = (sum_m G, sum_m G * xhat),  xhat = (pre - mean)*invstd        the previous BatchNorm's backward sums
//
// Before, gemm_tn (dW) and gemm_nt with the BNBWD / EDGE_BNBWD epilogue (G, stats) each read g, y and pre: at the EdgeBlock's size
// (655,360 edges x 128 channels) 2 x 840 MB; both launches ran at 32-39 % of the fp32 matrix peak, bound by that traffic.  Here a
// workgroup owns a contiguous run of 32-row chunks; per chunk the dy tile [32 x 128] and the pre tile [32 x 64] are staged ONCE into
// LDS and feed both products:
//   input gradient  [32 x 64]  = dy [32 x 128] . W [128 x 64]      v_mfma_f32_16x16x4_f32, 2 tiles per wave, K = 128 (W stays in LDS)
//   weight gradient [128 x 64] += dy^T [128 x 32] . a [32 x 64]    v_mfma_f32_32x32x2_f32, 2 tiles per wave, accumulated over all chunks
// -- 2048 + 2048 matrix-pipe cycles per wave and chunk.  The next chunk's global loads are in flight under the MFMAs (raw values in
// registers; the affine maps are applied at the LDS store, csrc/gemm.hip "prologues run at the LDS store"), the per-edge neighbour
// indices are fetched one chunk further ahead.  LDS: dy 32 x 130 | pre 32 x 68 | W 128 x 80 | G tile 32 x 68 floats = 75 KB -> two
// workgroups per CU.  The masked input-gradient tile goes through LDS so that it leaves as 16-byte stores of full 256-byte rows.
// Bank maths (ds_read_b32: 32-lane groups, 32 banks): dy pitch 130 = 2 mod 32 (16 rows x 2 k-values of a 16x16x4 A fragment land on 32
// banks; a 32x32x2 A fragment is 32 consecutive floats of one row), W pitch 80 = 16 mod 32 (2 k-rows x 16 columns), pre pitch 68
// (4*68 = 16 mod 32: the epilogue's two rows 4 apart x 16 columns).
// Deterministic: fixed chunk -> workgroup assignment, fixed-order sums, no atomics.  fp32 operands only (the "f16" operand mode keeps the
// two-launch route with its 16-bit storage).
#include "common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int R = 32;      // rows per chunk
constexpr int NA = 128;    // columns of dy  (= output channels of the layer)
constexpr int NB = 64;     // columns of pre (= input channels of the layer)
constexpr int LDY = 130, LDP = 68, LDW = 80, LDG = 68;
constexpr int SM_DY = R * LDY, SM_PRE = R * LDP, SM_W = NA * LDW, SM_G = R * LDG;

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int EK>   // 0: plain pre tensor; > 0: per-edge operand with EK edges per point
__global__ __launch_bounds__(256, 2) void gemm_dual_kernel(const spgan_gemm_dual_args p, int chunks_per_wg, int wgs) {
  __shared__ __attribute__((aligned(16))) float sm[SM_DY + SM_PRE + SM_W + SM_G];
  float* dys = sm;
  float* pres = sm + SM_DY;
  float* ws_ = sm + SM_DY + SM_PRE;
  float* gzs = sm + SM_DY + SM_PRE + SM_W;      // the masked input-gradient tile [32 x 64] on its way to coalesced 16-byte stores

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;     // 16x16x4 fragment coordinates
  const int l31 = lane & 31, lh = lane >> 5;     // 32x32x2 fragment coordinates
  const int chunks = p.M / R;
  // XCD-aware: the hardware deals consecutive workgroup ids round-robin to the 8 XCDs; logical workgroup (= run of rows) xcd*per + t goes
  // to XCD xcd, so that an XCD works on CONSECUTIVE rows -- the per-edge operand's neighbour rows of one shape (0.5 MB of the point tensor)
  // are then gathered through ONE L2 instead of all eight
  const int per_xcd = gridDim.x >> 3;
  const int wg = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (wg >= wgs) return;
  const int c0 = wg * chunks_per_wg;
  const int c1 = min(chunks, c0 + chunks_per_wg);
  const bool a2 = p.A2 != nullptr;

  // ---- W [NA, NB] -> LDS once (row = column of dy, pitch LDW)
#pragma unroll
  for (int i = 0; i < NA * NB / 4 / 256; ++i) {
    const int s = tid + 256 * i, row = s >> 4, c4 = (s & 15) * 4;
    *reinterpret_cast<float4*>(&ws_[row * LDW + c4]) = ldg4(p.W + (size_t)row * p.ldw + c4);
  }

  // ---- per-thread constants
  // staging slots: dy tile 32 x 128 -> 4 float4 per thread and tensor (row = s >> 5, column quad fixed per thread); pre tile 32 x 64 -> 2
  const int ycol = (tid & 31) * 4, yrow0 = tid >> 5;      // rows yrow0 + 8 i
  const int pcol = (tid & 15) * 4, prow0 = tid >> 4;      // rows prow0 + 16 i
  float4 cp = make_float4(1.f, 1.f, 1.f, 1.f), cq = make_float4(0.f, 0.f, 0.f, 0.f), cr = cq, eb = cq;
  if (a2) {
    cp = ldg4(p.p + ycol); cq = ldg4(p.q + ycol); cr = ldg4(p.r + ycol);
  }
  if (EK > 0) eb = ldg4(p.e_bias + pcol);
  // weight-gradient B fragments: columns 32 cj + l31 of pre; input-gradient tiles: columns 16 tn + l15, tn = 2 (w & 1) + t
  const int ti = w >> 1;
  float sc_w[2], sh_w[2], sc_d[2], sh_d[2], mu_d[2], iv_d[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    sc_w[j] = p.b_scale[32 * j + l31]; sh_w[j] = p.b_shift[32 * j + l31];
    const int c = 16 * (2 * (w & 1) + j) + l15;
    sc_d[j] = p.b_scale[c]; sh_d[j] = p.b_shift[c]; mu_d[j] = p.b_mean[c]; iv_d[j] = p.b_invstd[c];
  }
  const float slope = p.slope;

  f32x16 accw[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[j][r] = 0.f;
  float s0[2] = {0.f, 0.f}, s1[2] = {0.f, 0.f};

  float4 ra[4], ra2[4], rb[2], rb2[2];
  int nidx[2] = {0, 0};      // neighbour rows of the chunk AFTER the one whose values are being loaded

  // global addressing: a uniform base per chunk (scalar registers) + a 32-bit per-thread element offset (the rows of a chunk are less
  // than 2^31 elements apart)
  const unsigned offA = (unsigned)yrow0 * (unsigned)p.lda + (unsigned)ycol, stepA = 8u * (unsigned)p.lda;
  const unsigned offA2 = (unsigned)yrow0 * (unsigned)p.lda2 + (unsigned)ycol, stepA2 = 8u * (unsigned)p.lda2;
  const unsigned offB = (unsigned)prow0 * (unsigned)p.ldb + (unsigned)pcol, stepB = 16u * (unsigned)p.ldb;
  auto iload = [&](int c) {  // neighbour indices of chunk c (clamped: a chunk past the end is never stored)
    if (EK > 0) {
      const int m0 = min(c, chunks - 1) * R;
#pragma unroll
      for (int i = 0; i < 2; ++i) nidx[i] = p.e_idx[m0 + prow0 + 16 * i];
    }
  };
  auto gload = [&](int c) {
    const float* Ab = p.A + (size_t)c * R * p.lda;
    const float* A2b = p.A2 + (size_t)c * R * p.lda2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = ldg4(Ab + (offA + i * stepA));
      if (a2) ra2[i] = ldg4(A2b + (offA2 + i * stepA2));
    }
    if (EK > 0) {
      const unsigned m0 = (unsigned)c * R + (unsigned)prow0;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        rb[i] = ldg4(p.B + ((unsigned)nidx[i] * (unsigned)p.ldb + (unsigned)pcol));
        rb2[i] = ldg4(p.B + (((m0 + 16u * i) / (unsigned)(EK > 0 ? EK : 1)) * (unsigned)p.ldb + (unsigned)pcol));
      }
    } else {
      const float* Bb = p.B + (size_t)c * R * p.ldb;
#pragma unroll
      for (int i = 0; i < 2; ++i) rb[i] = ldg4(Bb + (offB + i * stepB));
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = ra[i];
      if (a2) {
        v.x = fmaf(v.x, cp.x, fmaf(ra2[i].x, cq.x, cr.x));
        v.y = fmaf(v.y, cp.y, fmaf(ra2[i].y, cq.y, cr.y));
        v.z = fmaf(v.z, cp.z, fmaf(ra2[i].z, cq.z, cr.z));
        v.w = fmaf(v.w, cp.w, fmaf(ra2[i].w, cq.w, cr.w));
      }
      float* d = &dys[(yrow0 + 8 * i) * LDY + ycol];      // pitch 130 floats: 8-byte aligned rows
      *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
      *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float4 v = rb[i];
      if (EK > 0) {
        v.x = (v.x - rb2[i].x) + eb.x;
        v.y = (v.y - rb2[i].y) + eb.y;
        v.z = (v.z - rb2[i].z) + eb.z;
        v.w = (v.w - rb2[i].w) + eb.w;
      }
      *reinterpret_cast<float4*>(&pres[(prow0 + 16 * i) * LDP + pcol]) = v;
    }
  };

  // Order of the memory operations of a chunk (vmcnt counts loads and stores in issue order, and a register a store still has to read
  // cannot be reloaded): the loads of chunk c+2 are issued right after chunk c+1's registers went to LDS and BEFORE chunk c's gradient
  // tile is stored -- they fly for a whole chunk, and waiting for them never waits for a store.
  if (c0 < c1) {
    iload(c0);
    gload(c0);
    iload(c0 + 1);
    sstore();
    if (c0 + 1 < c1) {
      gload(c0 + 1);
      iload(c0 + 2);
    }
    __syncthreads();
    for (int c = c0; c < c1; ++c) {
      const bool more = c + 1 < c1;
      // ---- input gradient: [32 x 64] = dy [32 x 128] . W [128 x 64]; this wave: rows 16 ti .. +16, columns 32 (w & 1) .. +32
      f32x4 accd[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) accd[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      {
        const float* ap = dys + (16 * ti + l15) * LDY + lq;
        const float* bp = ws_ + lq * LDW + 32 * (w & 1) + l15;
#pragma unroll 8
        for (int kq = 0; kq < NA / 4; ++kq) {
          const float af = ap[4 * kq];
          const float b0 = bp[4 * kq * LDW], b1 = bp[4 * kq * LDW + 16];
          accd[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, b0, accd[0], 0, 0, 0);
          accd[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, b1, accd[1], 0, 0, 0);
        }
      }
      // ---- weight gradient: [128 x 64] += dy^T [128 x 32] . a [32 x 64]; this wave: dy columns 32 w .. +32, both column halves of a.
      // The epilogue of the input gradient (LeakyReLU mask of the previous layer, BatchNorm-backward sums, tile -> LDS) is interleaved,
      // one accumulator element behind every second k-step: its VALU work issues while the matrix pipe works off the MFMAs before it.
      {
        const float* ap = dys + lh * LDY + 32 * w + l31;
        const float* bp = pres + lh * LDP + l31;
        const float* ep = pres + (16 * ti + 4 * lq) * LDP + 32 * (w & 1) + l15;
        float* gp = gzs + (16 * ti + 4 * lq) * LDG + 32 * (w & 1) + l15;
#pragma unroll
        for (int kq = 0; kq < R / 2; ++kq) {
          const float af = ap[2 * kq * LDY];
          float b0 = fmaf(bp[2 * kq * LDP], sc_w[0], sh_w[0]);
          float b1 = fmaf(bp[2 * kq * LDP + 32], sc_w[1], sh_w[1]);
          b0 = b0 > 0.f ? b0 : b0 * slope;
          b1 = b1 > 0.f ? b1 : b1 * slope;
          accw[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, b0, accw[0], 0, 0, 0);
          accw[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, b1, accw[1], 0, 0, 0);
          if (kq & 1) {
            const int e = kq >> 1, j = e >> 2, r = e & 3;      // compile-time after unrolling
            const float pre = ep[r * LDP + 16 * j];
            const float z = fmaf(pre, sc_d[j], sh_d[j]);
            const float g = accd[j][r] * (z > 0.f ? 1.f : slope);
            const float xh = (pre - mu_d[j]) * iv_d[j];
            s0[j] += g;
            s1[j] = fmaf(g, xh, s1[j]);
            gp[r * LDG + 16 * j] = g;
          }
        }
      }
      __syncthreads();      // every read of this chunk's tiles is done; the masked input-gradient tile is complete in LDS
      if (more) {
        sstore();                    // chunk c+1: registers -> LDS
        if (c + 2 < c1) {
          gload(c + 2);              // uses the indices fetched one iteration ago
          iload(c + 3);
        }
      }
      {
        float* gout = p.G + (size_t)c * R * p.ldg;
        const unsigned og = (unsigned)prow0 * (unsigned)p.ldg + (unsigned)pcol;
#pragma unroll
        for (int i = 0; i < 2; ++i)
          *reinterpret_cast<float4*>(gout + (og + i * 16u * (unsigned)p.ldg)) = *reinterpret_cast<const float4*>(&gzs[(prow0 + 16 * i) * LDG + pcol]);
      }
      __syncthreads();
    }
  }

  // ---- statistics: lanes with the same column (lq = 0..3), then the two waves that share the columns (ti = 0, 1), fixed order
  float* red = dys;         // the tiles are dead (the loop ended with a barrier)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float a = s0[j], b = s1[j];
    a += __shfl_xor(a, 16); b += __shfl_xor(b, 16);
    a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
    if (lq == 0) {
      red[(w * 32 + 16 * j + l15) * 2] = a;
      red[(w * 32 + 16 * j + l15) * 2 + 1] = b;
    }
  }
  __syncthreads();
  if (tid < NB) {
    const int half = tid >> 5, cc = tid & 31;         // columns 32 half + cc: waves `half` (ti = 0) and `half + 2` (ti = 1)
    const float a = red[(half * 32 + cc) * 2] + red[((half + 2) * 32 + cc) * 2];
    const float b = red[(half * 32 + cc) * 2 + 1] + red[((half + 2) * 32 + cc) * 2 + 1];
    float* st = p.stats + ((size_t)wg * NB + tid) * 2;
    st[0] = a; st[1] = b;
  }
  // ---- weight-gradient partial of this workgroup -> ws[wg][NA][NB]
  float* out = p.ws + (size_t)wg * NA * NB + (size_t)(32 * w + 4 * lh) * NB + l31;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2)) * NB + 32 * j] = accw[j][r];
}

// chunk -> workgroup plan: two workgroups per CU resident (LDS), enough chunks per workgroup to amortise the W load and the partial store
inline void dual_plan(int M, int* wgs, int* cpw) {
  const int chunks = M / R;
  int per = (chunks + 511) / 512;
  if (per < 4) per = 4;
  *cpw = per;
  *wgs = (chunks + per - 1) / per;
}

}  // namespace

extern "C" int spgan_gemm_dual_wgs(int M, int Na, int Nb, int e_k) {
  if (M < 8192 || M % R || Na != NA || Nb != NB || !(e_k == 0 || e_k == 10)) return 0;
  int wgs, cpw;
  dual_plan(M, &wgs, &cpw);
  return wgs;
}

extern "C" int spgan_gemm_dual_rows_per_wg(int M) {
  if (M < R) return 0;
  int wgs, cpw;
  dual_plan(M, &wgs, &cpw);
  return cpw * R;
}

extern "C" int spgan_gemm_dual(const spgan_gemm_dual_args* a, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(a && a->A && a->W && a->B && a->G && a->stats && a->ws && a->b_scale && a->b_shift && a->b_mean && a->b_invstd);
  const int ek = a->e_idx ? a->e_k : 0;
  SPGAN_CHECK_ARG(spgan_gemm_dual_wgs(a->M, a->Na, a->Nb, ek) > 0);
  SPGAN_CHECK_ARG(!a->A2 || (a->p && a->q && a->r));
  SPGAN_CHECK_ARG(!a->e_idx || a->e_bias);
  // 16-byte aligned rows everywhere (float4 loads)
  auto al = [](const void* q, int ld) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0 && (ld & 3) == 0; };
  SPGAN_CHECK_ARG(al(a->A, a->lda) && (!a->A2 || al(a->A2, a->lda2)) && al(a->W, a->ldw) && al(a->B, a->ldb) && al(a->G, a->ldg) && a->ldg >= NB);
  SPGAN_CHECK_ARG(a->lda >= NA && (!a->A2 || a->lda2 >= NA) && a->ldw >= NB && a->ldb >= NB);
  SPGAN_CHECK_ARG(!a->A2 || (al(a->p, 4) && al(a->q, 4) && al(a->r, 4)));
  SPGAN_CHECK_ARG(!a->e_idx || al(a->e_bias, 4));
  int wgs, cpw;
  dual_plan(a->M, &wgs, &cpw);
  hipStream_t s = (hipStream_t)s_;
  const int grid = ((wgs + 7) / 8) * 8;
  if (ek == 0) hipLaunchKernelGGL((gemm_dual_kernel<0>), dim3(grid), dim3(256), 0, s, *a, cpw, wgs);
  else hipLaunchKernelGGL((gemm_dual_kernel<10>), dim3(grid), dim3(256), 0, s, *a, cpw, wgs);
  return spgan_launch_status();
}
