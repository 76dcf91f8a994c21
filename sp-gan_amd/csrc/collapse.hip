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

// spgan_collapse_prep: everything the collapsed backward needs before its big launch, in ONE launch -- one to four W^T diag(alpha) W problems
// (72 or 64 workgroups each, latency-bound on the cold weight matrix: 29 us alone) and one to four sparse-row products S.W (2048 workgroups
// each that stream 67 MB out: 33 us alone); several passes over the same W (the grouped D step: real, fake, the double backward) share it.  The weight workgroups come first in the grid, so they are resident from the start and finish under the
// streaming ones.  Both parts run the device functions of their stand-alone kernels: bit-identical results.
__global__ __launch_bounds__(256) void collapse_prep_kernel(const spgan_collapse_prep_args a, int chunks, int RB) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  int id = blockIdx.x;
  const int kx = a.K / 32;
  // weight problems first (nprob <= SPGAN_GROUP_MAX: the real / fake passes and the double backward's two of a grouped D step)
  for (int q = 0; q < a.nprob; ++q) {
    const int wgs = kx * (kx + (a.cvec[q] ? 1 : 0));
    if (id < wgs) {
      wt_diag_w_body(id % kx, id / kx, wgs / kx, dyn, a.W, a.ldw, a.C, a.K, a.alpha[q], a.beta[q], a.bias[q], a.G[q], a.ldg, a.cvec[q]);
      return;
    }
    id -= wgs;
  }
  // then the sparse-row products, set after set (each: chunks x B workgroups)
  const int per_set = chunks * a.B;
  const int set = id / per_set;
  id -= set * per_set;
  sparse_rows_nt_body(id % chunks, id / chunks, reinterpret_cast<unsigned*>(dyn), a.sp_val[set], a.sp_arg[set], a.rows, a.C, a.W, a.ldw, a.K, a.E[set], a.lde,
                      RB);
}

// spgan_wgrad_collapse: the weight gradient of the collapsed layer, all of its terms in ONE launch
//   out[a, n] (+)= a1[a] * sum_k W[a,k] X1[n,k]  +  (a1[a]*b1[a] + d1[a]) * v1[n]  +  a2[a] * sum_k W[a,k] X2[n,k]
//                 + sum_b val[b,a] * pro(Bm)[arg[b,a], n]
// (before: gemm_nt on 32 workgroups, rowscale_outer, a transpose + a second gemm_nt + axpby in the double backward, tn_sparse_rows: three to six
// launches of 5-15 us each).  One workgroup per 32 x 32 output tile (256 for [1024, 256]); the W tile and the X tiles [32, K <= 256] are staged
// once -- every load of the launch, the arg-max indices of the sparse term included, is in flight together --, the four waves split K
// (v_mfma_f32_32x32x2_f32, fp32 operands), their partial tiles are summed in wave order through LDS, and the epilogue's gathers (one B row
// segment per shape and output row, shapes ascending: deterministic) are issued eight shapes at a time.
constexpr int WG_KMAX = 256, WG_BMAX = 64;
// LDS floats in front of the sparse term's index / value tables: the operand tiles, later overlaid by the waves' partial tiles (4096 per product)
constexpr int wgrad_front(int K, int nx) { return (1 + nx) * 32 * (K + 4) > nx * 4096 ? (1 + nx) * 32 * (K + 4) : nx * 4096; }
constexpr size_t wgrad_lds_bytes(int K, int nx) { return (size_t)(wgrad_front(K, nx) + 2 * 32 * WG_BMAX) * sizeof(float); }

__device__ __forceinline__ void wgrad_collapse_body(const spgan_wgrad_collapse_args& p, float* dyn, int bx, int by) {
  const int K = p.K, LDK = K + 4, K4 = K / 4;
  const bool two = p.X2 != nullptr;
  float* Ws = dyn;                                   // [32][LDK]
  float* X1s = Ws + 32 * LDK;                        // [32][LDK]
  float* X2s = X1s + 32 * LDK;                       // [32][LDK] (two terms)
  int* sarg = reinterpret_cast<int*>(dyn + wgrad_front(K, two ? 2 : 1));  // [B][32]
  float* sval = reinterpret_cast<float*>(sarg + 32 * WG_BMAX);          // [B][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int a0 = bx * 32, n0 = by * 32;
  // ---- stage.  K == 256 (the layer this kernel exists for): every load of the launch is issued before the first LDS store, the sparse
  // term's index / value tables first (the gathers of the epilogue wait for them); other K: tile by tile
  const bool x2_rows = two && !p.x2_t;
  const int nsp = p.sp_val ? 32 * p.B : 0;                 // <= 2048 table entries: <= 8 per thread
  if (K == WG_KMAX) {
    int si[8];
    float sv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = min(tid + 256 * i, max(nsp - 1, 0));
      const size_t off = (size_t)(e >> 5) * p.C + a0 + (e & 31);
      si[i] = nsp ? p.sp_arg[off] : -1;
      sv[i] = nsp ? p.sp_val[off] : 0.f;
    }
    float4 vw[8], vx[8], vy[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + 256 * i, r = e >> 6, c = (e & 63) * 4;
      vw[i] = *reinterpret_cast<const float4*>(p.W + (size_t)(a0 + r) * p.ldw + c);
      vx[i] = *reinterpret_cast<const float4*>(p.X1 + (size_t)(n0 + r) * p.ldx1 + c);
      if (x2_rows) vy[i] = *reinterpret_cast<const float4*>(p.X2 + (size_t)(n0 + r) * p.ldx2 + c);
      else if (two) vy[i] = *reinterpret_cast<const float4*>(p.X2 + (size_t)(e >> 3) * p.ldx2 + n0 + (e & 7) * 4);   // X2 [K,N]: row k = e >> 3, 4 of the tile's columns
      else vy[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + 256 * i;
      if (e < nsp) {
        sarg[e] = (si[i] >= 0 && si[i] < p.B * p.rows) ? si[i] : -1;
        sval[e] = sv[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + 256 * i, r = e >> 6, c = (e & 63) * 4;
      *reinterpret_cast<float4*>(Ws + r * LDK + c) = vw[i];
      *reinterpret_cast<float4*>(X1s + r * LDK + c) = vx[i];
    }
    if (x2_rows) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = tid + 256 * i, r = e >> 6, c = (e & 63) * 4;
        *reinterpret_cast<float4*>(X2s + r * LDK + c) = vy[i];
      }
    } else if (two) {   // transposed into the tile: X2s[n][k]
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = tid + 256 * i, k = e >> 3, c4 = (e & 7) * 4;
        X2s[(c4 + 0) * LDK + k] = vy[i].x; X2s[(c4 + 1) * LDK + k] = vy[i].y; X2s[(c4 + 2) * LDK + k] = vy[i].z; X2s[(c4 + 3) * LDK + k] = vy[i].w;
      }
    }
  } else {
    for (int e = tid; e < nsp; e += 256) {
      const int r = p.sp_arg[(size_t)(e >> 5) * p.C + a0 + (e & 31)];
      sarg[e] = (r >= 0 && r < p.B * p.rows) ? r : -1;
      sval[e] = p.sp_val[(size_t)(e >> 5) * p.C + a0 + (e & 31)];
    }
    for (int e = tid; e < 32 * K4; e += 256) {
      const int r = e / K4, c = (e % K4) * 4;
      *reinterpret_cast<float4*>(Ws + r * LDK + c) = *reinterpret_cast<const float4*>(p.W + (size_t)(a0 + r) * p.ldw + c);
      *reinterpret_cast<float4*>(X1s + r * LDK + c) = *reinterpret_cast<const float4*>(p.X1 + (size_t)(n0 + r) * p.ldx1 + c);
      if (x2_rows) *reinterpret_cast<float4*>(X2s + r * LDK + c) = *reinterpret_cast<const float4*>(p.X2 + (size_t)(n0 + r) * p.ldx2 + c);
    }
    if (two && p.x2_t) {   // X2 given as [K, N]: rows k, the tile's 32 columns n0.. -> LDS transposed
      for (int e = tid; e < K * 8; e += 256) {
        const int k = e >> 3, c = (e & 7) * 4;
        const float4 q = *reinterpret_cast<const float4*>(p.X2 + (size_t)k * p.ldx2 + n0 + c);
        X2s[(c + 0) * LDK + k] = q.x; X2s[(c + 1) * LDK + k] = q.y; X2s[(c + 2) * LDK + k] = q.z; X2s[(c + 3) * LDK + k] = q.w;
      }
    }
  }
  __syncthreads();
  // this thread's four output rows (accumulator registers 4w .. 4w+3 of the tile: row = (r & 3) + 8*(r >> 2) + 4*lh) and its column
  const int col = n0 + l31;
  int rl[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) rl[u] = ((4 * wave + u) & 3) + 8 * ((4 * wave + u) >> 2) + 4 * lh;
  // the sparse term's gathers, GS shapes x 4 rows per round, the first round in flight under the MFMA phase (one workgroup per CU: the
  // memory-level parallelism has to come from the loads a thread keeps in flight)
  constexpr int GS = 16;
  float gv[GS][4];
  int gr[GS][4];
#define WG_GATHER(b0_)                                                                         \
  _Pragma("unroll") for (int j = 0; j < GS; ++j) _Pragma("unroll") for (int u = 0; u < 4; ++u) { \
    gr[j][u] = ((b0_) + j < p.B) ? sarg[((b0_) + j) * 32 + rl[u]] : -1;                           \
    gv[j][u] = gr[j][u] >= 0 ? p.Bm[(size_t)gr[j][u] * p.ldb + col] : 0.f;                        \
  }
  if (p.sp_val) { WG_GATHER(0) }
  // ---- wave w: k in [w*K/4, (w+1)*K/4); a 16-byte read at k + 4*lh feeds four MFMA steps
  f32x16 acc1, acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc1[r] = acc2[r] = 0.f;
  const int ks = wave * (K / 4), ke = ks + K / 4;
  const float* wp = Ws + l31 * LDK + 4 * lh;
  const float* x1p = X1s + l31 * LDK + 4 * lh;
  const float* x2p = X2s + l31 * LDK + 4 * lh;
  for (int k = ks; k < ke; k += 8) {
    const float4 a = *reinterpret_cast<const float4*>(wp + k), x = *reinterpret_cast<const float4*>(x1p + k);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, x.x, acc1, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, x.y, acc1, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, x.z, acc1, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, x.w, acc1, 0, 0, 0);
    if (two) {
      const float4 y = *reinterpret_cast<const float4*>(x2p + k);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, y.x, acc2, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, y.y, acc2, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, y.z, acc2, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, y.w, acc2, 0, 0, 0);
    }
  }
  __syncthreads();           // the operand tiles are dead: their LDS holds the partial tiles [term][wave][r][lane] (wgrad_front covers them)
  float* red = dyn;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    red[(wave * 16 + r) * 64 + lane] = acc1[r];
    if (two) red[4096 + (wave * 16 + r) * 64 + lane] = acc2[r];
  }
  __syncthreads();
  // ---- epilogue
  const float vcol = p.v1 ? p.v1[col] : 0.f;
  float sc = 1.f, sh = 0.f;
  if (p.p_scale) { sc = p.p_scale[col]; sh = p.p_shift[col]; }
  float o[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int r = 4 * wave + u;
    const int row = a0 + rl[u];
    const float s1 = (red[(0 * 16 + r) * 64 + lane] + red[(1 * 16 + r) * 64 + lane]) + (red[(2 * 16 + r) * 64 + lane] + red[(3 * 16 + r) * 64 + lane]);
    if (p.T) p.T[(size_t)row * p.ldt + col] = s1;
    const float al = p.a1[row];
    float v = al * s1;
    if (p.v1) v = fmaf(fmaf(al, p.b1[row], p.d1[row]), vcol, v);
    if (two) {
      const float s2 = (red[4096 + (0 * 16 + r) * 64 + lane] + red[4096 + (1 * 16 + r) * 64 + lane]) +
                       (red[4096 + (2 * 16 + r) * 64 + lane] + red[4096 + (3 * 16 + r) * 64 + lane]);
      v = fmaf(p.a2[row], s2, v);
    }
    o[u] = v;
  }
  if (p.sp_val) {
    float add[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b0 = 0; b0 < p.B; b0 += GS) {
      if (b0) { WG_GATHER(b0) }
#pragma unroll
      for (int j = 0; j < GS; ++j)       // shapes ascending: a fixed summation order
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float x = gv[j][u];
          if (p.p_scale) x = lrelu_f(fmaf(x, sc, sh), p.p_slope);
          if (gr[j][u] >= 0) add[u] = fmaf(sval[(b0 + j) * 32 + rl[u]], x, add[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) o[u] += add[u];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    float* q = p.out + (size_t)(a0 + rl[u]) * p.ldo + col;
    *q = p.accumulate ? *q + o[u] : o[u];
  }
}

#undef WG_GATHER

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wgrad_collapse_kernel(const spgan_wgrad_collapse_args p) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  wgrad_collapse_body(p, dyn, blockIdx.x, blockIdx.y);
}

// spgan_wgrad_collapse_multi: the same layer's weight gradient for several passes (blockIdx.z; equal C, N, K -- the real and the fake pass and
// phase B of the double backward of a grouped D step) as one launch; every pass runs the stand-alone kernel's body: bit-identical results.
struct WgradMulti {
  spgan_wgrad_collapse_args a[SPGAN_GROUP_MAX];
};
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wgrad_collapse_multi_kernel(const WgradMulti m) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  wgrad_collapse_body(m.a[blockIdx.z], dyn, blockIdx.x, blockIdx.y);
}

}  // namespace

static int wgrad_check(const spgan_wgrad_collapse_args* a) {
  SPGAN_CHECK_ARG(a && a->W && a->X1 && a->a1 && a->out && a->C > 0 && a->N > 0 && a->K >= 32 && a->K <= WG_KMAX);
  SPGAN_CHECK_ARG(a->C % 32 == 0 && a->N % 32 == 0 && a->K % 32 == 0 && a->ldw >= a->K && a->ldx1 >= a->K && a->ldo >= a->N);
  SPGAN_CHECK_ARG((a->ldw % 4 == 0) && (a->ldx1 % 4 == 0) && ((reinterpret_cast<uintptr_t>(a->W) | reinterpret_cast<uintptr_t>(a->X1)) & 15) == 0);
  SPGAN_CHECK_ARG(!a->v1 || (a->b1 && a->d1));
  SPGAN_CHECK_ARG(!a->X2 || (a->a2 && (a->ldx2 % 4 == 0) && (reinterpret_cast<uintptr_t>(a->X2) & 15) == 0 && a->ldx2 >= (a->x2_t ? a->N : a->K)));
  SPGAN_CHECK_ARG(!a->sp_val || (a->sp_arg && a->Bm && a->B > 0 && a->B <= WG_BMAX && a->rows > 0 && a->ldb >= a->N && (!a->p_scale == !a->p_shift)));
  SPGAN_CHECK_ARG(!a->T || a->ldt >= a->N);
  return SPGAN_OK;
}

extern "C" int spgan_wgrad_collapse(const spgan_wgrad_collapse_args* a, spgan_stream_t s_) {
  const int rc = wgrad_check(a);
  if (rc != SPGAN_OK) return rc;
  static LdsOptIn opt;  // > 64 KB of dynamic LDS: once per kernel and device
  opt.ensure(reinterpret_cast<const void*>(&wgrad_collapse_kernel), (int)wgrad_lds_bytes(WG_KMAX, 2));
  const size_t lds = wgrad_lds_bytes(a->K, a->X2 ? 2 : 1);
  hipLaunchKernelGGL(wgrad_collapse_kernel, dim3(a->C / 32, a->N / 32), dim3(256), lds, (hipStream_t)s_, *a);
  return spgan_launch_status();
}

extern "C" int spgan_wgrad_collapse_multi(const spgan_wgrad_collapse_args* a, int count, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(a && count >= 1 && count <= SPGAN_GROUP_MAX);
  if (count == 1) return spgan_wgrad_collapse(a, s_);
  WgradMulti m;
  bool two = false;
  for (int g = 0; g < count; ++g) {
    const int rc = wgrad_check(a + g);
    if (rc != SPGAN_OK) return rc;
    SPGAN_CHECK_ARG(a[g].C == a[0].C && a[g].N == a[0].N && a[g].K == a[0].K);
    m.a[g] = a[g];
    two = two || a[g].X2 != nullptr;
  }
  static LdsOptIn opt;
  opt.ensure(reinterpret_cast<const void*>(&wgrad_collapse_multi_kernel), (int)wgrad_lds_bytes(WG_KMAX, 2));
  const size_t lds = wgrad_lds_bytes(a->K, two ? 2 : 1);
  hipLaunchKernelGGL(wgrad_collapse_multi_kernel, dim3(a->C / 32, a->N / 32, count), dim3(256), lds, (hipStream_t)s_, m);
  return spgan_launch_status();
}

extern "C" int spgan_collapse_prep(const spgan_collapse_prep_args* a, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(a && a->W && a->C > 0 && a->K > 0 && a->C % 256 == 0 && a->K % 32 == 0 && a->ldw >= a->K && a->ldg >= a->K && a->C <= 8192);
  SPGAN_CHECK_ARG(a->nprob >= 1 && a->nprob <= SPGAN_GROUP_MAX && a->nsparse >= 1 && a->nsparse <= SPGAN_GROUP_MAX && a->B > 0 && a->rows > 0 &&
                  a->lde >= a->K);
  int wgs = 0;
  for (int p = 0; p < a->nprob; ++p) {
    SPGAN_CHECK_ARG(a->alpha[p] && a->G[p] && (!a->cvec[p] || (a->beta[p] && a->bias[p])));
    wgs += (a->K / 32) * (a->K / 32 + (a->cvec[p] ? 1 : 0));
  }
  for (int q = 0; q < a->nsparse; ++q) SPGAN_CHECK_ARG(a->sp_val[q] && a->sp_arg[q] && a->E[q]);
  const int RB = sparse_rows_nt_rb(a->rows, a->C);
  const int chunks = cdiv(a->rows, RB);
  size_t lds = sparse_rows_nt_lds(RB, a->C);
  if (lds < WDW_LDS_FLOATS * sizeof(float)) lds = WDW_LDS_FLOATS * sizeof(float);
  hipLaunchKernelGGL(collapse_prep_kernel, dim3(wgs + chunks * a->B * a->nsparse), dim3(256), lds, (hipStream_t)s_, *a, chunks, RB);
  return spgan_launch_status();
}

extern "C" int spgan_wt_diag_w(const float* W, int ldw, int C, int K, const float* alpha, const float* beta, const float* bias, float* G, int ldg,
                               float* cvec, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(W && alpha && G && C > 0 && K > 0 && C % 256 == 0 && K % 32 == 0 && ldw >= K && ldg >= K);
  SPGAN_CHECK_ARG(!cvec || (beta && bias));
  hipLaunchKernelGGL(wt_diag_w_kernel, dim3(K / 32, K / 32 + (cvec ? 1 : 0)), dim3(256), 0, (hipStream_t)s_, W, ldw, C, K, alpha, beta, bias, G, ldg, cvec);
  return spgan_launch_status();
}
