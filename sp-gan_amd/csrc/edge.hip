// EdgeBlock (Generation/Generator.py:47-88) gather-side kernels.
//
// The reference materialises ee = cat[x_i, x_j - x_i] [B,2C,N,k] and runs three per-edge 1x1 convs
// on it.  Here conv_w.0 and conv_x.0 are restructured per POINT:
//     conv_w.0(x_j - x_i) = P_j - P_i + b1,            P = W1 x
//     conv_x.0([x_i, x_j - x_i]) = R_i + Q_j + bx,     Q = Wd x,  R = (Wc - Wd) x,  Wx = [Wc | Wd]
// so one per-point GEMM produces PQR[M, H+2F] (H = F/2) and every per-edge quantity is a gather of
// 128-512 B rows of PQR -- coalesced, L2-resident, no ee tensor.  The kernels below do the per-edge
// elementwise work between the MFMA contractions: train-mode BatchNorm statistics over edges, the
// softmax over the k neighbours times conv_x's activation, and their backward passes.  The backward
// "scatter" to neighbours is a gather over the CSR in-edge lists (deterministic; no float atomics).
//
// Thread mapping everywhere: one wave per point, lanes along channels.
#include <stdlib.h>
#include "common.hpp"

namespace {

// XCD-aware workgroup order for the gather kernels: the hardware deals consecutive workgroup ids round-robin to the 8 XCDs, each with its own
// 4 MB L2.  In launch order, neighbouring point chunks -- which gather the same rows of the [B*N, H+2F] point tensor, 2.6 MB per shape at
// F = 128 -- would sit on eight different L2s and every shape's rows would be fetched eight times.  Logical block xcd * per + t runs on XCD
// xcd: an XCD works through a CONTIGUOUS eighth of the points (4 shapes at B = 32), whose rows it fetches once.  Grids are rounded up to a
// multiple of 8; logical blocks past the end find no points.
// Measured in the replayed step (profiles/r04_step_sequence.txt): edge_stats 65 -> 43-51 us, edge_attend_bwd 280 -> 272, edge_scatter 237 -> 228
// (9.24 -> 9.18-9.20 ms/step on one box); the same order changed nothing for the kNN scan and the per-edge-operand GEMM (not kept there).
__device__ __forceinline__ int xcd_block() {
  const int per = gridDim.x >> 3;
  return (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
}
inline int grid8(long n) { return (int)((n + 7) / 8 * 8); }

// ------------------------------------------------------------------------------------------ weights
__global__ void edge_wcat_kernel(const float* __restrict__ Ww0, const float* __restrict__ Wx, int H, int F, int C, float* __restrict__ Wcat,
                                 float* __restrict__ WcatT) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (H + 2 * F) * C) return;
  const int row = t / C, c = t % C;
  float v;
  if (row < H) v = Ww0[row * C + c];
  else if (row < H + F) v = Wx[(row - H) * 2 * C + C + c];                       // Wd
  else v = Wx[(row - H - F) * 2 * C + c] - Wx[(row - H - F) * 2 * C + C + c];     // Wc - Wd
  Wcat[t] = v;
  if (WcatT) WcatT[c * (H + 2 * F) + row] = v;
}
// conv_out.weight [F,F,1,k] -> Wo [F, k*F] (K index r*F + c) and Wo^T
__global__ void conv_out_weight_pm_kernel(const float* __restrict__ w, int F, int k, float* __restrict__ Wo, float* __restrict__ WoT) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= F * F * k) return;
  const int o = t / (k * F), q = t % (k * F), r = q / F, c = q % F;      // Wo[o, r*F + c] = w[o, c, 0, r]
  const float v = w[((size_t)o * F + c) * k + r];
  Wo[t] = v;
  if (WoT) WoT[(size_t)q * F + o] = v;
}
__global__ void edge_wcat_bwd_kernel(const float* __restrict__ dWcat, int H, int F, int C, float* __restrict__ dWw0, float* __restrict__ dWx) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < H * C) dWw0[t] = dWcat[t];
  if (t < F * C) {
    const int f = t / C, c = t % C;
    const float dq = dWcat[(H + f) * C + c], dr = dWcat[(H + F + f) * C + c];
    dWx[f * 2 * C + c] = dr;            // d/dWc
    dWx[f * 2 * C + C + c] = dq - dr;   // d/dWd
  }
}

// ------------------------------------------------------------------------------------------ statistics over edges
// partials [tiles][H+F][2] = (sum, centred M2) of  h1pre = (P_j - P_i) + b1  (channels [0,H))
//                                             and  ypre  = (R_i + Q_j) + bx  (channels [H,H+F))
constexpr int ES_PT = 32;  // points per workgroup (tile = ES_PT*k edges)

__device__ __forceinline__ float edge_pre(const float* __restrict__ PQR, int ld, int H, int F, int c, int i, int j, float bias) {
  if (c < H) return (PQR[(size_t)j * ld + c] - PQR[(size_t)i * ld + c]) + bias;
  const int f = c - H;
  return (PQR[(size_t)i * ld + H + F + f] + PQR[(size_t)j * ld + H + f]) + bias;
}

// KT > 0: compile-time k -- a point's neighbour indices are wave-uniform (scalar loads), its centre value is read once
// and the k neighbour gathers are all in flight together.
template <int KT>
__global__ __launch_bounds__(256) void edge_stats_kernel(const float* __restrict__ PQR, int ld, const int32_t* __restrict__ idx, int M, int k_,
                                                         int H, int F, const float* __restrict__ b1, const float* __restrict__ bx,
                                                         float* __restrict__ part) {
  constexpr int KU = KT > 0 ? KT : 1;
  const int k = KT > 0 ? KT : k_;
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bxid = xcd_block();
  const int p0 = bxid * ES_PT;
  if (p0 >= M) return;
  const int np = min(ES_PT, M - p0);
  const int CH = H + F;
  const float cnt = (float)(np * k);
  // Single gather pass with shifted sums: d = v - v0 (v0 = the tile's first edge value of this channel), so
  // M2 = sum d^2 - (sum d)^2/n is free of catastrophic cancellation (|d| is of the order of the std).
  __shared__ float red2[4][64];
  for (int c0 = 0; c0 < CH; c0 += 64) {
    const int c = c0 + lane;
    const bool ok = c < CH;
    const float bias = ok ? (c < H ? b1[c] : bx[c - H]) : 0.f;
    float s1 = 0.f, s2 = 0.f, v0 = 0.f;
    if (ok) {
      v0 = edge_pre(PQR, ld, H, F, c, p0, idx[(size_t)p0 * k], bias);
      if (KT > 0) {
        const int colI = c < H ? c : c + F, colJ = c;  // centre column: P | R, neighbour column: P | Q
        const float sgn = c < H ? -1.f : 1.f;
        for (int p = w; p < np; p += 4) {
          const int i = p0 + p;
          int jn[KU];
#pragma unroll
          for (int r = 0; r < KU; ++r) jn[r] = idx[(size_t)i * KT + r];
          const float vi = sgn * PQR[(size_t)i * ld + colI];
          float vj[KU];
#pragma unroll
          for (int r = 0; r < KU; ++r) vj[r] = PQR[(size_t)jn[r] * ld + colJ];
#pragma unroll
          for (int r = 0; r < KU; ++r) {
            const float d = ((c < H ? vj[r] + vi : vi + vj[r]) + bias) - v0;  // (P_j - P_i) + b1  |  (R_i + Q_j) + bx
            s1 += d;
            s2 = fmaf(d, d, s2);
          }
        }
      } else {
        for (int p = w; p < np; p += 4) {
          const int i = p0 + p;
          for (int r = 0; r < k; ++r) {
            const float d = edge_pre(PQR, ld, H, F, c, i, idx[(size_t)i * k + r], bias) - v0;
            s1 += d;
            s2 = fmaf(d, d, s2);
          }
        }
      }
    }
    red[w][lane] = s1;
    red2[w][lane] = s2;
    __syncthreads();
    if (w == 0 && ok) {
      const float t1 = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
      const float t2 = (red2[0][lane] + red2[1][lane]) + (red2[2][lane] + red2[3][lane]);
      float* o = part + ((size_t)bxid * CH + c) * 2;
      o[0] = fmaf(cnt, v0, t1);
      o[1] = fmaxf(t2 - t1 * t1 / cnt, 0.f);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------ attention over the k neighbours
// T[i, r*F + f] = softmax_r( lrelu(bn2(h2pre[i,r,f])) ) * lrelu(bnx(ypre[i,r,f]))     Generator.py:79,81-82
constexpr int EA_PT = 16;  // points per workgroup: 4 waves x 4 points

__global__ __launch_bounds__(256) void edge_attend_fwd_kernel(const float* __restrict__ h2, const float* __restrict__ sc2,
                                                              const float* __restrict__ sh2, const float* __restrict__ PQR, int ld, int H,
                                                              int F, const int32_t* __restrict__ idx, int M, int k,
                                                              const float* __restrict__ bx, const float* __restrict__ scx,
                                                              const float* __restrict__ shx, float slope, float* __restrict__ T) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int pp = 0; pp < EA_PT / 4; ++pp) {
    const int i = blockIdx.x * EA_PT + pp * 4 + w;
    if (i >= M) return;
    const float* h2i = h2 + (size_t)i * k * F;
    for (int f = lane; f < F; f += 64) {
      const float a2 = sc2[f], c2 = sh2[f], ax = scx[f], cx = shx[f], bb = bx[f];
      const float Ri = PQR[(size_t)i * ld + H + F + f];
      float mx = -INFINITY;
      for (int r = 0; r < k; ++r) mx = fmaxf(mx, lrelu_f(fmaf(h2i[(size_t)r * F + f], a2, c2), slope));
      float den = 0.f;
      for (int r = 0; r < k; ++r) den += expf(lrelu_f(fmaf(h2i[(size_t)r * F + f], a2, c2), slope) - mx);
      const float rden = 1.0f / den;
      for (int r = 0; r < k; ++r) {
        const float wgt = expf(lrelu_f(fmaf(h2i[(size_t)r * F + f], a2, c2), slope) - mx) * rden;
        const int j = idx[(size_t)i * k + r];
        const float yv = lrelu_f(fmaf((Ri + PQR[(size_t)j * ld + H + f]) + bb, ax, cx), slope);
        T[((size_t)i * k + r) * F + f] = yv * wgt;
      }
    }
  }
}

// Backward of the above: given dT, produce the gradients w.r.t. the two BatchNorm OUTPUTS (pre-LeakyReLU)
//   g2[e,f] (conv_w.4 branch) and gy[e,f] (conv_x.1 branch), plus the column sums BatchNorm backward needs:
//   partials [tiles][2F][2]: column f      -> (sum g2, sum g2*xhat2)
//                            column F + f  -> (sum gy, sum gy*xhaty)
constexpr int EB_PT = 32;  // points per workgroup: 4 waves x 8 points

__global__ __launch_bounds__(256) void edge_attend_bwd_kernel(
    const float* __restrict__ dT, const float* __restrict__ h2, const float* __restrict__ sc2, const float* __restrict__ sh2,
    const float* __restrict__ mean2, const float* __restrict__ inv2, const float* __restrict__ PQR, int ld, int H, int F,
    const int32_t* __restrict__ idx, int M, int k, const float* __restrict__ bx, const float* __restrict__ scx,
    const float* __restrict__ shx, const float* __restrict__ meanx, const float* __restrict__ invx, float slope, float* __restrict__ g2,
    float* __restrict__ gy, float* __restrict__ part) {
  __shared__ float red[4][4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int f0 = 0; f0 < F; f0 += 64) {
    const int f = f0 + lane;
    const bool ok = f < F;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (ok) {
      const float a2 = sc2[f], c2 = sh2[f], ax = scx[f], cx = shx[f], bb = bx[f];
      const float m2 = mean2[f], i2 = inv2[f], mxm = meanx[f], ixv = invx[f];
      for (int pp = 0; pp < EB_PT / 4; ++pp) {
        const int i = blockIdx.x * EB_PT + pp * 4 + w;
        if (i >= M) break;
        const float* h2i = h2 + (size_t)i * k * F;
        const float* dTi = dT + (size_t)i * k * F;
        const float Ri = PQR[(size_t)i * ld + H + F + f];
        float mx = -INFINITY;
        for (int r = 0; r < k; ++r) mx = fmaxf(mx, lrelu_f(fmaf(h2i[(size_t)r * F + f], a2, c2), slope));
        float den = 0.f;
        for (int r = 0; r < k; ++r) den += expf(lrelu_f(fmaf(h2i[(size_t)r * F + f], a2, c2), slope) - mx);
        const float rden = 1.0f / den;
        float dot = 0.f;  // sum_r dw[r]*w[r]
        for (int r = 0; r < k; ++r) {
          const float wgt = expf(lrelu_f(fmaf(h2i[(size_t)r * F + f], a2, c2), slope) - mx) * rden;
          const int j = idx[(size_t)i * k + r];
          const float yv = lrelu_f(fmaf((Ri + PQR[(size_t)j * ld + H + f]) + bb, ax, cx), slope);
          dot = fmaf(dTi[(size_t)r * F + f] * yv, wgt, dot);
        }
        for (int r = 0; r < k; ++r) {
          const float hp = h2i[(size_t)r * F + f];
          const float z2 = fmaf(hp, a2, c2);
          const float wgt = expf(lrelu_f(z2, slope) - mx) * rden;
          const int j = idx[(size_t)i * k + r];
          const float yp = (Ri + PQR[(size_t)j * ld + H + f]) + bb;
          const float zy = fmaf(yp, ax, cx);
          const float yv = lrelu_f(zy, slope);
          const float d = dTi[(size_t)r * F + f];
          const float ds = wgt * (d * yv - dot);            // softmax backward
          const float o2 = ds * lrelu_mask(z2, slope);
          const float oy = d * wgt * lrelu_mask(zy, slope);
          g2[((size_t)i * k + r) * F + f] = o2;
          gy[((size_t)i * k + r) * F + f] = oy;
          s0 += o2;
          s1 = fmaf(o2, (hp - m2) * i2, s1);
          s2 += oy;
          s3 = fmaf(oy, (yp - mxm) * ixv, s3);
        }
      }
    }
    red[0][w][lane] = s0; red[1][w][lane] = s1; red[2][w][lane] = s2; red[3][w][lane] = s3;
    __syncthreads();
    if (w == 0 && ok) {
      float* o = part + (size_t)blockIdx.x * (2 * F) * 2;
      o[(size_t)f * 2 + 0] = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
      o[(size_t)f * 2 + 1] = (red[2][0][lane] + red[2][1][lane]) + (red[2][2][lane] + red[2][3][lane]);
      o[(size_t)(F + f) * 2 + 0] = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
      o[(size_t)(F + f) * 2 + 1] = (red[3][0][lane] + red[3][1][lane]) + (red[3][2][lane] + red[3][3][lane]);
    }
    __syncthreads();
  }
}

// ---- register-resident variants for compile-time k (k = 10 is the reference default, nk//2, Generator.py:96) ----
// One wave per point; a lane owns VEC consecutive channels and keeps all k pre-activations in registers, so
// h2pre / dT are read exactly once (the generic kernels above re-read them for max, sum and output passes).
template <int VEC> struct VecT;
template <> struct VecT<1> { typedef float T; };
template <> struct VecT<2> { typedef float2 T; };
template <> struct VecT<4> { typedef float4 T; };

template <int VEC>
__device__ __forceinline__ void ldv(const float* p, float (&o)[VEC]) {
  const typename VecT<VEC>::T v = *reinterpret_cast<const typename VecT<VEC>::T*>(p);
  const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
  for (int q = 0; q < VEC; ++q) o[q] = f[q];
}
template <int VEC>
__device__ __forceinline__ void stv(float* p, const float (&o)[VEC]) {
  typename VecT<VEC>::T v;
  float* f = reinterpret_cast<float*>(&v);
#pragma unroll
  for (int q = 0; q < VEC; ++q) f[q] = o[q];
  *reinterpret_cast<typename VecT<VEC>::T*>(p) = v;
}

// 16-bit storage of the two tensors only GEMMs consume / produce (spgan_edge_attend_fwd_h / _bwd_b; the "f16" operand mode):
// T is written as fp16 (an activation of magnitude O(1): what the fp16 MFMA would round it to anyway), dT is read as bfloat16
// (a gradient: needs fp32's exponent range).
template <int VEC>
__device__ __forceinline__ void stv_h(_Float16* p, const float (&o)[VEC]) {
  _Float16 h[VEC];
#pragma unroll
  for (int q = 0; q < VEC; ++q) h[q] = (_Float16)o[q];
  if (VEC == 2) *reinterpret_cast<unsigned*>(p) = *reinterpret_cast<const unsigned*>(h);
  else p[0] = h[0];
}
template <int VEC>
__device__ __forceinline__ void ldv_b(const __bf16* p, float (&o)[VEC]) {
  __bf16 h[VEC];
  if (VEC == 2) *reinterpret_cast<unsigned*>(h) = *reinterpret_cast<const unsigned*>(p);
  else h[0] = p[0];
#pragma unroll
  for (int q = 0; q < VEC; ++q) o[q] = (float)h[q];
}

template <int VEC>
__device__ __forceinline__ void ldv_h(const _Float16* p, float (&o)[VEC]) {
  _Float16 h[VEC];
  if (VEC == 2) *reinterpret_cast<unsigned*>(h) = *reinterpret_cast<const unsigned*>(p);
  else h[0] = p[0];
#pragma unroll
  for (int q = 0; q < VEC; ++q) o[q] = (float)h[q];
}
template <int VEC>
__device__ __forceinline__ void stv_b(__bf16* p, const float (&o)[VEC]) {
  __bf16 h[VEC];
#pragma unroll
  for (int q = 0; q < VEC; ++q) h[q] = (__bf16)o[q];
  if (VEC == 2) *reinterpret_cast<unsigned*>(p) = *reinterpret_cast<const unsigned*>(h);
  else p[0] = h[0];
}

// HH = 1: h2pre lies in memory as fp16 (written by the edge GEMM with spgan_gemm_nt_args.y_half)
template <int K, int VEC, int TH = 0, int HH = 0>
__global__ __launch_bounds__(256) void edge_attend_fwd_k_kernel(const float* __restrict__ h2, const float* __restrict__ sc2,
                                                                const float* __restrict__ sh2, const float* __restrict__ PQR, int ld, int H,
                                                                int F, const int32_t* __restrict__ idx, int M, const float* __restrict__ bx,
                                                                const float* __restrict__ scx, const float* __restrict__ shx, float slope,
                                                                float* __restrict__ T) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = __builtin_amdgcn_readfirstlane(xcd_block() * 4 + w);
  if (i >= M) return;
  int nb[K];
#pragma unroll
  for (int r = 0; r < K; ++r) nb[r] = __builtin_amdgcn_readfirstlane(idx[(size_t)i * K + r]);
  const float* h2i = h2 + (size_t)i * K * F;
  const _Float16* h2h = reinterpret_cast<const _Float16*>(h2) + (size_t)i * K * F;
  for (int f = lane * VEC; f < F; f += 64 * VEC) {
    float a2[VEC], c2[VEC], ax[VEC], cx[VEC], bb[VEC], Ri[VEC];
    ldv<VEC>(sc2 + f, a2); ldv<VEC>(sh2 + f, c2); ldv<VEC>(scx + f, ax); ldv<VEC>(shx + f, cx); ldv<VEC>(bx + f, bb);
    ldv<VEC>(PQR + (size_t)i * ld + H + F + f, Ri);
    float z[K][VEC], qv[K][VEC];
#pragma unroll
    for (int r = 0; r < K; ++r) {
      if (HH) ldv_h<VEC>(h2h + (size_t)r * F + f, z[r]);
      else ldv<VEC>(h2i + (size_t)r * F + f, z[r]);
      ldv<VEC>(PQR + (size_t)nb[r] * ld + H + f, qv[r]);
    }
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < K; ++r) {
        z[r][q] = lrelu_f(fmaf(z[r][q], a2[q], c2[q]), slope);
        mx = fmaxf(mx, z[r][q]);
      }
      float den = 0.f;
#pragma unroll
      for (int r = 0; r < K; ++r) {
        z[r][q] = expf(z[r][q] - mx);
        den += z[r][q];
      }
      const float rden = 1.0f / den;
#pragma unroll
      for (int r = 0; r < K; ++r) {
        const float yv = lrelu_f(fmaf((Ri[q] + qv[r][q]) + bb[q], ax[q], cx[q]), slope);
        z[r][q] = yv * (z[r][q] * rden);
      }
    }
#pragma unroll
    for (int r = 0; r < K; ++r) {
      if (TH) stv_h<VEC>(reinterpret_cast<_Float16*>(T) + ((size_t)i * K + r) * F + f, z[r]);
      else stv<VEC>(T + ((size_t)i * K + r) * F + f, z[r]);
    }
  }
}

template <int K, int VEC, int TB = 0, int HH = 0>
__global__ __launch_bounds__(256) void edge_attend_bwd_k_kernel(
    const float* __restrict__ dT, const float* __restrict__ h2, const float* __restrict__ sc2, const float* __restrict__ sh2,
    const float* __restrict__ mean2, const float* __restrict__ inv2, const float* __restrict__ PQR, int ld, int H, int F,
    const int32_t* __restrict__ idx, int M, const float* __restrict__ bx, const float* __restrict__ scx, const float* __restrict__ shx,
    const float* __restrict__ meanx, const float* __restrict__ invx, float slope, float* __restrict__ g2, float* __restrict__ gy,
    float* __restrict__ part) {
  // EB_PT points per workgroup (4 waves x EB_PT/4 points) so the partial format matches the generic kernel
  __shared__ float red[4][4][64 * VEC];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int bxid = xcd_block();
  if (bxid * EB_PT >= M) return;
  for (int f0 = 0; f0 < F; f0 += 64 * VEC) {
    const int f = f0 + lane * VEC;
    const bool ok = f < F;
    float s0[VEC], s1[VEC], s2[VEC], s3[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) s0[q] = s1[q] = s2[q] = s3[q] = 0.f;
    if (ok) {
      float a2[VEC], c2[VEC], ax[VEC], cx[VEC], bb[VEC], m2[VEC], i2[VEC], mxm[VEC], ixv[VEC];
      ldv<VEC>(sc2 + f, a2); ldv<VEC>(sh2 + f, c2); ldv<VEC>(scx + f, ax); ldv<VEC>(shx + f, cx); ldv<VEC>(bx + f, bb);
      ldv<VEC>(mean2 + f, m2); ldv<VEC>(inv2 + f, i2); ldv<VEC>(meanx + f, mxm); ldv<VEC>(invx + f, ixv);
      for (int pp = 0; pp < EB_PT / 4; ++pp) {
        const int i = __builtin_amdgcn_readfirstlane(bxid * EB_PT + pp * 4 + w);
        if (i >= M) break;
        float Ri[VEC];
        ldv<VEC>(PQR + (size_t)i * ld + H + F + f, Ri);
        float hp[K][VEC], d[K][VEC], yp[K][VEC];
#pragma unroll
        for (int r = 0; r < K; ++r) {
          const int j = __builtin_amdgcn_readfirstlane(idx[(size_t)i * K + r]);
          if (HH) ldv_h<VEC>(reinterpret_cast<const _Float16*>(h2) + ((size_t)i * K + r) * F + f, hp[r]);
          else ldv<VEC>(h2 + ((size_t)i * K + r) * F + f, hp[r]);
          if (TB) ldv_b<VEC>(reinterpret_cast<const __bf16*>(dT) + ((size_t)i * K + r) * F + f, d[r]);
          else ldv<VEC>(dT + ((size_t)i * K + r) * F + f, d[r]);
          ldv<VEC>(PQR + (size_t)j * ld + H + f, yp[r]);
        }
        float o2[K][VEC], oy[K][VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
          float e[K], mx = -INFINITY;
#pragma unroll
          for (int r = 0; r < K; ++r) {
            e[r] = lrelu_f(fmaf(hp[r][q], a2[q], c2[q]), slope);
            mx = fmaxf(mx, e[r]);
          }
          float den = 0.f;
#pragma unroll
          for (int r = 0; r < K; ++r) {
            e[r] = expf(e[r] - mx);
            den += e[r];
          }
          const float rden = 1.0f / den;
          float dot = 0.f;
#pragma unroll
          for (int r = 0; r < K; ++r) {
            yp[r][q] = (Ri[q] + yp[r][q]) + bb[q];
            e[r] *= rden;
            const float yv = lrelu_f(fmaf(yp[r][q], ax[q], cx[q]), slope);
            dot = fmaf(d[r][q] * yv, e[r], dot);
          }
#pragma unroll
          for (int r = 0; r < K; ++r) {
            const float z2 = fmaf(hp[r][q], a2[q], c2[q]);
            const float zy = fmaf(yp[r][q], ax[q], cx[q]);
            const float yv = lrelu_f(zy, slope);
            const float ds = e[r] * (d[r][q] * yv - dot);
            const float v2 = ds * lrelu_mask(z2, slope);
            const float vy = d[r][q] * e[r] * lrelu_mask(zy, slope);
            o2[r][q] = v2; oy[r][q] = vy;
            s0[q] += v2;
            s1[q] = fmaf(v2, (hp[r][q] - m2[q]) * i2[q], s1[q]);
            s2[q] += vy;
            s3[q] = fmaf(vy, (yp[r][q] - mxm[q]) * ixv[q], s3[q]);
          }
        }
#pragma unroll
        for (int r = 0; r < K; ++r) {
          if (TB) stv_b<VEC>(reinterpret_cast<__bf16*>(g2) + ((size_t)i * K + r) * F + f, o2[r]);   // consumed as a GEMM operand only
          else stv<VEC>(g2 + ((size_t)i * K + r) * F + f, o2[r]);
          if (TB) stv_b<VEC>(reinterpret_cast<__bf16*>(gy) + ((size_t)i * K + r) * F + f, oy[r]);   // consumed by edge_scatter only
          else stv<VEC>(gy + ((size_t)i * K + r) * F + f, oy[r]);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      red[0][w][lane * VEC + q] = s0[q]; red[1][w][lane * VEC + q] = s1[q];
      red[2][w][lane * VEC + q] = s2[q]; red[3][w][lane * VEC + q] = s3[q];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 64 * VEC; c += 256) {
      const int ff = f0 + c;
      if (ff < F) {
        float* o = part + (size_t)bxid * (2 * F) * 2;
        o[(size_t)ff * 2 + 0] = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
        o[(size_t)ff * 2 + 1] = (red[2][0][c] + red[2][1][c]) + (red[2][2][c] + red[2][3][c]);
        o[(size_t)(F + ff) * 2 + 0] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
        o[(size_t)(F + ff) * 2 + 1] = (red[3][0][c] + red[3][1][c]) + (red[3][2][c] + red[3][3][c]);
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------ edge -> point gradients
// BatchNorm backward (train mode) of the two per-edge pre-activations fused with the gather-style
// reduction onto points:
//   dh1[e,c] = gam1*inv1*(g1[e,c] - S1[c]/E - xhat1[e,c]*S1x[c]/E)        xhat1 from (P_j - P_i) + b1
//   dyp[e,f] = gamx*invx*(gy[e,f] - Sy[f]/E - xhaty[e,f]*Syx[f]/E)        xhaty from (R_i + Q_j) + bx
//   dP[j] = sum_{e in in(j)} dh1[e] - sum_r dh1[(j,r)];  dQ[j] = sum_{e in in(j)} dyp[e];  dR[i] = sum_r dyp[(i,r)]
// One wave per point, lanes over channels.  The loads of a point's k out-edges (KT > 0: compile-time k) and of its
// in-edges (chunks of 4) are issued together before they are consumed -- a serial edge loop leaves one row in flight per
// wave and ran at 1.7 TB/s; sums are still taken in edge order (deterministic).
// In-edges whose rows are in flight together.  The kernel's time is the dependent chain source list -> gathered rows, not its bytes (a
// variant that read 35 % fewer bytes ran no faster: DESIGN.md section 10.4): chunks of 4 left ~3 serial round trips per point and channel
// pass (mean in-degree k = 10); 16 covers almost every point in one.  Measured in the replayed step: 239 -> 206 us (4 -> 16), sums still
// taken in edge order (bit-identical).  72 VGPRs, 7 waves per SIMD.
#ifndef SPGAN_SCATTER_CHUNK
#define SPGAN_SCATTER_CHUNK 16
#endif
constexpr int SCH = SPGAN_SCATTER_CHUNK;
template <int KT, int GB = 0>
__global__ __launch_bounds__(256) void edge_scatter_kernel(
    const float* __restrict__ g1, const float* __restrict__ gy, const float* __restrict__ PQR, int ld, int H, int F,
    const int32_t* __restrict__ idx, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src, int M, int k_,
    const float* __restrict__ b1, const float* __restrict__ mean1, const float* __restrict__ inv1, const float* __restrict__ gam1,
    const float* __restrict__ sums1, const float* __restrict__ bx, const float* __restrict__ meanx, const float* __restrict__ invx,
    const float* __restrict__ gamx, const float* __restrict__ sumsx, float rE, float* __restrict__ dPQR) {
  constexpr int KU = KT > 0 ? KT : 1;
  const int k = KT > 0 ? KT : k_;
  const int lane = threadIdx.x & 63;
  const int j = __builtin_amdgcn_readfirstlane(xcd_block() * 4 + (threadIdx.x >> 6));  // wave-uniform: scalar index loads
  if (j >= M) return;
  auto ldgy = [&](size_t off) -> float {   // GB: gy lies in memory as bfloat16 (written by spgan_edge_attend_bwd_b)
    return GB ? (float)reinterpret_cast<const __bf16*>(gy)[off] : gy[off];
  };
  const int t0 = rowptr[j], t1 = rowptr[j + 1];
  int jo[KU];  // out-edge neighbours
  if (KT > 0) {
#pragma unroll
    for (int r = 0; r < KU; ++r) jo[r] = idx[j * KT + r];
  }
  // channel c of segment `seg` (0: P/g1, width H; 1: Q,R/gy, width F)
  for (int c = lane; c < H; c += 64) {
    const float bb = b1[c], mu = mean1[c], iv = inv1[c];
    const float coef = gam1[c] * iv, a0 = sums1[c] * rE, a1 = sums1[H + c] * rE;
    const float Pj = PQR[(size_t)j * ld + c];
    float acc = 0.f;
    if (KT > 0) {
      float pv[KU], gv[KU];
#pragma unroll
      for (int r = 0; r < KU; ++r) {
        pv[r] = PQR[(size_t)jo[r] * ld + c];
        gv[r] = g1[(size_t)(j * KT + r) * H + c];
      }
#pragma unroll
      for (int r = 0; r < KU; ++r) {
        const float xh = (((pv[r] - Pj) + bb) - mu) * iv;
        acc -= coef * (gv[r] - a0 - xh * a1);
      }
    } else {
      for (int r = 0; r < k; ++r) {  // out-edges: j is the centre
        const int e = j * k + r;
        const float xh = (((PQR[(size_t)idx[e] * ld + c] - Pj) + bb) - mu) * iv;
        acc -= coef * (g1[(size_t)e * H + c] - a0 - xh * a1);
      }
    }
    for (int t = t0; t < t1; t += SCH) {  // in-edges: j is the neighbour
      float pv[SCH], gv[SCH];
#pragma unroll
      for (int u = 0; u < SCH; ++u) {
        const int e = src[min(t + u, t1 - 1)];
        pv[u] = PQR[(size_t)(e / k) * ld + c];
        gv[u] = g1[(size_t)e * H + c];
      }
#pragma unroll
      for (int u = 0; u < SCH; ++u) {
        if (t + u < t1) {
          const float xh = (((Pj - pv[u]) + bb) - mu) * iv;
          acc += coef * (gv[u] - a0 - xh * a1);
        }
      }
    }
    dPQR[(size_t)j * ld + c] = acc;
  }
  for (int f = lane; f < F; f += 64) {
    const float bb = bx[f], mu = meanx[f], iv = invx[f];
    const float coef = gamx[f] * iv, a0 = sumsx[f] * rE, a1 = sumsx[F + f] * rE;
    const float Rj = PQR[(size_t)j * ld + H + F + f], Qj = PQR[(size_t)j * ld + H + f];
    float accR = 0.f, accQ = 0.f;
    if (KT > 0) {
      float qv[KU], gv[KU];
#pragma unroll
      for (int r = 0; r < KU; ++r) {
        qv[r] = PQR[(size_t)jo[r] * ld + H + f];
        gv[r] = ldgy((size_t)(j * KT + r) * F + f);
      }
#pragma unroll
      for (int r = 0; r < KU; ++r) {
        const float xh = (((Rj + qv[r]) + bb) - mu) * iv;
        accR += coef * (gv[r] - a0 - xh * a1);
      }
    } else {
      for (int r = 0; r < k; ++r) {
        const int e = j * k + r;
        const float xh = (((Rj + PQR[(size_t)idx[e] * ld + H + f]) + bb) - mu) * iv;
        accR += coef * (ldgy((size_t)e * F + f) - a0 - xh * a1);
      }
    }
    for (int t = t0; t < t1; t += SCH) {
      float rv[SCH], gv[SCH];
#pragma unroll
      for (int u = 0; u < SCH; ++u) {
        const int e = src[min(t + u, t1 - 1)];
        rv[u] = PQR[(size_t)(e / k) * ld + H + F + f];
        gv[u] = ldgy((size_t)e * F + f);
      }
#pragma unroll
      for (int u = 0; u < SCH; ++u) {
        if (t + u < t1) {
          const float xh = (((rv[u] + Qj) + bb) - mu) * iv;
          accQ += coef * (gv[u] - a0 - xh * a1);
        }
      }
    }
    dPQR[(size_t)j * ld + H + f] = accQ;
    dPQR[(size_t)j * ld + H + F + f] = accR;
  }
}

}  // namespace

extern "C" int spgan_edge_wcat(const float* Ww0, const float* Wx, int H, int F, int C, float* Wcat, float* WcatT, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(Ww0 && Wx && Wcat && H > 0 && F > 0 && C > 0);
  hipLaunchKernelGGL(edge_wcat_kernel, dim3(cdiv((H + 2 * F) * C, 256)), dim3(256), 0, (hipStream_t)s_, Ww0, Wx, H, F, C, Wcat, WcatT);
  return spgan_launch_status();
}
extern "C" int spgan_conv_out_weight_pm(const float* w, int F, int k, float* Wo, float* WoT, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(w && Wo && F > 0 && k > 0);
  hipLaunchKernelGGL(conv_out_weight_pm_kernel, dim3(cdiv((long)F * F * k, 256)), dim3(256), 0, (hipStream_t)s_, w, F, k, Wo, WoT);
  return spgan_launch_status();
}
extern "C" int spgan_edge_wcat_bwd(const float* dWcat, int H, int F, int C, float* dWw0, float* dWx, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(dWcat && dWw0 && dWx && H > 0 && F > 0 && C > 0);
  hipLaunchKernelGGL(edge_wcat_bwd_kernel, dim3(cdiv((H > F ? H : F) * C, 256)), dim3(256), 0, (hipStream_t)s_, dWcat, H, F, C, dWw0, dWx);
  return spgan_launch_status();
}

extern "C" int spgan_edge_stats_tile_rows(int k) { return ES_PT * k; }
extern "C" int spgan_edge_attend_bwd_tile_points(void) { return EB_PT; }

extern "C" int spgan_edge_stats(const float* PQR, int ld, const int32_t* idx, int M, int k, int H, int F, const float* b1, const float* bx,
                                float* partials, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(PQR && idx && b1 && bx && partials && M > 0 && k > 0 && H > 0 && F > 0 && ld >= H + 2 * F);
  if (k == 10) hipLaunchKernelGGL(edge_stats_kernel<10>, dim3(grid8(cdiv(M, ES_PT))), dim3(256), 0, (hipStream_t)s_, PQR, ld, idx, M, k, H, F, b1, bx, partials);
  else hipLaunchKernelGGL(edge_stats_kernel<0>, dim3(grid8(cdiv(M, ES_PT))), dim3(256), 0, (hipStream_t)s_, PQR, ld, idx, M, k, H, F, b1, bx, partials);
  return spgan_launch_status();
}

extern "C" int spgan_edge_attend_fwd(const float* h2pre, const float* sc2, const float* sh2, const float* PQR, int ld, int H, int F,
                                     const int32_t* idx, int M, int k, const float* bx, const float* scx, const float* shx, float slope,
                                     float* T, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(h2pre && sc2 && sh2 && PQR && idx && bx && scx && shx && T && M > 0 && k > 0 && ld >= H + 2 * F);
  const bool al = ((ld | H | F) % 2 == 0);
  if (k == 10 && al && F % 128 == 0)
    hipLaunchKernelGGL((edge_attend_fwd_k_kernel<10, 2>), dim3(grid8(cdiv(M, 4))), dim3(256), 0, (hipStream_t)s_, h2pre, sc2, sh2, PQR, ld, H, F, idx, M, bx,
                       scx, shx, slope, T);
  else if (k == 10)
    hipLaunchKernelGGL((edge_attend_fwd_k_kernel<10, 1>), dim3(grid8(cdiv(M, 4))), dim3(256), 0, (hipStream_t)s_, h2pre, sc2, sh2, PQR, ld, H, F, idx, M, bx,
                       scx, shx, slope, T);
  else
    hipLaunchKernelGGL(edge_attend_fwd_kernel, dim3(cdiv(M, EA_PT)), dim3(256), 0, (hipStream_t)s_, h2pre, sc2, sh2, PQR, ld, H, F, idx, M, k,
                       bx, scx, shx, slope, T);
  return spgan_launch_status();
}

extern "C" int spgan_edge_attend_fwd_h(const void* h2pre, int h2_half, const float* sc2, const float* sh2, const float* PQR, int ld, int H, int F,
                                       const int32_t* idx, int M, int k, const float* bx, const float* scx, const float* shx, float slope,
                                       uint16_t* T_f16, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(h2pre && sc2 && sh2 && PQR && idx && bx && scx && shx && T_f16 && M > 0 && ld >= H + 2 * F);
  SPGAN_CHECK_ARG(k == 10 && F % 4 == 0);   // the k = 10 kernels only; rows of T stay 8-byte aligned for the consumers' loads
  const float* h2 = reinterpret_cast<const float*>(h2pre);
  float* T = reinterpret_cast<float*>(T_f16);
  const dim3 g(grid8(cdiv(M, 4))), b(256);
  hipStream_t s = (hipStream_t)s_;
  const bool v2 = ((ld | H) % 2 == 0) && F % 128 == 0;
  if (v2 && h2_half) hipLaunchKernelGGL((edge_attend_fwd_k_kernel<10, 2, 1, 1>), g, b, 0, s, h2, sc2, sh2, PQR, ld, H, F, idx, M, bx, scx, shx, slope, T);
  else if (v2) hipLaunchKernelGGL((edge_attend_fwd_k_kernel<10, 2, 1, 0>), g, b, 0, s, h2, sc2, sh2, PQR, ld, H, F, idx, M, bx, scx, shx, slope, T);
  else if (h2_half) hipLaunchKernelGGL((edge_attend_fwd_k_kernel<10, 1, 1, 1>), g, b, 0, s, h2, sc2, sh2, PQR, ld, H, F, idx, M, bx, scx, shx, slope, T);
  else hipLaunchKernelGGL((edge_attend_fwd_k_kernel<10, 1, 1, 0>), g, b, 0, s, h2, sc2, sh2, PQR, ld, H, F, idx, M, bx, scx, shx, slope, T);
  return spgan_launch_status();
}

extern "C" int spgan_edge_attend_bwd_b(const uint16_t* dT_bf16, const void* h2pre, int h2_half, const float* sc2, const float* sh2,
                                       const float* mean2, const float* inv2, const float* PQR, int ld, int H, int F, const int32_t* idx, int M,
                                       int k, const float* bx, const float* scx, const float* shx, const float* meanx, const float* invx,
                                       float slope, uint16_t* g2_bf16, uint16_t* gy_bf16, float* partials, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(dT_bf16 && h2pre && sc2 && sh2 && mean2 && inv2 && PQR && idx && bx && scx && shx && meanx && invx && g2_bf16 && gy_bf16 && partials);
  SPGAN_CHECK_ARG(M > 0 && ld >= H + 2 * F && k == 10 && F % 4 == 0);
  const float* dT = reinterpret_cast<const float*>(dT_bf16);
  const float* h2 = reinterpret_cast<const float*>(h2pre);
  float* g2 = reinterpret_cast<float*>(g2_bf16);
  float* gy = reinterpret_cast<float*>(gy_bf16);
  const dim3 g(grid8(cdiv(M, EB_PT))), b(256);
  hipStream_t s = (hipStream_t)s_;
  const bool v2 = ((ld | H) % 2 == 0) && F % 128 == 0;
#define SPGAN_ATT_BWD(V, HHV)                                                                                                               \
  hipLaunchKernelGGL((edge_attend_bwd_k_kernel<10, V, 1, HHV>), g, b, 0, s, dT, h2, sc2, sh2, mean2, inv2, PQR, ld, H, F, idx, M, bx, scx, shx, \
                     meanx, invx, slope, g2, gy, partials)
  if (v2 && h2_half) SPGAN_ATT_BWD(2, 1);
  else if (v2) SPGAN_ATT_BWD(2, 0);
  else if (h2_half) SPGAN_ATT_BWD(1, 1);
  else SPGAN_ATT_BWD(1, 0);
#undef SPGAN_ATT_BWD
  return spgan_launch_status();
}

extern "C" int spgan_edge_attend_bwd(const float* dT, const float* h2pre, const float* sc2, const float* sh2, const float* mean2,
                                     const float* inv2, const float* PQR, int ld, int H, int F, const int32_t* idx, int M, int k,
                                     const float* bx, const float* scx, const float* shx, const float* meanx, const float* invx,
                                     float slope, float* g2, float* gy, float* partials, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(dT && h2pre && sc2 && sh2 && mean2 && inv2 && PQR && idx && bx && scx && shx && meanx && invx && g2 && gy && partials);
  SPGAN_CHECK_ARG(M > 0 && k > 0 && ld >= H + 2 * F);
  const bool al = ((ld | H | F) % 2 == 0);
  if (k == 10 && al && F % 128 == 0)
    hipLaunchKernelGGL((edge_attend_bwd_k_kernel<10, 2>), dim3(grid8(cdiv(M, EB_PT))), dim3(256), 0, (hipStream_t)s_, dT, h2pre, sc2, sh2, mean2, inv2, PQR,
                       ld, H, F, idx, M, bx, scx, shx, meanx, invx, slope, g2, gy, partials);
  else if (k == 10)
    hipLaunchKernelGGL((edge_attend_bwd_k_kernel<10, 1>), dim3(grid8(cdiv(M, EB_PT))), dim3(256), 0, (hipStream_t)s_, dT, h2pre, sc2, sh2, mean2, inv2, PQR,
                       ld, H, F, idx, M, bx, scx, shx, meanx, invx, slope, g2, gy, partials);
  else
    hipLaunchKernelGGL(edge_attend_bwd_kernel, dim3(cdiv(M, EB_PT)), dim3(256), 0, (hipStream_t)s_, dT, h2pre, sc2, sh2, mean2, inv2, PQR, ld,
                       H, F, idx, M, k, bx, scx, shx, meanx, invx, slope, g2, gy, partials);
  return spgan_launch_status();
}

extern "C" int spgan_edge_scatter_b(const float* g1, const uint16_t* gy_bf16, const float* PQR, int ld, int H, int F, const int32_t* idx,
                                    const int32_t* rowptr, const int32_t* src, int M, int k, const float* b1, const float* mean1,
                                    const float* inv1, const float* gam1, const float* sums1, const float* bx, const float* meanx,
                                    const float* invx, const float* gamx, const float* sumsx, float* dPQR, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(g1 && gy_bf16 && PQR && idx && rowptr && src && b1 && mean1 && inv1 && gam1 && sums1 && bx && meanx && invx && gamx && sumsx && dPQR);
  SPGAN_CHECK_ARG(M > 0 && k == 10 && ld >= H + 2 * F);
  hipLaunchKernelGGL((edge_scatter_kernel<10, 1>), dim3(grid8(cdiv(M, 4))), dim3(256), 0, (hipStream_t)s_, g1, reinterpret_cast<const float*>(gy_bf16), PQR, ld,
                     H, F, idx, rowptr, src, M, k, b1, mean1, inv1, gam1, sums1, bx, meanx, invx, gamx, sumsx, 1.0f / ((float)M * (float)k), dPQR);
  return spgan_launch_status();
}

extern "C" int spgan_edge_scatter(const float* g1, const float* gy, const float* PQR, int ld, int H, int F, const int32_t* idx,
                                  const int32_t* rowptr, const int32_t* src, int M, int k, const float* b1, const float* mean1,
                                  const float* inv1, const float* gam1, const float* sums1, const float* bx, const float* meanx,
                                  const float* invx, const float* gamx, const float* sumsx, float* dPQR, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(g1 && gy && PQR && idx && rowptr && src && b1 && mean1 && inv1 && gam1 && sums1 && bx && meanx && invx && gamx && sumsx && dPQR);
  SPGAN_CHECK_ARG(M > 0 && k > 0 && ld >= H + 2 * F);
  if (k == 10)
    hipLaunchKernelGGL(edge_scatter_kernel<10>, dim3(grid8(cdiv(M, 4))), dim3(256), 0, (hipStream_t)s_, g1, gy, PQR, ld, H, F, idx, rowptr, src, M, k,
                       b1, mean1, inv1, gam1, sums1, bx, meanx, invx, gamx, sumsx, 1.0f / ((float)M * (float)k), dPQR);
  else
    hipLaunchKernelGGL(edge_scatter_kernel<0>, dim3(grid8(cdiv(M, 4))), dim3(256), 0, (hipStream_t)s_, g1, gy, PQR, ld, H, F, idx, rowptr, src, M, k,
                       b1, mean1, inv1, gam1, sums1, bx, meanx, invx, gamx, sumsx, 1.0f / ((float)M * (float)k), dPQR);
  return spgan_launch_status();
}
