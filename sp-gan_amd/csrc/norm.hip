// Column reductions and the BatchNorm / max-pool bookkeeping around the MFMA contractions.
// All of these are HBM-bound: lanes run along the channel dimension (coalesced rows of the
// point-major tensors), row slices are combined through LDS in a fixed order (deterministic).
#include "common.hpp"

namespace {

constexpr int RT = 128;  // rows per partial tile (== gemm_nt's BM, so fused and standalone partials share a format)

// partials layout: [groups * tiles_per_group][C][2]
// MODE 0: (sum, centred M2) of f(x);  MODE 1: (sum f(x), 0)
// direct0 != NULL (only with tiles_per_group == 1: a group is a single tile): the tile's result IS the group's -- written as
// out[g][c] = sum (MODE 1) or (mean, biased var) (MODE 0) right here, no finalize launch.
template <int MODE>
__global__ __launch_bounds__(256) void colpartials_kernel(const float* __restrict__ X, int ldx, int C, int G, int tiles_per_group,
                                                          float slope, float* __restrict__ part, float* __restrict__ direct0 = nullptr,
                                                          float* __restrict__ direct1 = nullptr) {
  __shared__ float red[4][64];
  const int tile = blockIdx.x;
  const int g = tile / tiles_per_group, q = tile % tiles_per_group;
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int sl = threadIdx.x >> 6;
  const int r0 = q * RT, cnt = min(RT, G - r0);
  const float* base = X + ((size_t)g * G + r0) * ldx;
  const bool cok = c < C;
  float s = 0.f;
  if (cok) {
    int r = sl;
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (; r + 28 < cnt; r += 32) {  // eight rows in flight per thread
#pragma unroll
      for (int u = 0; u < 8; ++u) a8[u] += lrelu_f(base[(size_t)(r + 4 * u) * ldx + c], slope);
    }
    s = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    for (; r < cnt; r += 4) s += lrelu_f(base[(size_t)r * ldx + c], slope);
  }
  red[sl][threadIdx.x & 63] = s;
  __syncthreads();
  const float tot = (red[0][threadIdx.x & 63] + red[1][threadIdx.x & 63]) + (red[2][threadIdx.x & 63] + red[3][threadIdx.x & 63]);
  float m2tot = 0.f;
  if (MODE == 0) {
    const float mean = tot / (float)cnt;
    __syncthreads();
    float m2 = 0.f;
    if (cok)
  #pragma unroll 8
    for (int r = sl; r < cnt; r += 4) {
        const float d = lrelu_f(base[(size_t)r * ldx + c], slope) - mean;
        m2 = fmaf(d, d, m2);
      }
    red[sl][threadIdx.x & 63] = m2;
    __syncthreads();
    m2tot = (red[0][threadIdx.x & 63] + red[1][threadIdx.x & 63]) + (red[2][threadIdx.x & 63] + red[3][threadIdx.x & 63]);
  }
  if (sl == 0 && cok) {
    if (direct0) {
      direct0[(size_t)tile * C + c] = MODE == 0 ? tot / (float)cnt : tot;
      if (MODE == 0 && direct1) direct1[(size_t)tile * C + c] = m2tot / (float)cnt;
      return;
    }
    float* o = part + ((size_t)tile * C + c) * 2;
    o[0] = tot;
    o[1] = m2tot;
  }
}

// Combine per-tile partials over the tiles of each group.  block = 32 slices x 8 columns (one 64-byte
// segment of the partial row per slice); slices are merged through LDS in a fixed tree order.
// <FS, FC> = <32, 8> by default; <128, 2> when a group has thousands of tiles (the per-edge GEMMs: 5120 records per column): four
// times the workgroups and a quarter of the serial record loop per thread (27 -> ~10 us at 5120 x 128).
// (the geometry is a template parameter of colfinalize_t_kernel; launch_colfinalize picks it)

__device__ __forceinline__ void chan_merge(float& n, float& a, float& b, float n2, float a2, float b2) {
  const float nn = n + n2;
  if (nn > 0.f) {
    const float d = a2 - a;
    a = a + d * (n2 / nn);
    b = b + b2 + d * d * (n * n2 / nn);
  }
  n = nn;
}

// Optional tail of the finalize kernel: train-mode BatchNorm bookkeeping from the just-computed (mean, var)
struct BnTail {
  const float* gamma; const float* beta;
  float* rmean; float* rvar;            // running stats (may be NULL)
  float* scale; float* shift; float* invstd; float* mean_out;   // NULL scale: no tail
  float eps, momentum;
  // Columns >= split belong to a SECOND BatchNorm layer (its own gamma/beta/running buffers, indexed from 0): the two per-edge
  // BatchNorms of an EdgeBlock (conv_w.1 over H channels, conv_x.1 over F) get their statistics from one record set.  split <= 0: off.
  int split;
  const float* gamma2; const float* beta2; float* rmean2; float* rvar2;
  int count_rep;                        // the rows stand for count_rep identical copies (unbiased-variance count of the running statistics)
  // seq_groups > 1 (grid.y == 1): the workgroup walks seq_groups row groups one after the other -- the same BatchNorm layer applied to
  // several passes: outputs of group g at offset g * group_stride, running statistics updated in group order by the same thread.
  int seq_groups; size_t group_stride;
  // mode 1 (plain sums S0 = out0, S1 = out1): also emit the BatchNorm-backward coefficient vectors coef[3,C] = [p | q | r] of
  // dy = p*g + q*y + r (spgan_bn_bwd_coeffs) -- the finalize of a BNBWD-epilogue GEMM hands the lazy operand to its consumers
  float* bwd_coef; const float* bwd_mean; const float* bwd_invstd; const float* bwd_gamma; float bwd_rcount;
  // mode 1: also phase B of the BatchNorm double backward (the arithmetic of spgan_bn_dbl_phaseb_sums) from the sums this launch merges:
  // s0 = out0, s1 = out1 -- the finalize behind the double backward's fused layer launch needs no per-channel launch after it
  const float* pb_U0; const float* pb_U1; const float* pb_Ugz; const float* pb_S0; const float* pb_S1; const float* pb_gamma; const float* pb_inv;
  float pb_rM; float* pb_sums; float* pb_dgamma;
  // ... and, from those phase-B sums (S0', S1'), the coefficients [p | q | r] of the lazy operand p*X + q*y + r = invstd*(X - S0'/M - xhat*S1'/M):
  // the BatchNorm backward (gamma = 1) of the adjoint X = xbarA + gamma*g, consumed on the next layer's operand loads (bn_bwd_coeffs_kernel's arithmetic)
  float* pb_coef; const float* pb_mean;
};

// workgroup (bx, by) of a (ceil(C / FC), groups) grid
template <int FS, int FC>
__device__ __forceinline__ void colfinalize_body(const float* __restrict__ part, int tiles_per_group, int C, int G, int mode, int tile_rows,
                                                 float* __restrict__ out0, float* __restrict__ out1, const BnTail& bn, int bx, int by) {
  static_assert(FS * FC == 256, "one thread per (slice, column)");
  __shared__ float sn[FS][FC], sa[FS][FC], sb[FS][FC];
  const int cl = threadIdx.x & (FC - 1), sl = threadIdx.x / FC;
  const int c = bx * FC + cl;
  const bool cok = c < C;
  const int g_end = bn.seq_groups > 1 ? bn.seq_groups : by + 1;
  for (int g = bn.seq_groups > 1 ? 0 : by; g < g_end; ++g) {
  const size_t goff = bn.seq_groups > 1 ? (size_t)g * bn.group_stride : 0;
  if (g > 0 && bn.seq_groups > 1) __syncthreads();  // the LDS exchange of the previous group is done
  float n = 0.f, a = 0.f, b = 0.f;  // mode 0: (count, mean, M2); mode 1: (-, s0, s1)
  if (cok) {
    const float2* base = reinterpret_cast<const float2*>(part) + (size_t)g * tiles_per_group * C + c;
    int t = sl;
    for (; t + 7 * FS < tiles_per_group; t += 8 * FS) {  // eight independent loads in flight (512 tiles: two rounds per thread)
      float2 q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) q[u] = base[(size_t)(t + u * FS) * C];
      if (mode == 0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float nb = (float)min(tile_rows, G - (t + u * FS) * tile_rows);
          chan_merge(n, a, b, nb, q[u].x / nb, q[u].y);
        }
      } else {
        a += ((q[0].x + q[1].x) + (q[2].x + q[3].x)) + ((q[4].x + q[5].x) + (q[6].x + q[7].x));
        b += ((q[0].y + q[1].y) + (q[2].y + q[3].y)) + ((q[4].y + q[5].y) + (q[6].y + q[7].y));
      }
    }
    for (; t + 3 * FS < tiles_per_group; t += 4 * FS) {  // four independent loads in flight
      const float2 p0 = base[(size_t)t * C], p1 = base[(size_t)(t + FS) * C], p2 = base[(size_t)(t + 2 * FS) * C],
                   p3 = base[(size_t)(t + 3 * FS) * C];
      if (mode == 0) {
        chan_merge(n, a, b, (float)min(tile_rows, G - t * tile_rows), p0.x / (float)min(tile_rows, G - t * tile_rows), p0.y);
        chan_merge(n, a, b, (float)min(tile_rows, G - (t + FS) * tile_rows), p1.x / (float)min(tile_rows, G - (t + FS) * tile_rows), p1.y);
        chan_merge(n, a, b, (float)min(tile_rows, G - (t + 2 * FS) * tile_rows), p2.x / (float)min(tile_rows, G - (t + 2 * FS) * tile_rows), p2.y);
        chan_merge(n, a, b, (float)min(tile_rows, G - (t + 3 * FS) * tile_rows), p3.x / (float)min(tile_rows, G - (t + 3 * FS) * tile_rows), p3.y);
      } else {
        a += (p0.x + p1.x) + (p2.x + p3.x);
        b += (p0.y + p1.y) + (p2.y + p3.y);
      }
    }
    for (; t < tiles_per_group; t += FS) {
      const float2 p0 = base[(size_t)t * C];
      if (mode == 0) {
        const float nb = (float)min(tile_rows, G - t * tile_rows);
        chan_merge(n, a, b, nb, p0.x / nb, p0.y);
      } else {
        a += p0.x;
        b += p0.y;
      }
    }
  }
  sn[sl][cl] = n; sa[sl][cl] = a; sb[sl][cl] = b;
  __syncthreads();
  for (int w = FS / 2; w > 0; w >>= 1) {
    if (sl < w) {
      const float n2 = sn[sl + w][cl], a2 = sa[sl + w][cl], b2 = sb[sl + w][cl];
      if (mode == 0) chan_merge(n, a, b, n2, a2, b2);
      else { a += a2; b += b2; }
      sn[sl][cl] = n; sa[sl][cl] = a; sb[sl][cl] = b;
    }
    __syncthreads();
  }
  if (sl == 0 && cok) {
    const float var = (mode == 0) ? b / (float)G : b;
    if (out0) out0[(size_t)g * C + c] = a;
    if (out1) out1[(size_t)g * C + c] = var;
    if (bn.bwd_coef) {  // single group, mode 1: the arithmetic of bn_bwd_coeffs_kernel
      const float inv = bn.bwd_invstd[c];
      const float pc = (bn.bwd_gamma ? bn.bwd_gamma[c] : 1.f) * inv;
      const float qc = -(pc * inv) * (b * bn.bwd_rcount);
      const float rc = -(pc * (a * bn.bwd_rcount)) - qc * bn.bwd_mean[c];
      bn.bwd_coef[c] = pc;
      bn.bwd_coef[C + c] = qc;
      bn.bwd_coef[2 * C + c] = rc;
    }
    if (bn.pb_sums) {  // single group, mode 1: the arithmetic of bn_dbl_phaseb_sums_kernel
      const float U0 = bn.pb_U0[c], U1 = bn.pb_U1[c], S0 = bn.pb_S0[c], S1 = bn.pb_S1[c], ga = bn.pb_gamma[c], iv = bn.pb_inv[c];
      const float core = bn.pb_Ugz[c] - (U0 * S0 + U1 * S1) * bn.pb_rM;
      const float gsM = ga * iv * bn.pb_rM;
      const float k0 = iv * core, k1 = ga * core, k2 = -gsM * (U0 * S1 + S0 * U1), k3 = -2.0f * gsM * (U1 * S1);
      const float S0n = k2 + ga * a, S1n = k3 + ga * b + iv * k1;
      bn.pb_sums[c] = S0n;
      bn.pb_sums[C + c] = S1n;
      bn.pb_dgamma[c] = k0 + b;
      if (bn.pb_coef) {
        const float pc = iv;
        const float qc = -(pc * iv) * (S1n * bn.pb_rM);
        bn.pb_coef[c] = pc;
        bn.pb_coef[C + c] = qc;
        bn.pb_coef[2 * C + c] = -(pc * (S0n * bn.pb_rM)) - qc * bn.pb_mean[c];
      }
    }
    if (bn.scale) {  // single group, mode 0: same arithmetic as bn_prepare_kernel
      const bool second = bn.split > 0 && c >= bn.split;
      const int cc = second ? c - bn.split : c;
      float* rm = second ? bn.rmean2 : bn.rmean;
      float* rv = second ? bn.rvar2 : bn.rvar;
      const float* gam = second ? bn.gamma2 : bn.gamma;
      const float* bet = second ? bn.beta2 : bn.beta;
      if (rm) {
        const float cnt = (float)G * (float)(bn.count_rep > 1 ? bn.count_rep : 1);
        const float unb = cnt > 1.f ? var * (cnt / (cnt - 1.f)) : var;
        rm[cc] = (1.f - bn.momentum) * rm[cc] + bn.momentum * a;
        rv[cc] = (1.f - bn.momentum) * rv[cc] + bn.momentum * unb;
      }
      const float inv = 1.0f / sqrtf(var + bn.eps);
      const float ga = gam ? gam[cc] : 1.f, be = bet ? bet[cc] : 0.f;
      const float sc = ga * inv;
      bn.scale[goff + c] = sc;
      bn.shift[goff + c] = be - a * sc;
      bn.invstd[goff + c] = inv;
      bn.mean_out[goff + c] = a;
    }
  }
  }
}

template <int FS, int FC>
__global__ __launch_bounds__(256) void colfinalize_t_kernel(const float* __restrict__ part, int tiles_per_group, int C, int G, int mode,
                                                            int tile_rows, float* __restrict__ out0, float* __restrict__ out1, const BnTail bn) {
  colfinalize_body<FS, FC>(part, tiles_per_group, C, G, mode, tile_rows, out0, out1, bn, blockIdx.x, blockIdx.y);
}

// spgan_colstats_finalize_multi: the mode-1 finalize launches (plain sums, optionally with the BatchNorm-backward coefficients or phase B of
// the double backward as tail) behind the `count` problems of a grouped GEMM launch, as ONE launch: problem blockIdx.y runs the stand-alone
// kernel's body on its own record set -- bit-identical results.
struct FinMulti {
  const float* part[SPGAN_GROUP_MAX];
  int tiles[SPGAN_GROUP_MAX], C[SPGAN_GROUP_MAX], G[SPGAN_GROUP_MAX], tile_rows[SPGAN_GROUP_MAX];
  float* out0[SPGAN_GROUP_MAX];
  float* out1[SPGAN_GROUP_MAX];
  BnTail bn[SPGAN_GROUP_MAX];
};

__global__ __launch_bounds__(256) void colfinalize_multi_kernel(const FinMulti m) {
  const int g = blockIdx.y;
  if ((int)blockIdx.x * 8 >= m.C[g]) return;
  colfinalize_body<32, 8>(m.part[g], m.tiles[g], m.C[g], m.G[g], 1, m.tile_rows[g], m.out0[g], m.out1[g], m.bn[g], blockIdx.x, 0);
}

// The grouped BatchNorm finalize (spgan_colstats_finalize_bn_groups) for a handful of groups: the record loads of ALL groups are in
// flight together (walking the groups one after the other costs one memory latency chain per group: 15.5 us for three passes of
// 512 tiles where one pass takes 7), every group is merged in exactly the order of the single-group kernel above (same results bit
// for bit), and the running statistics are advanced group after group by the thread that owns the column.
template <int FS, int FC, int NG>
__global__ __launch_bounds__(256) void colfinalize_bn_groups_kernel(const float* __restrict__ part, int tiles_per_group, int C, int G, int tile_rows,
                                                                   const BnTail bn) {
  static_assert(FS * FC == 256, "one thread per (slice, column)");
  __shared__ float sn[NG][FS][FC], sa[NG][FS][FC], sb[NG][FS][FC];
  const int cl = threadIdx.x & (FC - 1), sl = threadIdx.x / FC;
  const int c = blockIdx.x * FC + cl;
  const bool cok = c < C;
  float n[NG], a[NG], b[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) { n[g] = 0.f; a[g] = 0.f; b[g] = 0.f; }
  if (cok) {
    const float2* base = reinterpret_cast<const float2*>(part) + c;
    const size_t gstride = (size_t)tiles_per_group * C;
    int t = sl;
    for (; t + 3 * FS < tiles_per_group; t += 4 * FS) {  // NG x 4 independent loads in flight
      float2 q[NG][4];
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int u = 0; u < 4; ++u) q[g][u] = base[g * gstride + (size_t)(t + u * FS) * C];
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float nb = (float)min(tile_rows, G - (t + u * FS) * tile_rows);
          chan_merge(n[g], a[g], b[g], nb, q[g][u].x / nb, q[g][u].y);
        }
    }
    for (; t < tiles_per_group; t += FS) {
      float2 q[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) q[g] = base[g * gstride + (size_t)t * C];
      const float nb = (float)min(tile_rows, G - t * tile_rows);
#pragma unroll
      for (int g = 0; g < NG; ++g) chan_merge(n[g], a[g], b[g], nb, q[g].x / nb, q[g].y);
    }
  }
#pragma unroll
  for (int g = 0; g < NG; ++g) { sn[g][sl][cl] = n[g]; sa[g][sl][cl] = a[g]; sb[g][sl][cl] = b[g]; }
  __syncthreads();
  for (int w = FS / 2; w > 0; w >>= 1) {
    if (sl < w) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        chan_merge(n[g], a[g], b[g], sn[g][sl + w][cl], sa[g][sl + w][cl], sb[g][sl + w][cl]);
        sn[g][sl][cl] = n[g]; sa[g][sl][cl] = a[g]; sb[g][sl][cl] = b[g];
      }
    }
    __syncthreads();
  }
  if (sl == 0 && cok) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {  // in group order: the running statistics see pass 0 first
      const size_t goff = (size_t)g * bn.group_stride;
      const float mean = a[g], var = b[g] / (float)G;
      if (bn.rmean) {
        const float cnt = (float)G;
        const float unb = cnt > 1.f ? var * (cnt / (cnt - 1.f)) : var;
        bn.rmean[c] = (1.f - bn.momentum) * bn.rmean[c] + bn.momentum * mean;
        bn.rvar[c] = (1.f - bn.momentum) * bn.rvar[c] + bn.momentum * unb;
      }
      const float inv = 1.0f / sqrtf(var + bn.eps);
      const float ga = bn.gamma ? bn.gamma[c] : 1.f, be = bn.beta ? bn.beta[c] : 0.f;
      const float sc = ga * inv;
      bn.scale[goff + c] = sc;
      bn.shift[goff + c] = be - mean * sc;
      bn.invstd[goff + c] = inv;
      bn.mean_out[goff + c] = mean;
    }
  }
}

inline void launch_colfinalize(hipStream_t s, const float* part, int groups, int tiles_per_group, int C, int G, int mode, int tile_rows,
                               float* out0, float* out1, const BnTail& bn) {
  if (tiles_per_group >= 2048)
    hipLaunchKernelGGL((colfinalize_t_kernel<128, 2>), dim3(cdiv(C, 2), groups), dim3(256), 0, s, part, tiles_per_group, C, G, mode, tile_rows, out0, out1, bn);
  else
    hipLaunchKernelGGL((colfinalize_t_kernel<32, 8>), dim3(cdiv(C, 8), groups), dim3(256), 0, s, part, tiles_per_group, C, G, mode, tile_rows, out0, out1, bn);
}

__global__ void bn_prepare_kernel(const float* mean, const float* var, const float* gamma, const float* beta, int C, int count,
                                  float eps, float momentum, int training, float* rmean, float* rvar, float* scale, float* shift,
                                  float* invstd, float* mean_used) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float m, v;
  if (training) {
    m = mean[c];
    v = fmaxf(var[c], 0.f);  // a variance obtained as a difference (d_advance_running_stats) may round below zero
    if (rmean) {
      const float unb = count > 1 ? v * ((float)count / (float)(count - 1)) : v;
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * m;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
    }
  } else {
    m = rmean[c];
    v = rvar[c];
  }
  const float inv = 1.0f / sqrtf(v + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const float sc = g * inv;
  scale[c] = sc;
  shift[c] = b - m * sc;
  invstd[c] = inv;
  mean_used[c] = m;
}

__global__ void bn_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ y, int ld, size_t M, int C,
                                    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                    const float* __restrict__ sums, float rcount, float* __restrict__ dy) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * C) return;
  const size_t m = t / C;
  const int c = t % C;
  const float inv = invstd[c];
  const float xh = (y[m * ld + c] - mean[c]) * inv;
  const float ga = gamma ? gamma[c] : 1.f;
  dy[m * ld + c] = ga * inv * (g[m * ld + c] - sums[c] * rcount - xh * (sums[C + c] * rcount));
}

// 4 channels per thread, 32-bit index arithmetic (the scalar kernel spends its time in 64-bit div/mod: 4.0-4.4 TB/s by PMC)
__global__ __launch_bounds__(256) void bn_bwd_apply_v4_kernel(const float* __restrict__ g, const float* __restrict__ y, int ld, unsigned total4,
                                                              int C, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ sums, float rcount,
                                                              float* __restrict__ dy, const float* __restrict__ g2,
                                                              const float* __restrict__ g2scale) {
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  if (t >= total4) return;
  const unsigned c4n = (unsigned)C >> 2;
  const unsigned m = t / c4n;
  const int c = (int)(t - m * c4n) * 4;
  const size_t o = (size_t)m * ld + c;
  float4 gv = *reinterpret_cast<const float4*>(g + o);
  const float4 yv = *reinterpret_cast<const float4*>(y + o);
  if (g2) {  // g + g2scale[c]*g2 formed on the fly (the double-backward's xbarA + gamma*g without a pass of its own)
    const float4 hv = *reinterpret_cast<const float4*>(g2 + (size_t)m * C + c), hs = *reinterpret_cast<const float4*>(g2scale + c);
    gv.x = fmaf(hs.x, hv.x, gv.x); gv.y = fmaf(hs.y, hv.y, gv.y); gv.z = fmaf(hs.z, hv.z, gv.z); gv.w = fmaf(hs.w, hv.w, gv.w);
  }
  const float4 mu = *reinterpret_cast<const float4*>(mean + c), iv = *reinterpret_cast<const float4*>(invstd + c);
  const float4 s0 = *reinterpret_cast<const float4*>(sums + c), s1 = *reinterpret_cast<const float4*>(sums + C + c);
  float4 ga = make_float4(1.f, 1.f, 1.f, 1.f);
  if (gamma) ga = *reinterpret_cast<const float4*>(gamma + c);
  float4 r;
  r.x = ga.x * iv.x * (gv.x - s0.x * rcount - ((yv.x - mu.x) * iv.x) * (s1.x * rcount));
  r.y = ga.y * iv.y * (gv.y - s0.y * rcount - ((yv.y - mu.y) * iv.y) * (s1.y * rcount));
  r.z = ga.z * iv.z * (gv.z - s0.z * rcount - ((yv.z - mu.z) * iv.z) * (s1.z * rcount));
  r.w = ga.w * iv.w * (gv.w - s0.w * rcount - ((yv.w - mu.w) * iv.w) * (s1.w * rcount));
  *reinterpret_cast<float4*>(dy + o) = r;
}

__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ y, int ld, int N, int C, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, float slope, float* __restrict__ out,
                                                      int32_t* __restrict__ argmax) {
  __shared__ float rv[4][64];
  __shared__ int ri[4][64];
  const int b = blockIdx.x;
  const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cl;
  const bool cok = c < C;
  const float sc = (cok && scale) ? scale[c] : 1.f, sh = (cok && shift) ? shift[c] : 0.f;
  const float* base = y + (size_t)b * N * ld;
  float best = -INFINITY;
  int bi = 0;
  if (cok)
    for (int n = sl; n < N; n += 4) {
      const float v = lrelu_f(fmaf(base[(size_t)n * ld + c], sc, sh), slope);
      if (v > best) {
        best = v;
        bi = n;
      }
    }
  rv[sl][cl] = best;
  ri[sl][cl] = bi;
  __syncthreads();
  if (sl == 0 && cok) {
#pragma unroll
    for (int s2 = 1; s2 < 4; ++s2) {
      const float v = rv[s2][cl];
      const int i2 = ri[s2][cl];
      if (v > best || (v == best && i2 < bi)) {
        best = v;
        bi = i2;
      }
    }
    out[(size_t)b * C + c] = best;
    if (argmax) argmax[(size_t)b * C + c] = b * N + bi;  // global row index
  }
}

// float4 variant: LQ lanes x 16 B cover LQ*4 channels of a row, 256/LQ row-slices per workgroup.  LQ = 16 (64 channels per workgroup)
// when shapes x channel blocks already fill the chip; LQ = 4 (16 channels, 64 row-slices: 4x the workgroups, a quarter of the serial
// row loop) for the generator's global feature (B x 128 channels: 64 workgroups of 2048 rows each were latency-bound at 33 us).
template <int LQ>
__global__ __launch_bounds__(256) void maxpool_v4_kernel(const float* __restrict__ y, int ld, int N, int C, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float slope, float* __restrict__ out,
                                                         int32_t* __restrict__ argmax) {
  constexpr int SLN = 256 / LQ, CW = LQ * 4;
  __shared__ float rv[SLN][CW];
  __shared__ int ri[SLN][CW];
  const int b = blockIdx.x;
  const int q = threadIdx.x % LQ, sl = threadIdx.x / LQ;
  const int c = blockIdx.y * CW + q * 4;
  const bool cok = c < C;  // C % 4 == 0
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cok && scale) { sc = *reinterpret_cast<const float4*>(scale + c); sh = *reinterpret_cast<const float4*>(shift + c); }
  const float* base = y + (size_t)b * N * ld + c;
  float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  int bi[4] = {0, 0, 0, 0};
  if (cok) {
    auto take = [&](const float4 v, int n) {
      const float t0 = lrelu_f(fmaf(v.x, sc.x, sh.x), slope), t1 = lrelu_f(fmaf(v.y, sc.y, sh.y), slope);
      const float t2 = lrelu_f(fmaf(v.z, sc.z, sh.z), slope), t3 = lrelu_f(fmaf(v.w, sc.w, sh.w), slope);
      if (t0 > best[0]) { best[0] = t0; bi[0] = n; }
      if (t1 > best[1]) { best[1] = t1; bi[1] = n; }
      if (t2 > best[2]) { best[2] = t2; bi[2] = n; }
      if (t3 > best[3]) { best[3] = t3; bi[3] = n; }
    };
    int n = sl;
    for (; n + 7 * SLN < N; n += 8 * SLN) {  // eight rows in flight, consumed in row order (strict '>' keeps the first maximum)
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(base + (size_t)(n + SLN * u) * ld);
#pragma unroll
      for (int u = 0; u < 8; ++u) take(v[u], n + SLN * u);
    }
    for (; n < N; n += SLN) take(*reinterpret_cast<const float4*>(base + (size_t)n * ld), n);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { rv[sl][q * 4 + j] = best[j]; ri[sl][q * 4 + j] = bi[j]; }
  __syncthreads();
  if (threadIdx.x < CW) {
    const int cc = blockIdx.y * CW + threadIdx.x;
    if (cc < C) {
      float bv = rv[0][threadIdx.x];
      int bn = ri[0][threadIdx.x];
      for (int s2 = 1; s2 < SLN; ++s2) {
        const float v = rv[s2][threadIdx.x];
        const int i2 = ri[s2][threadIdx.x];
        if (v > bv || (v == bv && i2 < bn)) { bv = v; bn = i2; }
      }
      out[(size_t)b * C + cc] = bv;
      if (argmax) argmax[(size_t)b * C + cc] = b * N + bn;
    }
  }
}

// Global max-pool over each shape from per-tile column (max, min, arg-max, arg-min) partials of the PRE-BatchNorm output
// (spgan_gemm_nt pooling epilogue): z = scale*y + shift is increasing in y for scale > 0 and decreasing for scale < 0, and
// LeakyReLU (slope > 0) is increasing, so the pooled value sits at the max resp. min of y.  scale == 0: every row ties, the
// first row wins (like torch.max).  Tiles are visited in ascending row order with strict compares: first row on ties.
__global__ void pool_finalize_kernel(const float* __restrict__ pv, const int32_t* __restrict__ pa, int B, int tiles, int C,
                                     const float* __restrict__ scale, const float* __restrict__ shift, float slope, int rows,
                                     float* __restrict__ pooled, int32_t* __restrict__ argmax, float* __restrict__ yarg,
                                     int group_stride, int shapes_per_group, int relative_rows) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (c >= C) return;
  const int grp = shapes_per_group > 0 ? b / shapes_per_group : 0;   // the pass this shape belongs to (its own BatchNorm vectors)
  const float sc = scale[(size_t)grp * group_stride + c], sh = shift[(size_t)grp * group_stride + c];
  const int pick = sc >= 0.f ? 0 : 1;
  float best = 0.f;
  int arg = -1;
  for (int t = 0; t < tiles; ++t) {
    const size_t o = ((size_t)(b * tiles + t) * C + c) * 2 + pick;
    const float v = pv[o];
    if (arg < 0 || (pick == 0 ? v > best : v < best)) { best = v; arg = pa[o]; }
  }
  if (sc == 0.f) arg = b * rows;
  if (relative_rows) arg -= grp * shapes_per_group * rows;
  pooled[(size_t)b * C + c] = lrelu_f(fmaf(best, sc, sh), slope);
  argmax[(size_t)b * C + c] = arg;
  if (yarg) yarg[(size_t)b * C + c] = best;
}

}  // namespace

extern "C" int spgan_pool_finalize(const float* pool_val, const int32_t* pool_arg, int B, int rows, int C, const float* scale, const float* shift,
                                   float slope, float* pooled, int32_t* argmax, float* yarg, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(pool_val && pool_arg && scale && shift && pooled && argmax && B > 0 && rows > 0 && C > 0 && rows % 128 == 0 && slope > 0.f);
  hipLaunchKernelGGL(pool_finalize_kernel, dim3(cdiv(C, 64), B), dim3(64), 0, (hipStream_t)s_, pool_val, pool_arg, B, rows / 128, C, scale, shift,
                     slope, rows, pooled, argmax, yarg, 0, 0, 0);
  return spgan_launch_status();
}

extern "C" int spgan_pool_finalize_groups(const float* pool_val, const int32_t* pool_arg, int B, int rows, int C, const float* scale,
                                          const float* shift, int group_stride, int shapes_per_group, float slope, float* pooled,
                                          int32_t* argmax, float* yarg, int relative_rows, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(pool_val && pool_arg && scale && shift && pooled && argmax && B > 0 && rows > 0 && C > 0 && rows % 128 == 0 && slope > 0.f);
  SPGAN_CHECK_ARG(shapes_per_group > 0 && B % shapes_per_group == 0 && group_stride >= C);
  hipLaunchKernelGGL(pool_finalize_kernel, dim3(cdiv(C, 64), B), dim3(64), 0, (hipStream_t)s_, pool_val, pool_arg, B, rows / 128, C, scale, shift,
                     slope, rows, pooled, argmax, yarg, group_stride, shapes_per_group, relative_rows);
  return spgan_launch_status();
}

extern "C" size_t spgan_colreduce_ws_bytes(int M, int C, int G) {
  if (M <= 0 || C <= 0 || G <= 0) return 0;
  const size_t groups = M / G;
  return groups * (size_t)cdiv(G, RT) * C * 2 * sizeof(float);
}

extern "C" int spgan_colstats_finalize(const float* partials, int groups, int tiles_per_group, int C, int G, int mode, int tile_rows,
                                       float* out0, float* out1, spgan_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  if (tile_rows <= 0) tile_rows = RT;
  SPGAN_CHECK_ARG(partials && out0 && out1 && groups > 0 && tiles_per_group > 0 && C > 0 && G > 0 && (mode == 0 || mode == 1));
  SPGAN_CHECK_ARG(tiles_per_group == cdiv(G, tile_rows));
  launch_colfinalize(s, partials, groups, tiles_per_group, C, G, mode, tile_rows, out0, out1, BnTail{});
  return spgan_launch_status();
}

// Finalize plain column sums (mode 1, one group) AND emit the BatchNorm-backward coefficients of spgan_bn_bwd_coeffs from them.
extern "C" int spgan_colstats_finalize_bnbwd(const float* partials, int tiles, int C, int G, int tile_rows, const float* mean, const float* invstd,
                                             const float* gamma, float count, float* s0, float* s1, float* coef, spgan_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  if (tile_rows <= 0) tile_rows = RT;
  SPGAN_CHECK_ARG(partials && s0 && s1 && coef && mean && invstd && tiles > 0 && C > 0 && G > 0 && count > 0.f && tiles == cdiv(G, tile_rows));
  BnTail bn{};
  bn.bwd_coef = coef; bn.bwd_mean = mean; bn.bwd_invstd = invstd; bn.bwd_gamma = gamma; bn.bwd_rcount = 1.0f / count;
  launch_colfinalize(s, partials, 1, tiles, C, G, 1, tile_rows, s0, s1, bn);
  return spgan_launch_status();
}

extern "C" int spgan_colstats_finalize_phaseb(const float* partials, int tiles, int C, int G, int tile_rows, const float* U0, const float* U1,
                                              const float* Ugz, const float* S0, const float* S1, const float* gamma, const float* invstd, int count,
                                              float* s0, float* s1, float* sums2C, float* dgamma, spgan_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  if (tile_rows <= 0) tile_rows = RT;
  SPGAN_CHECK_ARG(partials && s0 && s1 && U0 && U1 && Ugz && S0 && S1 && gamma && invstd && sums2C && dgamma);
  SPGAN_CHECK_ARG(tiles > 0 && C > 0 && G > 0 && count > 0 && tiles == cdiv(G, tile_rows));
  BnTail bn{};
  bn.pb_U0 = U0; bn.pb_U1 = U1; bn.pb_Ugz = Ugz; bn.pb_S0 = S0; bn.pb_S1 = S1; bn.pb_gamma = gamma; bn.pb_inv = invstd;
  bn.pb_rM = 1.0f / (float)count; bn.pb_sums = sums2C; bn.pb_dgamma = dgamma;
  launch_colfinalize(s, partials, 1, tiles, C, G, 1, tile_rows, s0, s1, bn);
  return spgan_launch_status();
}

extern "C" int spgan_colstats_finalize_multi(const spgan_colfinalize_args* a, int count, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(a && count >= 1 && count <= SPGAN_GROUP_MAX);
  FinMulti m{};
  int cmax = 0;
  for (int g = 0; g < count; ++g) {
    const spgan_colfinalize_args& q = a[g];
    const int tr = q.tile_rows > 0 ? q.tile_rows : RT;
    SPGAN_CHECK_ARG(q.partials && q.s0 && q.s1 && q.tiles > 0 && q.tiles < 2048 && q.C > 0 && q.G > 0 && q.tiles == cdiv(q.G, tr));
    SPGAN_CHECK_ARG(q.kind >= 0 && q.kind <= 2);
    m.part[g] = q.partials; m.tiles[g] = q.tiles; m.C[g] = q.C; m.G[g] = q.G; m.tile_rows[g] = tr; m.out0[g] = q.s0; m.out1[g] = q.s1;
    BnTail& bn = m.bn[g];
    if (q.kind == 1) {
      SPGAN_CHECK_ARG(q.coef && q.mean && q.invstd && q.count > 0.f);
      bn.bwd_coef = q.coef; bn.bwd_mean = q.mean; bn.bwd_invstd = q.invstd; bn.bwd_gamma = q.gamma; bn.bwd_rcount = 1.0f / q.count;
    } else if (q.kind == 2) {
      SPGAN_CHECK_ARG(q.U0 && q.U1 && q.Ugz && q.S0 && q.S1 && q.gamma && q.invstd && q.sums && q.dgamma && q.count > 0.f);
      bn.pb_U0 = q.U0; bn.pb_U1 = q.U1; bn.pb_Ugz = q.Ugz; bn.pb_S0 = q.S0; bn.pb_S1 = q.S1; bn.pb_gamma = q.gamma; bn.pb_inv = q.invstd;
      bn.pb_rM = 1.0f / q.count; bn.pb_sums = q.sums; bn.pb_dgamma = q.dgamma;
      SPGAN_CHECK_ARG(!q.pb_coef || q.mean);
      bn.pb_coef = q.pb_coef; bn.pb_mean = q.mean;
    }
    if (q.C > cmax) cmax = q.C;
  }
  hipLaunchKernelGGL(colfinalize_multi_kernel, dim3(cdiv(cmax, 8), count), dim3(256), 0, (hipStream_t)s_, m);
  return spgan_launch_status();
}

// Finalize the (sum, M2) partials of ONE group of G rows and do the train-mode BatchNorm bookkeeping in the same launch:
// scale = gamma*invstd, shift = beta - mean*scale, invstd, mean; running stats updated in place when given.
extern "C" int spgan_colstats_finalize_bn(const float* partials, int tiles, int C, int G, int tile_rows, const float* gamma, const float* beta,
                                          float eps, float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                                          float* invstd, float* mean_out, spgan_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  if (tile_rows <= 0) tile_rows = RT;
  SPGAN_CHECK_ARG(partials && scale && shift && invstd && mean_out && tiles > 0 && C > 0 && G > 0 && tiles == cdiv(G, tile_rows));
  SPGAN_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr));
  BnTail bn{gamma, beta, running_mean, running_var, scale, shift, invstd, mean_out, eps, momentum, 0, nullptr, nullptr, nullptr, nullptr, 1, 0, 0};
  launch_colfinalize(s, partials, 1, tiles, C, G, 0, tile_rows, (float*)nullptr,
                     (float*)nullptr, bn);
  return spgan_launch_status();
}

extern "C" int spgan_colstats_finalize_bn_groups(const float* partials, int groups, int tiles_per_group, int C, int G, int tile_rows,
                                                 const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                                 float* running_var, float* out, spgan_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  if (tile_rows <= 0) tile_rows = RT;
  SPGAN_CHECK_ARG(partials && out && groups > 0 && tiles_per_group > 0 && C > 0 && G > 0 && tiles_per_group == cdiv(G, tile_rows));
  SPGAN_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr));
  // one workgroup row (grid.y == 1) walks the groups in order; with a single group this is spgan_colstats_finalize_bn
  const size_t gc = (size_t)groups * C;  // out [4, groups, C]: the scale (shift, ...) vectors of all groups are one contiguous [groups, C] block
  BnTail bn{gamma, beta, running_mean, running_var, out, out + gc, out + 2 * gc, out + 3 * gc, eps, momentum,
            0, nullptr, nullptr, nullptr, nullptr, 1, groups > 1 ? groups : 0, (size_t)C};
  if (groups == 2 && tiles_per_group < 2048)
    hipLaunchKernelGGL((colfinalize_bn_groups_kernel<32, 8, 2>), dim3(cdiv(C, 8)), dim3(256), 0, s, partials, tiles_per_group, C, G, tile_rows, bn);
  else if (groups == 3 && tiles_per_group < 2048)
    hipLaunchKernelGGL((colfinalize_bn_groups_kernel<32, 8, 3>), dim3(cdiv(C, 8)), dim3(256), 0, s, partials, tiles_per_group, C, G, tile_rows, bn);
  else  // any number of groups: one workgroup row walks them one after the other
    launch_colfinalize(s, partials, 1, tiles_per_group, C, G, 0, tile_rows, (float*)nullptr, (float*)nullptr, bn);
  return spgan_launch_status();
}

// The same for TWO BatchNorm layers whose channels sit side by side in one record set (columns [0,split) and [split,C)):
// out4 [4,C] = scale | shift | invstd | mean over all C columns.  count_rep as in spgan_gemm_nt's fan-in descriptor.
extern "C" int spgan_colstats_finalize_bn2(const float* partials, int tiles, int C, int G, int tile_rows, int split, const float* gammaA,
                                           const float* betaA, float* rmeanA, float* rvarA, const float* gammaB, const float* betaB,
                                           float* rmeanB, float* rvarB, float eps, float momentum, int count_rep, float* out4,
                                           spgan_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  if (tile_rows <= 0) tile_rows = RT;
  SPGAN_CHECK_ARG(partials && out4 && tiles > 0 && C > 0 && G > 0 && tiles == cdiv(G, tile_rows) && split > 0 && split < C && count_rep >= 1);
  SPGAN_CHECK_ARG((rmeanA == nullptr) == (rvarA == nullptr) && (rmeanB == nullptr) == (rvarB == nullptr));
  BnTail bn{gammaA, betaA, rmeanA, rvarA, out4, out4 + C, out4 + 2 * (size_t)C, out4 + 3 * (size_t)C, eps, momentum,
            split, gammaB, betaB, rmeanB, rvarB, count_rep, 0, 0};
  launch_colfinalize(s, partials, 1, tiles, C, G, 0, tile_rows, (float*)nullptr,
                     (float*)nullptr, bn);
  return spgan_launch_status();
}

extern "C" int spgan_colstats(const float* X, int ldx, int M, int C, int G, float slope, float* out_mean, float* out_var, float* ws,
                              size_t ws_bytes, spgan_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  SPGAN_CHECK_ARG(X && out_mean && out_var && ws && M > 0 && C > 0 && G > 0 && M % G == 0 && ldx >= C);
  SPGAN_CHECK_ARG(ws_bytes >= spgan_colreduce_ws_bytes(M, C, G));
  const int groups = M / G, tpg = cdiv(G, RT);
  if (tpg == 1) {  // groups of <= 128 rows: one launch
    hipLaunchKernelGGL((colpartials_kernel<0>), dim3(groups, cdiv(C, 64)), dim3(256), 0, s, X, ldx, C, G, 1, slope, ws, out_mean, out_var);
    return spgan_launch_status();
  }
  hipLaunchKernelGGL((colpartials_kernel<0>), dim3(groups * tpg, cdiv(C, 64)), dim3(256), 0, s, X, ldx, C, G, tpg, slope, ws, (float*)nullptr, (float*)nullptr);
  launch_colfinalize(s, ws, groups, tpg, C, G, 0, RT, out_mean, out_var, BnTail{});
  return spgan_launch_status();
}

extern "C" int spgan_colsum(const float* X, int ldx, int M, int C, int G, float* out, float* ws, size_t ws_bytes, spgan_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  SPGAN_CHECK_ARG(X && out && ws && M > 0 && C > 0 && G > 0 && M % G == 0 && ldx >= C);
  const int groups = M / G, tpg = cdiv(G, RT);
  // ws: partials + a scratch row block for the unused second output
  SPGAN_CHECK_ARG(ws_bytes >= spgan_colreduce_ws_bytes(M, C, G) + (size_t)groups * C * sizeof(float));
  float* scratch = ws + (size_t)groups * tpg * C * 2;
  if (tpg == 1) {  // groups of <= 128 rows (per-shape linears: M = batch): one launch
    hipLaunchKernelGGL((colpartials_kernel<1>), dim3(groups, cdiv(C, 64)), dim3(256), 0, s, X, ldx, C, G, 1, 1.0f, ws, out, (float*)nullptr);
    return spgan_launch_status();
  }
  hipLaunchKernelGGL((colpartials_kernel<1>), dim3(groups * tpg, cdiv(C, 64)), dim3(256), 0, s, X, ldx, C, G, tpg, 1.0f, ws, (float*)nullptr, (float*)nullptr);
  launch_colfinalize(s, ws, groups, tpg, C, G, 1, RT, out, scratch, BnTail{});
  return spgan_launch_status();
}

extern "C" int spgan_bn_prepare(const float* mean, const float* var, const float* gamma, const float* beta, int C, int count, float eps,
                                float momentum, int training, float* running_mean, float* running_var, float* scale, float* shift,
                                float* invstd, float* mean_used, spgan_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  SPGAN_CHECK_ARG(scale && shift && invstd && mean_used && C > 0 && count > 0);
  if (training) SPGAN_CHECK_ARG(mean && var && ((running_mean == nullptr) == (running_var == nullptr)));
  else SPGAN_CHECK_ARG(running_mean && running_var);
  hipLaunchKernelGGL(bn_prepare_kernel, dim3(cdiv(C, 128)), dim3(128), 0, s, mean, var, gamma, beta, C, count, eps, momentum, training,
                     running_mean, running_var, scale, shift, invstd, mean_used);
  return spgan_launch_status();
}

extern "C" int spgan_bn_bwd_apply2(const float* g, const float* g2, const float* g2scale, const float* y, int M, int C, const float* mean,
                                   const float* invstd, const float* gamma, const float* sums, int count, float* dy, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(g && g2 && g2scale && y && dy && mean && invstd && sums && M > 0 && C > 0 && C % 4 == 0 && count > 0);
  const size_t total = (size_t)M * C;
  auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  SPGAN_CHECK_ARG(al(g) && al(g2) && al(g2scale) && al(y) && al(dy) && al(mean) && al(invstd) && al(sums) && (!gamma || al(gamma)) &&
                  total / 4 < (1ull << 32));
  hipLaunchKernelGGL(bn_bwd_apply_v4_kernel, dim3(cdiv(total / 4, 256)), dim3(256), 0, (hipStream_t)s_, g, y, C, (unsigned)(total / 4), C, mean,
                     invstd, gamma, sums, 1.0f / (float)count, dy, g2, g2scale);
  return spgan_launch_status();
}

extern "C" int spgan_bn_bwd_apply(const float* g, const float* y, int ld, int M, int C, const float* mean, const float* invstd,
                                  const float* gamma, const float* sums, int count, float* dy, spgan_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  SPGAN_CHECK_ARG(g && y && dy && mean && invstd && sums && M > 0 && C > 0 && ld >= C && count > 0);
  const size_t total = (size_t)M * C;
  auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool v4 = (C % 4 == 0) && (ld % 4 == 0) && al(g) && al(y) && al(dy) && al(mean) && al(invstd) && al(sums) && (!gamma || al(gamma)) &&
                  total / 4 < (1ull << 32);
  if (v4)
    hipLaunchKernelGGL(bn_bwd_apply_v4_kernel, dim3(cdiv(total / 4, 256)), dim3(256), 0, s, g, y, ld, (unsigned)(total / 4), C, mean, invstd, gamma,
                       sums, 1.0f / (float)count, dy, (const float*)nullptr, (const float*)nullptr);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, g, y, ld, (size_t)M, C, mean, invstd, gamma, sums,
                       1.0f / (float)count, dy);
  return spgan_launch_status();
}

extern "C" int spgan_maxpool(const float* y, int ld, int B, int N, int C, const float* scale, const float* shift, float slope, float* out,
                             int32_t* argmax, spgan_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  SPGAN_CHECK_ARG(y && out && B > 0 && N > 0 && C > 0 && ld >= C);
  const bool v4 = (C % 4 == 0) && (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0) &&
                  (!scale || (((reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift)) & 15) == 0));
  if (v4 && (long)B * cdiv(C, 64) < 512 && N >= 512)
    hipLaunchKernelGGL(maxpool_v4_kernel<4>, dim3(B, cdiv(C, 16)), dim3(256), 0, s, y, ld, N, C, scale, shift, slope, out, argmax);
  else if (v4) hipLaunchKernelGGL(maxpool_v4_kernel<16>, dim3(B, cdiv(C, 64)), dim3(256), 0, s, y, ld, N, C, scale, shift, slope, out, argmax);
  else hipLaunchKernelGGL(maxpool_kernel, dim3(B, cdiv(C, 64)), dim3(256), 0, s, y, ld, N, C, scale, shift, slope, out, argmax);
  return spgan_launch_status();
}

// BatchNorm backward as an affine combination of two tensors: dy = p*g + q*y + r (spgan_hip.h: spgan_bn_bwd_coeffs).  One thread per
// channel; the consumers (spgan_gemm_nt_args.A2 / spgan_gemm_tn_args.A2) evaluate the combination on their operand loads.
namespace {
__global__ void bn_bwd_coeffs_kernel(const float* __restrict__ sums, const float* __restrict__ mean, const float* __restrict__ invstd,
                                     const float* __restrict__ gamma, int C, float rcount, float* __restrict__ coef) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float inv = invstd[c];
  const float p = (gamma ? gamma[c] : 1.f) * inv;
  const float q = -(p * inv) * (sums[C + c] * rcount);
  const float r = -(p * (sums[c] * rcount)) - q * mean[c];
  coef[c] = p;
  coef[C + c] = q;
  coef[2 * C + c] = r;
}
}  // namespace

extern "C" int spgan_bn_bwd_coeffs(const float* sums, const float* mean, const float* invstd, const float* gamma, int C, float count, float* coef,
                                   spgan_stream_t s) {
  SPGAN_CHECK_ARG(sums && mean && invstd && coef && C > 0 && count > 0.f);
  hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)s, sums, mean, invstd, gamma, C, 1.0f / count, coef);
  return spgan_launch_status();
}
