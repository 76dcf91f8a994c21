// HBM-bound per-point kernels: AdaIN (AdaptivePointNorm, Generation/Generator.py:24-45) forward and
// backward, the sparse BatchNorm backward behind D's global max-pool (Discriminator.py:77-81,104),
// max-pool gradient routing, tanh backward and the flat Adam update.  Lanes run along channels.
#include "common.hpp"

namespace {

constexpr int RT = 128;  // rows per partial tile (shared partial format, see norm.hip)

// out = gamma * xhat + beta,  xhat = (lrelu(x, slope) - mean[b,c]) * rsqrt(var[b,c] + eps),  [gamma|beta] = gb[m, 0:2C]
__global__ void adain_fwd_kernel(const float* __restrict__ x, size_t M, int C, int N, float slope, const float* __restrict__ imean,
                                 const float* __restrict__ ivar, float eps, const float* __restrict__ gb, float* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * C) return;
  const size_t m = t / C;
  const int c = t % C;
  const size_t b = m / N;
  const float xh = (lrelu_f(x[t], slope) - imean[b * C + c]) * rsqrtf(ivar[b * C + c] + eps);
  out[t] = fmaf(gb[m * 2 * C + c], xh, gb[m * 2 * C + C + c]);
}

// stage 1: dgb = [dout*xhat | dout];  partials over each shape's N rows of (sum dxh, sum dxh*xhat), dxh = dout*gamma
__global__ __launch_bounds__(256) void adain_bwd1_kernel(const float* __restrict__ dout, const float* __restrict__ x, int C, int N, float slope,
                                                         const float* __restrict__ imean, const float* __restrict__ ivar, float eps,
                                                         const float* __restrict__ gb, int tiles_per_group, float* __restrict__ dgb,
                                                         float* __restrict__ part) {
  __shared__ float r0[4][64], r1[4][64];
  const int tile = blockIdx.x;
  const int b = tile / tiles_per_group, q = tile % tiles_per_group;
  const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + lane;
  const bool ok = c < C;
  const int n0 = q * RT, cnt = min(RT, N - n0);
  float s0 = 0.f, s1 = 0.f;
  if (ok) {
    const float mu = imean[(size_t)b * C + c], iv = rsqrtf(ivar[(size_t)b * C + c] + eps);
    for (int r = sl; r < cnt; r += 4) {  // (two stores per row: unrolling this one by 8 made it slower, 30 -> 46 us)
      const size_t m = (size_t)b * N + n0 + r;
      const float d = dout[m * C + c];
      const float xh = (lrelu_f(x[m * C + c], slope) - mu) * iv;
      dgb[m * 2 * C + c] = d * xh;
      dgb[m * 2 * C + C + c] = d;
      const float dxh = d * gb[m * 2 * C + c];
      s0 += dxh;
      s1 = fmaf(dxh, xh, s1);
    }
  }
  r0[sl][lane] = s0; r1[sl][lane] = s1;
  __syncthreads();
  if (sl == 0 && ok) {
    float* o = part + ((size_t)tile * C + c) * 2;
    o[0] = (r0[0][lane] + r0[1][lane]) + (r0[2][lane] + r0[3][lane]);
    o[1] = (r1[0][lane] + r1[1][lane]) + (r1[2][lane] + r1[3][lane]);
  }
}

// stage 2: dx = invstd * (dxh - S0/N - xhat*S1/N) * lrelu'(x)
__global__ void adain_bwd2_kernel(const float* __restrict__ dout, const float* __restrict__ x, size_t M, int C, int N, float slope,
                                  const float* __restrict__ imean, const float* __restrict__ ivar, float eps, const float* __restrict__ gb,
                                  const float* __restrict__ S0, const float* __restrict__ S1, float* __restrict__ dx) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * C) return;
  const size_t m = t / C;
  const int c = t % C;
  const size_t b = m / N;
  const float xv = x[t];
  const float iv = rsqrtf(ivar[b * C + c] + eps);
  const float xh = (lrelu_f(xv, slope) - imean[b * C + c]) * iv;
  const float dxh = dout[t] * gb[m * 2 * C + c];
  const float rn = 1.0f / (float)N;
  dx[t] = iv * (dxh - S0[b * C + c] * rn - xh * (S1[b * C + c] * rn)) * lrelu_mask(xv, slope);
}

// gval = gpool * lrelu'(pooled);  sums[c] = sum_b gval, sums[C+c] = sum_b gval * xhat(argmax row)
// 64 channels x 4 shape-lanes per workgroup: the B gathers of a channel are independent loads spread over four threads
// (one thread per channel walked the shapes one dependent-latency at a time: 25 us for 32 x 1024 values); the four partial
// sums are combined in a fixed order.
__device__ __forceinline__ void pool_bwd_stats_body(int bx, const float* __restrict__ gpool, const float* __restrict__ pooled,
                                                    const int32_t* __restrict__ argmax, const float* __restrict__ y, int ld,
                                                    const float* __restrict__ mean, const float* __restrict__ invstd, float slope, int B,
                                                    int C, float* __restrict__ gval, float* __restrict__ sums,
                                                    const float* __restrict__ gamma, float rM, float* __restrict__ alpha,
                                                    float* __restrict__ beta, float* __restrict__ cg) {
  __shared__ float r0[4][64], r1[4][64];
  const int cl = threadIdx.x & 63, bl = threadIdx.x >> 6;
  const int c = bx * 64 + cl;
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    const float mu = mean[c], iv = invstd[c];
#pragma unroll 4
    for (int b = bl; b < B; b += 4) {
      const float g = gpool[(size_t)b * C + c] * lrelu_mask(pooled[(size_t)b * C + c], slope);
      gval[(size_t)b * C + c] = g;
      const float xh = (y[(size_t)argmax[(size_t)b * C + c] * ld + c] - mu) * iv;
      s0 += g;
      s1 = fmaf(g, xh, s1);
    }
  }
  r0[bl][cl] = s0;
  r1[bl][cl] = s1;
  __syncthreads();
  if (c >= C) return;
  const float t0 = (r0[0][cl] + r0[1][cl]) + (r0[2][cl] + r0[3][cl]);
  const float t1 = (r1[0][cl] + r1[1][cl]) + (r1[2][cl] + r1[3][cl]);
  if (bl == 0) {
    sums[c] = t0;
    sums[C + c] = t1;
  }
  if (alpha) {  // the coefficients of the lazily evaluated BatchNorm backward (sparse_bn_prep_kernel's arithmetic) in the same launch
    const float iv = invstd[c], coef = gamma[c] * iv;
    if (bl == 0) {
      const float al = -(coef * iv) * (t1 * rM);
      alpha[c] = al;
      beta[c] = -(coef * (t0 * rM)) - al * mean[c];
    }
    for (int b = bl; b < B; b += 4) cg[(size_t)b * C + c] = gval[(size_t)b * C + c] * coef;   // this thread wrote gval[b,c] itself
  }
}

__global__ __launch_bounds__(256) void pool_bwd_stats_kernel(const float* __restrict__ gpool, const float* __restrict__ pooled,
                                                             const int32_t* __restrict__ argmax, const float* __restrict__ y, int ld,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd, float slope, int B,
                                                             int C, float* __restrict__ gval, float* __restrict__ sums,
                                                             const float* __restrict__ gamma = nullptr, float rM = 0.f, float* __restrict__ alpha = nullptr,
                                                             float* __restrict__ beta = nullptr, float* __restrict__ cg = nullptr) {
  pool_bwd_stats_body(blockIdx.x, gpool, pooled, argmax, y, ld, mean, invstd, slope, B, C, gval, sums, gamma, rM, alpha, beta, cg);
}

// spgan_pool_bwd_stats_prep for several passes (blockIdx.y) as one launch: the stand-alone kernel's body per pass
struct PoolBwdMulti {
  spgan_pool_bwd_args a[SPGAN_GROUP_MAX];
};
__global__ __launch_bounds__(256) void pool_bwd_stats_multi_kernel(const PoolBwdMulti m) {
  const spgan_pool_bwd_args& q = m.a[blockIdx.y];
  if ((int)blockIdx.x * 64 >= q.C) return;
  pool_bwd_stats_body(blockIdx.x, q.gpool, q.pooled, q.argmax, q.y, q.ld, q.mean, q.invstd, q.slope, q.B, q.C, q.gval, q.sums, q.gamma,
                      1.0f / (float)q.count, q.alpha, q.beta, q.cg);
}

// dy[m,c] = gamma*invstd*( (argmax[b,c]==m ? gval[b,c] : 0) - sums[c]/count - xhat*sums[C+c]/count )
__global__ void bn_bwd_apply_sparse_kernel(const float* __restrict__ gval, const int32_t* __restrict__ argmax, const float* __restrict__ y,
                                           int ld, size_t M, int C, int N, const float* __restrict__ mean, const float* __restrict__ invstd,
                                           const float* __restrict__ gamma, const float* __restrict__ sums, float rcount,
                                           float* __restrict__ dy) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * C) return;
  const size_t m = t / C;
  const int c = t % C;
  const size_t b = m / N;
  const float iv = invstd[c];
  const float xh = (y[m * ld + c] - mean[c]) * iv;
  const float g = ((size_t)argmax[b * C + c] == m) ? gval[b * C + c] : 0.f;
  dy[m * C + c] = gamma[c] * iv * (g - sums[c] * rcount - xh * (sums[C + c] * rcount));
}

__global__ void maxpool_bwd_add_kernel(const float* __restrict__ dpool, const int32_t* __restrict__ argmax, int BC, int C,
                                       float* __restrict__ dst, int ld) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= BC) return;
  const int c = t % C;
  dst[(size_t)argmax[t] * ld + c] += dpool[t];  // (b,c) pairs hit distinct elements: no race
}

__global__ void tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, size_t n, float* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) out[t] = dy[t] * (1.f - y[t] * y[t]);
}

// dpre = dy * act'(y) from the activation OUTPUT y (LeakyReLU in place / tanh)
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, size_t n, int act, float slope, float* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const float yv = y[t];
  float d = dy[t];
  if (act == SPGAN_ACT_LRELU) d *= lrelu_mask(yv, slope);
  else if (act == SPGAN_ACT_TANH) d *= (1.f - yv * yv);
  out[t] = d;
}

// out[M,C] = 0 except out[argmax[b,c], c] = val[b,c]
__global__ void scatter_rows_kernel(const float* __restrict__ val, const int32_t* __restrict__ argmax, int BC, int C, float* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= BC) return;
  out[(size_t)argmax[t] * C + (t % C)] = val[t];
}
__global__ void gather_rows_kernel(const float* __restrict__ src, int ld, const int32_t* __restrict__ argmax, int BC, int C, float* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= BC) return;
  out[t] = src[(size_t)argmax[t] * ld + (t % C)];
}

// WGAN-GP double backward through train-mode BatchNorm: column sums of
//   u, u*xhat, u*gz  ->  partials [tiles][2C][2]: col c -> (sum u, sum u*xhat), col C+c -> (sum u*gz, 0)
__global__ __launch_bounds__(256) void bn_dbl_stats_kernel(const float* __restrict__ u, const float* __restrict__ y, const float* __restrict__ gz,
                                                           int M, int C, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           float* __restrict__ part) {
  __shared__ float r0[4][64], r1[4][64], r2[4][64];
  const int tile = blockIdx.x;
  const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + lane;
  const bool ok = c < C;
  const int m0 = tile * RT, cnt = min(RT, M - m0);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  if (ok) {
    const float mu = mean[c], iv = invstd[c];
#pragma unroll 8
    for (int r = sl; r < cnt; r += 4) {
      const size_t o = (size_t)(m0 + r) * C + c;
      const float uv = u[o];
      s0 += uv;
      s1 = fmaf(uv, (y[o] - mu) * iv, s1);
      s2 = fmaf(uv, gz[o], s2);
    }
  }
  r0[sl][lane] = s0; r1[sl][lane] = s1; r2[sl][lane] = s2;
  __syncthreads();
  if (sl == 0 && ok) {
    float* o = part + (size_t)tile * (2 * C) * 2;
    o[(size_t)c * 2 + 0] = (r0[0][lane] + r0[1][lane]) + (r0[2][lane] + r0[3][lane]);
    o[(size_t)c * 2 + 1] = (r1[0][lane] + r1[1][lane]) + (r1[2][lane] + r1[3][lane]);
    o[(size_t)(C + c) * 2 + 0] = (r2[0][lane] + r2[1][lane]) + (r2[2][lane] + r2[3][lane]);
    o[(size_t)(C + c) * 2 + 1] = 0.f;
  }
}

//   q     = gamma*invstd*(u - U0/M - xhat*U1/M) * lrelu'(z),  z = y*scale + shift      (adjoint handed to the next layer)
//   xbarA = -(gamma*invstd/M) * (u*S1 + gz*U1)                                          (adjoint deposited on xhat)
__global__ void bn_dbl_apply_kernel(const float* __restrict__ u, const float* __restrict__ y, const float* __restrict__ gz, size_t M, int C,
                                    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ scale,
                                    const float* __restrict__ shift, float slope, const float* __restrict__ gamma, const float* __restrict__ S1,
                                    const float* __restrict__ U0, const float* __restrict__ U1, float rM, float* __restrict__ q,
                                    float* __restrict__ xbar) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * C) return;
  const int c = t % C;
  const float yv = y[t], uv = u[t];
  const float iv = invstd[c];
  const float xh = (yv - mean[c]) * iv;
  const float gs = gamma[c] * iv;
  const float z = fmaf(yv, scale[c], shift[c]);
  q[t] = gs * (uv - U0[c] * rM - xh * (U1[c] * rM)) * lrelu_mask(z, slope);
  xbar[t] = -(gs * rM) * (uv * S1[c] + gz[t] * U1[c]);
}

// Per-channel scalar algebra of the double backward, one launch each instead of ~15 elementwise ones.
//   phase A: core = Ugz - (U0*S0 + U1*S1)/M;  out[0]=dgammaA = inv*core;  out[1]=sbarA = gamma*core;
//            out[2]=xsum0 = -(gamma*inv/M)*(U0*S1 + S0*U1);  out[3]=xsum1 = -2*(gamma*inv/M)*U1*S1
__global__ void bn_dbl_coeffs_kernel(const float* U0, const float* U1, const float* Ugz, const float* S0, const float* S1, const float* gamma,
                                     const float* inv, int C, float rM, float* out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float core = Ugz[c] - (U0[c] * S0[c] + U1[c] * S1[c]) * rM;
  const float gsM = gamma[c] * inv[c] * rM;
  out[c] = inv[c] * core;
  out[C + c] = gamma[c] * core;
  out[2 * C + c] = -gsM * (U0[c] * S1[c] + S0[c] * U1[c]);
  out[3 * C + c] = -2.0f * gsM * (U1[c] * S1[c]);
}
//   phase B: sums[0:C] = xsum0 + gamma*s0;  sums[C:2C] = xsum1 + gamma*s1 + inv*sbarA;  dgamma = dgammaA + s1   (s0/s1 NULL: zeros)
__global__ void bn_dbl_phaseb_kernel(const float* coeffs, const float* gamma, const float* inv, const float* s0, const float* s1, int C,
                                     float* sums, float* dgamma) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float a0 = s0 ? s0[c] : 0.f, a1 = s1 ? s1[c] : 0.f;
  sums[c] = coeffs[2 * C + c] + gamma[c] * a0;
  sums[C + c] = coeffs[3 * C + c] + gamma[c] * a1 + inv[c] * coeffs[C + c];
  dgamma[c] = coeffs[c] + a1;
}
// the two above in one launch: phase B's sums straight from phase A's per-channel sums (the coefficient vectors are not stored)
__global__ void bn_dbl_phaseb_sums_kernel(const float* U0, const float* U1, const float* Ugz, const float* S0, const float* S1, const float* gamma,
                                          const float* inv, const float* s0, const float* s1, int C, float rM, float* sums, float* dgamma) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float core = Ugz[c] - (U0[c] * S0[c] + U1[c] * S1[c]) * rM;
  const float gsM = gamma[c] * inv[c] * rM;
  const float k0 = inv[c] * core, k1 = gamma[c] * core, k2 = -gsM * (U0[c] * S1[c] + S0[c] * U1[c]), k3 = -2.0f * gsM * (U1[c] * S1[c]);
  const float a0 = s0 ? s0[c] : 0.f, a1 = s1 ? s1[c] : 0.f;
  sums[c] = k2 + gamma[c] * a0;
  sums[C + c] = k3 + gamma[c] * a1 + inv[c] * k1;
  dgamma[c] = k0 + a1;
}
// ---- collapsed double backward of the layer in front of the max-pool: everything that used to need the dense
// [M,C] tensors u = q.W^T, y, gz and q_out is only needed (a) as per-channel sums and (b) at the B*C arg-max positions.
// out[b,c] = Q[arg[b,c], :] . W[c, :]     (u at the arg-max rows); 16 lanes per (b,c), fixed shuffle tree
__global__ __launch_bounds__(256) void gather_rowdot_kernel(const float* __restrict__ Q, int ldq, const int32_t* __restrict__ arg,
                                                            const float* __restrict__ W, int ldw, int BC, int C, int K, float* __restrict__ out) {
  const int pair = blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15;
  float s = 0.f;
  if (pair < BC) {
    const float* q = Q + (size_t)arg[pair] * ldq;
    const float* w = W + (size_t)(pair % C) * ldw;
    for (int k = l; k < K; k += 16) s = fmaf(q[k], w[k], s);
  }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
  if (pair < BC && l == 0) out[pair] = s;
}
// out[r] = X[r,:] . Y[r,:]
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ Y, int ldy, int R, int K,
                                                     float* __restrict__ out) {
  const int r = blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15;
  float s = 0.f;
  if (r < R)
    for (int k = l; k < K; k += 16) s = fmaf(X[(size_t)r * ldx + k], Y[(size_t)r * ldy + k], s);
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
  if (r < R && l == 0) out[r] = s;
}
// The three independent dot-product launches of the collapsed double backward's phase A in one grid: quad[c] = W[c,:].T[c,:] and
// U0[c] = W[c,:].cq (the first C/16 workgroups, 16 lanes per row) and uarg[b,c] = Q[arg[b,c],:].W[c,:] (gather_rowdot_kernel's arithmetic)
__global__ __launch_bounds__(256) void dbl_top_dots_kernel(const float* __restrict__ Q, int ldq, const int32_t* __restrict__ arg,
                                                           const float* __restrict__ W, int ldw, const float* __restrict__ T, int ldt,
                                                           const float* __restrict__ cq, int BC, int C, int K, int row_blocks,
                                                           float* __restrict__ uarg, float* __restrict__ quad, float* __restrict__ U0) {
  const int l = threadIdx.x & 15;
  if ((int)blockIdx.x < row_blocks) {
    const int r = blockIdx.x * 16 + (threadIdx.x >> 4);
    float s = 0.f, u = 0.f;
    if (r < C)
      for (int k = l; k < K; k += 16) {
        const float w = W[(size_t)r * ldw + k];
        s = fmaf(w, T[(size_t)r * ldt + k], s);
        u = fmaf(w, cq[k], u);
      }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { s += __shfl_xor(s, o); u += __shfl_xor(u, o); }
    if (r < C && l == 0) { quad[r] = s; U0[r] = u; }
    return;
  }
  const int pair = ((int)blockIdx.x - row_blocks) * 16 + (threadIdx.x >> 4);
  float s = 0.f;
  if (pair < BC) {
    const float* q = Q + (size_t)arg[pair] * ldq;
    const float* w = W + (size_t)(pair % C) * ldw;
    for (int k = l; k < K; k += 16) s = fmaf(q[k], w[k], s);
  }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
  if (pair < BC && l == 0) uarg[pair] = s;
}
// One thread per channel, loop over the B shapes.  With u = q.W^T, xhat = (y-mean)*inv, gz = scatter(gval):
//   U0 = sum_m u (given), U1 = sum_m u*xhat = inv*(quad + (bias-mean)*U0)   (quad[c] = w_c^T (q^T a) w_c), Ugz = sum_b gval*uarg
//   phase A coefficients as bn_dbl_coeffs; top adjoint t[b,c] = gamma*inv*(uarg - U0/M - xhat_arg*U1/M)*lrelu'(pooled)
//   phase B: ybar = c1*u + c2*y + c3 + scatter(spB)  with sum0 = xsum0, sum1 = xsum1 + inv*sbarA:
//     c1 = -gamma*inv^2*S1/M,  c2 = -inv^2*sum1/M,  c3 = -inv*sum0/M + inv^2*mean*sum1/M,  spB = -(gamma*inv^2/M)*U1*gval
__global__ __launch_bounds__(256) void bn_dbl_pool_kernel(const float* __restrict__ uarg, const float* __restrict__ gval,
                                                          const float* __restrict__ yarg, const float* __restrict__ pooled,
                                                          const float* __restrict__ U0, const float* __restrict__ quad,
                                                          const float* __restrict__ bias, const float* __restrict__ mean,
                                                          const float* __restrict__ inv, const float* __restrict__ gamma,
                                                          const float* __restrict__ S0, const float* __restrict__ S1, int B, int C, float rM,
                                                          float slope, float* __restrict__ t, float* __restrict__ spB, float* __restrict__ out4) {
  // 64 channels x 4 shape-lanes per workgroup (see pool_bwd_stats_kernel)
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, bl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const bool ok = c < C;
  float part = 0.f;
  if (ok)
    for (int b = bl; b < B; b += 4) part = fmaf(gval[(size_t)b * C + c], uarg[(size_t)b * C + c], part);
  red[bl][cl] = part;
  __syncthreads();
  if (!ok) return;
  const float ugz = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
  const float iv = inv[c], ga = gamma[c], mu = mean[c], u0 = U0[c];
  const float u1 = iv * (quad[c] + (bias[c] - mu) * u0);
  const float core = ugz - (u0 * S0[c] + u1 * S1[c]) * rM;
  const float gsM = ga * iv * rM;
  const float sbarA = ga * core;
  const float sum0 = -gsM * (u0 * S1[c] + S0[c] * u1);
  const float sum1 = -2.0f * gsM * (u1 * S1[c]) + iv * sbarA;
  if (bl == 0) {
    out4[c] = iv * core;                                       // dgamma
    out4[C + c] = -gsM * iv * S1[c];                           // c1
    out4[2 * C + c] = -iv * iv * sum1 * rM;                    // c2
    out4[3 * C + c] = -iv * sum0 * rM + iv * iv * mu * sum1 * rM;  // c3
  }
  const float gs = ga * iv, spc = -gsM * iv * u1;
  for (int b = bl; b < B; b += 4) {
    const size_t i = (size_t)b * C + c;
    const float xh = (yarg[i] - mu) * iv;
    t[i] = gs * (uarg[i] - u0 * rM - xh * (u1 * rM)) * lrelu_mask(pooled[i], slope);
    spB[i] = spc * gval[i];
  }
}

// BatchNorm backward behind the max-pool as a lazy operand: alpha = -(gamma*inv)*inv*S1/M, beta = -(gamma*inv)*S0/M - alpha*mean,
// cg[b,c] = gamma*inv*gval[b,c]
__global__ __launch_bounds__(256) void sparse_bn_prep_kernel(const float* gval, const float* mean, const float* inv, const float* gamma,
                                                             const float* sums, int B, int C, float rM, float* alpha, float* beta, float* cg) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), bl = threadIdx.x >> 6;
  if (c >= C) return;
  const float coef = gamma[c] * inv[c];
  if (bl == 0) {
    const float al = -(coef * inv[c]) * (sums[C + c] * rM);
    alpha[c] = al;
    beta[c] = -(coef * (sums[c] * rM)) - al * mean[c];
  }
  for (int b = bl; b < B; b += 4) cg[(size_t)b * C + c] = gval[(size_t)b * C + c] * coef;
}

// out = a + gamma[c]*b
__global__ void col_scale_add_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ gamma, size_t M, int C,
                                     float* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < M * C) out[t] = fmaf(gamma[t % C], b[t], a[t]);
}

// out[m,c] = lrelu(X[m,c]*scale[c] + shift[c], slope): a train-mode BatchNorm + LeakyReLU output materialised
// (Discriminator.py:57-64); 4 channels per thread when the layout allows.
template <int VEC>
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ X, int ldx, size_t M, int C, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float slope, float* __restrict__ out) {
  const size_t per = (size_t)(C / VEC);
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * per) return;
  const size_t m = t / per;
  const int c = (int)(t % per) * VEC;
  if (VEC == 4) {
    const float4 v = *reinterpret_cast<const float4*>(X + m * ldx + c);
    const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
    float4 o;
    o.x = lrelu_f(fmaf(v.x, sc.x, sh.x), slope);
    o.y = lrelu_f(fmaf(v.y, sc.y, sh.y), slope);
    o.z = lrelu_f(fmaf(v.z, sc.z, sh.z), slope);
    o.w = lrelu_f(fmaf(v.w, sc.w, sh.w), slope);
    *reinterpret_cast<float4*>(out + m * C + c) = o;
  } else {
    out[m * C + c] = lrelu_f(fmaf(X[m * ldx + c], scale[c], shift[c]), slope);
  }
}

// out[r,c] = a[r]*X[r,c] + (a[r]*b[r] + d[r])*v[c]   (v == NULL: first term only).  Small [R,C] weight-shaped tensors.
__global__ void rowscale_outer_kernel(const float* __restrict__ X, int ldx, int R, int C, const float* __restrict__ a, const float* __restrict__ b,
                                      const float* __restrict__ d, const float* __restrict__ v, float* __restrict__ out, int ldo, int accumulate) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= R * C) return;
  const int r = t / C, c = t % C;
  float o = a[r] * X[(size_t)r * ldx + c];
  if (v) o = fmaf(fmaf(a[r], b[r], d[r]), v[c], o);
  float* q = out + (size_t)r * ldo + c;
  *q = accumulate ? *q + o : o;
}

__global__ void axpby_kernel(float a, const float* __restrict__ x, float b, float* __restrict__ y, size_t n) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) y[t] = a * x[t] + (b == 0.f ? 0.f : b * y[t]);
}

// torch.optim.Adam (no weight decay / amsgrad): Generation/model.py:94-97
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                            float lr, float b1, float b2, float eps, float bc1, float rsqrt_bc2, float gscale) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const float gr = g[t] * gscale;
  const float mm = b1 * m[t] + (1.f - b1) * gr;
  const float vv = b2 * v[t] + (1.f - b2) * gr * gr;
  m[t] = mm;
  v[t] = vv;
  const float denom = sqrtf(vv) * rsqrt_bc2 + eps;
  p[t] -= (lr / bc1) * (mm / denom);
}

// Capturable variant (hipGraph replay): the step count lives in device memory.  state[0] = step (as float bits of an int),
// state[1] = 1 - beta1^step, state[2] = 1/sqrt(1 - beta2^step); adam_prep advances it once per optimiser step.
__global__ void adam_prep_kernel(float beta1, float beta2, float* __restrict__ state) {
  int* st = reinterpret_cast<int*>(state);
  const int t = st[0] + 1;
  st[0] = t;
  state[1] = (float)(1.0 - pow((double)beta1, (double)t));
  state[2] = (float)(1.0 / sqrt(1.0 - pow((double)beta2, (double)t)));
}
__global__ void adam_dev_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                                float lr, float b1, float b2, float eps, const float* __restrict__ state, float gscale, int zero_grad) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const float bc1 = state[1], rsqrt_bc2 = state[2];
  lr *= state[3];  // learning-rate multiplier in device memory: a schedule changes it without re-capturing the step
  const float gr = g[t] * gscale;
  if (zero_grad) g[t] = 0.f;   // optimizer.zero_grad() of the NEXT step, for free (the gradient is in a register anyway)
  const float mm = b1 * m[t] + (1.f - b1) * gr;
  const float vv = b2 * v[t] + (1.f - b2) * gr * gr;
  m[t] = mm;
  v[t] = vv;
  const float denom = sqrtf(vv) * rsqrt_bc2 + eps;
  p[t] -= (lr / bc1) * (mm / denom);
}

// Device timestamps INSIDE a replayed hipGraph (HIP events cannot be queried there): a one-thread launch in front of the kernel under
// measurement stores the constant-rate wall clock (s_memrealtime; hipDeviceAttributeWallClockRate), a one-thread launch behind it adds
// the elapsed ticks to an accumulator.  Stream order makes the pair bracket exactly that kernel (plus two launch boundaries: an EMPTY
// pair in the same graph measures those).  Measurement plumbing of bench.py's `roofline` entry -- no product path calls it.
__global__ void stamp_begin_kernel(unsigned long long* slot) { *slot = wall_clock64(); }
__global__ void stamp_end_kernel(const unsigned long long* slot, unsigned long long* acc) {
  const unsigned long long t = wall_clock64();
  acc[0] += t - *slot;
  acc[1] += 1ull;
}

}  // namespace

extern "C" int spgan_stamp_begin(uint64_t* slot, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(slot);
  hipLaunchKernelGGL(stamp_begin_kernel, dim3(1), dim3(1), 0, (hipStream_t)s_, reinterpret_cast<unsigned long long*>(slot));
  return spgan_launch_status();
}

extern "C" int spgan_stamp_end(const uint64_t* slot, uint64_t* acc2, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(slot && acc2);
  hipLaunchKernelGGL(stamp_end_kernel, dim3(1), dim3(1), 0, (hipStream_t)s_, reinterpret_cast<const unsigned long long*>(slot),
                     reinterpret_cast<unsigned long long*>(acc2));
  return spgan_launch_status();
}

extern "C" int spgan_wall_clock_khz(void) {
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return 0;
  return khz;
}

extern "C" int spgan_adain_fwd(const float* x, int M, int C, int N, float slope, const float* imean, const float* ivar, float eps,
                               const float* gb, float* out, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(x && imean && ivar && gb && out && M > 0 && C > 0 && N > 0 && M % N == 0);
  const size_t total = (size_t)M * C;
  hipLaunchKernelGGL(adain_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)s_, x, (size_t)M, C, N, slope, imean, ivar, eps, gb, out);
  return spgan_launch_status();
}

extern "C" int spgan_adain_bwd1(const float* dout, const float* x, int M, int C, int N, float slope, const float* imean, const float* ivar,
                                float eps, const float* gb, float* dgb, float* partials, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(dout && x && imean && ivar && gb && dgb && partials && M > 0 && C > 0 && N > 0 && M % N == 0);
  const int tpg = cdiv(N, RT);
  hipLaunchKernelGGL(adain_bwd1_kernel, dim3((M / N) * tpg, cdiv(C, 64)), dim3(256), 0, (hipStream_t)s_, dout, x, C, N, slope, imean, ivar, eps,
                     gb, tpg, dgb, partials);
  return spgan_launch_status();
}

extern "C" int spgan_adain_bwd2(const float* dout, const float* x, int M, int C, int N, float slope, const float* imean, const float* ivar,
                                float eps, const float* gb, const float* S0, const float* S1, float* dx, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(dout && x && imean && ivar && gb && S0 && S1 && dx && M > 0 && C > 0 && N > 0 && M % N == 0);
  const size_t total = (size_t)M * C;
  hipLaunchKernelGGL(adain_bwd2_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)s_, dout, x, (size_t)M, C, N, slope, imean, ivar, eps,
                     gb, S0, S1, dx);
  return spgan_launch_status();
}

extern "C" int spgan_pool_bwd_stats(const float* gpool, const float* pooled, const int32_t* argmax, const float* y, int ld, const float* mean,
                                    const float* invstd, float slope, int B, int C, float* gval, float* sums, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(gpool && pooled && argmax && y && mean && invstd && gval && sums && B > 0 && C > 0 && ld >= C);
  hipLaunchKernelGGL(pool_bwd_stats_kernel, dim3(cdiv(C, 64)), dim3(256), 0, (hipStream_t)s_, gpool, pooled, argmax, y, ld, mean, invstd, slope, B,
                     C, gval, sums, (const float*)nullptr, 0.f, (float*)nullptr, (float*)nullptr, (float*)nullptr);
  return spgan_launch_status();
}

extern "C" int spgan_pool_bwd_stats_prep(const float* gpool, const float* pooled, const int32_t* argmax, const float* y, int ld, const float* mean,
                                         const float* invstd, float slope, int B, int C, const float* gamma, int count, float* gval, float* sums,
                                         float* alpha, float* beta, float* cg, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(gpool && pooled && argmax && y && mean && invstd && gval && sums && gamma && alpha && beta && cg && B > 0 && C > 0 && ld >= C &&
                  count > 0);
  hipLaunchKernelGGL(pool_bwd_stats_kernel, dim3(cdiv(C, 64)), dim3(256), 0, (hipStream_t)s_, gpool, pooled, argmax, y, ld, mean, invstd, slope, B,
                     C, gval, sums, gamma, 1.0f / (float)count, alpha, beta, cg);
  return spgan_launch_status();
}

extern "C" int spgan_pool_bwd_stats_prep_multi(const spgan_pool_bwd_args* a, int count, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(a && count >= 1 && count <= SPGAN_GROUP_MAX);
  PoolBwdMulti m;
  int cmax = 0;
  for (int g = 0; g < count; ++g) {
    const spgan_pool_bwd_args& q = a[g];
    SPGAN_CHECK_ARG(q.gpool && q.pooled && q.argmax && q.y && q.mean && q.invstd && q.gval && q.sums && q.gamma && q.alpha && q.beta && q.cg &&
                    q.B > 0 && q.C > 0 && q.ld >= q.C && q.count > 0);
    m.a[g] = q;
    if (q.C > cmax) cmax = q.C;
  }
  hipLaunchKernelGGL(pool_bwd_stats_multi_kernel, dim3(cdiv(cmax, 64), count), dim3(256), 0, (hipStream_t)s_, m);
  return spgan_launch_status();
}

extern "C" int spgan_bn_bwd_apply_sparse(const float* gval, const int32_t* argmax, const float* y, int ld, int M, int C, int N,
                                         const float* mean, const float* invstd, const float* gamma, const float* sums, int count, float* dy,
                                         spgan_stream_t s_) {
  SPGAN_CHECK_ARG(gval && argmax && y && mean && invstd && gamma && sums && dy && M > 0 && C > 0 && N > 0 && M % N == 0 && ld >= C && count > 0);
  const size_t total = (size_t)M * C;
  hipLaunchKernelGGL(bn_bwd_apply_sparse_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)s_, gval, argmax, y, ld, (size_t)M, C, N,
                     mean, invstd, gamma, sums, 1.0f / (float)count, dy);
  return spgan_launch_status();
}

extern "C" int spgan_maxpool_bwd_add(const float* dpool, const int32_t* argmax, int B, int C, float* dst, int ld, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(dpool && argmax && dst && B > 0 && C > 0 && ld >= C);
  hipLaunchKernelGGL(maxpool_bwd_add_kernel, dim3(cdiv(B * C, 256)), dim3(256), 0, (hipStream_t)s_, dpool, argmax, B * C, C, dst, ld);
  return spgan_launch_status();
}

extern "C" int spgan_tanh_bwd(const float* dy, const float* y, size_t n, float* out, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(dy && y && out && n > 0);
  hipLaunchKernelGGL(tanh_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)s_, dy, y, n, out);
  return spgan_launch_status();
}

extern "C" int spgan_act_bwd(const float* dy, const float* y, size_t n, int act, float slope, float* out, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(dy && y && out && n > 0);
  hipLaunchKernelGGL(act_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)s_, dy, y, n, act, slope, out);
  return spgan_launch_status();
}

extern "C" int spgan_scatter_rows(const float* val, const int32_t* argmax, int B, int C, int M, float* out, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(val && argmax && out && B > 0 && C > 0 && M > 0);
  hipError_t e = hipMemsetAsync(out, 0, (size_t)M * C * sizeof(float), (hipStream_t)s_);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv(B * C, 256)), dim3(256), 0, (hipStream_t)s_, val, argmax, B * C, C, out);
  return spgan_launch_status();
}
extern "C" int spgan_gather_rows(const float* src, int ld, const int32_t* argmax, int B, int C, float* out, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(src && argmax && out && B > 0 && C > 0 && ld >= C);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(B * C, 256)), dim3(256), 0, (hipStream_t)s_, src, ld, argmax, B * C, C, out);
  return spgan_launch_status();
}

extern "C" int spgan_bn_dbl_stats(const float* u, const float* y, const float* gz, int M, int C, const float* mean, const float* invstd,
                                  float* partials, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(u && y && gz && mean && invstd && partials && M > 0 && C > 0);
  hipLaunchKernelGGL(bn_dbl_stats_kernel, dim3(cdiv(M, RT), cdiv(C, 64)), dim3(256), 0, (hipStream_t)s_, u, y, gz, M, C, mean, invstd, partials);
  return spgan_launch_status();
}
extern "C" int spgan_bn_dbl_apply(const float* u, const float* y, const float* gz, int M, int C, const float* mean, const float* invstd,
                                  const float* scale, const float* shift, float slope, const float* gamma, const float* S1, const float* U0,
                                  const float* U1, float* q, float* xbar, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(u && y && gz && mean && invstd && scale && shift && gamma && S1 && U0 && U1 && q && xbar && M > 0 && C > 0);
  const size_t total = (size_t)M * C;
  hipLaunchKernelGGL(bn_dbl_apply_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)s_, u, y, gz, (size_t)M, C, mean, invstd, scale,
                     shift, slope, gamma, S1, U0, U1, 1.0f / (float)M, q, xbar);
  return spgan_launch_status();
}
extern "C" int spgan_bn_dbl_coeffs(const float* U0, const float* U1, const float* Ugz, const float* S0, const float* S1, const float* gamma,
                                   const float* invstd, int C, int count, float* out4C, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(U0 && U1 && Ugz && S0 && S1 && gamma && invstd && out4C && C > 0 && count > 0);
  hipLaunchKernelGGL(bn_dbl_coeffs_kernel, dim3(cdiv(C, 128)), dim3(128), 0, (hipStream_t)s_, U0, U1, Ugz, S0, S1, gamma, invstd, C,
                     1.0f / (float)count, out4C);
  return spgan_launch_status();
}
extern "C" int spgan_bn_dbl_phaseb(const float* coeffs4C, const float* gamma, const float* invstd, const float* s0, const float* s1, int C,
                                   float* sums2C, float* dgamma, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(coeffs4C && gamma && invstd && sums2C && dgamma && C > 0 && ((s0 == nullptr) == (s1 == nullptr)));
  hipLaunchKernelGGL(bn_dbl_phaseb_kernel, dim3(cdiv(C, 128)), dim3(128), 0, (hipStream_t)s_, coeffs4C, gamma, invstd, s0, s1, C, sums2C, dgamma);
  return spgan_launch_status();
}
extern "C" int spgan_bn_dbl_phaseb_sums(const float* U0, const float* U1, const float* Ugz, const float* S0, const float* S1, const float* gamma,
                                        const float* invstd, const float* s0, const float* s1, int C, int count, float* sums2C, float* dgamma,
                                        spgan_stream_t s_) {
  SPGAN_CHECK_ARG(U0 && U1 && Ugz && S0 && S1 && gamma && invstd && sums2C && dgamma && C > 0 && count > 0 && ((s0 == nullptr) == (s1 == nullptr)));
  hipLaunchKernelGGL(bn_dbl_phaseb_sums_kernel, dim3(cdiv(C, 128)), dim3(128), 0, (hipStream_t)s_, U0, U1, Ugz, S0, S1, gamma, invstd, s0, s1, C,
                     1.0f / (float)count, sums2C, dgamma);
  return spgan_launch_status();
}
extern "C" int spgan_gather_rowdot(const float* Q, int ldq, const int32_t* arg, const float* W, int ldw, int B, int C, int K, float* out,
                                   spgan_stream_t s_) {
  SPGAN_CHECK_ARG(Q && arg && W && out && B > 0 && C > 0 && K > 0 && ldq >= K && ldw >= K);
  hipLaunchKernelGGL(gather_rowdot_kernel, dim3(cdiv(B * C, 16)), dim3(256), 0, (hipStream_t)s_, Q, ldq, arg, W, ldw, B * C, C, K, out);
  return spgan_launch_status();
}
extern "C" int spgan_dbl_top_dots(const float* Q, int ldq, const int32_t* arg, const float* W, int ldw, const float* T, int ldt, const float* cq,
                                  int B, int C, int K, float* uarg, float* quad, float* U0, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(Q && arg && W && T && cq && uarg && quad && U0 && B > 0 && C > 0 && K > 0 && ldq >= K && ldw >= K && ldt >= K);
  const int row_blocks = cdiv(C, 16);
  hipLaunchKernelGGL(dbl_top_dots_kernel, dim3(row_blocks + cdiv(B * C, 16)), dim3(256), 0, (hipStream_t)s_, Q, ldq, arg, W, ldw, T, ldt, cq, B * C, C, K,
                     row_blocks, uarg, quad, U0);
  return spgan_launch_status();
}
extern "C" int spgan_rowdot(const float* X, int ldx, const float* Y, int ldy, int R, int K, float* out, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(X && Y && out && R > 0 && K > 0 && ldx >= K && ldy >= K);
  hipLaunchKernelGGL(rowdot_kernel, dim3(cdiv(R, 16)), dim3(256), 0, (hipStream_t)s_, X, ldx, Y, ldy, R, K, out);
  return spgan_launch_status();
}
extern "C" int spgan_bn_dbl_pool(const float* uarg, const float* gval, const float* yarg, const float* pooled, const float* U0, const float* quad,
                                 const float* bias, const float* mean, const float* invstd, const float* gamma, const float* S0, const float* S1,
                                 int B, int C, int count, float slope, float* t, float* spB, float* out4C, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(uarg && gval && yarg && pooled && U0 && quad && bias && mean && invstd && gamma && S0 && S1 && t && spB && out4C);
  SPGAN_CHECK_ARG(B > 0 && C > 0 && count > 0);
  hipLaunchKernelGGL(bn_dbl_pool_kernel, dim3(cdiv(C, 64)), dim3(256), 0, (hipStream_t)s_, uarg, gval, yarg, pooled, U0, quad, bias, mean, invstd,
                     gamma, S0, S1, B, C, 1.0f / (float)count, slope, t, spB, out4C);
  return spgan_launch_status();
}
extern "C" int spgan_sparse_bn_prep(const float* gval, const float* mean, const float* invstd, const float* gamma, const float* sums, int B,
                                    int C, int count, float* alpha, float* beta, float* cg, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(gval && mean && invstd && gamma && sums && alpha && beta && cg && B > 0 && C > 0 && count > 0);
  hipLaunchKernelGGL(sparse_bn_prep_kernel, dim3(cdiv(C, 64)), dim3(256), 0, (hipStream_t)s_, gval, mean, invstd, gamma, sums, B, C,
                     1.0f / (float)count, alpha, beta, cg);
  return spgan_launch_status();
}

extern "C" int spgan_col_scale_add(const float* a, const float* b, const float* gamma, int M, int C, float* out, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(a && b && gamma && out && M > 0 && C > 0);
  const size_t total = (size_t)M * C;
  hipLaunchKernelGGL(col_scale_add_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)s_, a, b, gamma, (size_t)M, C, out);
  return spgan_launch_status();
}

// dst[t][i] += src[t][i] for up to SPGAN_MULTI_MAX tensors in one launch (parameter-gradient accumulation into the
// flat gradient buffer: replaces one elementwise launch per parameter tensor).
// (16-byte accesses when both pointers and the length allow, two of them in flight per thread: the launch lasts as long as its LARGEST
// pair -- 0.5 M elements for D's first head layer -- and a thread's trips are a chain of load latencies)
__device__ __forceinline__ void add_span(float* __restrict__ d, const float* __restrict__ s, int n, int gtid, int gthreads) {
  if ((((uintptr_t)d | (uintptr_t)s) & 15) == 0 && (n & 3) == 0) {
    float4* d4 = reinterpret_cast<float4*>(d);
    const float4* s4 = reinterpret_cast<const float4*>(s);
    const int n4 = n >> 2;
    int i = gtid;
    for (; i + gthreads < n4; i += 2 * gthreads) {
      float4 x0 = d4[i], x1 = d4[i + gthreads];
      const float4 y0 = s4[i], y1 = s4[i + gthreads];
      x0.x += y0.x; x0.y += y0.y; x0.z += y0.z; x0.w += y0.w;
      x1.x += y1.x; x1.y += y1.y; x1.z += y1.z; x1.w += y1.w;
      d4[i] = x0; d4[i + gthreads] = x1;
    }
    if (i < n4) {
      float4 x0 = d4[i];
      const float4 y0 = s4[i];
      x0.x += y0.x; x0.y += y0.y; x0.z += y0.z; x0.w += y0.w;
      d4[i] = x0;
    }
    return;
  }
  for (int i = gtid; i < n; i += gthreads) d[i] += s[i];
}

__global__ void multi_add_kernel(const spgan_multi_add_args a) {
  const int t = blockIdx.y;
  add_span(a.dst[t], a.src[t], a.n[t], blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// dst[t][i] = src[t][i]: the same argument block, assignment instead of accumulation (a train step's input tensors copied into the
// static buffers of its captured graph with one launch instead of one memcpy per tensor)
__global__ void multi_copy_kernel(const spgan_multi_add_args a) {
  const int t = blockIdx.y;
  const int n = a.n[t];
  float* __restrict__ d = a.dst[t];
  const float* __restrict__ s = a.src[t];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) d[i] = s[i];
}

extern "C" int spgan_multi_copy(const spgan_multi_add_args* a, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(a && a->count > 0 && a->count <= SPGAN_MULTI_MAX);
  int nmax = 0;
  for (int t = 0; t < a->count; ++t) {
    SPGAN_CHECK_ARG(a->dst[t] && a->src[t] && a->n[t] > 0);
    nmax = a->n[t] > nmax ? a->n[t] : nmax;
  }
  int bx = cdiv(nmax, 256 * 4);
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(multi_copy_kernel, dim3(bx, a->count), dim3(256), 0, (hipStream_t)s_, *a);
  return spgan_launch_status();
}

// The same accumulation for pairs that are 3-D strided views of each other's shape [n0, n1, n2]: a permuted source (the conv_out weight
// gradient leaves its GEMM as [F, k, F] and belongs into [F, F, 1, k]), a column slice of the destination (a weight whose gradient is
// computed in two column blocks).  Plain pairs have n1 = n2 = 1.
__global__ void multi_add3_kernel(const spgan_multi_add3_args a) {
  const int t = blockIdx.y;
  const int n = a.n[t], n1 = a.n1[t], n2 = a.n2[t];
  float* __restrict__ d = a.dst[t];
  const float* __restrict__ s = a.src[t];
  if (n1 == 1 && n2 == 1) {
    add_span(d, s, n, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
    return;
  }
  const long ds0 = a.ds[0][t], ds1 = a.ds[1][t], ds2 = a.ds[2][t], ss0 = a.ss[0][t], ss1 = a.ss[1][t], ss2 = a.ss[2][t];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int i2 = i % n2, r = i / n2, i1 = r % n1, i0 = r / n1;
    d[i0 * ds0 + i1 * ds1 + i2 * ds2] += s[i0 * ss0 + i1 * ss1 + i2 * ss2];
  }
}

extern "C" int spgan_multi_add3(const spgan_multi_add3_args* a, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(a && a->count > 0 && a->count <= SPGAN_MULTI_MAX);
  int nmax = 0;
  for (int t = 0; t < a->count; ++t) {
    SPGAN_CHECK_ARG(a->dst[t] && a->src[t] && a->n[t] > 0 && a->n1[t] > 0 && a->n2[t] > 0 && a->n[t] % (a->n1[t] * a->n2[t]) == 0);
    nmax = a->n[t] > nmax ? a->n[t] : nmax;
  }
  int bx = cdiv(nmax, 256 * 4);
  if (bx > 128) bx = 128;
  hipLaunchKernelGGL(multi_add3_kernel, dim3(bx, a->count), dim3(256), 0, (hipStream_t)s_, *a);
  return spgan_launch_status();
}

// dst[t] = ((dst[t] + src[0][t]) + src[1][t]) + src[2][t] (nsrc[t] of them, in this order): the parameter gradients of several passes that one grouped
// backward produced separately, accumulated in ONE launch with the sums of as many successive spgan_multi_add launches (same order per element)
__global__ void multi_addn_kernel(const spgan_multi_addn_args a) {
  const int t = blockIdx.y;
  const int n = a.n[t], ns = a.nsrc[t];
  float* __restrict__ d = a.dst[t];
  const float* __restrict__ s0 = a.src[0][t];
  const float* __restrict__ s1 = a.src[1][t];
  const float* __restrict__ s2 = a.src[2][t];
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gthreads = gridDim.x * blockDim.x;
  bool al = (((uintptr_t)d | (uintptr_t)s0) & 15) == 0 && (n & 3) == 0;
  if (ns > 1) al = al && ((uintptr_t)s1 & 15) == 0;
  if (ns > 2) al = al && ((uintptr_t)s2 & 15) == 0;
  if (al) {
    float4* d4 = reinterpret_cast<float4*>(d);
    for (int i = gtid; i < (n >> 2); i += gthreads) {
      float4 x = d4[i];
      const float4 y0 = reinterpret_cast<const float4*>(s0)[i];
      float4 y1 = make_float4(0.f, 0.f, 0.f, 0.f), y2 = y1;
      if (ns > 1) y1 = reinterpret_cast<const float4*>(s1)[i];
      if (ns > 2) y2 = reinterpret_cast<const float4*>(s2)[i];
      x.x += y0.x; x.y += y0.y; x.z += y0.z; x.w += y0.w;
      if (ns > 1) { x.x += y1.x; x.y += y1.y; x.z += y1.z; x.w += y1.w; }
      if (ns > 2) { x.x += y2.x; x.y += y2.y; x.z += y2.z; x.w += y2.w; }
      d4[i] = x;
    }
    return;
  }
  for (int i = gtid; i < n; i += gthreads) {
    float x = d[i] + s0[i];
    if (ns > 1) x += s1[i];
    if (ns > 2) x += s2[i];
    d[i] = x;
  }
}

extern "C" int spgan_multi_addn(const spgan_multi_addn_args* a, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(a && a->count > 0 && a->count <= SPGAN_MULTI_ADDN_MAX);
  int nmax = 0;
  for (int t = 0; t < a->count; ++t) {
    SPGAN_CHECK_ARG(a->dst[t] && a->n[t] > 0 && a->nsrc[t] >= 1 && a->nsrc[t] <= 3);
    for (int j = 0; j < a->nsrc[t]; ++j) SPGAN_CHECK_ARG(a->src[j][t]);
    nmax = a->n[t] > nmax ? a->n[t] : nmax;
  }
  int bx = cdiv(nmax, 256 * 4);
  if (bx > 128) bx = 128;
  hipLaunchKernelGGL(multi_addn_kernel, dim3(bx, a->count), dim3(256), 0, (hipStream_t)s_, *a);
  return spgan_launch_status();
}

extern "C" int spgan_multi_add(const spgan_multi_add_args* a, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(a && a->count > 0 && a->count <= SPGAN_MULTI_MAX);
  int nmax = 0;
  for (int t = 0; t < a->count; ++t) {
    SPGAN_CHECK_ARG(a->dst[t] && a->src[t] && a->n[t] > 0);
    nmax = a->n[t] > nmax ? a->n[t] : nmax;
  }
  int bx = cdiv(nmax, 256 * 4);
  if (bx > 128) bx = 128;
  hipLaunchKernelGGL(multi_add_kernel, dim3(bx, a->count), dim3(256), 0, (hipStream_t)s_, *a);
  return spgan_launch_status();
}

extern "C" int spgan_affine_act(const float* X, int ldx, size_t M, int C, const float* scale, const float* shift, float slope, float* out,
                                spgan_stream_t s_) {
  SPGAN_CHECK_ARG(X && scale && shift && out && M > 0 && C > 0 && ldx >= C);
  const bool v4 = (C % 4 == 0) && (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(out) |
                                                     reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift)) & 15) == 0;
  if (v4) hipLaunchKernelGGL(affine_act_kernel<4>, dim3(cdiv(M * (size_t)(C / 4), 256)), dim3(256), 0, (hipStream_t)s_, X, ldx, M, C, scale, shift, slope, out);
  else hipLaunchKernelGGL(affine_act_kernel<1>, dim3(cdiv(M * (size_t)C, 256)), dim3(256), 0, (hipStream_t)s_, X, ldx, M, C, scale, shift, slope, out);
  return spgan_launch_status();
}

extern "C" int spgan_rowscale_outer(const float* X, int ldx, int R, int C, const float* a, const float* b, const float* d, const float* v,
                                    float* out, int ldo, int accumulate, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(X && a && out && R > 0 && C > 0 && ldx >= C && ldo >= C && (!v || (b && d)));
  hipLaunchKernelGGL(rowscale_outer_kernel, dim3(cdiv(R * C, 256)), dim3(256), 0, (hipStream_t)s_, X, ldx, R, C, a, b, d, v, out, ldo, accumulate);
  return spgan_launch_status();
}

// out[i] = sum_{j < parts} recv[j*n + i], j ascending: the local step of a one-hop all-reduce (every rank owns one chunk and has
// received that chunk from all ranks).  Streaming: each thread sums `parts` float4 values with up to eight loads in flight.
__global__ __launch_bounds__(256) void reduce_chunks_kernel(const float* __restrict__ recv, int parts, size_t n, float* __restrict__ out) {
  const size_t i4 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= n) return;
  if (i4 + 3 < n && (n & 3) == 0) {
    float4 s = *reinterpret_cast<const float4*>(recv + i4);
    for (int j = 1; j < parts; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(recv + (size_t)j * n + i4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(out + i4) = s;
  } else {
    for (size_t i = i4; i < n && i < i4 + 4; ++i) {
      float s = recv[i];
      for (int j = 1; j < parts; ++j) s += recv[(size_t)j * n + i];
      out[i] = s;
    }
  }
}

extern "C" int spgan_reduce_chunks(const float* recv, int parts, size_t n, float* out, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(recv && out && parts > 0 && n > 0);
  SPGAN_CHECK_ARG((reinterpret_cast<uintptr_t>(recv) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0);
  hipLaunchKernelGGL(reduce_chunks_kernel, dim3(cdiv(cdiv(n, 4), 256)), dim3(256), 0, (hipStream_t)s_, recv, parts, n, out);
  return spgan_launch_status();
}

extern "C" int spgan_axpby(float a, const float* x, float b, float* y, size_t n, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(x && y && n > 0);
  hipLaunchKernelGGL(axpby_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)s_, a, x, b, y, n);
  return spgan_launch_status();
}

extern "C" int spgan_adam_step_dev(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                                   float* state3, float grad_scale, int zero_grad, spgan_stream_t s_) {  // state3: 4 floats, see spgan_hip.h
  SPGAN_CHECK_ARG(p && g && m && v && state3 && n > 0);
  hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(1), 0, (hipStream_t)s_, beta1, beta2, state3);
  hipLaunchKernelGGL(adam_dev_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)s_, p, g, m, v, n, lr, beta1, beta2, eps, state3, grad_scale, zero_grad);
  return spgan_launch_status();
}

extern "C" int spgan_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, int step,
                               float grad_scale, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(p && g && m && v && n > 0 && step > 0);
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  hipLaunchKernelGGL(adam_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)s_, p, g, m, v, n, lr, beta1, beta2, eps, (float)bc1,
                     (float)(1.0 / sqrt(bc2)), grad_scale);
  return spgan_launch_status();
}
