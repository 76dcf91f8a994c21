// GAN losses on the [B,1] logits (Common/loss_utils.py:727-802, 854-972) with their gradients,
// and the WGAN-GP penalty pieces (Common/gradient_penalty.py:19-37).  O(B) .. O(B^2) work: one
// workgroup, one launch per loss (value + both logit gradients).
#include "common.hpp"

namespace {

enum { GAN_LS = 0, GAN_WGAN = 1, GAN_HINGE = 2, GAN_BCE = 3 };

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
  return t;
}

__device__ __forceinline__ float bce_logits(float x, float t) { return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// which: 0 = discriminator loss, 1 = generator loss.
// out[0] = loss, out[1] = fake-term, out[2] = real-term, out[3] = real_acc (logit >= t), out[4] = fake_acc (logit < t)
// Labels (ls only): real_label / fake_label [B] or NULL (ones / zeros; generator: fake label = ones).
// F.mse_loss([B,1] logits, [B] labels) broadcasts to [B,B] in the reference (loss_utils.py:923-924,763): kept.
__global__ __launch_bounds__(256) void gan_loss_kernel(int mode, int which, const float* __restrict__ d_real, const float* __restrict__ d_fake,
                                                       const float* __restrict__ real_label, const float* __restrict__ fake_label, int B,
                                                       float* __restrict__ out, float* __restrict__ g_real, float* __restrict__ g_fake) {
  __shared__ float red[4];
  const float rB = 1.f / (float)B;
  float lf = 0.f, lr = 0.f, accr = 0.f, accf = 0.f;
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    const float xf = d_fake[i];
    const float xr = d_real ? d_real[i] : 0.f;
    float gf = 0.f, gr = 0.f;
    if (mode == GAN_LS) {
      float sf = 0.f, gsf = 0.f, sr = 0.f, gsr = 0.f;
      for (int j = 0; j < B; ++j) {
        const float tf = fake_label ? fake_label[j] : (which == 1 ? 1.f : 0.f);
        const float df = xf - tf;
        sf = fmaf(df, df, sf);
        gsf += df;
        if (which == 0) {
          const float tr = real_label ? real_label[j] : 1.f;
          const float dr = xr - tr;
          sr = fmaf(dr, dr, sr);
          gsr += dr;
        }
      }
      const float w = (which == 0) ? 0.5f : 1.f;
      lf += sf * rB * rB;
      gf = w * 2.f * gsf * rB * rB;
      if (which == 0) {
        lr += sr * rB * rB;
        gr = w * 2.f * gsr * rB * rB;
        accr += (xr >= 0.5f) ? 1.f : 0.f;
        accf += (xf < 0.5f) ? 1.f : 0.f;
      }
    } else if (mode == GAN_WGAN) {
      if (which == 0) {
        lf += xf * rB; lr += xr * rB;
        gf = rB; gr = -rB;
      } else {
        lf += -xf * rB;
        gf = -rB;
      }
    } else if (mode == GAN_HINGE) {
      if (which == 0) {
        lr += fmaxf(1.f - xr, 0.f) * rB;
        lf += fmaxf(1.f + xf, 0.f) * rB;
        gr = (1.f - xr > 0.f) ? -rB : 0.f;
        gf = (1.f + xf > 0.f) ? rB : 0.f;
      } else {
        lf += -xf * rB;
        gf = -rB;
      }
      accr += (xr >= 0.f) ? 1.f : 0.f;
      accf += (xf < 0.f) ? 1.f : 0.f;
    } else {  // GAN_BCE
      if (which == 0) {
        lf += bce_logits(xf, 0.f) * rB; lr += bce_logits(xr, 1.f) * rB;
        gf = 0.5f * (sigmoidf_(xf) - 0.f) * rB;
        gr = 0.5f * (sigmoidf_(xr) - 1.f) * rB;
      } else {
        lf += bce_logits(xf, 1.f) * rB;
        gf = (sigmoidf_(xf) - 1.f) * rB;
      }
    }
    g_fake[i] = gf;
    if (g_real) g_real[i] = gr;
  }
  lf = block_sum(lf, red); lr = block_sum(lr, red); accr = block_sum(accr, red); accf = block_sum(accf, red);
  if (threadIdx.x == 0) {
    float loss;
    if (which == 0) loss = (mode == GAN_LS || mode == GAN_BCE) ? 0.5f * (lf + lr) : ((mode == GAN_WGAN) ? lf - lr : lf + lr);
    else loss = lf;
    out[0] = loss; out[1] = lf; out[2] = lr; out[3] = accr * rB; out[4] = accf * rB;
  }
}

// x_hat[b,:] = real[b,:] + alpha[b]*(fake[b,:] - real[b,:])     gradient_penalty.py:26
__global__ void lerp_rows_kernel(const float* __restrict__ real, const float* __restrict__ fake, const float* __restrict__ alpha, size_t L,
                                 size_t total, float* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const float r = real[t];
  out[t] = r + alpha[t / L] * (fake[t] - r);
}

// norms[b] = ||g[b,:]||_2  (one workgroup per sample)
__global__ __launch_bounds__(256) void row_norm_kernel(const float* __restrict__ g, size_t L, float* __restrict__ norms) {
  __shared__ float red[4];
  const float* gb = g + (size_t)blockIdx.x * L;
  float s = 0.f;
  for (size_t i = threadIdx.x; i < L; i += blockDim.x) s = fmaf(gb[i], gb[i], s);
  s = block_sum(s, red);
  if (threadIdx.x == 0) norms[blockIdx.x] = sqrtf(s);
}
// loss = lambda * mean_b ((norm_b - gamma)/gamma)^2      gradient_penalty.py:35
__global__ __launch_bounds__(256) void gp_value_kernel(const float* __restrict__ norms, int B, float gamma, float lambda, float* __restrict__ loss) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    const float d = (norms[i] - gamma) / gamma;
    s = fmaf(d, d, s);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) loss[0] = lambda * s / (float)B;
}
// v[b,:] = upstream * lambda * (2/B) * (norm_b - gamma)/gamma^2 * g[b,:]/norm_b
__global__ void gp_bwd_kernel(const float* __restrict__ g, const float* __restrict__ norms, size_t L, size_t total, int B, float gamma,
                              float lambda, const float* __restrict__ upstream, float* __restrict__ v) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const float n = norms[t / L];
  const float up = upstream ? upstream[0] : 1.f;
  const float c = (n > 0.f) ? up * lambda * (2.f / (float)B) * ((n - gamma) / (gamma * gamma)) / n : 0.f;
  v[t] = c * g[t];
}

// gp_bwd_kernel and gp_value_kernel as ONE launch: every workgroup writes its part of v; workgroup 0 also forms the penalty (the arithmetic and
// summation order of gp_value_kernel) and, with loss_add, total[0] = loss_add[0] + penalty (the D step's reported loss: no add launch)
__global__ __launch_bounds__(256) void gp_value_bwd_kernel(const float* __restrict__ g, const float* __restrict__ norms, size_t L, size_t total, int B,
                                                           float gamma, float lambda, float* __restrict__ loss, const float* __restrict__ loss_add,
                                                           float* __restrict__ loss_total, float* __restrict__ v) {
  __shared__ float red[4];
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < total) {
    const float n = norms[t / L];
    const float c = (n > 0.f) ? 1.f * lambda * (2.f / (float)B) * ((n - gamma) / (gamma * gamma)) / n : 0.f;
    v[t] = c * g[t];
  }
  if (blockIdx.x == 0) {
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
      const float d = (norms[i] - gamma) / gamma;
      s = fmaf(d, d, s);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
      const float pen = lambda * s / (float)B;
      loss[0] = pen;
      if (loss_total) loss_total[0] = loss_add[0] + pen;
    }
  }
}

}  // namespace

extern "C" int spgan_gp_penalty_fwd_bwd(const float* g, int B, size_t L, float gamma, float lambda, float* norms, float* loss, const float* loss_add,
                                        float* loss_total, float* v, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(g && norms && loss && v && B > 0 && L > 0 && gamma != 0.f && (!loss_total == !loss_add));
  const size_t total = (size_t)B * L;
  hipLaunchKernelGGL(row_norm_kernel, dim3(B), dim3(256), 0, (hipStream_t)s_, g, L, norms);
  hipLaunchKernelGGL(gp_value_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)s_, g, norms, L, total, B, gamma, lambda, loss, loss_add,
                     loss_total, v);
  return spgan_launch_status();
}

extern "C" int spgan_gan_loss(int mode, int which, const float* d_real, const float* d_fake, const float* real_label, const float* fake_label,
                              int B, float* out5, float* g_real, float* g_fake, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(mode >= 0 && mode <= 3 && (which == 0 || which == 1) && d_fake && out5 && g_fake && B > 0);
  if (which == 0) SPGAN_CHECK_ARG(d_real && g_real);
  hipLaunchKernelGGL(gan_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)s_, mode, which, d_real, d_fake, real_label, fake_label, B, out5,
                     g_real, g_fake);
  return spgan_launch_status();
}

extern "C" int spgan_lerp_rows(const float* real, const float* fake, const float* alpha, int B, size_t L, float* out, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(real && fake && alpha && out && B > 0 && L > 0);
  const size_t total = (size_t)B * L;
  hipLaunchKernelGGL(lerp_rows_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)s_, real, fake, alpha, L, total, out);
  return spgan_launch_status();
}

extern "C" int spgan_gp_penalty_fwd(const float* g, int B, size_t L, float gamma, float lambda, float* norms, float* loss, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(g && norms && loss && B > 0 && L > 0 && gamma != 0.f);
  hipLaunchKernelGGL(row_norm_kernel, dim3(B), dim3(256), 0, (hipStream_t)s_, g, L, norms);
  hipLaunchKernelGGL(gp_value_kernel, dim3(1), dim3(256), 0, (hipStream_t)s_, norms, B, gamma, lambda, loss);
  return spgan_launch_status();
}

extern "C" int spgan_gp_penalty_bwd(const float* g, const float* norms, int B, size_t L, float gamma, float lambda, const float* upstream,
                                    float* v, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(g && norms && v && B > 0 && L > 0 && gamma != 0.f);
  const size_t total = (size_t)B * L;
  hipLaunchKernelGGL(gp_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)s_, g, norms, L, total, B, gamma, lambda, upstream, v);
  return spgan_launch_status();
}
