// EXPERIMENT: phase ablation of the ping-pong kNN kernel.  ABL 0 full, 1 no selection, 2 no MFMA, 3 neither
#include "common.hpp"
namespace {
template <int KP, int CP, int ABL>
__global__ __launch_bounds__(512) void knn_pp(const float* __restrict__ x, int N, int C, int k, int32_t* __restrict__ idx) {
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  constexpr int LDC = CP + 4;
  constexpr int H = CP / 2;
  constexpr int F4 = CP / 4;            // float4 per row
  constexpr int SL = (32 * F4 + 511) / 512;  // staging slots per thread (1 for CP = 64, 2 for CP = 128)
  __shared__ __attribute__((aligned(16))) float cand[2][32 * LDC];
  __shared__ __attribute__((aligned(16))) float cn[2][32];
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = wave >> 2;
  const int l31 = lane & 31, lh = lane >> 5;
  const float* xb = x + (size_t)b * N * C;
  const int qbase = blockIdx.x * 256;
  const bool vec = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);

  float4 st[SL];
  auto stage = [&](int row0) {
#pragma unroll
    for (int i = 0; i < SL; ++i) {
      const int e = tid + 512 * i, r = e / F4, c = (e % F4) * 4;
      const int row = row0 + r;
      st[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < 32 && row < N && c < C) {
        const float* p = xb + (size_t)row * C + c;
        if (vec) st[i] = *reinterpret_cast<const float4*>(p);
        else {
          st[i].x = p[0];
          if (c + 1 < C) st[i].y = p[1];
          if (c + 2 < C) st[i].z = p[2];
          if (c + 3 < C) st[i].w = p[3];
        }
      }
    }
  };
  auto commit = [&](int buf) {  // registers -> LDS tile + squared norms of its rows: the routine of knn_mfma_kernel (same shuffle tree)
#pragma unroll
    for (int i = 0; i < SL; ++i) {
      const int e = tid + 512 * i, r = e / F4, c = (e % F4) * 4;
      float s = fmaf(st[i].w, st[i].w, fmaf(st[i].z, st[i].z, fmaf(st[i].y, st[i].y, st[i].x * st[i].x)));
#pragma unroll
      for (int o = 1; o < F4; o <<= 1) s += __shfl_xor(s, o);
      if (r < 32) {
        *reinterpret_cast<float4*>(&cand[buf][r * LDC + c]) = st[i];
        if ((e % F4) == 0) cn[buf][r] = s;
      }
    }
  };

  // queries: 8 x 32 rows through the same staging path, two tiles per round (one per LDS buffer), wave w keeps tile w
  float qv[H];
  float qn = 0.f;
  for (int w2 = 0; w2 < 4; ++w2) {
    __syncthreads();
    stage(qbase + 32 * (2 * w2));
    commit(0);
    stage(qbase + 32 * (2 * w2 + 1));
    commit(1);
    __syncthreads();
    if ((wave >> 1) == w2) {
      const int bq = wave & 1;
#pragma unroll
      for (int t = 0; t < H; t += 4) {
        const float4 v = *reinterpret_cast<const float4*>(&cand[bq][l31 * LDC + lh * H + t]);
        qv[t] = v.x; qv[t + 1] = v.y; qv[t + 2] = v.z; qv[t + 3] = v.w;
      }
      qn = cn[bq][l31];
    }
  }
  __syncthreads();

  float bd[KP];
  int bi[KP];
#pragma unroll
  for (int t = 0; t < KP; ++t) {
    bd[t] = INFINITY;
    bi[t] = 0x7fffffff;
  }
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;

  const int ntiles = (N + 31) / 32;
  stage(0);
  commit(0);
  __syncthreads();
  for (int h = 0; h <= 2 * ntiles; ++h) {
    // staging of the next tile by all 512 threads: loads at the even half-step, LDS stores at the odd one (buffer (t+1)&1 was last
    // read -- tile t-1, by group 1 -- in the previous odd half-step)
    const int tnext = (h >> 1) + 1;
    if ((h & 1) == 0) {
      if (tnext < ntiles) stage(tnext * 32);
    } else if (tnext < ntiles) {
      commit(tnext & 1);
    }
    const int role = (ABL == 5) ? (h & 1) : ((h + grp) & 1);  // uniform per wave
    if (ABL == 4 && grp == 1) { __syncthreads(); continue; }
    if (role == 0) {
      const int tile = (ABL == 5) ? (h >> 1) : ((h - grp) >> 1);
      if (tile < ntiles && ABL != 2 && ABL != 3) {
        const float* a = &cand[tile & 1][l31 * LDC + lh * H];
        if (ABL == 6 || ABL == 7) __builtin_amdgcn_s_setprio(ABL == 6 ? 1 : 3);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
#pragma unroll
        for (int t = 0; t < H; t += 8) {
          const float4 v0 = *reinterpret_cast<const float4*>(a + t), v1 = *reinterpret_cast<const float4*>(a + t + 4);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0.x, qv[t], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0.y, qv[t + 1], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0.z, qv[t + 2], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0.w, qv[t + 3], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1.x, qv[t + 4], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1.y, qv[t + 5], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1.z, qv[t + 6], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1.w, qv[t + 7], acc1, 0, 0, 0);
        }
        if (ABL == 6 || ABL == 7) __builtin_amdgcn_s_setprio(0);
      }
    } else {
      const int tile = (ABL == 5) ? ((h - 1) >> 1) : ((h - 1 - grp) >> 1);
      if (tile >= 0 && (ABL == 5 ? h - 1 : h - 1 - grp) >= 0 && ABL != 1 && ABL != 3) {
        const int buf = tile & 1, j0 = tile * 32;
        float d[16];
        unsigned live = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 nrm = *reinterpret_cast<const float4*>(&cn[buf][8 * g + 4 * lh]);
          const float nn[4] = {nrm.x, nrm.y, nrm.z, nrm.w};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int r = 4 * g + u;
            const float dot = acc0[r] + acc1[r];
            d[r] = (-2.f * dot + qn) + nn[u];
            if (j0 + 8 * g + 4 * lh + u >= N) d[r] = INFINITY;
            live |= (d[r] < bd[KP - 1]) ? (1u << r) : 0u;
          }
        }
        while (live) {
          const int r = __ffs(live) - 1;
          live &= live - 1;
          float dv = d[0];
#pragma unroll
          for (int u = 1; u < 16; ++u) dv = (r == u) ? d[u] : dv;
          if (dv < bd[KP - 1]) {
            bd[KP - 1] = dv;
            bi[KP - 1] = j0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
            for (int t = KP - 1; t > 0; --t) {
              if (bd[t] < bd[t - 1]) {
                const float td = bd[t]; bd[t] = bd[t - 1]; bd[t - 1] = td;
                const int ti = bi[t]; bi[t] = bi[t - 1]; bi[t - 1] = ti;
              }
            }
          }
        }
      }
    }
    __syncthreads();
  }

#pragma unroll
  for (int t = 0; t < KP; ++t) {
    const float od = __shfl_xor(bd[t], 32);
    const int oi = __shfl_xor(bi[t], 32);
    if (lh == 0 && (od < bd[KP - 1] || (od == bd[KP - 1] && oi < bi[KP - 1]))) {
      bd[KP - 1] = od;
      bi[KP - 1] = oi;
#pragma unroll
      for (int u = KP - 1; u > 0; --u) {
        if (bd[u] < bd[u - 1] || (bd[u] == bd[u - 1] && bi[u] < bi[u - 1])) {
          const float td = bd[u]; bd[u] = bd[u - 1]; bd[u - 1] = td;
          const int ti = bi[u]; bi[u] = bi[u - 1]; bi[u - 1] = ti;
        }
      }
    }
  }
  if (ABL) { float sacc = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc += acc0[r] + acc1[r];
    bd[1] += sacc * 1e-30f; bi[1] += (int)(sacc * 1e-30f); }
  const int q = qbase + 32 * wave + l31;
  if (lh == 0 && q < N) {
    int32_t* o = idx + ((size_t)b * N + q) * k;
#pragma unroll
    for (int t = 1; t < KP; ++t)
      if (t <= k) o[t - 1] = b * N + bi[t];
  }
}

}
extern "C" int exp_knn_pp(const float* x, int B, int N, int C, int k, int abl, int32_t* idx, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  dim3 g(cdiv(N, 256), B), b(512);
  switch (abl) {
    case 0: hipLaunchKernelGGL((knn_pp<11, 64, 0>), g, b, 0, s, x, N, C, k, idx); break;
    case 1: hipLaunchKernelGGL((knn_pp<11, 64, 1>), g, b, 0, s, x, N, C, k, idx); break;
    case 2: hipLaunchKernelGGL((knn_pp<11, 64, 2>), g, b, 0, s, x, N, C, k, idx); break;
    case 3: hipLaunchKernelGGL((knn_pp<11, 64, 3>), g, b, 0, s, x, N, C, k, idx); break;
    case 4: hipLaunchKernelGGL((knn_pp<11, 64, 4>), g, b, 0, s, x, N, C, k, idx); break;
    case 5: hipLaunchKernelGGL((knn_pp<11, 64, 5>), g, b, 0, s, x, N, C, k, idx); break;
    case 6: hipLaunchKernelGGL((knn_pp<11, 64, 6>), g, b, 0, s, x, N, C, k, idx); break;
    case 7: hipLaunchKernelGGL((knn_pp<11, 64, 7>), g, b, 0, s, x, N, C, k, idx); break;
  }
  return (int)hipGetLastError();
}
