// Ball-query / grouping family of Common/pointnet_util.py and Common/pointconv_util.py (orphans in the
// reference, named by the north star).  Inputs follow the reference layout: xyz [B,N,C]
// (point-major already), index outputs int64 like torch.  All kernels are HBM/latency-bound integer+fp32 work:
// one thread per query streaming the candidate set through LDS tiles (as spgan_knn), no N x M matrix unless the
// caller asks for it (square_distance).
#include "common.hpp"

namespace {

// ((-2*<a,b>) + |a|^2) + |b|^2 in fp32, products and sums rounded separately like torch's matmul/sum on 3-vectors.
template <int C>
__device__ __forceinline__ float sqdist_expanded(const float (&a)[C], float an, const float* __restrict__ b, float bn) {
  float dot = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) dot = fmaf(a[c], b[c], dot);
  return (-2.f * dot + an) + bn;
}

__device__ __forceinline__ float norm2(const float* __restrict__ p, int C) {
  float s = 0.f;
  for (int c = 0; c < C; ++c) s = fmaf(p[c], p[c], s);
  return s;
}

// dist[b,n,m] = square_distance(src[b,n], dst[b,m])      Common/pointnet_util.py:19-40
__global__ void square_distance_kernel(const float* __restrict__ src, const float* __restrict__ dst, int N, int M, int C, float* __restrict__ out) {
  const int b = blockIdx.z;
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = blockIdx.y;
  if (m >= M) return;
  const float* a = src + ((size_t)b * N + n) * C;
  const float* q = dst + ((size_t)b * M + m) * C;
  float dot = 0.f;
  for (int c = 0; c < C; ++c) dot = fmaf(a[c], q[c], dot);
  out[((size_t)b * N + n) * M + m] = (-2.f * dot + norm2(a, C)) + norm2(q, C);
}

// out[b,s,:] = points[b, idx[b,s], :]      pointnet_util.py:43-60 (idx may be [B,S] or [B,S,K] flattened)
__global__ void index_points_kernel(const float* __restrict__ points, const int64_t* __restrict__ idx, int N, int C, size_t S, size_t total,
                                    float* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const size_t row = t / C;
  const int c = t % C;
  const size_t b = row / S;
  out[t] = points[((size_t)b * N + idx[row]) * C + c];
}

// Iterative farthest point sampling, one workgroup per shape      pointnet_util.py:63-84 / pointconv_util.py:60-83
__global__ __launch_bounds__(512) void fps_kernel(const float* __restrict__ xyz, int N, int npoint, const int64_t* __restrict__ start,
                                                  int64_t* __restrict__ out, float* __restrict__ dist_ws) {
  __shared__ float sv[8];
  __shared__ int si[8];
  __shared__ int far_s;
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const float* x = xyz + (size_t)b * N * 3;
  float* dist = dist_ws + (size_t)b * N;
  for (int i = tid; i < N; i += nt) dist[i] = 1e10f;
  if (tid == 0) far_s = start ? (int)start[b] : 0;
  __syncthreads();
  for (int it = 0; it < npoint; ++it) {
    const int far = far_s;
    if (tid == 0) out[(size_t)b * npoint + it] = far;
    const float cx = x[far * 3], cy = x[far * 3 + 1], cz = x[far * 3 + 2];
    float best = -1.f;
    int bi = 0;
    for (int i = tid; i < N; i += nt) {
      const float dx = x[i * 3] - cx, dy = x[i * 3 + 1] - cy, dz = x[i * 3 + 2] - cz;
      const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));   // sum((xyz-c)**2,-1), no contraction
      const float nd = fminf(dist[i], d);
      dist[i] = nd;
      if (nd > best) { best = nd; bi = i; }    // ascending i: first maximum wins inside a thread
    }
    // block arg-max, ties -> lowest index
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o);
      const int oi = __shfl_xor(bi, o);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    __syncthreads();
    if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
      float bv = sv[0];
      int bj = si[0];
      for (int w = 1; w < (nt >> 6); ++w)
        if (sv[w] > bv || (sv[w] == bv && si[w] < bj)) { bv = sv[w]; bj = si[w]; }
      far_s = bj;
    }
    __syncthreads();
  }
}

// First `nsample` indices (ascending) with d^2 <= r^2, padded with the first hit      pointnet_util.py:87-107
template <int C>
__global__ __launch_bounds__(256) void ball_query_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz, int N, int S,
                                                         float r2, int nsample, int64_t* __restrict__ out) {
  constexpr int TC = 256;
  __shared__ float cand[TC * C];
  __shared__ float cn[TC];
  const int b = blockIdx.y;
  const int s = blockIdx.x * 256 + threadIdx.x;
  const bool ok = s < S;
  float q[C];
#pragma unroll
  for (int c = 0; c < C; ++c) q[c] = ok ? new_xyz[((size_t)b * S + s) * C + c] : 0.f;
  float qn = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) qn = fmaf(q[c], q[c], qn);
  int64_t* o = out + ((size_t)b * S + (ok ? s : 0)) * nsample;
  int cnt = 0;
  int64_t first = N;   // reference pads with group_first; if nothing is inside the ball every slot is N (as there)
  for (int c0 = 0; c0 < N; c0 += TC) {
    const int nc = min(TC, N - c0);
    __syncthreads();
    for (int e = threadIdx.x; e < nc * C; e += 256) cand[e] = xyz[((size_t)b * N + c0) * C + e];
    __syncthreads();
    if (threadIdx.x < nc) cn[threadIdx.x] = norm2(cand + threadIdx.x * C, C);
    __syncthreads();
    if (ok && cnt < nsample)
      for (int j = 0; j < nc && cnt < nsample; ++j) {
        const float d = sqdist_expanded<C>(q, qn, cand + j * C, cn[j]);
        if (!(d > r2)) {
          if (cnt == 0) first = c0 + j;
          o[cnt++] = c0 + j;
        }
      }
  }
  if (ok)
    for (; cnt < nsample; ++cnt) o[cnt] = first;
}

// nsample nearest of xyz for every query (self included), ascending (distance, index)      pointconv_util.py:107-118
template <int KP, int C>
__global__ __launch_bounds__(256) void knn_point_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz, int N, int S, int k,
                                                        int64_t* __restrict__ out) {
  constexpr int TC = 256;
  __shared__ float cand[TC * C];
  __shared__ float cn[TC];
  const int b = blockIdx.y;
  const int s = blockIdx.x * 256 + threadIdx.x;
  const bool ok = s < S;
  float q[C];
#pragma unroll
  for (int c = 0; c < C; ++c) q[c] = ok ? new_xyz[((size_t)b * S + s) * C + c] : 0.f;
  float qn = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) qn = fmaf(q[c], q[c], qn);
  float bd[KP];
  int bi[KP];
#pragma unroll
  for (int t = 0; t < KP; ++t) { bd[t] = INFINITY; bi[t] = 0x7fffffff; }
  for (int c0 = 0; c0 < N; c0 += TC) {
    const int nc = min(TC, N - c0);
    __syncthreads();
    for (int e = threadIdx.x; e < nc * C; e += 256) cand[e] = xyz[((size_t)b * N + c0) * C + e];
    __syncthreads();
    if (threadIdx.x < nc) cn[threadIdx.x] = norm2(cand + threadIdx.x * C, C);
    __syncthreads();
    if (!ok) continue;
    for (int j = 0; j < nc; ++j) {
      const float d = sqdist_expanded<C>(q, qn, cand + j * C, cn[j]);
      if (d < bd[KP - 1]) {
        bd[KP - 1] = d;
        bi[KP - 1] = c0 + j;
#pragma unroll
        for (int t = KP - 1; t > 0; --t)
          if (bd[t] < bd[t - 1]) {
            const float td = bd[t]; bd[t] = bd[t - 1]; bd[t - 1] = td;
            const int ti = bi[t]; bi[t] = bi[t - 1]; bi[t - 1] = ti;
          }
      }
    }
  }
  if (ok) {
#pragma unroll
    for (int t = 0; t < KP; ++t)
      if (t < k) out[((size_t)b * S + s) * k + t] = bi[t];
  }
}

// out[b,s,j,:] = [xyz[idx] - center[b,s] | feat[idx]]      pointnet_util.py:127-139, pointconv_util.py:186-195
__global__ void group_concat_kernel(const float* __restrict__ xyz, const float* __restrict__ center, const float* __restrict__ feat,
                                    const int64_t* __restrict__ idx, int N, int S, int K, int C, int D, size_t total, float* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int W = C + D;
  const int c = t % W;
  const size_t row = t / W;         // (b, s, j)
  const size_t bs = row / K;        // (b, s)
  const size_t b = bs / S;
  const int64_t j = idx[row];
  out[t] = c < C ? xyz[((size_t)b * N + j) * C + c] - center[bs * C + c] : feat[((size_t)b * N + j) * D + (c - C)];
}

// ------------------------------------------------------------------------------------------ adjoints of the gathers
// The reference's gathers (index_points pointnet_util.py:43-60, grouping pointconv_util.py:174-197, the neighbour gather of
// get_edge_features modules.py:708-720) are differentiable through torch indexing; autograd's backward is an index_put with
// float atomics (and the CUDA side-car's grouping_backward an atomicAdd scatter, metrics/pointops/src/grouping/grouping_cuda_kernel.cu:28-45).
// Here: a CSR of "which gather slots read point n" (one workgroup per shape, LDS integer atomics, every segment sorted
// ascending), then a gather-style sum over each point's slots in slot order -- deterministic, no float atomics.
__global__ __launch_bounds__(1024) void gather_csr_kernel(const int64_t* __restrict__ idx, int S, int N, int32_t* __restrict__ rowptr,
                                                          int32_t* __restrict__ src, int* __restrict__ bad) {
  extern __shared__ int ism[];
  int* deg = ism;          // [N]
  int* start = ism + N;    // [N]
  int* wsum = ism + 2 * N; // [32]
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int64_t* ib = idx + (size_t)b * S;
  for (int i = tid; i < N; i += nt) deg[i] = 0;
  __syncthreads();
  for (int e = tid; e < S; e += nt) {
    const int64_t j = ib[e];
    if (j < 0 || j >= N) { if (bad) atomicOr(bad, 1); continue; }
    atomicAdd(&deg[(int)j], 1);
  }
  __syncthreads();
  const int chunk = (N + nt - 1) / nt;
  const int lo = min(N, tid * chunk), hi = min(N, lo + chunk);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += deg[i];
  int incl = s;
  const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int w = 0; w < (nt + 63) / 64; ++w) {
      const int v = wsum[w];
      wsum[w] = run;
      run += v;
    }
    wsum[31] = run;   // valid slots of this shape
  }
  __syncthreads();
  int run = wsum[wv] + incl - s;
  for (int i = lo; i < hi; ++i) {
    start[i] = run;
    run += deg[i];
  }
  __syncthreads();
  // segments of shape b live in src[b*S, (b+1)*S) (invalid indices leave the tail of that range unused)
  for (int i = tid; i < N; i += nt) {
    rowptr[((size_t)b * N + i) * 2] = b * S + start[i];
    rowptr[((size_t)b * N + i) * 2 + 1] = b * S + start[i] + deg[i];
  }
  __syncthreads();
  for (int i = tid; i < N; i += nt) deg[i] = 0;  // reuse as cursor
  __syncthreads();
  for (int e = tid; e < S; e += nt) {
    const int64_t j = ib[e];
    if (j < 0 || j >= N) continue;
    const int slot = atomicAdd(&deg[(int)j], 1);
    src[(size_t)b * S + start[(int)j] + slot] = b * S + e;
  }
  __threadfence_block();
  __syncthreads();
  for (int i = tid; i < N; i += nt) {
    int32_t* seg = src + (size_t)b * S + start[i];
    const int n = deg[i];
    for (int a = 1; a < n; ++a) {
      const int v = seg[a];
      int c = a - 1;
      while (c >= 0 && seg[c] > v) {
        seg[c + 1] = seg[c];
        --c;
      }
      seg[c + 1] = v;
    }
  }
}

// dpoints[n, c] = sum over the slots e of point n (ascending) of dout[e*ld + col0 + c]
__global__ void scatter_slots_kernel(const float* __restrict__ dout, int ld, int col0, int C, const int32_t* __restrict__ rowptr,
                                     const int32_t* __restrict__ src, size_t total, float* __restrict__ dpoints) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const size_t n = t / C;
  const int c = t % C;
  float acc = 0.f;
  for (int e = rowptr[2 * n], e1 = rowptr[2 * n + 1]; e < e1; ++e) acc += dout[(size_t)src[e] * ld + col0 + c];
  dpoints[t] = acc;
}

// dcenter[q, c] = -sum_j dout[(q*K + j)*ld + c]   (the "- center" term of the grouping, pointnet_util.py:128 / pointconv_util.py:189)
__global__ void group_center_bwd_kernel(const float* __restrict__ dout, int ld, int K, int C, size_t total, float* __restrict__ dcenter) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const size_t q = t / C;
  const int c = t % C;
  float acc = 0.f;
  for (int j = 0; j < K; ++j) acc += dout[(q * K + j) * ld + c];
  dcenter[t] = -acc;
}

// dx[b,c,i] = sum_r dE[b,c,i,r] - sum_r dE[b,C+c,i,r] + sum over the slots (i',r) that gathered point i of dE[b,C+c,i',r]    modules.py:708-720
__global__ void edge_features_cm_bwd_kernel(const float* __restrict__ dE, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src,
                                            int C, int N, int k, float* __restrict__ dx) {
  const int b = blockIdx.y;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)C * N) return;
  const int i = t % N;
  const int c = t / N;
  const size_t S = (size_t)N * k;
  const float* ec = dE + ((size_t)b * 2 * C + c) * S;
  const float* ed = dE + ((size_t)b * 2 * C + C + c) * S;
  float acc = 0.f;
  for (int r = 0; r < k; ++r) acc += ec[(size_t)i * k + r];
  float dsum = 0.f;
  for (int r = 0; r < k; ++r) dsum += ed[(size_t)i * k + r];
  acc -= dsum;
  const size_t n = (size_t)b * N + i;
  float in = 0.f;
  for (int e = rowptr[2 * n], e1 = rowptr[2 * n + 1]; e < e1; ++e) in += ed[(size_t)src[e] - (size_t)b * S];
  dx[((size_t)b * C + c) * N + i] = acc + in;
}

}  // namespace

extern "C" int spgan_square_distance(const float* src, const float* dst, int B, int N, int M, int C, float* out, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(src && dst && out && B > 0 && N > 0 && M > 0 && C > 0 && N <= 65535);
  hipLaunchKernelGGL(square_distance_kernel, dim3(cdiv(M, 256), N, B), dim3(256), 0, (hipStream_t)s_, src, dst, N, M, C, out);
  return spgan_launch_status();
}

extern "C" int spgan_index_points(const float* points, const int64_t* idx, int B, int N, int C, int S, float* out, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(points && idx && out && B > 0 && N > 0 && C > 0 && S > 0);
  const size_t total = (size_t)B * S * C;
  hipLaunchKernelGGL(index_points_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)s_, points, idx, N, C, (size_t)S, total, out);
  return spgan_launch_status();
}

extern "C" int spgan_farthest_point_sample(const float* xyz, int B, int N, int npoint, const int64_t* start, int64_t* out, float* dist_ws,
                                           spgan_stream_t s_) {
  SPGAN_CHECK_ARG(xyz && out && dist_ws && B > 0 && N > 0 && npoint > 0);
  hipLaunchKernelGGL(fps_kernel, dim3(B), dim3(512), 0, (hipStream_t)s_, xyz, N, npoint, start, out, dist_ws);
  return spgan_launch_status();
}

extern "C" int spgan_query_ball_point(float radius, int nsample, const float* xyz, const float* new_xyz, int B, int N, int S, int C,
                                      int64_t* out, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(xyz && new_xyz && out && B > 0 && N > 0 && S > 0 && nsample > 0 && (C == 2 || C == 3));
  const float r2 = radius * radius;
  dim3 grid(cdiv(S, 256), B);
  if (C == 3) hipLaunchKernelGGL((ball_query_kernel<3>), grid, dim3(256), 0, (hipStream_t)s_, xyz, new_xyz, N, S, r2, nsample, out);
  else hipLaunchKernelGGL((ball_query_kernel<2>), grid, dim3(256), 0, (hipStream_t)s_, xyz, new_xyz, N, S, r2, nsample, out);
  return spgan_launch_status();
}

extern "C" int spgan_knn_point(int nsample, const float* xyz, const float* new_xyz, int B, int N, int S, int C, int64_t* out,
                               spgan_stream_t s_) {
  SPGAN_CHECK_ARG(xyz && new_xyz && out && B > 0 && N > 0 && S > 0 && nsample > 0 && nsample <= 32 && nsample <= N && C == 3);
  dim3 grid(cdiv(S, 256), B);
  if (nsample <= 10) hipLaunchKernelGGL((knn_point_kernel<10, 3>), grid, dim3(256), 0, (hipStream_t)s_, xyz, new_xyz, N, S, nsample, out);
  else if (nsample <= 20) hipLaunchKernelGGL((knn_point_kernel<20, 3>), grid, dim3(256), 0, (hipStream_t)s_, xyz, new_xyz, N, S, nsample, out);
  else hipLaunchKernelGGL((knn_point_kernel<32, 3>), grid, dim3(256), 0, (hipStream_t)s_, xyz, new_xyz, N, S, nsample, out);
  return spgan_launch_status();
}

extern "C" int spgan_group_concat(const float* xyz, const float* center, const float* feat, const int64_t* idx, int B, int N, int S, int K,
                                  int C, int D, float* out, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(xyz && center && idx && out && B > 0 && N > 0 && S > 0 && K > 0 && C > 0 && D >= 0 && (D == 0 || feat));
  const size_t total = (size_t)B * S * K * (C + D);
  hipLaunchKernelGGL(group_concat_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)s_, xyz, center, feat, idx, N, S, K, C, D, total, out);
  return spgan_launch_status();
}

/* CSR of the gather slots per point: rowptr int32 [B*N, 2] = (begin, end) into src int32 [B*S] (global slot ids b*S + s, ascending per
 * point); bad (optional, one int, zeroed by the caller) is set when an index lies outside [0, N). */
extern "C" int spgan_gather_csr(const int64_t* idx, int B, int S, int N, int32_t* rowptr, int32_t* src, int32_t* bad, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(idx && rowptr && src && B > 0 && S > 0 && N > 0 && N <= 16384 && (size_t)B * S < 2147483647u);
  const size_t sh = (size_t)(2 * N + 32) * sizeof(int);      // N <= 16384: <= 128.1 KB of the CU's 160 KB
  static LdsOptIn opt;  // > 64 KB of dynamic LDS (N >= 8177): once per kernel and device
  if (sh > 64 * 1024) opt.ensure(reinterpret_cast<const void*>(&gather_csr_kernel), 160 * 1024);
  hipLaunchKernelGGL(gather_csr_kernel, dim3(B), dim3(1024), sh, (hipStream_t)s_, idx, S, N, rowptr, src, (int*)bad);
  return spgan_launch_status();
}

extern "C" int spgan_scatter_slots(const float* dout, int ld, int col0, int C, const int32_t* rowptr, const int32_t* src, int BN,
                                   float* dpoints, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(dout && rowptr && src && dpoints && ld > 0 && col0 >= 0 && C > 0 && col0 + C <= ld && BN > 0);
  const size_t total = (size_t)BN * C;
  hipLaunchKernelGGL(scatter_slots_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)s_, dout, ld, col0, C, rowptr, src, total, dpoints);
  return spgan_launch_status();
}

extern "C" int spgan_group_center_bwd(const float* dout, int ld, int Q, int K, int C, float* dcenter, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(dout && dcenter && ld >= C && Q > 0 && K > 0 && C > 0);
  const size_t total = (size_t)Q * C;
  hipLaunchKernelGGL(group_center_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)s_, dout, ld, K, C, total, dcenter);
  return spgan_launch_status();
}

extern "C" int spgan_edge_features_cm_bwd(const float* dE, const int32_t* rowptr, const int32_t* src, int B, int C, int N, int k, float* dx,
                                          spgan_stream_t s_) {
  SPGAN_CHECK_ARG(dE && rowptr && src && dx && B > 0 && C > 0 && N > 0 && k > 0);
  hipLaunchKernelGGL(edge_features_cm_bwd_kernel, dim3(cdiv((size_t)C * N, 256), B), dim3(256), 0, (hipStream_t)s_, dE, rowptr, src, C, N, k, dx);
  return spgan_launch_status();
}
