// Chamfer distance for the evaluation metrics: nearest-neighbour distances between two clouds
// (metrics/CD_EMD/cd/chamferdist/chamfer.cu:12-113), their gradient (chamfer.cu:155-195, here as a deterministic gather
// instead of float atomics), and the all-pairs Chamfer matrix between two SETS of clouds that MMD-CD / COV-CD / 1-NNA-CD
// are computed from (metrics/evaluation_metrics.py:89-126 calls the pairwise routine once per sample cloud).
// 3-D coordinates: VALU work on direct differences (the reference CUDA kernel's formula), candidates staged in LDS.
#include "common.hpp"

namespace {

constexpr int CT = 512;  // candidate points per LDS chunk

// dist[b,i] = min_j |x1[b,i] - x2[b,j]|^2, idx[b,i] = the first j attaining it (strict <, ascending j: chamfer.cu:40-49)
__global__ __launch_bounds__(256) void nn_distance_kernel(const float* __restrict__ x1, const float* __restrict__ x2, int N, int M,
                                                          float* __restrict__ dist, int32_t* __restrict__ idx) {
  __shared__ float4 buf[CT];  // (x, y, z, -): one wave-wide broadcast ds_read_b128 per candidate
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool ok = i < N;
  float px = 0.f, py = 0.f, pz = 0.f;
  if (ok) {
    const float* p = x1 + ((size_t)b * N + i) * 3;
    px = p[0]; py = p[1]; pz = p[2];
  }
  float best = INFINITY;
  int bi = 0;
  for (int c0 = 0; c0 < M; c0 += CT) {
    const int nc = min(CT, M - c0);
    __syncthreads();
    for (int e = threadIdx.x; e < nc; e += 256) {
      const float* q = x2 + ((size_t)b * M + c0 + e) * 3;
      buf[e] = make_float4(q[0], q[1], q[2], 0.f);
    }
    __syncthreads();
    if (ok) {
      for (int j = 0; j < nc; ++j) {
        const float4 c = buf[j];
        const float dx = c.x - px, dy = c.y - py, dz = c.z - pz;
        const float d = dx * dx + dy * dy + dz * dz;
        if (d < best) { best = d; bi = c0 + j; }
      }
    }
  }
  if (ok) {
    dist[(size_t)b * N + i] = best;
    idx[(size_t)b * N + i] = bi;
  }
}

// grad_a[b,i] = 2*ga[b,i]*(xa_i - xb[idxa[b,i]]) + sum_{j: idxb[b,j] == i} 2*gb[b,j]*(xa_i - xb_j)      (ascending j)
// -- both terms of d(sum ga*dist_a + sum gb*dist_b)/d xa_i: its own nearest neighbour, and every xb_j that chose xa_i.
__global__ __launch_bounds__(256) void chamfer_bwd_kernel(const float* __restrict__ xa, const float* __restrict__ xb, int Na, int Nb,
                                                          const float* __restrict__ ga, const int32_t* __restrict__ idxa,
                                                          const float* __restrict__ gb, const int32_t* __restrict__ idxb,
                                                          float* __restrict__ grad_a) {
  __shared__ float buf[CT * 3];
  __shared__ float gbuf[CT];
  __shared__ int ibuf[CT];
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool ok = i < Na;
  float px = 0.f, py = 0.f, pz = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
  if (ok) {
    const float* p = xa + ((size_t)b * Na + i) * 3;
    px = p[0]; py = p[1]; pz = p[2];
    const float* q = xb + ((size_t)b * Nb + idxa[(size_t)b * Na + i]) * 3;
    const float g = 2.f * ga[(size_t)b * Na + i];
    ax = g * (px - q[0]); ay = g * (py - q[1]); az = g * (pz - q[2]);
  }
  for (int c0 = 0; c0 < Nb; c0 += CT) {
    const int nc = min(CT, Nb - c0);
    __syncthreads();
    for (int e = threadIdx.x; e < nc * 3; e += 256) buf[e] = xb[((size_t)b * Nb + c0) * 3 + e];
    for (int e = threadIdx.x; e < nc; e += 256) {
      gbuf[e] = gb[(size_t)b * Nb + c0 + e];
      ibuf[e] = idxb[(size_t)b * Nb + c0 + e];
    }
    __syncthreads();
    if (ok) {
      for (int j = 0; j < nc; ++j) {
        if (ibuf[j] == i) {
          const float g = 2.f * gbuf[j];
          ax += g * (px - buf[j * 3]); ay += g * (py - buf[j * 3 + 1]); az += g * (pz - buf[j * 3 + 2]);
        }
      }
    }
  }
  if (ok) {
    float* o = grad_a + ((size_t)b * Na + i) * 3;
    o[0] = ax; o[1] = ay; o[2] = az;
  }
}

// out[s,r] = mean_i min_j |a_i - b_j|^2 + mean_j min_i |a_i - b_j|^2   for cloud a = A[s] (N points), b = Bc[r] (M points).
// One workgroup per pair; both directions share the LDS copies of the two clouds (float4 per point; N + M <= 8192: 128 KB);
// fixed-order sums.
__global__ __launch_bounds__(256) void chamfer_pairs_kernel(const float* __restrict__ A, const float* __restrict__ Bc, int N, int M, int R,
                                                            float* __restrict__ out) {
  extern __shared__ float4 sm4[];
  float4* pa = sm4;       // [N] (x, y, z, -)
  float4* pb = sm4 + N;   // [M]
  __shared__ float red[4];
  const int s = blockIdx.x / R, r = blockIdx.x % R;
  for (int e = threadIdx.x; e < N; e += 256) {
    const float* q = A + ((size_t)s * N + e) * 3;
    pa[e] = make_float4(q[0], q[1], q[2], 0.f);
  }
  for (int e = threadIdx.x; e < M; e += 256) {
    const float* q = Bc + ((size_t)r * M + e) * 3;
    pb[e] = make_float4(q[0], q[1], q[2], 0.f);
  }
  __syncthreads();
  float total = 0.f;
  for (int dir = 0; dir < 2; ++dir) {
    const float4* q = dir == 0 ? pa : pb;
    const float4* c = dir == 0 ? pb : pa;
    const int nq = dir == 0 ? N : M, nc = dir == 0 ? M : N;
    float acc = 0.f;
    // 4 queries per thread and pass: one broadcast LDS read of a candidate feeds 4 distance evaluations
    for (int i0 = threadIdx.x; i0 < nq; i0 += 4 * 256) {
      float4 p[4];
      float best[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        p[u] = q[min(i0 + u * 256, nq - 1)];
        best[u] = INFINITY;
      }
#pragma unroll 2
      for (int j = 0; j < nc; ++j) {
        const float4 cj = c[j];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float dx = cj.x - p[u].x, dy = cj.y - p[u].y, dz = cj.z - p[u].z;
          best[u] = fminf(best[u], dx * dx + dy * dy + dz * dz);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i0 + u * 256 < nq) acc += best[u];
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) total += ((red[0] + red[1]) + (red[2] + red[3])) / (float)nq;
    __syncthreads();
  }
  if (threadIdx.x == 0) out[(size_t)s * R + r] = total;
}


// Occupancy-grid statistics behind the JSD metric (metrics/evaluation_metrics.py:247-283): cell[s,n] = nearest grid cell of point n
// of cloud s.  counters[g] += number of points in g, bernoulli[g] += number of clouds with at least one point in g.
// One workgroup per cloud; the "seen" set is a bitmap in LDS.  Integer atomics only: the result does not depend on the order.
__global__ __launch_bounds__(256) void occupancy_counts_kernel(const int32_t* __restrict__ cell, int N, int G, int32_t* __restrict__ counters,
                                                                int32_t* __restrict__ bernoulli) {
  extern __shared__ unsigned seen[];
  const int words = (G + 31) >> 5;
  for (int w = threadIdx.x; w < words; w += blockDim.x) seen[w] = 0u;
  __syncthreads();
  const int32_t* c = cell + (size_t)blockIdx.x * N;
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    const int g = c[n];
    if (g >= 0 && g < G) {
      atomicAdd(counters + g, 1);
      atomicOr(seen + (g >> 5), 1u << (g & 31));
    }
  }
  __syncthreads();
  for (int w = threadIdx.x; w < words; w += blockDim.x) {
    unsigned m = seen[w];
    while (m) {
      const int b = __ffs(m) - 1;
      m &= m - 1;
      atomicAdd(bernoulli + (w << 5) + b, 1);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Set-level statistics over a distance matrix between two sets of clouds (metrics/evaluation_metrics.py:129-173): the
// minimum-matching-distance / coverage numbers and the leave-one-out k-NN two-sample test.  Small matrices (a few hundred to
// a few thousand clouds per side); what matters is that no value leaves the device and every sum has a fixed order.

// one workgroup per sample row: its smallest distance, and the reference cloud attaining it (first one on ties) is marked covered
__global__ __launch_bounds__(256) void mmd_rows_kernel(const float* __restrict__ dist, int R, float* __restrict__ rowmin, int32_t* __restrict__ hit) {
  __shared__ float sv[256];
  __shared__ int si[256];
  const float* row = dist + (size_t)blockIdx.x * R;
  float best = INFINITY;
  int bi = 0x7fffffff;
  for (int r = threadIdx.x; r < R; r += 256) {
    const float v = row[r];
    if (v < best) { best = v; bi = r; }  // ascending r per thread: first occurrence
  }
  sv[threadIdx.x] = best;
  si[threadIdx.x] = bi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      const float v = sv[threadIdx.x + o];
      const int i = si[threadIdx.x + o];
      if (v < sv[threadIdx.x] || (v == sv[threadIdx.x] && i < si[threadIdx.x])) { sv[threadIdx.x] = v; si[threadIdx.x] = i; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    rowmin[blockIdx.x] = sv[0];
    if (si[0] < R) hit[si[0]] = 1;  // every writer stores the same value
  }
}

// one thread per reference column (coalesced over the columns)
__global__ __launch_bounds__(256) void mmd_cols_kernel(const float* __restrict__ dist, int S, int R, float* __restrict__ colmin) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  float best = INFINITY;
  for (int s = 0; s < S; ++s) best = fminf(best, dist[(size_t)s * R + r]);
  colmin[r] = best;
}

__device__ __forceinline__ double block_sum_d(double v, double* red) {  // fixed-order tree over 256 threads
  red[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const double t = red[0];
  __syncthreads();
  return t;
}

// out3 = [mean_r colmin, |{covered r}| / R, mean_s rowmin]
__global__ __launch_bounds__(256) void mmd_final_kernel(const float* __restrict__ rowmin, const float* __restrict__ colmin, const int32_t* __restrict__ hit,
                                                        int S, int R, float* __restrict__ out3) {
  __shared__ double red[256];
  double a = 0., b = 0., c = 0.;
  for (int r = threadIdx.x; r < R; r += 256) { a += colmin[r]; c += hit[r] ? 1. : 0.; }
  for (int s = threadIdx.x; s < S; s += 256) b += rowmin[s];
  a = block_sum_d(a, red);
  b = block_sum_d(b, red);
  c = block_sum_d(c, red);
  if (threadIdx.x == 0) {
    out3[0] = (float)(a / R);
    out3[1] = (float)(c / R);
    out3[2] = (float)(b / S);
  }
}

// Joint matrix of the two-sample test, never materialised: M = [[Mxx, Mxy], [Mxy^T, Myy]] over n = n0 + n1 clouds.
__device__ __forceinline__ float joint_at(const float* Mxx, const float* Mxy, const float* Myy, int n0, int n1, int i, int j, int take_sqrt) {
  float v;
  if (i < n0) v = j < n0 ? Mxx[(size_t)i * n0 + j] : Mxy[(size_t)i * n1 + (j - n0)];
  else v = j < n0 ? Mxy[(size_t)j * n1 + (i - n0)] : Myy[(size_t)(i - n0) * n1 + (j - n0)];
  return take_sqrt ? sqrtf(fabsf(v)) : v;
}

// one workgroup per cloud j: its k nearest OTHER clouds (column j of M, smallest first, lower index on ties), the vote of their
// labels (label 1 = first set), pred[j] = votes >= k/2
__global__ __launch_bounds__(256) void two_sample_vote_kernel(const float* __restrict__ Mxx, const float* __restrict__ Mxy, const float* __restrict__ Myy,
                                                              int n0, int n1, int k, int take_sqrt, int32_t* __restrict__ pred) {
  __shared__ float sv[256];
  __shared__ int si[256];
  const int n = n0 + n1, j = blockIdx.x;
  float pv = -INFINITY;  // the pair selected in the previous round: candidates must be lexicographically greater
  int pi = -1, votes = 0;
  for (int round = 0; round < k; ++round) {
    float best = INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += 256) {
      if (i == j) continue;
      const float v = joint_at(Mxx, Mxy, Myy, n0, n1, i, j, take_sqrt);
      const bool after = v > pv || (v == pv && i > pi);
      if (after && (v < best || (v == best && i < bi))) { best = v; bi = i; }
    }
    sv[threadIdx.x] = best;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) {
        const float v = sv[threadIdx.x + o];
        const int i = si[threadIdx.x + o];
        if (v < sv[threadIdx.x] || (v == sv[threadIdx.x] && i < si[threadIdx.x])) { sv[threadIdx.x] = v; si[threadIdx.x] = i; }
      }
      __syncthreads();
    }
    pv = sv[0];
    pi = si[0];
    __syncthreads();
    if (pi < n0) ++votes;
  }
  if (threadIdx.x == 0) pred[j] = (2 * votes >= k) ? 1 : 0;
}

// out9 = [tp, fp, fn, tn, precision, recall, acc_t, acc_f, acc] (evaluation_metrics.py:144-158)
__global__ __launch_bounds__(256) void two_sample_final_kernel(const int32_t* __restrict__ pred, int n0, int n1, float* __restrict__ out9) {
  __shared__ double red[256];
  double tp = 0., fp = 0.;
  for (int j = threadIdx.x; j < n0 + n1; j += 256) {
    if (pred[j]) { if (j < n0) tp += 1.; else fp += 1.; }
  }
  tp = block_sum_d(tp, red);
  fp = block_sum_d(fp, red);
  if (threadIdx.x == 0) {
    const float ftp = (float)tp, ffp = (float)fp, ffn = (float)(n0 - tp), ftn = (float)(n1 - fp);
    out9[0] = ftp; out9[1] = ffp; out9[2] = ffn; out9[3] = ftn;
    out9[4] = ftp / (ftp + ffp + 1e-10f);
    out9[5] = ftp / (ftp + ffn + 1e-10f);
    out9[6] = ftp / (ftp + ffn + 1e-10f);
    out9[7] = ftn / (ftn + ffp + 1e-10f);
    out9[8] = (ftp + ftn) / (float)(n0 + n1);
  }
}

}  // namespace

extern "C" int spgan_nn_distance(const float* xyz1, const float* xyz2, int B, int N, int M, float* dist, int32_t* idx, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(xyz1 && xyz2 && dist && idx && B > 0 && N > 0 && M > 0);
  hipLaunchKernelGGL(nn_distance_kernel, dim3(cdiv(N, 256), B), dim3(256), 0, (hipStream_t)s_, xyz1, xyz2, N, M, dist, idx);
  return spgan_launch_status();
}

extern "C" int spgan_chamfer_bwd(const float* xa, const float* xb, int B, int Na, int Nb, const float* ga, const int32_t* idxa, const float* gb,
                                 const int32_t* idxb, float* grad_a, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(xa && xb && ga && idxa && gb && idxb && grad_a && B > 0 && Na > 0 && Nb > 0);
  hipLaunchKernelGGL(chamfer_bwd_kernel, dim3(cdiv(Na, 256), B), dim3(256), 0, (hipStream_t)s_, xa, xb, Na, Nb, ga, idxa, gb, idxb, grad_a);
  return spgan_launch_status();
}

extern "C" int spgan_chamfer_pairs(const float* A, const float* Bc, int S, int R, int N, int M, float* out, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(A && Bc && out && S > 0 && R > 0 && N > 0 && M > 0 && N <= 4096 && M <= 4096 && (long)S * R < (1L << 31));
  const size_t lds = (size_t)(N + M) * sizeof(float4);
  static LdsOptIn opt;  // > 64 KB of dynamic LDS: once per kernel and device
  if (lds > 64 * 1024) opt.ensure(reinterpret_cast<const void*>(&chamfer_pairs_kernel), 128 * 1024);
  hipLaunchKernelGGL(chamfer_pairs_kernel, dim3(S * R), dim3(256), lds, (hipStream_t)s_, A, Bc, N, M, R, out);
  return spgan_launch_status();
}

extern "C" int spgan_occupancy_counts(const int32_t* cell, int S, int N, int G, int32_t* counters, int32_t* bernoulli, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(cell && counters && bernoulli && S > 0 && N > 0 && G > 0 && G <= 64 * 1024 * 8);
  const size_t lds = (size_t)((G + 31) / 32) * sizeof(unsigned);
  hipLaunchKernelGGL(occupancy_counts_kernel, dim3(S), dim3(256), lds, (hipStream_t)s_, cell, N, G, counters, bernoulli);
  return spgan_launch_status();
}

extern "C" int spgan_mmd_cov(const float* dist, int S, int R, float* out3, float* ws, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(dist && out3 && ws && S > 0 && R > 0);
  hipStream_t s = (hipStream_t)s_;
  float* rowmin = ws;                // [S]
  float* colmin = ws + S;            // [R]
  int32_t* hit = reinterpret_cast<int32_t*>(ws + S + R);  // [R]
  (void)hipMemsetAsync(hit, 0, (size_t)R * sizeof(int32_t), s);
  hipLaunchKernelGGL(mmd_rows_kernel, dim3(S), dim3(256), 0, s, dist, R, rowmin, hit);
  hipLaunchKernelGGL(mmd_cols_kernel, dim3(cdiv(R, 256)), dim3(256), 0, s, dist, S, R, colmin);
  hipLaunchKernelGGL(mmd_final_kernel, dim3(1), dim3(256), 0, s, rowmin, colmin, hit, S, R, out3);
  return spgan_launch_status();
}

extern "C" int spgan_two_sample_knn(const float* Mxx, const float* Mxy, const float* Myy, int n0, int n1, int k, int take_sqrt, float* out9,
                                    int32_t* pred, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(Mxx && Mxy && Myy && out9 && pred && n0 > 0 && n1 > 0 && k > 0 && k < n0 + n1);
  hipStream_t s = (hipStream_t)s_;
  hipLaunchKernelGGL(two_sample_vote_kernel, dim3(n0 + n1), dim3(256), 0, s, Mxx, Mxy, Myy, n0, n1, k, take_sqrt, pred);
  hipLaunchKernelGGL(two_sample_final_kernel, dim3(1), dim3(256), 0, s, pred, n0, n1, out9);
  return spgan_launch_status();
}
