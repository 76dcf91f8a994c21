// Approximate earth mover's distance by a synchronous auction: the algorithm of the reference's emd module
// (metrics/emd/emd_cuda.cu:93-236, metrics/CD_EMD/emd_/emd_module.py:33-75) restated so that its result does not depend on
// thread timing:
//   every unassigned point i of cloud 1 bids for the object k of cloud 2 with the best value 3 - |x_i - y_k| - price_k; its bid
//   increment is best - second best + eps (emd_cuda.cu:141-176);
//   every object takes its highest bidder (the reference finds it with a float atomicMax and a 1e-6 tolerance test whose winner
//   among near-ties is whichever thread writes last, :179-192; here: exact maximum, lowest bidder index on ties -- one of the
//   outcomes the reference can produce), evicts its previous owner and raises its price by the increment (:194-214);
//   in the last iteration every still unassigned point simply takes the object it bid for (`last`, :200).
// The arithmetic that decides the assignment (difference, squares, sum, sqrt, subtractions) is written without contraction into
// fused multiply-adds, so that the numpy restatement in oracle/ reproduces the assignments bit for bit.
#include "common.hpp"

// hipcc contracts a*b + c into fused multiply-adds by default (__fmul_rn / __fadd_rn are plain, contractable operators in the HIP headers):
// switch that off for this file (and use plain operators below, so that no pre-flagged header inline is involved) -- the assignment must not depend on it (see above).
#pragma clang fp contract(off)

namespace {

constexpr int EMD_TILE = 1024;  // objects staged in LDS per sweep step
constexpr float NEG_BIG = -1e9f;

// order-preserving map float -> uint32 (for the packed 64-bit "highest increment, then lowest bidder" maximum)
__device__ __forceinline__ unsigned ordered_bits(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct Best {
  float best, better;
  int idx;
};

__device__ __forceinline__ void merge(Best& a, float b_best, float b_better, int b_idx) {
  if (b_best > a.best || (b_best == a.best && b_idx < a.idx && b_idx >= 0)) {
    a.better = fmaxf(a.best, b_better);
    a.best = b_best;
    a.idx = b_idx;
  } else {
    a.better = fmaxf(a.better, b_best);
  }
}

// One wave per bidder slot; blockDim = 256 (4 bidders per workgroup), grid (ceil(n/4), B).  A workgroup whose four points are all
// assigned leaves after one load.
__global__ __launch_bounds__(256) void emd_bid_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2, int n, float eps,
                                                      const int32_t* __restrict__ assignment, const float* __restrict__ price,
                                                      int32_t* __restrict__ bid, float* __restrict__ bid_inc,
                                                      unsigned long long* __restrict__ winner) {
  __shared__ float sx[EMD_TILE], sy[EMD_TILE], sz[EMD_TILE], sp[EMD_TILE];
  __shared__ int any_unassigned;
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wave;
  const size_t base = (size_t)b * n;
  const bool active = i < n && assignment[base + i] < 0;
  if (threadIdx.x == 0) any_unassigned = 0;
  __syncthreads();
  if (active && lane == 0) any_unassigned = 1;
  __syncthreads();
  if (!any_unassigned) return;
  float x1 = 0.f, y1 = 0.f, z1 = 0.f;
  if (active) {
    x1 = xyz1[(base + i) * 3 + 0]; y1 = xyz1[(base + i) * 3 + 1]; z1 = xyz1[(base + i) * 3 + 2];
  }
  Best me{NEG_BIG, NEG_BIG, -1};
  for (int k0 = 0; k0 < n; k0 += EMD_TILE) {
    const int cnt = min(EMD_TILE, n - k0);
    __syncthreads();
    for (int j = threadIdx.x; j < cnt; j += 256) {
      sx[j] = xyz2[(base + k0 + j) * 3 + 0]; sy[j] = xyz2[(base + k0 + j) * 3 + 1]; sz[j] = xyz2[(base + k0 + j) * 3 + 2];
      sp[j] = price[base + k0 + j];
    }
    __syncthreads();
    if (active) {
      for (int j = lane; j < cnt; j += 64) {  // ascending k per lane: strict '>' keeps the lowest index among equal values
        const float dx = sx[j] - x1, dy = sy[j] - y1, dz = sz[j] - z1;
        const float d2 = (dx * dx + dy * dy) + dz * dz;  // plain operators: the pragma above keeps them unfused
        const float v = (3.0f - sqrtf(d2)) - sp[j];
        if (v > me.best) {
          me.better = me.best; me.best = v; me.idx = k0 + j;
        } else if (v > me.better) {
          me.better = v;
        }
      }
    }
  }
  if (!active) return;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(me.best, o), os = __shfl_xor(me.better, o);
    const int oi = __shfl_xor(me.idx, o);
    merge(me, ob, os, oi);
  }
  if (lane == 0) {
    const float inc = (me.best - me.better) + eps;
    bid[base + i] = me.idx;
    bid_inc[base + i] = inc;
    const unsigned long long key = ((unsigned long long)ordered_bits(inc) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
    atomicMax(winner + base + me.idx, key);
  }
}

// One thread per object: the winner (if any bid arrived) takes it.
__global__ void emd_assign_kernel(int n, int32_t* __restrict__ assignment, int32_t* __restrict__ assignment_inv, float* __restrict__ price,
                                  const float* __restrict__ bid_inc, unsigned long long* __restrict__ winner) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const size_t base = (size_t)blockIdx.y * n;
  const unsigned long long key = winner[base + k];
  if (key == 0ull) return;
  winner[base + k] = 0ull;
  const int i = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
  const int prev = assignment_inv[base + k];
  if (prev >= 0) assignment[base + prev] = -1;
  assignment_inv[base + k] = i;
  assignment[base + i] = k;
  price[base + k] = price[base + k] + bid_inc[base + i];
}

// last iteration: every unassigned point takes the object it bid for (emd_cuda.cu:200, `last`)
__global__ void emd_take_bids_kernel(int n, int32_t* __restrict__ assignment, const int32_t* __restrict__ bid) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t base = (size_t)blockIdx.y * n;
  if (assignment[base + i] < 0) assignment[base + i] = bid[base + i];
}

__global__ void emd_dist_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2, int n, const int32_t* __restrict__ assignment,
                                float* __restrict__ dist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t base = (size_t)blockIdx.y * n;
  const int k = assignment[base + i];
  const float dx = xyz1[(base + i) * 3 + 0] - xyz2[(base + k) * 3 + 0];
  const float dy = xyz1[(base + i) * 3 + 1] - xyz2[(base + k) * 3 + 1];
  const float dz = xyz1[(base + i) * 3 + 2] - xyz2[(base + k) * 3 + 2];
  dist[base + i] = (dx * dx + dy * dy) + dz * dz;
}

// grad_xyz1[i] = 2 * g[i] * (x_i - y_assignment[i])  (emd_cuda.cu:279-296; each point owns its row: no atomics needed)
__global__ void emd_bwd_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2, int n, const float* __restrict__ g,
                               const int32_t* __restrict__ assignment, float* __restrict__ grad1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t base = (size_t)blockIdx.y * n;
  const int k = assignment[base + i];
  const float s = 2.f * g[base + i];
#pragma unroll
  for (int c = 0; c < 3; ++c) grad1[(base + i) * 3 + c] = s * (xyz1[(base + i) * 3 + c] - xyz2[(base + k) * 3 + c]);
}

}  // namespace

extern "C" size_t spgan_emd_ws_bytes(int B, int n) {
  // price f32 | bid_inc f32 | bid i32 | assignment_inv i32 | winner u64
  return (size_t)B * n * (4 + 4 + 4 + 4 + 8);
}

extern "C" int spgan_emd_forward(const float* xyz1, const float* xyz2, int B, int n, float eps, int iters, float* dist, int32_t* assignment,
                                 void* ws, size_t ws_bytes, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(xyz1 && xyz2 && dist && assignment && ws && B > 0 && n > 0 && iters > 0 && B <= 65535);
  SPGAN_CHECK_ARG(ws_bytes >= spgan_emd_ws_bytes(B, n) && ((uintptr_t)ws & 7) == 0);
  hipStream_t s = (hipStream_t)s_;
  const size_t bn = (size_t)B * n;
  unsigned long long* winner = (unsigned long long*)ws;
  float* price = (float*)(winner + bn);
  float* bid_inc = price + bn;
  int32_t* bid = (int32_t*)(bid_inc + bn);
  int32_t* inv = bid + bn;
  (void)hipMemsetAsync(winner, 0, bn * (8 + 4), s);               // winner = 0, price = 0
  (void)hipMemsetAsync(inv, 0xFF, bn * 4, s);                     // assignment_inv = -1
  (void)hipMemsetAsync(assignment, 0xFF, bn * 4, s);              // assignment = -1
  const dim3 per_point(cdiv(n, 256), B);
  for (int it = 0; it < iters; ++it) {
    hipLaunchKernelGGL(emd_bid_kernel, dim3(cdiv(n, 4), B), dim3(256), 0, s, xyz1, xyz2, n, eps, assignment, price, bid, bid_inc, winner);
    if (it == iters - 1) hipLaunchKernelGGL(emd_take_bids_kernel, per_point, dim3(256), 0, s, n, assignment, bid);
    else hipLaunchKernelGGL(emd_assign_kernel, per_point, dim3(256), 0, s, n, assignment, inv, price, bid_inc, winner);
  }
  hipLaunchKernelGGL(emd_dist_kernel, per_point, dim3(256), 0, s, xyz1, xyz2, n, assignment, dist);
  return spgan_launch_status();
}

extern "C" int spgan_emd_backward(const float* xyz1, const float* xyz2, int B, int n, const float* grad_dist, const int32_t* assignment,
                                  float* grad_xyz1, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(xyz1 && xyz2 && grad_dist && assignment && grad_xyz1 && B > 0 && n > 0 && B <= 65535);
  hipLaunchKernelGGL(emd_bwd_kernel, dim3(cdiv(n, 256), B), dim3(256), 0, (hipStream_t)s_, xyz1, xyz2, n, grad_dist, assignment, grad_xyz1);
  return spgan_launch_status();
}
