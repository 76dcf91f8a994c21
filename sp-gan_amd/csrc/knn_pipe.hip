// k nearest neighbours of 64-channel feature rows (Generation/modules.py:683-703 on EdgeBlock 2's input), the round-4 geometry.
//
// What the PMC passes over csrc/graph.hip's knn_mfma3_kernel showed (profiles/r04_knn_pmc.txt, tools/knn_pmc.sh): the launch is bound by
// VALU issue, not by the matrix pipe -- 513 VALU instructions per 32-candidate tile and wave (612 before the selection loop's LDS
// extraction + median-of-three insertion), MFMA pipe ~20 % busy -- and the two do not overlap: a wave issues its 24 MFMAs back to back
// and only then starts on the distances.  ~110 of those instructions converted the candidate tile to its three bfloat16 planes, in
// each of the 16 workgroups that scan the same shape.  Here (285 VALU instructions per tile and wave, 213 -> 172 us with the pre-pass)
//   * a pre-pass (knn_split_kernel) writes every 32-row tile once as the exact LDS image the main kernel wants: three bf16 planes per
//     row at the conflict-free row pitch, then the 32 squared norms (+inf for rows past N, which removes the bounds test from the
//     distance loop).  12.9 KB per tile, 26 MB at B=32, N=2048: one 10 us launch;
//   * the main kernel copies images global -> LDS with 16-byte loads (a ring of three, one barrier per tile) and takes its query
//     fragments straight from the images;
//   * the tile loop is software-pipelined inside a wave: the MFMAs of tile t+1 are issued between the distance instructions of
//     tile t (two accumulator sets, alternating), so the matrix pipe runs under the VALU work instead of before it;
//   * the survivor mask is built from sign bits (v_sub + v_alignbit per candidate instead of cmp + cndmask + or), and the selection
//     loop runs once per tile pair (32 candidates per lane): fewer max-over-lanes trips.
// What is left (same PMC file): VALU issue 58 % of a SIMD's cycles, matrix pipe 31 %, ~29 % of a wave's life in s_waitcnt/s_barrier --
// the four waves of a workgroup meet once per tile and wait for the one with the most survivors.
// Distances, norms and the selection are the arithmetic of knn_mfma3_kernel, operation for operation: the indices are identical
// (tools/knn_ab.py prints the same checksum for both).
#include <math.h>
#include <type_traits>
#include "common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

namespace {

constexpr int CP = 64;               // channels per row (C <= 64, zero padded)
constexpr int PW = CP / 2;           // words per bf16 plane of a row
constexpr int LDC = 3 * PW + 4;      // 100 words: 25 four-word groups per row (odd) -> conflict-free 16-byte fragment reads
constexpr int KS = CP / 16;          // MFMA k-steps per tile
constexpr int IMGW = 32 * LDC + 32;  // words per tile image: 32 rows, then 32 squared norms
constexpr int IMG4 = IMGW / 4;       // 808 float4
constexpr int KP = 11;               // list length: rank 0 (the query itself) + k <= 10

// One workgroup per 32-row tile: x [B][N][C] fp32 -> img [B][tiles][IMGW].  v = hi + mid + lo exactly (three bf16 terms carry fp32's
// 24 significand bits); the squared norm is the fp32 fma chain + butterfly of knn_mfma3_kernel's commit().
__global__ __launch_bounds__(256) void knn_split_kernel(const float* __restrict__ x, int N, int C, int tiles, float* __restrict__ img) {
  const int b = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
  const float* xb = x + (size_t)b * N * C;
  float* out = img + ((size_t)b * tiles + tile) * IMGW;
  const bool vec = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
#pragma unroll
  for (int i = 0; i < CP / 32; ++i) {
    const int e = tid + 256 * i, r = e / (CP / 4), c = (e % (CP / 4)) * 4;
    const int row = tile * 32 + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < N && c < C) {
      const float* p = xb + (size_t)row * C + c;
      if (vec) v = *reinterpret_cast<const float4*>(p);
      else {
        v.x = p[0];
        if (c + 1 < C) v.y = p[1];
        if (c + 2 < C) v.z = p[2];
        if (c + 3 < C) v.w = p[3];
      }
    }
    const f32x4v f = {v.x, v.y, v.z, v.w};
    const bf16x4 hi = __builtin_convertvector(f, bf16x4);
    const f32x4v r1 = f - __builtin_convertvector(hi, f32x4v);
    const bf16x4 mid = __builtin_convertvector(r1, bf16x4);
    const f32x4v r2 = r1 - __builtin_convertvector(mid, f32x4v);
    const bf16x4 lo = __builtin_convertvector(r2, bf16x4);
    float* o = out + r * LDC + (c >> 1);
    *reinterpret_cast<bf16x4*>(o) = hi;
    *reinterpret_cast<bf16x4*>(o + PW) = mid;
    *reinterpret_cast<bf16x4*>(o + 2 * PW) = lo;
    float s = fmaf(v.w, v.w, fmaf(v.z, v.z, fmaf(v.y, v.y, v.x * v.x)));
#pragma unroll
    for (int o2 = 1; o2 < CP / 4; o2 <<= 1) s += __shfl_xor(s, o2);  // the CP/4 lanes of a row are consecutive
    if ((e % (CP / 4)) == 0) out[32 * LDC + r] = row < N ? s : INFINITY;
  }
  if (tid < 32) {  // the 4 pad words of a row are copied with the image: keep them defined
    *reinterpret_cast<float4*>(out + tid * LDC + 3 * PW) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// grid (ceil(N/128), B), 256 threads: wave w owns the 32 queries of tile 4*blockIdx.x + w and scans every tile of its shape.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void knn_pipe_kernel(const float* __restrict__ img, int N, int tiles, int k, int32_t* __restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* cand = smem;               // [3][IMGW]
  float* dsc = smem + 3 * IMGW;     // [4][32*64]: per wave, the 2 x 16 distances of a tile pair of every lane, [r][lane]
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const float* ib = img + (size_t)b * tiles * IMGW;
  const int ntiles = (N + 31) / 32;

  // the query fragments: lane (l31, lh) holds channels 16*t + 8*lh .. +7 of k-step t of row l31, for each plane
  bf16x8 qv[3][KS];
  float qn;
  {
    const float* q = ib + (size_t)(4 * blockIdx.x + wave) * IMGW;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int t = 0; t < KS; ++t) qv[pl][t] = *reinterpret_cast<const bf16x8*>(q + l31 * LDC + pl * PW + 8 * t + 4 * lh);
    qn = q[32 * LDC + l31];
  }

  // image -> registers -> ring slot, 16-byte accesses; the fourth round covers 40 threads.  Two register sets: an image is loaded one
  // whole phase before it is committed (an L2 round trip under load is about as long as a phase).
  float4 xa, xb4, xc, xd = make_float4(0.f, 0.f, 0.f, 0.f), ya, yb, yc, yd = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool tail = tid < IMG4 - 768;
  auto gload = [&](int tile, float4& a, float4& b_, float4& c, float4& d) {
    const float4* src = reinterpret_cast<const float4*>(ib + (size_t)tile * IMGW) + tid;
    a = src[0];
    b_ = src[256];
    c = src[512];
    if (tail) d = src[768];
  };
  auto commit = [&](int slot, const float4& a, const float4& b_, const float4& c, const float4& d) {
    float4* dst = reinterpret_cast<float4*>(cand + slot * IMGW) + tid;
    dst[0] = a;
    dst[256] = b_;
    dst[512] = c;
    if (tail) dst[768] = d;
  };
  // six split-bf16 cross terms per k-step, small terms first (knn_mfma3_kernel): two accumulators alternate over the k-steps
  auto mfma_tile = [&](int slot, f32x16& c0, f32x16& c1) {
    const float* a = cand + slot * IMGW + l31 * LDC + 4 * lh;
#pragma unroll
    for (int r = 0; r < 16; ++r) c0[r] = c1[r] = 0.f;
#pragma unroll
    for (int t = 0; t < KS; ++t) {
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(a + 8 * t), am = *reinterpret_cast<const bf16x8*>(a + PW + 8 * t),
                   al = *reinterpret_cast<const bf16x8*>(a + 2 * PW + 8 * t);
      f32x16& c = (t & 1) ? c1 : c0;
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, qv[0][t], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qv[2][t], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, qv[1][t], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, qv[0][t], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qv[1][t], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qv[0][t], c, 0, 0, 0);
    }
  };

  float bd[KP];
  int bi[KP];
#pragma unroll
  for (int t = 0; t < KP; ++t) {
    bd[t] = INFINITY;
    bi[t] = 0x7fffffff;
  }
  float* sc = dsc + wave * (32 * 64) + lane;
  unsigned live = 0;  // survivor bits of the current tile pair: bit 16*h + r = candidate r of the pair's tile h

  // One phase: distances of tile `tile` (accumulators c0/c1, image in ring slot s0) while the matrix pipe works on tile+1 (ring slot s1 ->
  // n0/n1), the image of tile+2 (registers c*, loaded last phase) goes to ring slot s2 and the load of tile+3 into l* starts.  Past the
  // last tile the same work runs on a clamped tile index and its results are dropped (no divergent control flow around the MFMAs).
  // The selection runs once per tile PAIR (H = 1, or the last tile): a wave iterates max-over-lanes(#survivors) times, and that maximum
  // over 32 candidates per lane is well below twice the one over 16 (PMC: 3.8 trips per tile before).  The pair's survivors are
  // popped in candidate order, so the lists see the same sequence of insertions as with a selection per tile -- the mask of the
  // pair's first tile is merely tested against an older (larger) threshold, and the insertion re-tests.
  auto phase = [&](auto half, int tile, int s0, int s1, int s2, f32x16& c0, f32x16& c1, f32x16& n0, f32x16& n1, float4& la, float4& lb, float4& lc,
                   float4& ld, const float4& ca, const float4& cb, const float4& cc, const float4& cd) {
    constexpr int H = decltype(half)::value;
    gload(min(tile + 3, ntiles - 1), la, lb, lc, ld);
    mfma_tile(s1, n0, n1);
    const float* cn = cand + s0 * IMGW + 32 * LDC + 4 * lh;
    const float thr = bd[KP - 1];
    unsigned m = 0;
#pragma unroll
    for (int g = 3; g >= 0; --g) {
      const float4 nrm = *reinterpret_cast<const float4*>(cn + 8 * g);
      const float nn[4] = {nrm.x, nrm.y, nrm.z, nrm.w};
#pragma unroll
      for (int u = 3; u >= 0; --u) {
        const int r = 4 * g + u;
        const float dot = c0[r] + c1[r];
        const float dr = (-2.f * dot + qn) + nn[u];  // +inf for a row past N (its norm)
        sc[(16 * H + r) * 64] = dr;
        // bit r = (dr < thr): the sign of dr - thr, shifted in from the right -- r runs downwards, so the last one lands on bit 0.
        // (inf - inf is a NaN of either sign: a false survivor costs one trip, the insertion test below decides.)
        m = __builtin_amdgcn_alignbit(m, __float_as_uint(dr - thr), 31);
      }
    }
    live = H ? (live | (m << 16)) : m;
    if (H == 1 || tile == ntiles - 1) {
      const int j0 = (tile - H) * 32 + 4 * lh;
      int rn = live ? __ffs(live) - 1 : 0;
      float dn = sc[rn * 64];  // this lane's own store: no barrier needed
      while (live) {
        const int r = rn;
        const float dv = dn;
        live &= live - 1;
        rn = live ? __ffs(live) - 1 : 0;  // the next candidate's LDS read is in flight during this one's insertion
        dn = sc[rn * 64];
        if (dv < bd[KP - 1]) {  // the threshold may have tightened since the mask was built
          const int iv = j0 + (r & 3) + 8 * ((r >> 2) & 3) + 2 * (r & 16);
          // sorted insertion, equal distances keep the lower (earlier) index first: with c[t] = dv < old[t] (c[KP-1] holds),
          //   new_d[t] = median(dv, old_d[t-1], old_d[t]),  new_i[t] = c[t-1] ? old_i[t-1] : (c[t] ? iv : old_i[t])
          bool ct = true;
#pragma unroll
          for (int t = KP - 1; t > 0; --t) {
            const bool cl = dv < bd[t - 1];
            bi[t] = cl ? bi[t - 1] : (ct ? iv : bi[t]);
            bd[t] = __builtin_amdgcn_fmed3f(dv, bd[t - 1], bd[t]);
            ct = cl;
          }
          bi[0] = ct ? iv : bi[0];
          bd[0] = fminf(bd[0], dv);
        }
      }
    }
    commit(s2, ca, cb, cc, cd);  // slot s2 held tile-1: its last readers (this wave's norms, last phase) are behind the previous barrier
    __syncthreads();
  };

  f32x16 a0, a1, b0, b1;
  gload(0, xa, xb4, xc, xd);
  gload(min(1, ntiles - 1), ya, yb, yc, yd);
  commit(0, xa, xb4, xc, xd);
  commit(1, ya, yb, yc, yd);
  gload(min(2, ntiles - 1), ya, yb, yc, yd);
  __syncthreads();
  mfma_tile(0, a0, a1);
  for (int tile = 0; tile < ntiles; tile += 2) {
    const int s = tile % 3;  // ring slots of tile, tile+1, tile+2
    const int s1 = s == 2 ? 0 : s + 1, s2 = s1 == 2 ? 0 : s1 + 1;
    phase(std::integral_constant<int, 0>{}, tile, s, s1, s2, a0, a1, b0, b1, xa, xb4, xc, xd, ya, yb, yc, yd);
    if (tile + 1 < ntiles) phase(std::integral_constant<int, 1>{}, tile + 1, s1, s2, s, b0, b1, a0, a1, ya, yb, yc, yd, xa, xb4, xc, xd);
  }

  // merge the two half-lists of a query into lane lh == 0 by (distance, index)
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
  const int q = blockIdx.x * 128 + 32 * wave + l31;
  if (lh == 0 && q < N) {
    int32_t* o = idx + ((size_t)b * N + q) * k;
#pragma unroll
    for (int t = 1; t < KP; ++t)
      if (t <= k) o[t - 1] = b * N + bi[t];
  }
}

inline int tiles_of(int N) { return 4 * cdiv(N, 128); }  // whole query groups: the tiles past N hold zero rows with +inf norms
constexpr size_t PIPE_LDS = (size_t)(3 * IMGW + 4 * 32 * 64) * sizeof(float);  // 71,552 bytes: two workgroups per CU

}  // namespace

extern "C" {

size_t spgan_knn_ws_bytes(int B, int N, int C, int k, int mode) {
  if (mode != 0 || C <= 16 || C > CP || k > KP - 1 || B <= 0 || N <= 0) return 0;
  return (size_t)B * tiles_of(N) * IMGW * sizeof(float);
}

int spgan_knn_ws(const float* x_pm, int B, int N, int C, int k, int mode, int32_t* idx, void* ws, size_t ws_bytes, spgan_stream_t s_) {
  const size_t need = spgan_knn_ws_bytes(B, N, C, k, mode);
  if (need == 0) return spgan_knn(x_pm, B, N, C, k, mode, idx, s_);
  SPGAN_CHECK_ARG(x_pm && idx && ws && ws_bytes >= need && (reinterpret_cast<uintptr_t>(ws) & 15) == 0);
  SPGAN_CHECK_ARG(k >= 1 && k + 1 <= N);
  hipStream_t s = static_cast<hipStream_t>(s_);
  const int tiles = tiles_of(N);
  float* img = static_cast<float*>(ws);
  static LdsOptIn opt;  // > 64 KB of dynamic LDS: once per kernel and device
  opt.ensure(reinterpret_cast<const void*>(&knn_pipe_kernel), (int)PIPE_LDS);
  hipLaunchKernelGGL(knn_split_kernel, dim3(tiles, B), dim3(256), 0, s, x_pm, N, C, tiles, img);
  hipLaunchKernelGGL(knn_pipe_kernel, dim3(cdiv(N, 128), B), dim3(256), PIPE_LDS, s, img, N, tiles, k, idx);
  return spgan_launch_status();
}

}  // extern "C"
