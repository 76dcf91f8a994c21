// kNN graph construction and the gather-side helpers of the EdgeConv path.
//
// spgan_knn never materialises the [B,N,N] distance tensor nor sorts full rows (the reference does
// both: Generation/modules.py:695-703, 16 bytes per pair).  One lane owns one query point and keeps
// its k+1 best (distance, index) pairs sorted in registers; candidate points are staged through LDS
// in tiles and broadcast to all lanes.  Order is the stable ascending (distance, index) order, rank 0
// is dropped positionally, exactly like `sort(dist)[..., 1:k+1]`.
#include "common.hpp"

namespace {

// ------------------------------------------------------------------------------------------ kNN
// KP = k+1 rounded up to a compiled capacity; CP = feature count padded to a compiled capacity
// (zero padding leaves every fmaf chain bit-identical).  One query per lane, held in registers;
// candidates are staged in LDS tiles of TC points and read as wave-wide broadcasts.
constexpr int KNN_T = 256;  // queries (threads) per workgroup (128 measured slower on MI355X: 1.22 vs 1.00 ms at C=64, B*N=65536)

template <int KP, int CP>
__global__ __launch_bounds__(KNN_T) void knn_f32_kernel(const float* __restrict__ x, int N, int C, int k,
                                                        int32_t* __restrict__ idx) {
  constexpr int TC = 64;
  __shared__ __attribute__((aligned(16))) float cand[TC * CP];
  __shared__ float cnorm[TC];
  const int b = blockIdx.y;
  const int q = blockIdx.x * KNN_T + threadIdx.x;  // query within the shape
  const float* xb = x + (size_t)b * N * C;
  const bool qok = q < N;

  float xq[CP];
#pragma unroll
  for (int c = 0; c < CP; ++c) xq[c] = (qok && c < C) ? xb[(size_t)q * C + c] : 0.f;
  float qn = 0.f;
#pragma unroll
  for (int c = 0; c < CP; ++c) qn = fmaf(xq[c], xq[c], qn);

  float bd[KP];
  int bi[KP];
#pragma unroll
  for (int t = 0; t < KP; ++t) {
    bd[t] = INFINITY;
    bi[t] = 0x7fffffff;
  }

  for (int c0 = 0; c0 < N; c0 += TC) {
    const int nc = min(TC, N - c0);
    __syncthreads();
    for (int e = threadIdx.x; e < TC * CP; e += KNN_T) {
      const int j = e / CP, c = e % CP;
      cand[e] = (j < nc && c < C) ? xb[(size_t)(c0 + j) * C + c] : 0.f;
    }
    __syncthreads();
    if (threadIdx.x < TC) {
      float sq = 0.f;
#pragma unroll
      for (int c = 0; c < CP; ++c) sq = fmaf(cand[threadIdx.x * CP + c], cand[threadIdx.x * CP + c], sq);
      cnorm[threadIdx.x] = sq;
    }
    __syncthreads();
    if (!qok) continue;
    for (int j = 0; j < nc; ++j) {
      const float4* v = reinterpret_cast<const float4*>(cand + j * CP);
      // four independent fmaf chains (channels c%4): a single chain is latency-bound at 64-128 dependent FMAs
      float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
      for (int c = 0; c < CP / 4; ++c) {
        const float4 a = v[c];
        d0 = fmaf(xq[4 * c], a.x, d0);
        d1 = fmaf(xq[4 * c + 1], a.y, d1);
        d2 = fmaf(xq[4 * c + 2], a.z, d2);
        d3 = fmaf(xq[4 * c + 3], a.w, d3);
      }
      const float dot = (d0 + d1) + (d2 + d3);
      // modules.py:699: dist = (-2*inner + |x_i|^2) + |x_j|^2
      const float d = (-2.f * dot + qn) + cnorm[j];
      if (d < bd[KP - 1]) {
        bd[KP - 1] = d;
        bi[KP - 1] = c0 + j;
#pragma unroll
        for (int t = KP - 1; t > 0; --t) {
          if (bd[t] < bd[t - 1]) {  // strict: equal distances keep the lower (earlier) index first
            const float td = bd[t]; bd[t] = bd[t - 1]; bd[t - 1] = td;
            const int ti = bi[t]; bi[t] = bi[t - 1]; bi[t - 1] = ti;
          }
        }
      }
    }
  }
  if (qok) {
    int32_t* o = idx + ((size_t)b * N + q) * k;
#pragma unroll
    for (int t = 1; t < KP; ++t)
      if (t <= k) o[t - 1] = b * N + bi[t];
  }
}

// Feature-space kNN on the matrix cores (C > 16; EdgeConv2, Generator.py:176-177 -> modules.py:695-703).
// The inner products are a [candidates x queries] GEMM; the VALU version above spends 55 % of its time on them.
//   workgroup = 4 waves x 32 queries of one shape; candidates come in tiles of 32 through a double-buffered LDS tile.
//   v_mfma_f32_32x32x2_f32 with A = candidate tile (row = candidate), B = query tile (col = query): lane (l31, lh) ends
//   up with the 16 inner products of query l31 against candidate rows (r&3) + 8*(r>>2) + 4*lh -- ascending in r, so the
//   lane's running top-(k+1) list sees its candidates in index order; the two lanes of a query are merged at the end by
//   (distance, index).  Lane (l31, lh) feeds k = lh*CP/2 + t at MFMA step t (any k permutation is legal when A and B
//   agree), i.e. it reads a contiguous half row.
//   |x|^2 comes from ONE routine (16 lanes x 4 elements, fixed shuffle tree) for queries and candidates alike, like the
//   reference's single `xx` vector (modules.py:697-699): dist = (-2*inner + xx_i) + xx_j.
template <int KP, int CP>
__global__ __launch_bounds__(256) void knn_mfma_kernel(const float* __restrict__ x, int N, int C, int k, int32_t* __restrict__ idx) {
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  constexpr int LDC = CP + 4;   // 68 / 132 floats: the 16 lanes of a b128 read phase hit 16 distinct 4-bank groups
  constexpr int H = CP / 2;     // k-slice per lane
  constexpr int SL = CP / 32;   // float4 staging slots per thread (32 rows x CP/4 float4 / 256 threads)
  __shared__ __attribute__((aligned(16))) float cand[2][32 * LDC];
  __shared__ __attribute__((aligned(16))) float cn[2][32];
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const float* xb = x + (size_t)b * N * C;
  const int qbase = blockIdx.x * 128;
  const bool vec = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);

  float4 st[SL];
  auto stage = [&](int row0) {  // global -> registers: rows row0..row0+31, zero padded
#pragma unroll
    for (int i = 0; i < SL; ++i) {
      const int e = tid + 256 * i, r = e / (CP / 4), c = (e % (CP / 4)) * 4;
      const int row = row0 + r;
      st[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < N && c < C) {
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
  auto commit = [&](int buf) {  // registers -> LDS tile + squared norms of its 32 rows
#pragma unroll
    for (int i = 0; i < SL; ++i) {
      const int e = tid + 256 * i, r = e / (CP / 4), c = (e % (CP / 4)) * 4;
      *reinterpret_cast<float4*>(&cand[buf][r * LDC + c]) = st[i];
      float s = fmaf(st[i].w, st[i].w, fmaf(st[i].z, st[i].z, fmaf(st[i].y, st[i].y, st[i].x * st[i].x)));
#pragma unroll
      for (int o = 1; o < CP / 4; o <<= 1) s += __shfl_xor(s, o);  // the CP/4 lanes of a row are consecutive
      if ((e % (CP / 4)) == 0) cn[buf][r] = s;
    }
  };

  // queries: the workgroup's 4 x 32 rows go through the same staging path, wave w keeps tile w
  float qv[H];
  float qn = 0.f;
  for (int w = 0; w < 4; ++w) {
    stage(qbase + 32 * w);
    __syncthreads();
    commit(0);
    __syncthreads();
    if (w == wave) {
#pragma unroll
      for (int t = 0; t < H; t += 4) {
        const float4 v = *reinterpret_cast<const float4*>(&cand[0][l31 * LDC + lh * H + t]);
        qv[t] = v.x; qv[t + 1] = v.y; qv[t + 2] = v.z; qv[t + 3] = v.w;
      }
      qn = cn[0][l31];
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

  const int ntiles = (N + 31) / 32;
  stage(0);
  commit(0);
  __syncthreads();
  for (int tile = 0; tile < ntiles; ++tile) {
    const int buf = tile & 1;
    if (tile + 1 < ntiles) stage((tile + 1) * 32);  // in flight under the MFMAs
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    const float* a = &cand[buf][l31 * LDC + lh * H];
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
    const int j0 = tile * 32;
    // Selection.  After the first tiles a candidate rarely beats the lane's current worst, but SOME lane of the wave
    // nearly always has one among its 16 -- running the insertion under a per-candidate branch would execute it almost
    // every time.  Instead: 16 compares build a per-lane bit mask of the survivors, and a per-lane loop pops them in
    // index order (the wave iterates max-over-lanes(#survivors) times, typically 1-2 instead of ~12).
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
      if (dv < bd[KP - 1]) {  // the threshold may have tightened since the mask was built
        bd[KP - 1] = dv;
        bi[KP - 1] = j0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
        for (int t = KP - 1; t > 0; --t) {
          if (bd[t] < bd[t - 1]) {  // strict: equal distances keep the lower (earlier) index first
            const float td = bd[t]; bd[t] = bd[t - 1]; bd[t - 1] = td;
            const int ti = bi[t]; bi[t] = bi[t - 1]; bi[t - 1] = ti;
          }
        }
      }
    }
    if (tile + 1 < ntiles) commit(buf ^ 1);  // its last readers passed the previous barrier
    __syncthreads();
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
  const int q = qbase + 32 * wave + l31;
  if (lh == 0 && q < N) {
    int32_t* o = idx + ((size_t)b * N + q) * k;
#pragma unroll
    for (int t = 1; t < KP; ++t)
      if (t <= k) o[t - 1] = b * N + bi[t];
  }
}

// The same kernel with the inner products on the bf16 matrix pipe at fp32-equivalent accuracy: every fp32 value is split exactly
// into three bfloat16 terms (hi + mid + lo) when a tile is staged into LDS, and a product is the six leading cross terms
// (v_mfma_f32_32x32x16_bf16, fp32 accumulation; the split-bf16 scheme of csrc/gemm.hip) -- 24 MFMAs of 8 passes per 32 x 32 x 64
// tile instead of 32 of 16 passes.  Distances and selection are unchanged; the squared norms still come from the fp32 values.
// The matrix-pipe time and the selection's VALU time of a wave add up on this kernel (profiles/r02_knn_phase_ablation.txt), so
// shrinking the former to 3/8 is a direct gain: see profiles/r03_knn_ab.txt.
template <int KP, int CP>
__global__ __launch_bounds__(256) void knn_mfma3_kernel(const float* __restrict__ x, int N, int C, int k, int32_t* __restrict__ idx) {
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  constexpr int PW = CP / 2;            // words per plane of a row (CP bf16)
  constexpr int LDC = 3 * PW + 4;       // 100 / 196 words: 25 / 49 four-word groups per row (odd) -> the 16 lanes of a b128 read phase hit 16 distinct groups
  constexpr int KS = CP / 16;           // MFMA k-steps (16 channels each) per tile
  constexpr int SL = CP / 32;           // float4 staging slots per thread (32 rows x CP/4 float4 / 256 threads)
  __shared__ __attribute__((aligned(16))) float cand[2][32 * LDC];
  __shared__ __attribute__((aligned(16))) float cn[2][32];
  __shared__ float dsc[4][16 * 64];     // per wave: the tile's 16 distances of every lane, [r][lane]
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const float* xb = x + (size_t)b * N * C;
  const int qbase = blockIdx.x * 128;
  const bool vec = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);

  float4 st[SL];
  auto stage = [&](int row0) {  // global -> registers: rows row0..row0+31, zero padded
#pragma unroll
    for (int i = 0; i < SL; ++i) {
      const int e = tid + 256 * i, r = e / (CP / 4), c = (e % (CP / 4)) * 4;
      const int row = row0 + r;
      st[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < N && c < C) {
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
  auto commit = [&](int buf) {  // registers -> LDS tile + squared norms of its 32 rows
#pragma unroll
    for (int i = 0; i < SL; ++i) {
      const int e = tid + 256 * i, r = e / (CP / 4), c = (e % (CP / 4)) * 4;
      {  // v = hi + mid + lo exactly (three bf16 terms = fp32's 24 significand bits); plane p at words [p*PW, (p+1)*PW) of the row
        const f32x4v f = {st[i].x, st[i].y, st[i].z, st[i].w};
        const bf16x4 hi = __builtin_convertvector(f, bf16x4);
        const f32x4v r1 = f - __builtin_convertvector(hi, f32x4v);
        const bf16x4 mid = __builtin_convertvector(r1, bf16x4);
        const f32x4v r2 = r1 - __builtin_convertvector(mid, f32x4v);
        const bf16x4 lo = __builtin_convertvector(r2, bf16x4);
        float* row = &cand[buf][r * LDC + (c >> 1)];
        *reinterpret_cast<bf16x4*>(row) = hi;
        *reinterpret_cast<bf16x4*>(row + PW) = mid;
        *reinterpret_cast<bf16x4*>(row + 2 * PW) = lo;
      }
      float s = fmaf(st[i].w, st[i].w, fmaf(st[i].z, st[i].z, fmaf(st[i].y, st[i].y, st[i].x * st[i].x)));
#pragma unroll
      for (int o = 1; o < CP / 4; o <<= 1) s += __shfl_xor(s, o);  // the CP/4 lanes of a row are consecutive
      if ((e % (CP / 4)) == 0) cn[buf][r] = s;
    }
  };

  // queries: the workgroup's 4 x 32 rows go through the same staging path, wave w keeps tile w
  bf16x8 qv[3][KS];   // the query's three planes: lane (l31, lh) holds channels 16*t + 8*lh .. +7 of k-step t
  float qn = 0.f;
  for (int w = 0; w < 4; ++w) {
    stage(qbase + 32 * w);
    __syncthreads();
    commit(0);
    __syncthreads();
    if (w == wave) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int t = 0; t < KS; ++t) qv[pl][t] = *reinterpret_cast<const bf16x8*>(&cand[0][l31 * LDC + pl * PW + 8 * t + 4 * lh]);
      qn = cn[0][l31];
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

  const int ntiles = (N + 31) / 32;
  stage(0);
  commit(0);
  __syncthreads();
  for (int tile = 0; tile < ntiles; ++tile) {
    const int buf = tile & 1;
    if (tile + 1 < ntiles) stage((tile + 1) * 32);  // in flight under the MFMAs
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    const float* a = &cand[buf][l31 * LDC + 4 * lh];
#pragma unroll
    for (int t = 0; t < KS; ++t) {
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(a + 8 * t), am = *reinterpret_cast<const bf16x8*>(a + PW + 8 * t),
                   al = *reinterpret_cast<const bf16x8*>(a + 2 * PW + 8 * t);
      // a*q ~ ah*qh + (ah*qm + am*qh) + (ah*ql + al*qh + am*qm): every partial product is exact in fp32, the dropped terms are
      // <= 3*2^-24 relative -- fp32-equivalent inner products at 6/16 of the fp32-MFMA time; small terms first
      f32x16& c = (t & 1) ? acc1 : acc0;
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, qv[0][t], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qv[2][t], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, qv[1][t], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, qv[0][t], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qv[1][t], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qv[0][t], c, 0, 0, 0);
    }
    const int j0 = tile * 32;
    // Selection.  After the first tiles a candidate rarely beats the lane's current worst, but SOME lane of the wave
    // nearly always has one among its 16 -- running the insertion under a per-candidate branch would execute it almost
    // every time.  Instead: 16 compares build a per-lane bit mask of the survivors, and a per-lane loop pops them in
    // index order (the wave iterates max-over-lanes(#survivors) times, typically 1-2 instead of ~12).
    // Round 4 (PMC: the kernel is VALU-issue-bound -- 612 VALU instructions per tile and wave, 69 % of a SIMD's cycles, of which ~3.8
    // trips x 89 were this loop): the 16 distances go to an LDS scratch [r][lane] (conflict-free 4-byte stores), so that a trip reads
    // its candidate by address instead of a 15-deep select chain on a per-lane register index, and the sorted insertion is one
    // v_med3_f32 per slot -- new[t] = median(dv, old[t-1], old[t]) -- with the indices following through two selects.
    float* sc = &dsc[wave][lane];
    unsigned live = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 nrm = *reinterpret_cast<const float4*>(&cn[buf][8 * g + 4 * lh]);
      const float nn[4] = {nrm.x, nrm.y, nrm.z, nrm.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = 4 * g + u;
        const float dot = acc0[r] + acc1[r];
        float dr = (-2.f * dot + qn) + nn[u];
        if (j0 + 8 * g + 4 * lh + u >= N) dr = INFINITY;
        sc[r * 64] = dr;
        live |= (dr < bd[KP - 1]) ? (1u << r) : 0u;
      }
    }
    int rn = live ? __ffs(live) - 1 : 0;
    float dn = sc[rn * 64];             // this lane's own store: no barrier needed
    while (live) {
      const int r = rn;
      const float dv = dn;
      live &= live - 1;
      rn = live ? __ffs(live) - 1 : 0;  // the next candidate's LDS read is in flight during this one's insertion
      dn = sc[rn * 64];
      if (dv < bd[KP - 1]) {  // the threshold may have tightened since the mask was built
        const int iv = j0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
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
    if (tile + 1 < ntiles) commit(buf ^ 1);  // its last readers passed the previous barrier
    __syncthreads();
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
  const int q = qbase + 32 * wave + l31;
  if (lh == 0 && q < N) {
    int32_t* o = idx + ((size_t)b * N + q) * k;
#pragma unroll
    for (int t = 1; t < KP; ++t)
      if (t <= k) o[t - 1] = b * N + bi[t];
  }
}

// fp64 direct differences for coordinate-space inputs (C <= 8): the query sits in registers.
template <int KP, int C>
__global__ __launch_bounds__(256) void knn_f64_kernel(const float* __restrict__ x, int N, int k, int32_t* __restrict__ idx) {
  constexpr int TC = 256;
  __shared__ float cand[TC * C];
  const int b = blockIdx.y;
  const int q = blockIdx.x * 256 + threadIdx.x;
  const float* xb = x + (size_t)b * N * C;
  const bool qok = q < N;
  double xq[C];
#pragma unroll
  for (int c = 0; c < C; ++c) xq[c] = qok ? (double)xb[(size_t)q * C + c] : 0.0;
  double bd[KP];
  int bi[KP];
#pragma unroll
  for (int t = 0; t < KP; ++t) {
    bd[t] = INFINITY;
    bi[t] = 0x7fffffff;
  }
  for (int c0 = 0; c0 < N; c0 += TC) {
    const int nc = min(TC, N - c0);
    __syncthreads();
    for (int e = threadIdx.x; e < nc * C; e += 256) cand[e] = xb[(size_t)c0 * C + e];
    __syncthreads();
    if (!qok) continue;
    for (int j = 0; j < nc; ++j) {
      double d = 0.0;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const double t = xq[c] - (double)cand[j * C + c];
        d = fma(t, t, d);
      }
      if (d < bd[KP - 1]) {
        bd[KP - 1] = d;
        bi[KP - 1] = c0 + j;
#pragma unroll
        for (int t = KP - 1; t > 0; --t) {
          if (bd[t] < bd[t - 1]) {
            const double td = bd[t]; bd[t] = bd[t - 1]; bd[t - 1] = td;
            const int ti = bi[t]; bi[t] = bi[t - 1]; bi[t - 1] = ti;
          }
        }
      }
    }
  }
  if (qok) {
    int32_t* o = idx + ((size_t)b * N + q) * k;
#pragma unroll
    for (int t = 1; t < KP; ++t)
      if (t <= k) o[t - 1] = b * N + bi[t];
  }
}

template <int KP>
int launch_knn(const float* x, int B, int N, int C, int k, int mode, int32_t* idx, hipStream_t s) {
  dim3 grid(cdiv(N, 256), B), block(256);
  if (mode == 1) {
    switch (C) {
      case 1: hipLaunchKernelGGL((knn_f64_kernel<KP, 1>), grid, block, 0, s, x, N, k, idx); break;
      case 2: hipLaunchKernelGGL((knn_f64_kernel<KP, 2>), grid, block, 0, s, x, N, k, idx); break;
      case 3: hipLaunchKernelGGL((knn_f64_kernel<KP, 3>), grid, block, 0, s, x, N, k, idx); break;
      case 4: hipLaunchKernelGGL((knn_f64_kernel<KP, 4>), grid, block, 0, s, x, N, k, idx); break;
      default: return SPGAN_EINVAL;
    }
  } else {
    grid = dim3(cdiv(N, KNN_T), B);
    block = dim3(KNN_T);
    if (C <= 8) hipLaunchKernelGGL((knn_f32_kernel<KP, 8>), grid, block, 0, s, x, N, C, k, idx);
    else if (C <= 16) hipLaunchKernelGGL((knn_f32_kernel<KP, 16>), grid, block, 0, s, x, N, C, k, idx);
    else if (C > 128) return SPGAN_EINVAL;
    else if constexpr (KP == 11) {  // k <= 10 (SP-GAN: nk/2 = 10): matrix-core distances; the longer lists spill there
      static const bool x3 = !(getenv("SPGAN_KNN_BF16X3") && atoi(getenv("SPGAN_KNN_BF16X3")) == 0);   // 0: the fp32-MFMA kernel (A/B measurements)
      if (x3) {
        if (C <= 64) hipLaunchKernelGGL((knn_mfma3_kernel<KP, 64>), dim3(cdiv(N, 128), B), dim3(256), 0, s, x, N, C, k, idx);
        else hipLaunchKernelGGL((knn_mfma3_kernel<KP, 128>), dim3(cdiv(N, 128), B), dim3(256), 0, s, x, N, C, k, idx);
      } else if (C <= 64) hipLaunchKernelGGL((knn_mfma_kernel<KP, 64>), dim3(cdiv(N, 128), B), dim3(256), 0, s, x, N, C, k, idx);
      else hipLaunchKernelGGL((knn_mfma_kernel<KP, 128>), dim3(cdiv(N, 128), B), dim3(256), 0, s, x, N, C, k, idx);
    } else {
      if (C <= 32) hipLaunchKernelGGL((knn_f32_kernel<KP, 32>), grid, block, 0, s, x, N, C, k, idx);
      else if (C <= 64) hipLaunchKernelGGL((knn_f32_kernel<KP, 64>), grid, block, 0, s, x, N, C, k, idx);
      else hipLaunchKernelGGL((knn_f32_kernel<KP, 128>), grid, block, 0, s, x, N, C, k, idx);
    }
  }
  return spgan_launch_status();
}

// ------------------------------------------------------------------------------------------ CSR of in-edges
// One workgroup per shape (edges never cross shapes, and each shape has exactly N*k edges, so its
// segment of `src` starts at b*N*k).  Degree count and slot assignment use LDS integer atomics
// (order-independent results after the per-segment sort).
// LDS_SEG: the segments are filled and sorted in LDS as 16-bit local edge ids (E <= 65536) and leave with one coalesced store --
// the per-point insertion sort is a chain of dependent accesses, ~10x shorter on LDS than on L2 (67 -> 2x us at B=32, N=2048).
template <bool LDS_SEG>
__global__ __launch_bounds__(1024) void csr_kernel(const int32_t* __restrict__ idx, int N, int k,
                                                   int32_t* __restrict__ rowptr, int32_t* __restrict__ src) {
  extern __shared__ int ism[];
  int* deg = ism;          // [N]
  int* start = ism + N;    // [N]   exclusive scan of deg
  int* wsum = ism + 2 * N; // [32]
  unsigned short* lseg = reinterpret_cast<unsigned short*>(ism + 2 * N + 32);   // [E]  (LDS_SEG)
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int E = N * k;
  const int32_t* ib = idx + (size_t)b * E;
  for (int i = tid; i < N; i += nt) deg[i] = 0;
  __syncthreads();
  for (int e = tid; e < E; e += nt) atomicAdd(&deg[ib[e] - b * N], 1);
  __syncthreads();
  // blocked exclusive scan: thread t owns a contiguous chunk
  const int chunk = (N + nt - 1) / nt;
  const int lo = min(N, tid * chunk), hi = min(N, lo + chunk);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += deg[i];
  // scan of per-thread sums: wave scan + cross-wave
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
  }
  __syncthreads();
  int run = wsum[wv] + incl - s;
  for (int i = lo; i < hi; ++i) {
    start[i] = run;
    run += deg[i];
  }
  __syncthreads();
  for (int i = tid; i < N; i += nt) {
    rowptr[(size_t)b * N + i] = b * E + start[i];
    deg[i] = 0;  // reuse as cursor
  }
  if (b == gridDim.x - 1 && tid == 0) rowptr[(size_t)gridDim.x * N] = gridDim.x * E;
  __syncthreads();
  if constexpr (LDS_SEG) {
    for (int e = tid; e < E; e += nt) {
      const int j = ib[e] - b * N;
      const int slot = atomicAdd(&deg[j], 1);
      lseg[start[j] + slot] = (unsigned short)e;
    }
    __syncthreads();
    // sort each segment ascending (deterministic summation order downstream)
    for (int i = tid; i < N; i += nt) {
      unsigned short* seg = lseg + start[i];
      const int n = deg[i];
      for (int a = 1; a < n; ++a) {
        const unsigned short v = seg[a];
        int c = a - 1;
        while (c >= 0 && seg[c] > v) {
          seg[c + 1] = seg[c];
          --c;
        }
        seg[c + 1] = v;
      }
    }
    __syncthreads();
    for (int e = tid; e < E; e += nt) src[(size_t)b * E + e] = b * E + (int)lseg[e];
  } else {
    for (int e = tid; e < E; e += nt) {
      const int j = ib[e] - b * N;
      const int slot = atomicAdd(&deg[j], 1);
      src[(size_t)b * E + start[j] + slot] = b * E + e;
    }
    __threadfence_block();
    __syncthreads();
    for (int i = tid; i < N; i += nt) {
      int32_t* seg = src + (size_t)b * E + start[i];
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
}

// ------------------------------------------------------------------------------------------ gathers / layout
__global__ void edge_features_cm_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx, int C, int N, int k,
                                        float* __restrict__ ee) {
  // one thread per (b, c, n, r); output [B,2C,N,k] -- modules.py:708-720
  const size_t total = (size_t)gridDim.y * C * N * k;  // gridDim.y == B
  const int b = blockIdx.y;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)C * N * k) return;
  (void)total;
  const int r = t % k;
  const int n = (t / k) % N;
  const int c = t / ((size_t)k * N);
  const float* xb = x + (size_t)b * C * N;
  const int64_t j = idx[(size_t)b * N * k + (size_t)n * k + r];
  const float ctr = xb[(size_t)c * N + n];
  const float nb = xb[(size_t)c * N + j];
  float* eb = ee + (size_t)b * 2 * C * N * k;
  eb[((size_t)c * N + n) * k + r] = ctr;
  eb[((size_t)(C + c) * N + n) * k + r] = nb - ctr;
}

__global__ void idx_to_local64_kernel(const int32_t* __restrict__ idx, int N, size_t total, size_t per_batch, int64_t* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int b = t / per_batch;
  out[t] = (int64_t)(idx[t] - b * N);
}
__global__ void idx_from_local64_kernel(const int64_t* __restrict__ idx, int N, size_t total, size_t per_batch, int32_t* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int b = t / per_batch;
  out[t] = (int32_t)idx[t] + b * N;
}

// [B,C,N] -> [B*N,C] through a 32x32 LDS tile (coalesced on both sides)
__global__ void cm_to_pm_kernel(const float* __restrict__ x, int C, int N, float* __restrict__ y) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, n = n0 + tx;
    tile[i][tx] = (c < C && n < N) ? x[((size_t)b * C + c) * N + n] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int n = n0 + i, c = c0 + tx;
    if (n < N && c < C) y[((size_t)b * N + n) * C + c] = tile[tx][i];
  }
}
__global__ void pm_to_cm_kernel(const float* __restrict__ x, int C, int N, float* __restrict__ y) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int n = n0 + i, c = c0 + tx;
    tile[i][tx] = (c < C && n < N) ? x[((size_t)b * N + n) * C + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, n = n0 + tx;
    if (n < N && c < C) y[((size_t)b * C + c) * N + n] = tile[tx][i];
  }
}

// dst[e] = src[e]^T for up to SPGAN_MULTI_MAX row-major matrices in one launch (the transposed weights the input-gradient GEMMs
// read, refreshed once per optimiser step): 32x32 tiles through LDS, block b serves entry e with tile_start[e] <= b < tile_start[e+1].
__global__ __launch_bounds__(256) void multi_transpose_kernel(const spgan_multi_transpose_args a) {
  __shared__ float tile[32][33];
  int e = 0;
  while (e + 1 < a.count && (int)blockIdx.x >= a.tile_start[e + 1]) ++e;
  const int rows = a.rows[e], cols = a.cols[e], ld = a.ld[e];
  const int tcols = (cols + 31) >> 5;
  const int t = (int)blockIdx.x - a.tile_start[e];
  const int r0 = (t / tcols) * 32, c0 = (t % tcols) * 32;
  const float* __restrict__ src = a.src[e];
  float* __restrict__ dst = a.dst[e];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[(size_t)r * ld + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) dst[(size_t)c * rows + r] = tile[tx][i];
  }
}

__global__ void concat2_kernel(const float* __restrict__ a, int Ca, const float* __restrict__ b, int Cb, size_t M,
                               float* __restrict__ out) {
  const int Ct = Ca + Cb;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * Ct) return;
  const size_t m = t / Ct;
  const int c = t % Ct;
  out[t] = c < Ca ? a[m * Ca + c] : b[m * Cb + (c - Ca)];
}

}  // namespace

extern "C" int spgan_knn(const float* x, int B, int N, int C, int k, int mode, int32_t* idx, spgan_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  SPGAN_CHECK_ARG(x && idx && B > 0 && N > 1 && C > 0 && k > 0 && k <= 32 && k + 1 <= N);
  SPGAN_CHECK_ARG(mode == 0 || (mode == 1 && C <= 4));
  SPGAN_CHECK_ARG(mode == 1 || C <= 128);
  if (k <= 10) return launch_knn<11>(x, B, N, C, k, mode, idx, s);
  if (k <= 20) return launch_knn<21>(x, B, N, C, k, mode, idx, s);
  return launch_knn<33>(x, B, N, C, k, mode, idx, s);
}

extern "C" int spgan_csr_build(const int32_t* idx, int B, int N, int k, int32_t* rowptr, int32_t* src, spgan_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  SPGAN_CHECK_ARG(idx && rowptr && src && B > 0 && N > 0 && k > 0 && N <= 16384);
  const size_t sh = (size_t)(2 * N + 32) * sizeof(int);
  const size_t E = (size_t)N * k, sh_seg = sh + ((E * sizeof(unsigned short) + 15) & ~(size_t)15);
  if (E <= 65536 && sh_seg <= 160 * 1024) {
    static LdsOptIn opt;  // > 64 KB of dynamic LDS: once per kernel and device
    opt.ensure(reinterpret_cast<const void*>(&csr_kernel<true>), 160 * 1024);
    hipLaunchKernelGGL(csr_kernel<true>, dim3(B), dim3(1024), sh_seg, s, idx, N, k, rowptr, src);
  } else {
    hipLaunchKernelGGL(csr_kernel<false>, dim3(B), dim3(1024), sh, s, idx, N, k, rowptr, src);
  }
  return spgan_launch_status();
}

extern "C" int spgan_edge_features_cm(const float* x, const int64_t* idx, int B, int C, int N, int k, float* ee, spgan_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  SPGAN_CHECK_ARG(x && idx && ee && B > 0 && C > 0 && N > 0 && k > 0);
  const size_t per = (size_t)C * N * k;
  hipLaunchKernelGGL(edge_features_cm_kernel, dim3(cdiv(per, 256), B), dim3(256), 0, s, x, idx, C, N, k, ee);
  return spgan_launch_status();
}

extern "C" int spgan_idx_to_local64(const int32_t* idx, int B, int N, int k, int64_t* out, spgan_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  SPGAN_CHECK_ARG(idx && out && B > 0 && N > 0 && k > 0);
  const size_t total = (size_t)B * N * k;
  hipLaunchKernelGGL(idx_to_local64_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, idx, N, total, (size_t)N * k, out);
  return spgan_launch_status();
}
extern "C" int spgan_idx_from_local64(const int64_t* idx, int B, int N, int k, int32_t* out, spgan_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  SPGAN_CHECK_ARG(idx && out && B > 0 && N > 0 && k > 0);
  const size_t total = (size_t)B * N * k;
  hipLaunchKernelGGL(idx_from_local64_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, idx, N, total, (size_t)N * k, out);
  return spgan_launch_status();
}

extern "C" int spgan_cm_to_pm(const float* x, int B, int C, int N, float* y, spgan_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  SPGAN_CHECK_ARG(x && y && B > 0 && C > 0 && N > 0);
  hipLaunchKernelGGL(cm_to_pm_kernel, dim3(cdiv(N, 32), cdiv(C, 32), B), dim3(256), 0, s, x, C, N, y);
  return spgan_launch_status();
}
extern "C" int spgan_pm_to_cm(const float* x, int B, int C, int N, float* y, spgan_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  SPGAN_CHECK_ARG(x && y && B > 0 && C > 0 && N > 0);
  hipLaunchKernelGGL(pm_to_cm_kernel, dim3(cdiv(N, 32), cdiv(C, 32), B), dim3(256), 0, s, x, C, N, y);
  return spgan_launch_status();
}
extern "C" int spgan_multi_transpose(const spgan_multi_transpose_args* a, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(a && a->count > 0 && a->count <= SPGAN_MULTI_MAX && a->tile_start[0] == 0);
  for (int e = 0; e < a->count; ++e) {
    SPGAN_CHECK_ARG(a->src[e] && a->dst[e] && a->rows[e] > 0 && a->cols[e] > 0 && a->ld[e] >= a->cols[e]);
    SPGAN_CHECK_ARG(a->tile_start[e + 1] - a->tile_start[e] == cdiv(a->rows[e], 32) * cdiv(a->cols[e], 32));
  }
  hipLaunchKernelGGL(multi_transpose_kernel, dim3(a->tile_start[a->count]), dim3(256), 0, (hipStream_t)s_, *a);
  return spgan_launch_status();
}
extern "C" int spgan_concat2(const float* a, int Ca, const float* b, int Cb, int M, float* out, spgan_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  SPGAN_CHECK_ARG(a && b && out && Ca > 0 && Cb > 0 && M > 0);
  const size_t total = (size_t)M * (Ca + Cb);
  hipLaunchKernelGGL(concat2_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, a, Ca, b, Cb, (size_t)M, out);
  return spgan_launch_status();
}
