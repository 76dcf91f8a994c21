// In-launch finish of column statistics by the LAST-ARRIVING workgroup (no second "finalize" launch).
//
// Producers (GEMM epilogues, column-partial kernels) write one record (x, y) per (row-tile, column) and used to leave the merge
// over the row tiles to a tiny follow-up kernel -- 86 of them per train step, ~6 us each inside a replayed graph.  Here every
// workgroup, after publishing its records, arrives on a counter; whoever arrives last merges.  Two levels keep the serial tail
// short: the tiles are cut into groups of g1 consecutive tiles, the last arriver of a group merges the group's records into one
// group record, the last arriving GROUP merges the group records and runs the tail (mean/var + BatchNorm bookkeeping, or plain
// sums).  Every merge walks its items in index order -> the result does not depend on who arrived when: bit-reproducible.
//
// Memory protocol (MI355X_MICROARCH.md "fanin" / "publish-large", cdna_hip_programming.md G16): the 8 XCD L2s are not coherent
// with each other and a CU's L1 is never refreshed by other CUs' stores, so
//   producer : records with write-through (sc1) stores -> s_waitcnt vmcnt(0) per wave -> __syncthreads() -> ONE lane:
//              relaxed agent-scope atomic add on the counter            (no release fence: nothing dirty to write back)
//   consumer : the lane that saw the last ticket does an agent-scope acquire fence -> __syncthreads() -> plain loads.
// Counters are zero when a launch starts and are left zero by the last arriver (self-resetting; the host never memsets).
#pragma once
#include "common.hpp"

namespace fanin {

__device__ __forceinline__ void st_record(float* p, float x, float y) {  // one 8-byte write-through store
  union { float f[2]; unsigned long long u; } v;
  v.f[0] = x; v.f[1] = y;
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void chan_merge(float& n, float& a, float& b, float n2, float a2, float b2) {
  const float nn = n + n2;
  if (nn > 0.f) {
    const float d = a2 - a;
    a = a + d * (n2 / nn);
    b = b + b2 + d * d * (n * n2 / nn);
  }
  n = nn;
}

// group size of level 1 for a launch with `tiles` row tiles (host and device must agree: spgan_fanin_groups)
__host__ __device__ __forceinline__ int group_size(int tiles) { return tiles <= 48 ? tiles : (tiles <= 1536 ? 32 : 64); }
__host__ __device__ __forceinline__ int group_count(int tiles) { const int g = group_size(tiles); return (tiles + g - 1) / g; }

// One arrival: returns true for the last of `expected` arrivals on *cnt (and resets the counter).  All threads call it.
__device__ __forceinline__ bool arrive_last(int32_t* cnt, int expected, int* s_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's record stores have left
  __syncthreads();
  if (threadIdx.x == 0) {
    const int prev = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = prev == expected - 1;
    if (last) {
      __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    *s_flag = last;
  }
  __syncthreads();
  return *s_flag != 0;
}

// Merge `count` records rec[(first + i) * stride + c] (i ascending) of column c for the nc columns [c0, c0+nc) with the 256
// threads of the workgroup: thread (sl, cl) walks items sl, sl+SL, ..., then the SL slices are merged through LDS in a fixed
// tree.  mode 0: records are (sum, centred M2) of `rows_of(i)` rows -> (n, mean, M2); mode 1: plain sums.
// Result in (n, a, b) of the threads with sl == 0.  lds: 3*256 floats.
template <typename RowsOf>
__device__ __forceinline__ void merge_records(const float* rec, int first, int count, size_t stride, int c0, int nc, int mode, bool rec_is_mean,
                                              RowsOf rows_of, float* lds, float& n, float& a, float& b, int& cl_out, int& sl_out) {
  int ncp = 32;
  while (ncp < nc) ncp <<= 1;  // nc <= 256
  const int SL = 256 / ncp;
  const int cl = threadIdx.x % ncp, sl = threadIdx.x / ncp;
  n = 0.f; a = 0.f; b = 0.f;
  if (cl < nc) {
    const float2* base = reinterpret_cast<const float2*>(rec) + (size_t)first * stride + c0 + cl;
    int i = sl;
    for (; i + 3 * SL < count; i += 4 * SL) {  // four independent loads in flight
      float2 q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) q[u] = base[(size_t)(i + u * SL) * stride];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (mode == 0) {
          const float nb = rows_of(first + i + u * SL);
          chan_merge(n, a, b, nb, rec_is_mean ? q[u].x : q[u].x / nb, q[u].y);
        } else {
          a += q[u].x; b += q[u].y;
        }
      }
    }
    for (; i < count; i += SL) {
      const float2 q = base[(size_t)i * stride];
      if (mode == 0) {
        const float nb = rows_of(first + i);
        chan_merge(n, a, b, nb, rec_is_mean ? q.x : q.x / nb, q.y);
      } else {
        a += q.x; b += q.y;
      }
    }
  }
  float* sn = lds; float* sa = lds + 256; float* sb = lds + 512;
  sn[threadIdx.x] = n; sa[threadIdx.x] = a; sb[threadIdx.x] = b;
  __syncthreads();
  for (int w = SL / 2; w > 0; w >>= 1) {
    if (sl < w) {
      const int o = (sl + w) * ncp + cl;
      if (mode == 0) chan_merge(n, a, b, sn[o], sa[o], sb[o]);
      else { a += sa[o]; b += sb[o]; }
      sn[threadIdx.x] = n; sa[threadIdx.x] = a; sb[threadIdx.x] = b;
    }
    __syncthreads();
  }
  cl_out = cl; sl_out = sl;
}

// The whole protocol.  Called by all 256 threads of every workgroup that wrote records part[tile][c0 .. c0+nc) (layout
// [tiles][C] float2, written with st_record) -- `cblock` numbers the column block (its counters), G = total rows, tile_rows =
// rows per tile (the last tile may be short).  lds: 3*256 floats + 1 int, free for use.
__device__ __forceinline__ void finalize(const spgan_fanin& f, const float* part, int tile, int tiles, int C, int c0, int nc, int cblock, int G,
                                         int tile_rows, float* lds) {
  int* s_flag = reinterpret_cast<int*>(lds + 768);
  const int g1 = group_size(tiles), ng = group_count(tiles);
  const int grp = tile / g1;
  const int in_grp = min(g1, tiles - grp * g1);
  int32_t* cnt = f.counters + (size_t)cblock * (ng + 1);
  if (!arrive_last(cnt + grp, in_grp, s_flag)) return;
  float n, a, b;
  int cl, sl;
  const int mode = f.mode;
  auto tile_rows_of = [&](int t) { return (float)min(tile_rows, G - t * tile_rows); };
  merge_records(part, grp * g1, in_grp, (size_t)C, c0, nc, mode, false, tile_rows_of, lds, n, a, b, cl, sl);
  if (ng > 1) {
    if (sl == 0 && cl < nc) st_record(f.group_part + ((size_t)grp * C + c0 + cl) * 2, a, b);  // (mean, M2) | (s0, s1)
    if (!arrive_last(cnt + ng, ng, s_flag)) return;
    const int grows = g1 * tile_rows;
    auto group_rows_of = [&](int g) { return (float)min(grows, G - g * grows); };
    merge_records(f.group_part, 0, ng, (size_t)C, c0, nc, mode, true, group_rows_of, lds, n, a, b, cl, sl);
  }
  if (sl != 0 || cl >= nc) return;
  const int c = c0 + cl;
  if (mode != 0) {
    f.out0[c] = a;
    f.out1[c] = b;
    return;
  }
  const float var = b / (float)G;
  if (f.out0) f.out0[c] = a;
  if (f.out1) f.out1[c] = var;
  if (f.scale) {  // train-mode BatchNorm bookkeeping: the arithmetic of bn_prepare_kernel
    if (f.rmean) {
      const float cnt_f = (float)G * (float)max(f.count_rep, 1);
      const float unb = cnt_f > 1.f ? var * (cnt_f / (cnt_f - 1.f)) : var;
      f.rmean[c] = (1.f - f.momentum) * f.rmean[c] + f.momentum * a;
      f.rvar[c] = (1.f - f.momentum) * f.rvar[c] + f.momentum * unb;
    }
    const float inv = 1.0f / sqrtf(var + f.eps);
    const float ga = f.gamma ? f.gamma[c] : 1.f, be = f.beta ? f.beta[c] : 0.f;
    const float sc = ga * inv;
    f.scale[c] = sc;
    f.shift[c] = be - a * sc;
    f.invstd[c] = inv;
    f.mean_out[c] = a;
  }
}

}  // namespace fanin
