// The row-sparse product S.W of the max-pool gradient pattern (see gemm.hip: spgan_sparse_rows_nt) as a device function, so that more than
// one kernel can run it.
#pragma once
#include "common.hpp"

// E[m, :] = sum over the channels c whose arg[b,c] == m of val[b,c] * W[c, :]   (b = m / rows): the row-sparse product
// S.W of the max-pool gradient pattern, written densely (zero rows included) so that a GEMM epilogue can add it as a
// per-row addend.  Workgroup = (shape, chunk of RB rows).  The incidence is inverted in LDS as one Cs-bit mask per row
// (atomicOr: the result does not depend on the order), then a wave walks its rows, skips the (mostly) empty masks with a
// single ballot and adds the W rows of the set bits in ascending c (deterministic), lanes over columns.
// (A body shared by the stand-alone kernel of gemm.hip and the fused collapse_prep launch of collapse.hip: bx = row chunk, b = shape.)
__device__ __forceinline__ void sparse_rows_nt_body(int bx, int b, unsigned* sm_u, const float* __restrict__ val, const int32_t* __restrict__ arg, int rows,
                                                    int Cs, const float* __restrict__ W, int ldw, int N, float* __restrict__ E, int lde, int RB) {
  const int words = (Cs + 63) / 64;                             // 64-bit words per row mask
  unsigned* mask = sm_u;                                        // [RB][2*words]
  float* sval = reinterpret_cast<float*>(sm_u + RB * 2 * words);  // [Cs]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = bx * RB;
  for (int i = tid; i < RB * 2 * words; i += 256) mask[i] = 0u;
  __syncthreads();
  for (int c = tid; c < Cs; c += 256) {
    sval[c] = val[(size_t)b * Cs + c];
    const int r = arg[(size_t)b * Cs + c] - b * rows - r0;
    if (r >= 0 && r < RB) atomicOr(&mask[r * 2 * words + (c >> 5)], 1u << (c & 31));
  }
  __syncthreads();
  for (int rl = wave; rl < RB && r0 + rl < rows; rl += 4) {
    float* e = E + ((size_t)b * rows + r0 + rl) * lde;
    const unsigned long long* mrow = reinterpret_cast<const unsigned long long*>(mask + rl * 2 * words);
    for (int n0 = 0; n0 < N; n0 += 256) {
      const int n = n0 + lane * 4;
      const bool vec = n0 + 256 <= N && (ldw & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0;   // all 64 lanes on whole, aligned float4s
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int w0 = 0; w0 < words; w0 += 64) {
        const unsigned long long mine = (w0 + lane < words) ? mrow[w0 + lane] : 0ull;
        unsigned long long nz = __ballot(mine != 0ull);
        while (nz) {
          const int wl = __ffsll((long long)nz) - 1;
          nz &= nz - 1;
          unsigned long long bits = __shfl(mine, wl);
          if (vec) {
            // up to 8 W rows in flight per round (the set bits are wave-uniform: scalar indices): a row that carries the arg-max of dozens of
            // channels -- a few extreme points of a shape do -- was a chain of that many dependent row loads; the sums keep ascending c
            while (bits) {
              int cc[8];
              float4 wv[8];
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                cc[u] = bits ? (w0 + wl) * 64 + __ffsll((long long)bits) - 1 : -1;
                bits &= bits - 1;   // 0 stays 0
              }
#pragma unroll
              for (int u = 0; u < 8; ++u) wv[u] = *reinterpret_cast<const float4*>(W + (size_t)max(cc[u], 0) * ldw + n);
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                if (cc[u] >= 0) {
                  const float v = sval[cc[u]];
                  acc.x = fmaf(v, wv[u].x, acc.x); acc.y = fmaf(v, wv[u].y, acc.y); acc.z = fmaf(v, wv[u].z, acc.z); acc.w = fmaf(v, wv[u].w, acc.w);
                }
              }
            }
          }
          while (bits) {
            const int cc = (w0 + wl) * 64 + __ffsll((long long)bits) - 1;
            bits &= bits - 1;
            const float v = sval[cc];
            const float* w = W + (size_t)cc * ldw + n;
            if (n + 3 < N) {
              acc.x = fmaf(v, w[0], acc.x); acc.y = fmaf(v, w[1], acc.y); acc.z = fmaf(v, w[2], acc.z); acc.w = fmaf(v, w[3], acc.w);
            } else {
              if (n < N) acc.x = fmaf(v, w[0], acc.x);
              if (n + 1 < N) acc.y = fmaf(v, w[1], acc.y);
              if (n + 2 < N) acc.z = fmaf(v, w[2], acc.z);
            }
          }
        }
      }
      if (n + 3 < N && (lde & 3) == 0 && (reinterpret_cast<uintptr_t>(E) & 15) == 0) {
        *reinterpret_cast<float4*>(e + n) = acc;
      } else {
        if (n < N) e[n] = acc.x;
        if (n + 1 < N) e[n + 1] = acc.y;
        if (n + 2 < N) e[n + 2] = acc.z;
        if (n + 3 < N) e[n + 3] = acc.w;
      }
    }
  }
}


// rows per workgroup and dynamic LDS bytes of sparse_rows_nt_body for Cs channels
static inline int sparse_rows_nt_rb(int rows, int Cs) {
  const int words = cdiv(Cs, 64);
  int RB = (32 * 1024) / (words * 8);  // at most 32 KB of row masks per workgroup
  if (RB > 32) RB = 32;               // 8 rows per wave: B*rows/32 workgroups (2048 at C2) -- with 256 rows a launch had one workgroup per CU and was latency-bound (61 us for 67 MB)
  if (RB > rows) RB = rows;
  return RB;
}
static inline size_t sparse_rows_nt_lds(int RB, int Cs) { return (size_t)RB * cdiv(Cs, 64) * 8 + (size_t)Cs * 4; }
