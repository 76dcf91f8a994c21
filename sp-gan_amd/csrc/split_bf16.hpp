// Exact split of fp32 values into three bfloat16 terms (hi + mid + lo, round to nearest at each level: every residual is representable) and
// the 8-byte-per-plane LDS store of four consecutive k-values -- shared by the split-bf16 kernels (gemm_wide3.hip, gemm_tn_wide3.hip).
#pragma once
#include "common.hpp"

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

// two fp32 values -> (hi, mid, lo) as packed bf16 pairs
__device__ __forceinline__ void split2(float x, float y, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  const f32x2v v = {x, y};
  const bf16x2 h = __builtin_convertvector(v, bf16x2);
  hi = *reinterpret_cast<const uint32_t*>(&h);
  const float rx = x - __uint_as_float(hi << 16), ry = y - __uint_as_float(hi & 0xffff0000u);
  const f32x2v r = {rx, ry};
  const bf16x2 m = __builtin_convertvector(r, bf16x2);
  mid = *reinterpret_cast<const uint32_t*>(&m);
  const f32x2v q = {rx - __uint_as_float(mid << 16), ry - __uint_as_float(mid & 0xffff0000u)};
  const bf16x2 l = __builtin_convertvector(q, bf16x2);
  lo = *reinterpret_cast<const uint32_t*>(&l);
}

// 4 consecutive k-values -> 8 bytes in each of the three planes (p points at the hi plane's slot, `plane` words between planes)
__device__ __forceinline__ void st_split4(uint32_t* p, int plane, float4 v) {
  uint32_t h0, m0, l0, h1, m1, l1;
  split2(v.x, v.y, h0, m0, l0);
  split2(v.z, v.w, h1, m1, l1);
  *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(p + plane) = make_uint2(m0, m1);
  *reinterpret_cast<uint2*>(p + 2 * plane) = make_uint2(l0, l1);
}

