// Epilogue of the 256-row-tile gemm_nt kernels (gemm_wide.hip: fp32 / fp16 operands; gemm_wide3.hip: split-bf16 operands): a wave owns
// 128 rows x (TJ*32) columns of the tile as TI x TJ MFMA accumulator tiles of 32x32 -- exactly one statistics / pooling record tile
// (the [M/128, N, 2] records of spgan_hip.h) per column, so nothing is exchanged between waves: lane halves merge by one shuffle.
// One body for every operand mode: the bits of an output depend on the accumulators only.
#pragma once
#include "common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// internal epilogue id (template argument only, never in spgan_gemm_nt_args.epi_mode): BNBWD whose stored tile is gout_add + gout_scale * g.
// Its own instantiation (gemm_wide3.hip), so that the BNBWD kernels keep their registers.
constexpr int SPGAN_WIDE_EPI_BNBWD_GOUT = 8;

#define ROFF(r) (((r) & 3) + 8 * ((r) >> 2))   // C/D layout of v_mfma_f32_32x32x*: register r of lane l holds row ROFF(r) + 4*(l >> 5), column l & 31

// m0 / n0: first row / column of the workgroup's tile; wm / wn: the wave's position in the tile; H16: the 16-bit result storage modes
// (y_bf16 / y_half of the fp16-operand mode) are compiled in.
template <int EPI, int TI, int TJ, bool H16>
__device__ __forceinline__ void wide_epilogue(const spgan_gemm_nt_args& p, f32x16 (&acc)[TI][TJ], int m0, int n0, int wm, int wn, int l31, int lh) {
  static_assert(TI * 32 == 128, "a wave's rows are one 128-row record tile");
  constexpr bool F16 = H16;
  const int rbase = m0 + wm * (TI * 32) + 4 * lh;
  const int cbase = n0 + wn * (TJ * 32) + l31;
  const int rec = (m0 >> 7) + wm;      // the 128-row record tile this wave's rows form
  constexpr float NL = (float)(16 * TI);  // rows a lane holds per column

  if constexpr (EPI == SPGAN_EPI_LINEAR) {
    const bool rb_dense = p.rowbias && p.rows_per_group == 1;
    const bool rb_group = p.rowbias && p.rows_per_group > 1;  // host: rows_per_group % 256 == 0 -> one group per tile
    const int g0 = rb_group ? m0 / p.rows_per_group : 0;
    float csum[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      float b = p.bias ? p.bias[cbase + j * 32] : 0.f;
      if (rb_group) b += p.rowbias[(size_t)g0 * p.ld_rowbias + cbase + j * 32];
      csum[j] = 0.f;
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        if (rb_dense) {
          const float* ab = p.rowbias + (size_t)(rbase + i * 32) * p.ld_rowbias + cbase + j * 32;
          const unsigned ld2 = (unsigned)p.ld_rowbias;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] += ab[(unsigned)ROFF(r) * ld2];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[i][j][r] + b;
          acc[i][j][r] = v;  // pre-activation value: what the statistics and the pooling see
          csum[j] += v;
        }
      }
    }
    if (F16 && p.Y && p.y_bf16) {   // 16-bit result storage (no activation: checked on the host): bfloat16 (the EdgeBlock's dT)
      __bf16* yb = reinterpret_cast<__bf16*>(p.Y) + (size_t)rbase * p.ldy + cbase;
      const unsigned ldy = (unsigned)p.ldy;
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) yb[(size_t)((unsigned)(i * 32 + ROFF(r)) * ldy + (unsigned)(j * 32))] = (__bf16)acc[i][j][r];
    } else if (F16 && p.Y && p.y_half) {   // ... or fp16
      _Float16* yb = reinterpret_cast<_Float16*>(p.Y) + (size_t)rbase * p.ldy + cbase;
      const unsigned ldy = (unsigned)p.ldy;
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) yb[(size_t)((unsigned)(i * 32 + ROFF(r)) * ldy + (unsigned)(j * 32))] = (_Float16)acc[i][j][r];
    } else if (p.Y) {
      float* yb = p.Y + (size_t)rbase * p.ldy + cbase;
      const unsigned ldy = (unsigned)p.ldy;
      auto store_all = [&](auto actf) {
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
          for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) yb[(size_t)((unsigned)(i * 32 + ROFF(r)) * ldy + (unsigned)(j * 32))] = actf(acc[i][j][r]);
      };
      if (p.act == SPGAN_ACT_LRELU) {
        const float sl = p.act_slope;
        store_all([sl](float v) { return lrelu_f(v, sl); });
      } else if (p.act == SPGAN_ACT_TANH) {
        store_all([](float v) { return tanhf(v); });
      } else {
        store_all([](float v) { return v; });
      }
    }
    if (p.stats || p.pool_val) {
      // every lane holds NL rows of a column: (sum, M2 about the lane's own mean), merged with the other lane half by Chan's formula
      const bool do_pool = p.pool_val != nullptr;
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const float sl = csum[j], mean = sl * (1.f / NL);
        float m2 = 0.f, vx = -INFINITY, vn = INFINITY;
        int ax = 0x7fffffff, an = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[i][j][r];
            const float d = v - mean;
            m2 = fmaf(d, d, m2);
            if (do_pool) {  // rows ascend with (i, r): strict compares keep the first
              const int row = rbase + i * 32 + ROFF(r);
              if (v > vx) { vx = v; ax = row; }
              if (v < vn) { vn = v; an = row; }
            }
          }
        const float so = __shfl_xor(sl, 32), m2o = __shfl_xor(m2, 32);
        const float dl = (so - sl) * (1.f / NL);
        const float S = sl + so, M2 = (m2 + m2o) + dl * dl * (0.5f * NL);
        if (do_pool) {
          const float ovx = __shfl_xor(vx, 32), ovn = __shfl_xor(vn, 32);
          const int oax = __shfl_xor(ax, 32), oan = __shfl_xor(an, 32);
          if (ovx > vx || (ovx == vx && oax < ax)) { vx = ovx; ax = oax; }
          if (ovn < vn || (ovn == vn && oan < an)) { vn = ovn; an = oan; }
        }
        if (lh == 0) {
          const size_t o = ((size_t)rec * p.N + cbase + j * 32) * 2;
          if (p.stats) { p.stats[o] = S; p.stats[o + 1] = M2; }
          if (do_pool) {
            p.pool_val[o] = vx; p.pool_val[o + 1] = vn;
            p.pool_arg[o] = ax; p.pool_arg[o + 1] = an;
          }
        }
      }
    }
  } else if constexpr (EPI == SPGAN_EPI_MASK_OUT) {
    const float* rb = p.ref + (size_t)rbase * p.ld_ref + cbase;
    float* yb = p.Y + (size_t)rbase * p.ldy + cbase;
    const unsigned ldr = (unsigned)p.ld_ref, ldy = (unsigned)p.ldy;
    const float sl = p.b_slope;
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = rb[(size_t)((unsigned)(i * 32 + ROFF(r)) * ldr + (unsigned)(j * 32))];
#pragma unroll
        for (int r = 0; r < 16; ++r) yb[(size_t)((unsigned)(i * 32 + ROFF(r)) * ldy + (unsigned)(j * 32))] = acc[i][j][r] * lrelu_mask(rv[r], sl);
      }
  } else {  // SPGAN_EPI_BNBWD (/ SPGAN_WIDE_EPI_BNBWD_GOUT)
    constexpr bool GOUT = EPI == SPGAN_WIDE_EPI_BNBWD_GOUT;
    static_assert(EPI == SPGAN_EPI_BNBWD || GOUT, "unknown epilogue");
    const float* rb = p.ref + (size_t)rbase * p.ld_ref + cbase;
    float* yb = p.Y + (size_t)rbase * p.ldy + cbase;
    const unsigned ldr = (unsigned)p.ld_ref, ldy = (unsigned)p.ldy;
    const float sl = p.b_slope;
    const float* ab = p.rowbias ? p.rowbias + (size_t)rbase * p.ld_rowbias + cbase : nullptr;  // host: rows_per_group == 1 (dense addend)
    const unsigned ld2 = (unsigned)p.ld_rowbias;
    const float* gx = GOUT ? p.gout_add + (size_t)rbase * p.ld_gout_add + cbase : nullptr;
    const unsigned ldx = GOUT ? (unsigned)p.ld_gout_add : 0u;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int col = cbase + j * 32;
      const float sc = p.b_scale[col], sh = p.b_shift[col], mu = p.b_mean[col], inv = p.b_invstd[col];
      const float bia = p.bias ? p.bias[col] : 0.f;
      const float gs = GOUT ? p.gout_scale[col] : 1.f;
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        float yv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) yv[r] = rb[(size_t)((unsigned)(i * 32 + ROFF(r)) * ldr + (unsigned)(j * 32))];
        if (ab) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] += ab[(size_t)((unsigned)(i * 32 + ROFF(r)) * ld2 + (unsigned)(j * 32))];
        }
        if constexpr (GOUT) {   // g into the accumulator, the statistics from it; then the stored value gout_add + gout_scale * g
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float y = yv[r];
            const float z = fmaf(y, sc, sh);
            const float g = (acc[i][j][r] + bia) * lrelu_mask(z, sl);
            acc[i][j][r] = g;
            s0 += g;
            s1 = fmaf(g, (y - mu) * inv, s1);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) yv[r] = gx[(size_t)((unsigned)(i * 32 + ROFF(r)) * ldx + (unsigned)(j * 32))];
#pragma unroll
          for (int r = 0; r < 16; ++r) yb[(size_t)((unsigned)(i * 32 + ROFF(r)) * ldy + (unsigned)(j * 32))] = fmaf(gs, acc[i][j][r], yv[r]);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float y = yv[r];
            const float z = fmaf(y, sc, sh);
            const float g = (acc[i][j][r] + bia) * lrelu_mask(z, sl);
            const float xh = (y - mu) * inv;
            yb[(size_t)((unsigned)(i * 32 + ROFF(r)) * ldy + (unsigned)(j * 32))] = g;
            s0 += g;
            s1 = fmaf(g, xh, s1);
          }
        }
      }
      if (p.stats) {
        s0 += __shfl_xor(s0, 32);
        s1 += __shfl_xor(s1, 32);
        if (lh == 0) {
          float* o = p.stats + ((size_t)rec * p.N + col) * 2;
          o[0] = s0; o[1] = s1;
        }
      }
    }
  }
}
