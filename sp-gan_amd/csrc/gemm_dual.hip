// spgan_gemm_dual: the backward of one 1x1-conv layer behind a train-mode BatchNorm + LeakyReLU as ONE launch -- the input-gradient
// product AND the weight-gradient product from ONE staging of the incoming gradient tile (round-3 review item 1).  The layers:
// conv_w.3 of an EdgeBlock (Generation/Generator.py:56-63,78), mlps.3 and mlps.6 of the Discriminator
// (Generation/Discriminator.py:55-65) and the collapsed form of its fc2.0 (Discriminator.py:77-81,104; DESIGN.md "collapsed backward").
//
//   dy[m, :]   = p*A[m, :] + q*A2[m, :] + r                 the layer's output gradient as the lazy BatchNorm-backward operand, or
//              = lrelu(p*A[m, :] + r, a_slope)               an activation formed on load (the collapsed layer: dy := a3), or A itself
//   a[m, :]    = lrelu(sc*pre[m, :] + sh)                    the layer's INPUT: the previous layer's BatchNorm + LeakyReLU applied on load;
//                                                            pre[m] = B[m] (plain) or B[idx[m]] - B[m / k] + e_bias (per-edge operand)
//   dW[Na,Nb]  = sum_m dy[m, :]^T a[m, :]                    weight gradient (split over row runs, partials summed in fixed order)
//   G[m, :]    = (dy[m, :] . W + bias + rowadd[m, :]) * lrelu'(sc*pre[m, :] + sh)      gradient w.r.t. the previous BatchNorm's output
//   stats      = (sum_m G, sum_m G * xhat),  xhat = (pre - mean)*invstd               the previous BatchNorm's backward sums
//   (optional: the STORED tile is gout_add[m, :] + gout_scale * G[m, :] -- phase B of the double backward hands X = xbarA + gamma*g to the next
//    layer's lazy BatchNorm-backward operand; the statistics stay those of G)
//   colsum     = sum_m dy[m, :]                              (optional by-product: the collapsed layer's colsum(a3))
//
// Before, gemm_tn (dW) and gemm_nt with the BNBWD / EDGE_BNBWD epilogue (G, stats) each read A, A2 and pre: at the EdgeBlock's size
// (655,360 edges x 128 channels) 2 x 840 MB; both launches ran at 32-39 % of the fp32 matrix peak, bound by that traffic.  Here a
// workgroup owns a contiguous run of 32-row chunks and ONE 64-column part of the Nb output columns; per chunk the dy tile [32 x Na] and
// the pre tile [32 x 64] are staged ONCE into LDS and feed both products:
//   input gradient  [32 x 64]  = dy [32 x Na] . W [Na x 64]        v_mfma_f32_16x16x4_f32, K = Na (the W part stays in LDS)
//   weight gradient [Na x 64] += dy^T [Na x 32] . a [32 x 64]      v_mfma_f32_32x32x2_f32, 2 tiles per wave, accumulated over all chunks
// with Na / 32 waves (Na = 128: 256 threads, two workgroups per CU; Na = 256: 512 threads, one per CU): 2048 + 2048 matrix-pipe cycles per
// wave and chunk in both geometries.  Nb = 128 / 256 run as 2 / 4 column parts (workgroups next to each other on one XCD: the dy tile of a
// run comes from HBM once and from that XCD's L2 for the other parts).  The next chunk's global loads are in flight under the MFMAs (raw
// values in registers; the affine maps are applied at the LDS store, csrc/gemm.hip "prologues run at the LDS store"), the per-edge
// neighbour indices are fetched one chunk further ahead.  The masked input-gradient tile goes through LDS (pre-loaded with the row addend
// when there is one) so that it leaves as 16-byte stores of full rows.
// Bank maths (ds_read_b32: 32-lane groups, 32 banks): dy pitch Na + 2 = 2 mod 32 (16 rows x 2 k-values of a 16x16x4 A fragment land on 32
// banks; a 32x32x2 A fragment is 32 consecutive floats of one row), W pitch 80 = 16 mod 32 (2 k-rows x 16 columns), pre / G pitch 68
// (4*68 = 16 mod 32: the epilogue's two rows 4 apart x 16 columns).
// Deterministic: fixed chunk -> workgroup assignment, fixed-order sums, no atomics.  fp32 operands only (the "f16" operand mode keeps the
// two-launch route with its 16-bit storage).
#include "common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int R = 32;      // rows per chunk
constexpr int NBW = 64;    // output columns per workgroup (one part of Nb)
constexpr int LDP = 68, LDW = 80, LDG = 68;

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// A-operand modes (spgan_gemm_dual_args.a_mode)
constexpr int A_DENSE = 0, A_LAZY2 = 1, A_ACT = 2;

// The body of one workgroup: `bid` of `nblk` workgroups of ONE problem (the stand-alone kernel passes blockIdx.x / gridDim.x; the grouped
// kernel the workgroup's position inside its problem's block range).
template <int EK, int NA>   // EK: 0 plain pre tensor, > 0 per-edge operand with EK edges per point (NA = 128 only); NA: columns of dy
__device__ __forceinline__ void gemm_dual_body(const spgan_gemm_dual_args& p, int chunks_per_wg, int runs, int parts, int a_mode, int bid, int nblk) {
  constexpr int T = 2 * NA;            // threads
  constexpr int W_ = NA / 32;          // waves
  constexpr int TPW = 8 / W_;          // 16 x 16 input-gradient tiles per wave (8 tiles: 2 row halves x 4 column sixteenths)
  constexpr int NE = 4 * TPW;          // input-gradient accumulator elements per lane
  constexpr int PS = 512 / T;          // float4 staging slots per thread of a 32 x 64 tile
  constexpr int LDY = NA + 2;
  constexpr int SM_DY = R * LDY, SM_PRE = R * LDP, SM_W = NA * LDW, SM_G = R * LDG;
  __shared__ __attribute__((aligned(16))) float sm[SM_DY + SM_PRE + SM_W + SM_G];
  float* dys = sm;
  float* pres = sm + SM_DY;
  float* ws_ = sm + SM_DY + SM_PRE;
  float* gzs = sm + SM_DY + SM_PRE + SM_W;      // the masked input-gradient tile [32 x 64] on its way to coalesced 16-byte stores

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;     // 16x16x4 fragment coordinates
  const int l31 = lane & 31, lh = lane >> 5;     // 32x32x2 fragment coordinates
  const int chunks = p.M / R;
  // XCD-aware: the hardware deals consecutive workgroup ids round-robin to the 8 XCDs; logical workgroup xcd*per + t goes to XCD xcd, so
  // that an XCD works on CONSECUTIVE row runs and all column parts of a run: the per-edge operand's neighbour rows of one shape (0.5 MB
  // of the point tensor) are gathered through ONE L2, and a run's dy tile is fetched from HBM once for all its parts
  const int per_xcd = nblk >> 3;
  const int L = (bid & 7) * per_xcd + (bid >> 3);
  if (L >= runs * parts) return;
  const int wg = L / parts, part = L - wg * parts;
  const int c0 = wg * chunks_per_wg;
  const int c1 = min(chunks, c0 + chunks_per_wg);
  const int col0 = part * NBW;                   // this workgroup's columns of pre / W / G
  const bool do_cs = p.colsum_ws != nullptr && part == 0;

  // ---- the W part [NA, 64] -> LDS once (row = column of dy, pitch LDW)
#pragma unroll
  for (int i = 0; i < NA * NBW / 4 / T; ++i) {
    const int s = tid + T * i, row = s >> 4, c4 = (s & 15) * 4;
    *reinterpret_cast<float4*>(&ws_[row * LDW + c4]) = ldg4(p.W + (size_t)row * p.ldw + col0 + c4);
  }

  // ---- per-thread constants
  // staging slots: dy tile 32 x NA -> 4 float4 per thread and tensor (column quad fixed per thread); 32 x 64 tiles -> PS per thread
  const int ycol = (tid % (NA / 4)) * 4, yrow0 = tid / (NA / 4);      // rows yrow0 + 8 i
  const int pcol = (tid & 15) * 4, prow0 = tid >> 4;                    // rows prow0 + (T / 16) i
  float4 cp = make_float4(1.f, 1.f, 1.f, 1.f), cq = make_float4(0.f, 0.f, 0.f, 0.f), cr = cq, eb = cq;
  if (a_mode != A_DENSE) {
    cp = ldg4(p.p + ycol); cr = ldg4(p.r + ycol);
    if (a_mode == A_LAZY2) cq = ldg4(p.q + ycol);
  }
  const float a_slope = p.a_slope;
  if (EK > 0) eb = ldg4(p.e_bias + pcol);
  // weight-gradient B fragments: columns 32 cj + l31 of the pre tile; input-gradient tiles: row half ti, column sixteenths tn0 .. tn0+TPW-1
  const int ti = w / (W_ / 2), tn0 = (w % (W_ / 2)) * TPW;
  float sc_w[2], sh_w[2], sc_d[TPW], sh_d[TPW], mu_d[TPW], iv_d[TPW], bi_d[TPW];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    sc_w[j] = p.b_scale[col0 + 32 * j + l31]; sh_w[j] = p.b_shift[col0 + 32 * j + l31];
  }
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int c = col0 + 16 * (tn0 + j) + l15;
    sc_d[j] = p.b_scale[c]; sh_d[j] = p.b_shift[c]; mu_d[j] = p.b_mean[c]; iv_d[j] = p.b_invstd[c];
    bi_d[j] = p.bias ? p.bias[c] : 0.f;
  }
  const float slope = p.slope;
  const bool radd = p.rowadd != nullptr;

  f32x16 accw[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[j][r] = 0.f;
  float s0[TPW], s1[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) s0[j] = s1[j] = 0.f;
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);      // column sums of dy over this thread's rows (ascending: deterministic)

  float4 ra[4], ra2[4], rb[PS], rb2[PS], re[PS];
  float4 rx[PS];     // post-mask addend of the gradient tile (gout_add) of the chunk that is stored NEXT
  const bool gadd = p.gout_add != nullptr;
  float4 gsc = make_float4(1.f, 1.f, 1.f, 1.f);
  if (gadd) gsc = ldg4(p.gout_scale + col0 + pcol);
  int nidx[PS];      // neighbour rows of the chunk AFTER the one whose values are being loaded
#pragma unroll
  for (int i = 0; i < PS; ++i) nidx[i] = 0;

  // global addressing: a uniform base per chunk (scalar registers) + a 32-bit per-thread element offset (the rows of a chunk are less
  // than 2^31 elements apart)
  const unsigned offA = (unsigned)yrow0 * (unsigned)p.lda + (unsigned)ycol, stepA = 8u * (unsigned)p.lda;
  const unsigned offA2 = (unsigned)yrow0 * (unsigned)p.lda2 + (unsigned)ycol, stepA2 = 8u * (unsigned)p.lda2;
  const unsigned offB = (unsigned)prow0 * (unsigned)p.ldb + (unsigned)(col0 + pcol), stepB = (unsigned)(T / 16) * (unsigned)p.ldb;
  const unsigned offE = (unsigned)prow0 * (unsigned)p.ld_rowadd + (unsigned)(col0 + pcol), stepE = (unsigned)(T / 16) * (unsigned)p.ld_rowadd;
  const unsigned offG = (unsigned)prow0 * (unsigned)p.ldg + (unsigned)(col0 + pcol), stepG = (unsigned)(T / 16) * (unsigned)p.ldg;
  const unsigned offX = (unsigned)prow0 * (unsigned)p.ld_gout_add + (unsigned)(col0 + pcol), stepX = (unsigned)(T / 16) * (unsigned)p.ld_gout_add;
  auto xload = [&](int c) {  // the stored tile's addend of chunk c: in flight under chunk c's MFMA phases
    if (gadd) {
      const float* Xb = p.gout_add + (size_t)c * R * p.ld_gout_add;
#pragma unroll
      for (int i = 0; i < PS; ++i) rx[i] = ldg4(Xb + (offX + i * stepX));
    }
  };
  auto iload = [&](int c) {  // neighbour indices of chunk c (clamped: a chunk past the end is never stored)
    if (EK > 0) {
      const int m0 = min(c, chunks - 1) * R;
#pragma unroll
      for (int i = 0; i < PS; ++i) nidx[i] = p.e_idx[m0 + prow0 + (T / 16) * i];
    }
  };
  auto gload = [&](int c) {
    const float* Ab = p.A + (size_t)c * R * p.lda;
    const float* A2b = p.A2 + (size_t)c * R * p.lda2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = ldg4(Ab + (offA + i * stepA));
      if (a_mode == A_LAZY2) ra2[i] = ldg4(A2b + (offA2 + i * stepA2));
    }
    if (EK > 0) {
      const unsigned m0 = (unsigned)c * R + (unsigned)prow0;
#pragma unroll
      for (int i = 0; i < PS; ++i) {
        rb[i] = ldg4(p.B + ((unsigned)nidx[i] * (unsigned)p.ldb + (unsigned)pcol));
        rb2[i] = ldg4(p.B + (((m0 + (unsigned)(T / 16) * i) / (unsigned)(EK > 0 ? EK : 1)) * (unsigned)p.ldb + (unsigned)pcol));
      }
    } else {
      const float* Bb = p.B + (size_t)c * R * p.ldb;
#pragma unroll
      for (int i = 0; i < PS; ++i) rb[i] = ldg4(Bb + (offB + i * stepB));
    }
    if (radd) {
      const float* Eb = p.rowadd + (size_t)c * R * p.ld_rowadd;
#pragma unroll
      for (int i = 0; i < PS; ++i) re[i] = ldg4(Eb + (offE + i * stepE));
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = ra[i];
      if (a_mode == A_LAZY2) {
        v.x = fmaf(v.x, cp.x, fmaf(ra2[i].x, cq.x, cr.x));
        v.y = fmaf(v.y, cp.y, fmaf(ra2[i].y, cq.y, cr.y));
        v.z = fmaf(v.z, cp.z, fmaf(ra2[i].z, cq.z, cr.z));
        v.w = fmaf(v.w, cp.w, fmaf(ra2[i].w, cq.w, cr.w));
      } else if (a_mode == A_ACT) {
        v.x = lrelu_f(fmaf(v.x, cp.x, cr.x), a_slope);
        v.y = lrelu_f(fmaf(v.y, cp.y, cr.y), a_slope);
        v.z = lrelu_f(fmaf(v.z, cp.z, cr.z), a_slope);
        v.w = lrelu_f(fmaf(v.w, cp.w, cr.w), a_slope);
      }
      if (do_cs) { cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w; }
      float* d = &dys[(yrow0 + 8 * i) * LDY + ycol];      // pitch NA + 2 floats: 8-byte aligned rows
      *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
      *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
    }
#pragma unroll
    for (int i = 0; i < PS; ++i) {
      float4 v = rb[i];
      if (EK > 0) {
        v.x = (v.x - rb2[i].x) + eb.x;
        v.y = (v.y - rb2[i].y) + eb.y;
        v.z = (v.z - rb2[i].z) + eb.z;
        v.w = (v.w - rb2[i].w) + eb.w;
      }
      *reinterpret_cast<float4*>(&pres[(prow0 + (T / 16) * i) * LDP + pcol]) = v;
      // the row addend of this chunk waits in the gradient tile's LDS slot: the epilogue adds it where it writes its own value
      if (radd) *reinterpret_cast<float4*>(&gzs[(prow0 + (T / 16) * i) * LDG + pcol]) = re[i];
    }
  };

  // Order of the memory operations of a chunk (vmcnt counts loads and stores in issue order, and a register a store still has to read
  // cannot be reloaded): chunk c's gradient tile is stored first (its LDS slots then take chunk c+1's row addend), chunk c+1's registers
  // go to LDS, and the loads of chunk c+2 are issued right behind -- they fly for a whole chunk.
  if (c0 < c1) {
    iload(c0);
    gload(c0);
    iload(c0 + 1);
    sstore();
    xload(c0);
    if (c0 + 1 < c1) {
      gload(c0 + 1);
      iload(c0 + 2);
    }
    __syncthreads();
    for (int c = c0; c < c1; ++c) {
      const bool more = c + 1 < c1;
      // ---- input gradient: [32 x 64] = dy [32 x NA] . W [NA x 64]; this wave: rows 16 ti .. +16, column sixteenths tn0 .. tn0+TPW-1
      f32x4 accd[TPW];
#pragma unroll
      for (int j = 0; j < TPW; ++j) accd[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      {
        const float* ap = dys + (16 * ti + l15) * LDY + lq;
        const float* bp = ws_ + lq * LDW + 16 * tn0 + l15;
#pragma unroll 8
        for (int kq = 0; kq < NA / 4; ++kq) {
          const float af = ap[4 * kq];
#pragma unroll
          for (int j = 0; j < TPW; ++j) accd[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bp[4 * kq * LDW + 16 * j], accd[j], 0, 0, 0);
        }
      }
      // ---- weight gradient: [NA x 64] += dy^T [NA x 32] . a [32 x 64]; this wave: dy columns 32 w .. +32, both column halves of a.
      // The epilogue of the input gradient (bias / row addend, LeakyReLU mask of the previous layer, BatchNorm-backward sums, tile -> LDS)
      // is interleaved, one accumulator element behind every (16 / NE)-th k-step: its VALU work issues while the matrix pipe works off
      // the MFMAs before it.
      {
        const float* ap = dys + lh * LDY + 32 * w + l31;
        const float* bp = pres + lh * LDP + l31;
        const float* ep = pres + (16 * ti + 4 * lq) * LDP + 16 * tn0 + l15;
        float* gp = gzs + (16 * ti + 4 * lq) * LDG + 16 * tn0 + l15;
#pragma unroll
        for (int kq = 0; kq < R / 2; ++kq) {
          const float af = ap[2 * kq * LDY];
          float b0 = fmaf(bp[2 * kq * LDP], sc_w[0], sh_w[0]);
          float b1 = fmaf(bp[2 * kq * LDP + 32], sc_w[1], sh_w[1]);
          b0 = b0 > 0.f ? b0 : b0 * slope;
          b1 = b1 > 0.f ? b1 : b1 * slope;
          accw[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, b0, accw[0], 0, 0, 0);
          accw[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, b1, accw[1], 0, 0, 0);
          if ((kq + 1) % (16 / NE) == 0) {
            const int e = kq / (16 / NE), j = e >> 2, r = e & 3;      // compile-time after unrolling
            const float pre = ep[r * LDP + 16 * j];
            const float z = fmaf(pre, sc_d[j], sh_d[j]);
            float acc = accd[j][r] + bi_d[j];
            if (radd) acc += gp[r * LDG + 16 * j];
            const float g = acc * (z > 0.f ? 1.f : slope);
            const float xh = (pre - mu_d[j]) * iv_d[j];
            s0[j] += g;
            s1[j] = fmaf(g, xh, s1[j]);
            gp[r * LDG + 16 * j] = g;
          }
        }
      }
      __syncthreads();      // every read of this chunk's tiles is done; the masked input-gradient tile is complete in LDS
      {
        float* gout = p.G + (size_t)c * R * p.ldg;
#pragma unroll
        for (int i = 0; i < PS; ++i) {
          float4 gq = *reinterpret_cast<const float4*>(&gzs[(prow0 + (T / 16) * i) * LDG + pcol]);
          if (gadd) {   // G = gout_add + gout_scale * g (the statistics above are those of g)
            gq.x = fmaf(gsc.x, gq.x, rx[i].x); gq.y = fmaf(gsc.y, gq.y, rx[i].y); gq.z = fmaf(gsc.z, gq.z, rx[i].z); gq.w = fmaf(gsc.w, gq.w, rx[i].w);
          }
          *reinterpret_cast<float4*>(gout + (offG + i * stepG)) = gq;
        }
      }
      if (more) {
        xload(c + 1);
        sstore();                    // chunk c+1: registers -> LDS (the same thread re-fills the gradient-tile slots it just stored from)
        if (c + 2 < c1) {
          gload(c + 2);              // uses the indices fetched one iteration ago
          iload(c + 3);
        }
      }
      __syncthreads();
    }
  }

  // ---- statistics: lanes with the same column (lq = 0..3), then the two waves that share the columns (ti = 0, 1), fixed order
  float* red = dys;         // the tiles are dead (the loop ended with a barrier)
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    float a = s0[j], b = s1[j];
    a += __shfl_xor(a, 16); b += __shfl_xor(b, 16);
    a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
    if (lq == 0) {
      red[(w * 32 + 16 * j + l15) * 2] = a;
      red[(w * 32 + 16 * j + l15) * 2 + 1] = b;
    }
  }
  __syncthreads();
  if (tid < NBW) {
    const int t16 = tid >> 4, cc = tid & 15;          // column sixteenth t16: waves t16 / TPW (ti = 0) and that + W_/2 (ti = 1), slot t16 % TPW
    const int wa = t16 / TPW, sl = t16 % TPW;
    const float a = red[(wa * 32 + 16 * sl + cc) * 2] + red[((wa + W_ / 2) * 32 + 16 * sl + cc) * 2];
    const float b = red[(wa * 32 + 16 * sl + cc) * 2 + 1] + red[((wa + W_ / 2) * 32 + 16 * sl + cc) * 2 + 1];
    float* st = p.stats + ((size_t)wg * p.Nb + col0 + tid) * 2;
    st[0] = a; st[1] = b;
  }
  // ---- column sums of dy over this run (8 row groups per column quad -> one value per column), part 0 only
  if (do_cs) {
    float* cred = ws_;       // [8][NA]: the W part is dead (every MFMA loop ended before the last barrier of the chunk loop)
    *reinterpret_cast<float4*>(&cred[yrow0 * NA + ycol]) = cs;
    __syncthreads();
    if (tid < NA) {
      float t = 0.f;
#pragma unroll
      for (int g8 = 0; g8 < 8; ++g8) t += cred[g8 * NA + tid];
      p.colsum_ws[(size_t)wg * NA + tid] = t;
    }
  }
  // ---- weight-gradient partial of this workgroup -> ws[run][NA][Nb], columns col0 .. col0+63
  float* out = p.ws + (size_t)wg * NA * p.Nb + (size_t)(32 * w + 4 * lh) * p.Nb + col0 + l31;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(size_t)((r & 3) + 8 * (r >> 2)) * p.Nb + 32 * j] = accw[j][r];
}

template <int EK, int NA>
__global__ __launch_bounds__(2 * NA, NA == 128 ? 2 : 1) void gemm_dual_kernel(const spgan_gemm_dual_args p, int chunks_per_wg, int runs, int parts,
                                                                               int a_mode) {
  gemm_dual_body<EK, NA>(p, chunks_per_wg, runs, parts, a_mode, blockIdx.x, gridDim.x);
}

// Grouped launch (spgan_gemm_dual_multi): `count` independent problems of ONE geometry (M, Na, Nb equal; operands, modes, addends and outputs
// per problem) as one grid -- problem g owns the workgroups [g*grid1, (g+1)*grid1) and runs exactly the stand-alone kernel's body there
// (grid1 is a multiple of 8: a workgroup's XCD is the same as in the stand-alone launch).  The D step's real / fake backward passes and
// phase B of the penalty's double backward issue the same layer's launch three times with different operands (Discriminator.py:97-115,
// gradient_penalty.py:19-37): one launch, the tail of one problem under the head of the next, results bit-identical to the separate ones.
struct DualMulti {
  spgan_gemm_dual_args a[SPGAN_GROUP_MAX];
};

template <int NA>
__global__ __launch_bounds__(2 * NA, NA == 128 ? 2 : 1) void gemm_dual_multi_kernel(const DualMulti m, int chunks_per_wg, int runs, int parts, int grid1) {
  const int g = blockIdx.x / grid1;
  const spgan_gemm_dual_args& p = m.a[g];
  gemm_dual_body<0, NA>(p, chunks_per_wg, runs, parts, p.a_mode, blockIdx.x - g * grid1, grid1);
}

// row run -> workgroup plan.  Na = 128: two workgroups per CU resident (LDS) -> 512 slots; Na = 256: one -> 256 slots; the slots are shared
// by the column parts of a run; at least 4 chunks per run to amortise the W load and the partial store
inline void dual_plan(int M, int Na, int Nb, int* runs, int* cpw) {
  const int chunks = M / R, parts = Nb / NBW;
  const int slots = (Na == 128 ? 512 : 256) / parts;
  int per = (chunks + slots - 1) / slots;
  if (per < 4) per = 4;
  *cpw = per;
  *runs = (chunks + per - 1) / per;
}

inline bool dual_shape_ok(int M, int Na, int Nb, int e_k) {
  if (M < 8192 || M % R) return false;
  if (Na == 128 && Nb == 64) return e_k == 0 || e_k == 10;
  if (Na == 256 && (Nb == 128 || Nb == 256)) return e_k == 0;
  return false;
}

}  // namespace

extern "C" int spgan_gemm_dual_wgs(int M, int Na, int Nb, int e_k) {
  if (!dual_shape_ok(M, Na, Nb, e_k)) return 0;
  int runs, cpw;
  dual_plan(M, Na, Nb, &runs, &cpw);
  return runs;
}

extern "C" int spgan_gemm_dual_rows_per_wg(int M, int Na, int Nb) {
  if (M < R || !(Na == 128 || Na == 256) || Nb < NBW || Nb % NBW) return 0;
  int runs, cpw;
  dual_plan(M, Na, Nb, &runs, &cpw);
  return cpw * R;
}

static int dual_check(const spgan_gemm_dual_args* a) {
  SPGAN_CHECK_ARG(a && a->A && a->W && a->B && a->G && a->stats && a->ws && a->b_scale && a->b_shift && a->b_mean && a->b_invstd);
  const int ek = a->e_idx ? a->e_k : 0;
  SPGAN_CHECK_ARG(dual_shape_ok(a->M, a->Na, a->Nb, ek));
  SPGAN_CHECK_ARG(a->a_mode == A_DENSE || a->a_mode == A_LAZY2 || a->a_mode == A_ACT);
  SPGAN_CHECK_ARG(a->a_mode == A_DENSE || (a->p && a->r));
  SPGAN_CHECK_ARG(a->a_mode != A_LAZY2 || (a->A2 && a->q));
  SPGAN_CHECK_ARG(!a->e_idx || a->e_bias);
  // 16-byte aligned rows everywhere (float4 loads / stores)
  auto al = [](const void* q, int ld) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0 && (ld & 3) == 0; };
  SPGAN_CHECK_ARG(al(a->A, a->lda) && (a->a_mode != A_LAZY2 || al(a->A2, a->lda2)) && al(a->W, a->ldw) && al(a->B, a->ldb) && al(a->G, a->ldg));
  SPGAN_CHECK_ARG(a->lda >= a->Na && (a->a_mode != A_LAZY2 || a->lda2 >= a->Na) && a->ldw >= a->Nb && a->ldg >= a->Nb && (ek > 0 ? a->ldb >= NBW : a->ldb >= a->Nb));
  SPGAN_CHECK_ARG(a->a_mode == A_DENSE || (al(a->p, 4) && al(a->r, 4) && (a->a_mode != A_LAZY2 || al(a->q, 4))));
  SPGAN_CHECK_ARG(!a->e_idx || al(a->e_bias, 4));
  SPGAN_CHECK_ARG(!a->rowadd || (al(a->rowadd, a->ld_rowadd) && a->ld_rowadd >= a->Nb));
  SPGAN_CHECK_ARG(!a->gout_add || (a->gout_scale && al(a->gout_add, a->ld_gout_add) && a->ld_gout_add >= a->Nb && al(a->gout_scale, 4)));
  return SPGAN_OK;
}

extern "C" int spgan_gemm_dual(const spgan_gemm_dual_args* a, spgan_stream_t s_) {
  const int rc = dual_check(a);
  if (rc != SPGAN_OK) return rc;
  const int ek = a->e_idx ? a->e_k : 0;
  int runs, cpw;
  dual_plan(a->M, a->Na, a->Nb, &runs, &cpw);
  const int parts = a->Nb / NBW;
  const int grid = ((runs * parts + 7) / 8) * 8;
  hipStream_t s = (hipStream_t)s_;
  if (a->Na == 128) {
    if (ek == 0) hipLaunchKernelGGL((gemm_dual_kernel<0, 128>), dim3(grid), dim3(256), 0, s, *a, cpw, runs, parts, a->a_mode);
    else hipLaunchKernelGGL((gemm_dual_kernel<10, 128>), dim3(grid), dim3(256), 0, s, *a, cpw, runs, parts, a->a_mode);
  } else {
    hipLaunchKernelGGL((gemm_dual_kernel<0, 256>), dim3(grid), dim3(512), 0, s, *a, cpw, runs, parts, a->a_mode);
  }
  return spgan_launch_status();
}

extern "C" int spgan_gemm_dual_multi(const spgan_gemm_dual_args* a, int count, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(a && count >= 1 && count <= SPGAN_GROUP_MAX);
  if (count == 1) return spgan_gemm_dual(a, s_);
  DualMulti m;
  for (int g = 0; g < count; ++g) {
    const int rc = dual_check(a + g);
    if (rc != SPGAN_OK) return rc;
    SPGAN_CHECK_ARG(!a[g].e_idx && a[g].M == a[0].M && a[g].Na == a[0].Na && a[g].Nb == a[0].Nb);   // one geometry, plain pre tensors
    m.a[g] = a[g];
  }
  int runs, cpw;
  dual_plan(a->M, a->Na, a->Nb, &runs, &cpw);
  const int parts = a->Nb / NBW;
  const int grid1 = ((runs * parts + 7) / 8) * 8;
  hipStream_t s = (hipStream_t)s_;
  if (a->Na == 128) hipLaunchKernelGGL((gemm_dual_multi_kernel<128>), dim3(grid1 * count), dim3(256), 0, s, m, cpw, runs, parts, grid1);
  else hipLaunchKernelGGL((gemm_dual_multi_kernel<256>), dim3(grid1 * count), dim3(512), 0, s, m, cpw, runs, parts, grid1);
  return spgan_launch_status();
}
