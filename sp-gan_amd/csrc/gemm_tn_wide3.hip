// gemm_tn with split-bf16 operands (spgan_gemm_tn_args.mfma_lp == 2, "bf16x3"): the fp32-equivalent weight gradient C[Na,Nb] = sum_m A[m,:]^T B[m,:]
// on the bf16 matrix pipe, in the row-pipelined form of gemm_wide3.hip.
//
// The MFMA wants, per lane, 8 consecutive k (= m-rows) of one operand column, while the operands lie [m][column] in memory.  A thread therefore
// stages "quads": 4 consecutive m-rows of ONE column (4 dword loads, the 64 lanes of a wave covering 16 consecutive columns x 4 row-quads: 64-byte
// segments of 4 rows per load instruction), applies the operand's prologue, splits the 4 values into three bf16 planes and writes 8 bytes per
// plane into the LDS tile [plane][column][16 m] -- the layout (and swizzle) of gemm_wide3.hip's tiles with "row" = operand column, so that the
// fragment reads, the six cross products and the sign conventions are the same.  A workgroup owns XM x XN of the output (WGM x WGN waves of 128 x 64;
// every staged value is split once per 128..256 columns of the other operand) over one split of the rows; k-tiles of 16 rows, double-buffered; the
// staging of tile kt+1 and the loads of tile kt+2 ride in the MFMA gaps of tile kt.  Partials go to ws[split][Na][Nb] like gemm_tn_kernel's; the
// bf16 MFMA's accumulation bias (toward -infinity) alternates in sign from split to split (odd splits multiply A by -1 and negate their partial),
// so it cancels in the split sum.
//
// Operand modes: B plain / BatchNorm + LeakyReLU per column; A plain / affine (+ LeakyReLU); the fp32 column sums of the transformed A as a
// by-product.  Everything else (per-edge B, the two-tensor lazy A operand -- its second quad per slot on top of 128 accumulators spilled and
// measured no faster than the fp32 kernel --, 16-bit storage, unaligned shapes) stays with gemm.hip.
#include <type_traits>
#include "gemm_tn_wide3.hpp"
#include "split_bf16.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int XK = 16;    // k-tile: m-rows
constexpr int PLW = 8;    // 4-byte words per column and plane (16 bf16)
constexpr int TI = 4, TJ = 2;

template <int WGM, int WGN>
struct T3 {
  static constexpr int XM = WGM * 128, XN = WGN * 64, THREADS = WGM * WGN * 64;
  static constexpr int CPP = THREADS / 4;                       // columns staged by one pass of the workgroup (4 row-quads per column)
  static constexpr int QA = XM / CPP, QB = XN / CPP;            // quads per thread, operand and k-tile
  static constexpr int PLANE_A = XM * PLW, PLANE_B = XN * PLW;  // words per plane
  static constexpr int BUF = 3 * (PLANE_A + PLANE_B);
  static constexpr size_t LDS = (size_t)2 * BUF * sizeof(uint32_t);
};

// APRO: 0 plain A, 1 a = A*a_scale + a_shift (a_lrelu: LeakyReLU behind it)
template <int BMODE, int APRO, int WGM, int WGN>
__global__ __launch_bounds__((T3<WGM, WGN>::THREADS), 2) void gemm_tn_wide3_kernel(const spgan_gemm_tn_args p_, int rows_per_split) {
  using X = T3<WGM, WGN>;
  const spgan_gemm_tn_args& p = p_;
  constexpr int XM = X::XM, XN = X::XN, QA = X::QA, QB = X::QB, CPP = X::CPP;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem_t3[];

  const int tilesB = p.Nb / XN;
  const int ta = blockIdx.x / tilesB, tb = blockIdx.x % tilesB;
  const int a0 = ta * XM, b0 = tb * XN;
  const int split = blockIdx.y;
  const int mbeg = split * rows_per_split, mend = min(p.M, mbeg + rows_per_split);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WGN, wn = wave % WGN;
  const int l31 = lane & 31, lh = lane >> 5;

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging: thread = (column cl + CPP*s, row-quad mq) of every pass s
  const int cl = tid >> 2, mq = tid & 3;
  float4 ra[QA], rb[QB];     // [slot] = the 4 m-rows of the quad
  float asc[QA], ash[QA], bsc[QB], bsh[QB], csum[QA];
  const float ssgn = (split & 1) ? -1.f : 1.f;
#pragma unroll
  for (int s = 0; s < QA; ++s) {
    const int c = a0 + cl + CPP * s;
    asc[s] = APRO ? p.a_scale[c] : 1.f;
    ash[s] = APRO ? p.a_shift[c] : 0.f;
    csum[s] = 0.f;
  }
#pragma unroll
  for (int s = 0; s < QB; ++s) {
    const int c = b0 + cl + CPP * s;
    bsc[s] = BMODE != SPGAN_A_PLAIN ? p.p_scale[c] : 1.f;
    bsh[s] = BMODE != SPGAN_A_PLAIN ? p.p_shift[c] : 0.f;
  }
  const float asl = (APRO == 1 && p.a_lrelu) ? p.a_slope : 1.f;     // host: 0 <= slope <= 1: lrelu(v) = max(v, v*slope); slope 1 = no activation
  const float bsl = BMODE != SPGAN_A_PLAIN ? p.p_slope : 1.f;
  // element offsets of row-quad mq's four rows at column cl, relative to tile row 0 (32-bit: host checks M*ld < 2^32)
  const unsigned oa = (unsigned)(mbeg + 4 * mq) * (unsigned)p.lda + (unsigned)(a0 + cl);
  const unsigned ob = (unsigned)(mbeg + 4 * mq) * (unsigned)p.ldb + (unsigned)(b0 + cl);
  const unsigned lda = (unsigned)p.lda, ldb = (unsigned)p.ldb;

  auto load_a = [&](int s, int kt) {
    const unsigned o = oa + (unsigned)(kt * XK) * lda + (unsigned)(CPP * s);
    ra[s] = make_float4(p.A[o], p.A[o + lda], p.A[o + 2 * lda], p.A[o + 3 * lda]);
  };
  auto load_b = [&](int s, int kt) {
    const unsigned o = ob + (unsigned)(kt * XK) * ldb + (unsigned)(CPP * s);
    rb[s] = make_float4(p.B[o], p.B[o + ldb], p.B[o + 2 * ldb], p.B[o + 3 * ldb]);
  };
  // LDS word offset of this thread's 8-byte slot inside a plane: column cl (+ CPP*s: a multiple of 32, bit 3 unchanged), row-quad mq
  const int st_off = cl * PLW + 4 * ((mq >> 1) ^ ((cl >> 3) & 1)) + (mq & 1) * 2;
  auto pin4 = [](float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); };
  auto piece_a = [&](int s, int buf) {
    float4 v = ra[s];
    if (APRO == 1) {
      v.x = fmaf(v.x, asc[s], ash[s]); v.y = fmaf(v.y, asc[s], ash[s]); v.z = fmaf(v.z, asc[s], ash[s]); v.w = fmaf(v.w, asc[s], ash[s]);
      v.x = fmaxf(v.x, v.x * asl); v.y = fmaxf(v.y, v.y * asl); v.z = fmaxf(v.z, v.z * asl); v.w = fmaxf(v.w, v.w * asl);
    }
    csum[s] += (v.x + v.y) + (v.z + v.w);
    v.x *= ssgn; v.y *= ssgn; v.z *= ssgn; v.w *= ssgn;
    st_split4(smem_t3 + buf * X::BUF + st_off + CPP * s * PLW, X::PLANE_A, v);
  };
  auto piece_b = [&](int s, int buf) {
    float4 v = rb[s];
    if (BMODE != SPGAN_A_PLAIN) {
      v.x = fmaf(v.x, bsc[s], bsh[s]); v.y = fmaf(v.y, bsc[s], bsh[s]); v.z = fmaf(v.z, bsc[s], bsh[s]); v.w = fmaf(v.w, bsc[s], bsh[s]);
      v.x = fmaxf(v.x, v.x * bsl); v.y = fmaxf(v.y, v.y * bsl); v.z = fmaxf(v.z, v.z * bsl); v.w = fmaxf(v.w, v.w * bsl);
    }
    st_split4(smem_t3 + buf * X::BUF + 3 * X::PLANE_A + st_off + CPP * s * PLW, X::PLANE_B, v);
  };

  // fragments: as gemm_wide3.hip with "row" = operand column
  const int fa_off = (wm * 128 + l31) * PLW + 4 * (lh ^ ((l31 >> 3) & 1));
  const int fb_off = 3 * X::PLANE_A + (wn * 64 + l31) * PLW + 4 * (lh ^ ((l31 >> 3) & 1));
  bf16x8 aq[2][3];
  constexpr bool XPF = X::THREADS == 512;   // one workgroup per CU: the next k-tile's first fragments are read across the barrier, under tile row 3
  bf16x8 bq[XPF ? 2 : 1][3][TJ];
  auto read_a = [&](bf16x8 (&dst)[3], int buf, int i) {
#pragma unroll
    for (int q = 0; q < 3; ++q) dst[q] = *reinterpret_cast<const bf16x8*>(smem_t3 + buf * X::BUF + fa_off + q * X::PLANE_A + i * 32 * PLW);
  };
  auto read_b = [&](bf16x8 (&dst)[3][TJ], int buf) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int j = 0; j < TJ; ++j) dst[q][j] = *reinterpret_cast<const bf16x8*>(smem_t3 + buf * X::BUF + fb_off + q * X::PLANE_B + j * 32 * PLW);
  };
  auto mfma_row = [&](int i, const bf16x8 (&a)[3], const bf16x8 (&b)[3][TJ]) {
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0][j], acc[i][j], 0, 0, 0);  // lo * hi
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2][j], acc[i][j], 0, 0, 0);  // hi * lo
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1][j], acc[i][j], 0, 0, 0);  // mid * mid
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0][j], acc[i][j], 0, 0, 0);  // mid * hi
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1][j], acc[i][j], 0, 0, 0);  // hi * mid
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0][j], acc[i][j], 0, 0, 0);  // hi * hi
  };
  auto row_schedule = [&](int quads) {   // 12 MFMAs: every gap carries the row's share of VALU, one LDS and one VMEM instruction
#pragma unroll
    for (int g = 0; g < 2 * 3 * TJ; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (quads >= 4) __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
      else if (quads >= 2) __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
      else if (quads == 1) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
  };

  auto ktile = [&](int kt, auto st, auto ld, auto cur) {
    constexpr bool ST = decltype(st)::value, LD = decltype(ld)::value;
    constexpr int C = decltype(cur)::value;
    const int buf = kt & 1, nb = buf ^ 1;
    constexpr int AH = QA / 2;
    // row 0
#pragma unroll
    for (int s = 0; s < AH; ++s) if (ST) pin4(ra[s]);
    read_a(aq[1], buf, 1);
    mfma_row(0, aq[0], bq[C]);
#pragma unroll
    for (int s = 0; s < AH; ++s) {
      if (ST) piece_a(s, nb);
      if (LD) load_a(s, kt + 2);
    }
    row_schedule(ST ? AH : 0);
    __builtin_amdgcn_sched_barrier(0);
    // row 1
#pragma unroll
    for (int s = AH; s < QA; ++s) if (ST) pin4(ra[s]);
    read_a(aq[0], buf, 2);
    mfma_row(1, aq[1], bq[C]);
#pragma unroll
    for (int s = AH; s < QA; ++s) {
      if (ST) piece_a(s, nb);
      if (LD) load_a(s, kt + 2);
    }
    row_schedule(ST ? AH : 0);
    __builtin_amdgcn_sched_barrier(0);
    // row 2
#pragma unroll
    for (int s = 0; s < QB; ++s) if (ST) pin4(rb[s]);
    read_a(aq[1], buf, 3);
    mfma_row(2, aq[0], bq[C]);
#pragma unroll
    for (int s = 0; s < QB; ++s) {
      if (ST) piece_b(s, nb);
      if (LD) load_b(s, kt + 2);
    }
    row_schedule(ST ? QB : 0);
    __builtin_amdgcn_sched_barrier(0);
    if (XPF) {
      __syncthreads();
      if (ST) {
        read_a(aq[0], nb, 0);
        read_b(bq[C ^ 1], nb);
      }
      mfma_row(3, aq[1], bq[C]);
      row_schedule(0);
    } else {
      mfma_row(3, aq[1], bq[C]);
      __syncthreads();
      if (ST) {
        read_a(aq[0], nb, 0);
        read_b(bq[C], nb);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, XPF ? 1 : 0>;
  const int nk = (mend - mbeg) / XK;   // even, >= 2 (host: M and the rows per split are multiples of 32)
  if (nk >= 2) {
#pragma unroll
    for (int s = 0; s < QA; ++s) load_a(s, 0);
#pragma unroll
    for (int s = 0; s < QB; ++s) load_b(s, 0);
#pragma unroll
    for (int s = 0; s < QA; ++s) {
      piece_a(s, 0);
      load_a(s, 1);
    }
#pragma unroll
    for (int s = 0; s < QB; ++s) {
      piece_b(s, 0);
      load_b(s, 1);
    }
    __syncthreads();
    read_a(aq[0], 0, 0);
    read_b(bq[0], 0);
    for (int kt = 0; kt < nk - 2; kt += 2) {
      ktile(kt, T_{}, T_{}, C0{});
      ktile(kt + 1, T_{}, T_{}, C1{});
    }
    ktile(nk - 2, T_{}, F_{}, C0{});
    ktile(nk - 1, F_{}, F_{}, C1{});
  }

  // ---- epilogue: this split's partial tile (C/D layout: col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5)); odd splits undo their sign
  if (p.a_colsum_ws != nullptr && tb == 0) {
    // fp32 column sums of the transformed A operand over this split's rows: the four row-quads of a column sit in four consecutive lanes
#pragma unroll
    for (int s = 0; s < QA; ++s) {
      float t = csum[s];
      t += __shfl_xor(t, 1);
      t += __shfl_xor(t, 2);
      if (mq == 0) p.a_colsum_ws[(size_t)split * p.Na + a0 + cl + CPP * s] = t;
    }
  }
  float* out = p.ws + (size_t)split * p.Na * p.Nb;
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = a0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int col = b0 + wn * 64 + j * 32 + l31;
        out[(size_t)row * p.Nb + col] = acc[i][j][r] * ssgn;
      }
}

template <int BMODE, int APRO, int WGM, int WGN>
int launch_cfg(const spgan_gemm_tn_args& a, int splits, int rows, hipStream_t s) {
  using X = T3<WGM, WGN>;
  static LdsOptIn opt;
  opt.ensure(reinterpret_cast<const void*>(&gemm_tn_wide3_kernel<BMODE, APRO, WGM, WGN>), (int)X::LDS);
  hipLaunchKernelGGL((gemm_tn_wide3_kernel<BMODE, APRO, WGM, WGN>), dim3((a.Na / X::XM) * (a.Nb / X::XN), splits), dim3(X::THREADS), X::LDS, s, a, rows);
  return spgan_launch_status();
}

template <int BMODE, int APRO>
int launch_apro(const spgan_gemm_tn_args& a, int cfg, int splits, int rows, hipStream_t s) {
  switch (cfg) {
    case 24: return launch_cfg<BMODE, APRO, 2, 4>(a, splits, rows, s);
    case 22: return launch_cfg<BMODE, APRO, 2, 2>(a, splits, rows, s);
    case 14: return launch_cfg<BMODE, APRO, 1, 4>(a, splits, rows, s);
    default: return SPGAN_EINVAL;
  }
}

template <int BMODE>
int launch_bmode(const spgan_gemm_tn_args& a, int cfg, int splits, int rows, hipStream_t s) {
  if (a.A2) return SPGAN_EINVAL;      // not eligible (spgan_tn_wide3_eligible): see there
  if (a.a_scale) return launch_apro<BMODE, 1>(a, cfg, splits, rows, s);
  return launch_apro<BMODE, 0>(a, cfg, splits, rows, s);
}

}  // namespace

// Tile configuration (WGM*10 + WGN: output tile 128*WGM x 64*WGN) for an [Na, Nb] weight gradient, 0 when the shape is not this kernel's
int spgan_tn_wide3_config(int M, int Na, int Nb) {
  if (M < 64 || M % 32 || Na % 128 || Nb % 128) return 0;
  const int cfg = (Na % 256 == 0 ? 20 : 10) + (Nb % 256 == 0 ? 4 : 2);
  return cfg == 12 ? 0 : cfg;      // 128 x 128 tiles (two waves per workgroup): measured slower than the fp32 kernel (38 -> 47 us at 65536 x 128 x 128)
}

// The split plan of the split-bf16 kernel: one 512-thread workgroup per CU, two 256-thread ones, four 128-thread ones
void spgan_tn_wide3_plan(int M, int Na, int Nb, int* splits, int* rows) {
  const int cfg = spgan_tn_wide3_config(M, Na, Nb);
  const int xm = (cfg / 10) * 128, xn = (cfg % 10) * 64, threads = xm * xn / 128;
  const int tiles = (Na / xm) * (Nb / xn);
  const int target = 256 * 512 / threads;
  int want = cdiv(target, tiles);
  int r = cdiv(M, want);
  if (r < 64) r = 64;
  r = cdiv(r, 32) * 32;
  *rows = r;
  *splits = cdiv(M, r);
}

bool spgan_tn_wide3_eligible(const spgan_gemm_tn_args& a) {
  auto al4 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 3) == 0; };
  if (a.mfma_lp != 2 || !spgan_tn_wide3_config(a.M, a.Na, a.Nb)) return false;
  if (a.b_mode == SPGAN_A_EDGE || a.a_half || a.b_half) return false;
  // the two-tensor lazy A operand doubles the A quads in flight (8 more registers per quad on top of 128 accumulators): the kernel spills and
  // measured no faster (65536 x 256 x 256: 112 us) or slower (256 x 128: 212 us against 63) than the fp32 kernel -- it keeps that one
  if (a.A2) return false;
  if (!al4(a.A) || !al4(a.B)) return false;
  if (a.b_mode != SPGAN_A_PLAIN && !(a.p_slope >= 0.f && a.p_slope <= 1.f)) return false;
  if (a.a_lrelu && !(a.a_slope >= 0.f && a.a_slope <= 1.f)) return false;
  if ((double)a.M * a.lda >= 4294967296.0 || (double)a.M * a.ldb >= 4294967296.0) return false;
  return true;
}

int spgan_launch_tn_wide3(const spgan_gemm_tn_args& a, int splits, int rows, hipStream_t s) {
  const int cfg = spgan_tn_wide3_config(a.M, a.Na, a.Nb);
  if (a.b_mode == SPGAN_A_PLAIN) return launch_bmode<SPGAN_A_PLAIN>(a, cfg, splits, rows, s);
  return launch_bmode<SPGAN_A_AFFINE_LRELU>(a, cfg, splits, rows, s);
}
