// gemm_nt for the large aligned products of the train step (M % 256 == 0, N % 256 == 0, K % 32 == 0, 16-byte aligned rows):
// 256 x 256 output tiles, 512 threads.
//
// Why a second geometry.  The 128-row kernels of gemm.hip run at a clock the chip's power limit sets (~1.8 GHz instead of 2.4 under
// fp32 MFMA load), so what a product costs is decided by the energy spent per MFMA besides the MFMA itself: LDS
// fragment reads, LDS stores and L2 -> LDS operand traffic.  Here
//   * 8 waves as 2 (rows) x 4 (columns), each owning 128 x 64 of the tile (4 x 2 MFMA tiles of 32x32, 128 accumulator registers):
//     6 fragment reads feed 16 MFMAs (0.375 LDS reads per MFMA, 0.5-0.75 in the 128-row kernels);
//   * every operand element is fetched once per 256 columns/rows of the other operand (128 or 64 before): half the L2 and
//     LDS-store traffic per FLOP;
//   * a wave's 128 rows are exactly one statistics / pooling record tile (the [M/128, N, 2] records of spgan_hip.h): the epilogue
//     needs no exchange between waves -- lane halves merge by one shuffle, nothing goes through LDS, no barrier after the k-loop.
// One workgroup per CU (139 KB of double-buffered LDS tiles, k-tiles of 32, one barrier per k-tile), two waves per SIMD (<= 256 VGPRs).
// Measured on MI355X inside the train step (tools/mfma_shapes.py, same box): D.fc2.0 (65536 x 1024 x 256 with BatchNorm+LeakyReLU
// prologue, statistics + pooling epilogue, output not stored) 327 -> 300 us = 73 % of the fp32 MFMA peak.
//
// Only straight-line code: shapes that do not tile exactly, the split-bf16 operand mode, the per-edge prologue/epilogue and the
// in-launch fan-in stay with gemm.hip (launch_nt falls through).  fp16 operands (mfma_f16 == 1) are served here: D.fc2.0 114 -> 99 us.
#include <type_traits>
#include "gemm_wide.hpp"
#include "gemm_wide_epi.hpp"

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

namespace {

constexpr int WM = 256, WN = 256, WK = 32, WLD = WK + 2;  // row stride 34 words == 2 (mod 32): conflict-free 8-byte fragment reads
constexpr int WTHREADS = 512;
constexpr int WGN = 4;           // waves along N (x 2 along M)
constexpr int TI = 4, TJ = 2;    // MFMA tiles per wave
constexpr int SLOTS = 4;         // float4 staging slots per thread and operand: 256 rows x 8 float4 / 512 threads
constexpr int RPP = 64;          // rows staged by one pass of the workgroup
constexpr int WLDH = WK / 2 + 2;  // fp16 operands: a row holds the 32 k-values as 16 words + 2 of padding (18 == 2 mod 16: conflict-free 8-byte reads)
template <int F16> constexpr size_t wide_lds() { return (size_t)2 * (WM + WN) * (F16 ? WLDH : WLD) * sizeof(float); }

// 4 consecutive k-values rounded to fp16 (round to nearest) into 2 LDS words
__device__ __forceinline__ void st_row4h(float* p, float4 v) {
  f32x4v f = {v.x, v.y, v.z, v.w};
  *reinterpret_cast<f16x4*>(p) = __builtin_convertvector(f, f16x4);
}

// F16 = 1: the operands are rounded to fp16 when they are staged into LDS (after the fp32 prologue) and multiplied with
// v_mfma_f32_32x32x8_f16, fp32 accumulation (spgan_gemm_nt_args.mfma_f16 == 1, BASELINE configs[4]); loads, prologues, epilogues fp32.
template <int AMODE, int EPI, int F16>
__global__ __launch_bounds__(WTHREADS, 2) void gemm_nt_wide_kernel(const spgan_gemm_nt_args p_) {
  const spgan_gemm_nt_args& p = p_;  // stays in the kernarg segment (scalar loads)
  constexpr bool affine = AMODE != SPGAN_A_PLAIN;
  constexpr bool sparse = AMODE == SPGAN_WIDE_A_SPARSE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LDX = F16 ? WLDH : WLD;  // LDS row stride in 4-byte words
  float* As = smem;                 // [2][WM*LDX]
  float* Bs = smem + 2 * WM * LDX;  // [2][WN*LDX]

  const int tilesN = p.N / WN, tilesM = p.M / WM;
  const int id = blockIdx.x, xcd = id & 7, t = id >> 3;  // XCD-aware: all N-tiles of one M-tile share an L2
  const int tn = t % tilesN, tm = xcd + 8 * (t / tilesN);
  if (tm >= tilesM) return;
  const int m0 = tm * WM, n0 = tn * WN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WGN, wn = wave % WGN;
  const int l31 = lane & 31, lh = lane >> 5;

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[SLOTS], rb[SLOTS];
  float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = make_float4(0.f, 0.f, 0.f, 0.f);
  int4 spa = make_int4(-1, -1, -1, -1);
  float4 spv = make_float4(0.f, 0.f, 0.f, 0.f);
  const int lrow = tid >> 3, lc4 = (tid & 7) * 4;
  // 32-bit element offsets (host: M*lda, N*ldw < 2^32): scalar base + 32-bit offset addressing
  const unsigned oa = (unsigned)(m0 + lrow) * (unsigned)p.lda + (unsigned)lc4, sa = (unsigned)RPP * (unsigned)p.lda;
  const unsigned ow = (unsigned)(n0 + lrow) * (unsigned)p.ldw + (unsigned)lc4, sw = (unsigned)RPP * (unsigned)p.ldw;
  const int sp_b = sparse ? m0 / p.sp_rows : 0;  // the whole tile lies in one shape (host: sp_rows % 256 == 0)
  // prologue vectors: one pair for all rows, or one pair per group of p_group_rows rows (host: p_group_rows % 256 == 0)
  const float* pPsc = p_.p_scale;
  const float* pPsh = p_.p_shift;
  if (affine && p.p_group_rows > 0) {
    for (int r = m0 - p.p_group_rows; r >= 0; r -= p.p_group_rows) {  // a handful of groups: scalar subtractions instead of a division
      pPsc += p.K;
      pPsh += p.K;
    }
  }

  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) ra[i] = *reinterpret_cast<const float4*>(p.A + (oa + (unsigned)i * sa + (unsigned)k0));
    if (affine) {
      psc = *reinterpret_cast<const float4*>(pPsc + k0 + lc4);
      psh = *reinterpret_cast<const float4*>(pPsh + k0 + lc4);
    }
    if (sparse) {
      const size_t off = (size_t)sp_b * p.K + k0 + lc4;
      spa = *reinterpret_cast<const int4*>(p.sp_arg + off);
      spv = *reinterpret_cast<const float4*>(p.sp_val + off);
    }
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) rb[i] = *reinterpret_cast<const float4*>(p.W + (ow + (unsigned)i * sw + (unsigned)k0));
  };
  // The prologue transform runs here, after the current tile's first MFMAs were issued: transforming at load time would put a
  // vmcnt wait in front of them.
  auto sstore = [&](int buf) {
    float* a = As + buf * WM * LDX + lrow * LDX + (F16 ? (lc4 >> 1) : lc4);
    float* b = Bs + buf * WN * LDX + lrow * LDX + (F16 ? (lc4 >> 1) : lc4);
    if (affine) {
      const float sl = p.p_slope;
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) {
        ra[i].x = lrelu_f(fmaf(ra[i].x, psc.x, psh.x), sl);
        ra[i].y = lrelu_f(fmaf(ra[i].y, psc.y, psh.y), sl);
        ra[i].z = lrelu_f(fmaf(ra[i].z, psc.z, psh.z), sl);
        ra[i].w = lrelu_f(fmaf(ra[i].w, psc.w, psh.w), sl);
      }
    }
    if (sparse) {
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) {
        const int m = m0 + lrow + RPP * i;
        ra[i].x += (spa.x == m) ? spv.x : 0.f;
        ra[i].y += (spa.y == m) ? spv.y : 0.f;
        ra[i].z += (spa.z == m) ? spv.z : 0.f;
        ra[i].w += (spa.w == m) ? spv.w : 0.f;
      }
    }
    if (F16) {
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) st_row4h(a + i * RPP * LDX, ra[i]);
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) st_row4h(b + i * RPP * LDX, rb[i]);
      return;
    }
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      *reinterpret_cast<float2*>(a + i * RPP * LDX) = make_float2(ra[i].x, ra[i].y);
      *reinterpret_cast<float2*>(a + i * RPP * LDX + 2) = make_float2(ra[i].z, ra[i].w);
    }
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      *reinterpret_cast<float2*>(b + i * RPP * LDX) = make_float2(rb[i].x, rb[i].y);
      *reinterpret_cast<float2*>(b + i * RPP * LDX + 2) = make_float2(rb[i].z, rb[i].w);
    }
  };
  // One lane's 8-byte read feeds the k-operands of two consecutive MFMAs (any permutation of k inside a tile is legal as long as
  // both operands agree).
  auto compute = [&](int buf, int kk0, int kk1) {
    const float* a = As + buf * WM * LDX + (wm * TI * 32 + l31) * LDX + 2 * lh;
    const float* b = Bs + buf * WN * LDX + (wn * TJ * 32 + l31) * LDX + 2 * lh;
#pragma unroll
    for (int kk = kk0; kk < kk1; ++kk) {
      if (F16) {  // two 8-byte reads = 8 halfs = the lane's k-operands of ONE v_mfma_f32_32x32x16_f16 (k-groups kk, kk+1; kk0/kk1 even)
        if (kk & 1) continue;
        f16x8 ah[TI], bh[TJ];
#pragma unroll
        for (int i = 0; i < TI; ++i)
          ah[i] = __builtin_shufflevector(*reinterpret_cast<const f16x4*>(a + i * 32 * LDX + kk * 4),
                                          *reinterpret_cast<const f16x4*>(a + i * 32 * LDX + kk * 4 + 4), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          bh[j] = __builtin_shufflevector(*reinterpret_cast<const f16x4*>(b + j * 32 * LDX + kk * 4),
                                          *reinterpret_cast<const f16x4*>(b + j * 32 * LDX + kk * 4 + 4), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        continue;
      }
      float2 af[TI], bf[TJ];
#pragma unroll
      for (int i = 0; i < TI; ++i) af[i] = *reinterpret_cast<const float2*>(a + i * 32 * LDX + kk * 4);
#pragma unroll
      for (int j = 0; j < TJ; ++j) bf[j] = *reinterpret_cast<const float2*>(b + j * 32 * LDX + kk * 4);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
    }
  };

  constexpr int KK = F16 ? WK / 8 : WK / 4;  // fragment reads per k-tile
  const int nk = p.K / WK;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload((kt + 1) * WK);  // next tile's loads fly under this tile's MFMAs
    compute(kt & 1, 0, KK / 2);
    if (kt + 1 < nk) sstore((kt + 1) & 1);  // into the other buffer: its last readers passed the previous barrier
    compute(kt & 1, KK / 2, KK);
    __syncthreads();
  }

  wide_epilogue<EPI, TI, TJ, F16 != 0>(p, acc, m0, n0, wm, wn, l31, lh);
}

template <int AMODE, int EPI, int F16>
int launch_op(const spgan_gemm_nt_args& a, hipStream_t s) {
  constexpr size_t lds = wide_lds<F16>();
  static LdsOptIn opt;  // > 64 KB of dynamic LDS: once per kernel and device
  opt.ensure(reinterpret_cast<const void*>(&gemm_nt_wide_kernel<AMODE, EPI, F16>), (int)lds);
  const int tm8 = cdiv(a.M / WM, 8) * 8;
  hipLaunchKernelGGL((gemm_nt_wide_kernel<AMODE, EPI, F16>), dim3(tm8 * (a.N / WN)), dim3(WTHREADS), lds, s, a);
  return spgan_launch_status();
}

template <int AMODE, int EPI>
int launch(const spgan_gemm_nt_args& a, hipStream_t s) {
  return a.mfma_f16 == 1 ? launch_op<AMODE, EPI, 1>(a, s) : launch_op<AMODE, EPI, 0>(a, s);
}

template <int AMODE>
int launch_epi(const spgan_gemm_nt_args& a, hipStream_t s) {
  switch (a.epi_mode) {
    case SPGAN_EPI_LINEAR: return launch<AMODE, SPGAN_EPI_LINEAR>(a, s);
    case SPGAN_EPI_MASK_OUT: return AMODE == SPGAN_A_PLAIN ? launch<SPGAN_A_PLAIN, SPGAN_EPI_MASK_OUT>(a, s) : SPGAN_EINVAL;
    case SPGAN_EPI_BNBWD: return launch<AMODE, SPGAN_EPI_BNBWD>(a, s);
  }
  return SPGAN_EINVAL;
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

bool spgan_nt_wide_eligible(const spgan_gemm_nt_args& a) {
  if (a.M < WM || a.M % WM || a.N % WN || a.K % WK || a.K < WK) return false;
  if ((a.mfma_f16 != 0 && a.mfma_f16 != 1) || a.tail.enabled || a.batch > 1 || a.A2) return false;  // fp32 or fp16 operands (not the split-bf16 mode)
  if (a.mfma_f16 == 1 && a.sp_val) return false;                                              // the sparse addend is an fp32-operand path (gemm.hip: same rule)
  if (a.a_mode == SPGAN_A_EDGE || a.epi_mode == SPGAN_EPI_EDGE_BNBWD) return false;
  if (a.lda % 4 || a.ldw % 4 || !al16(a.A) || !al16(a.W)) return false;
  if (a.a_mode != SPGAN_A_PLAIN && (!al16(a.p_scale) || !al16(a.p_shift))) return false;
  if (a.p_group_rows > 0 && a.p_group_rows % WM) return false;
  if (a.sp_val) {
    if (a.a_mode != SPGAN_A_AFFINE_LRELU || a.sp_rows % WM || !al16(a.sp_val) || !al16(a.sp_arg)) return false;
  }
  if (a.epi_mode == SPGAN_EPI_LINEAR) {
    if (a.rowbias && a.rows_per_group != 1 && a.rows_per_group % WM) return false;
    if (!a.Y && !a.stats && !a.pool_val) return false;
  } else {
    if (a.epi_mode == SPGAN_EPI_MASK_OUT && a.a_mode != SPGAN_A_PLAIN) return false;
    if (a.epi_mode == SPGAN_EPI_BNBWD && a.rowbias && a.rows_per_group != 1) return false;
    if (a.epi_mode == SPGAN_EPI_MASK_OUT && (a.bias || a.rowbias)) return false;
  }
  return true;
}

// Measured inside the train step (tools/mfma_shapes.py, MI355X): one workgroup per CU means a launch whose tiles make a single round
// runs its load / MFMA / epilogue phases in lockstep on all CUs -- with a heavy epilogue (the BatchNorm-backward one reads and writes
// [M,N]) that costs what the leaner main loop gains (65536 x 256 x 256: 103 -> 107 us), and k-loops of two tiles are all ramp
// (K = 64: 40 -> 45 us).  Wins: 65536 x 1024 x 256 + statistics/pooling 327 -> 300 us, 65536 x 256 x 128: 57 -> 54, 65536 x 1280 x 128: 224 -> 214.
bool spgan_nt_wide_pays(const spgan_gemm_nt_args& a) {
  // decided from the rows of ONE group: a grouped launch (several passes of a layer as one product) must run the kernel its passes
  // would run as separate calls -- the two kernels sum in different orders, and the grouped form is specified as bit-identical
  const long tiles = (long)((a.p_group_rows > 0 ? a.p_group_rows : a.M) / WM) * (a.N / WN);
  if (tiles < 256 || a.K < 128) return false;
  return a.epi_mode == SPGAN_EPI_LINEAR || tiles >= 512;
}

bool spgan_nt_wide_selected(const spgan_gemm_nt_args& a) {
  static const bool off = getenv("SPGAN_NT_WIDE") && atoi(getenv("SPGAN_NT_WIDE")) == 0;
  if (off || a.tile_hint == 1 || !spgan_nt_wide_eligible(a)) return false;
  return a.tile_hint == 2 || spgan_nt_wide_pays(a);
}

int spgan_launch_nt_wide(const spgan_gemm_nt_args& a, hipStream_t s) {
  if (a.a_mode == SPGAN_A_PLAIN) return launch_epi<SPGAN_A_PLAIN>(a, s);
  if (a.sp_val) return launch_epi<SPGAN_WIDE_A_SPARSE>(a, s);
  return launch_epi<SPGAN_A_AFFINE_LRELU>(a, s);
}
