// gemm_nt with fp16 operands (spgan_gemm_nt_args.mfma_f16 == 1, BASELINE configs[4] "fp16 MFMA MLPs") on 256-row tiles, in the row-pipelined
// form of gemm_wide3.hip: the fp32 operands are rounded to fp16 (round to nearest) where the prologue runs -- between the global load and the LDS
// store of the NEXT k-tile, slot by slot in the gaps of the current tile's MFMAs -- and multiplied with v_mfma_f32_32x32x16_f16, fp32 accumulation.
//
// Why a second fp16 kernel.  gemm_wide.hip's fp16 instantiation issues a k-tile's loads, MFMAs and LDS stores as three blocks between two
// barriers; with one MFMA where the fp32 form has eight, its k-loop is all staging (D.fc2.0: 86 us for 13.7 us of matrix work).  Here
//   * k-tiles of 32 (two MFMA k-steps), double-buffered in LDS: 64 KB for 256 x 256 tiles; a row holds its 32 fp16 as four 16-byte units
//     (k-step, lane half), unit u stored at position u ^ ((row / 4) % 4): the 16 lanes of a ds_read_b128 group hit 16 different bank groups and
//     the 8-byte staging stores of 16 consecutive lanes cover two whole rows;
//   * a k-tile is four tile rows of 4 MFMAs; rows 0-2 carry the staging of tile kt+1 (a slot: transform, convert, one 8-byte store, then the
//     slot's registers are reloaded with tile kt+2: every load has a whole k-tile to land), the barrier sits in front of row 3, whose gaps carry
//     the first fragment reads of tile kt+1;
//   * the operand stream is the bound: 64 KB of fp32 rows per k-tile and workgroup through the vector cache.
// Geometry as gemm_wide3.hip's 256 x 256 form: 2 x 4 waves of 128 x 64 (the statistics / pooling record tile of gemm_wide_epi.hpp), 512 threads.
// Every epilogue of gemm_wide.hip, incl. the 16-bit result storage.  Measured (tools/nt16_bench.py): D.fc2.0 325 -> 381 TF (one pass), 358 -> 428 TF
// (three passes); the operand stream (fp32 rows, 64 KB per k-tile and workgroup) is what is left.
#include <type_traits>
#include "gemm_wide.hpp"
#include "gemm_wide_epi.hpp"

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4w __attribute__((ext_vector_type(4)));

namespace {

constexpr int HK = 32;    // k-tile: two MFMA k-steps
constexpr int HLW = 16;   // 4-byte words per LDS row (32 fp16)
constexpr int TI = 4, TJ = 2;

template <int WGN>
struct X16 {
  static constexpr int XM = 256, XN = WGN * 64, THREADS = 2 * WGN * 64;
  static constexpr int RPP = THREADS / 8;                                       // rows staged by one pass of the workgroup (8 float4 per row)
  static constexpr int A_SLOTS = XM / RPP, B_SLOTS = XN / RPP;                  // float4 staging slots per thread, operand and k-tile: 4 / 4 (WGN 4), 8 / 4 (WGN 2)
  static constexpr int TILE_A = XM * HLW, TILE_B = XN * HLW;                    // words
  static constexpr int BUF = TILE_A + TILE_B;
  static constexpr size_t LDS = (size_t)2 * BUF * sizeof(uint32_t);             // 64 KB / 48 KB
};

template <int AMODE, int EPI, int WGN>
__global__ __launch_bounds__((X16<WGN>::THREADS), 2) void gemm_nt_wide16_kernel(const spgan_gemm_nt_args p_) {
  using X = X16<WGN>;
  const spgan_gemm_nt_args& p = p_;  // stays in the kernarg segment (scalar loads)
  constexpr bool affine = AMODE != SPGAN_A_PLAIN;
  constexpr int XM = X::XM, XN = X::XN, AS = X::A_SLOTS, BS = X::B_SLOTS, RPP = X::RPP;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem16[];

  const int tilesN = p.N / XN, tilesM = p.M / XM;
  const int id = blockIdx.x, xcd = id & 7, t = id >> 3;  // XCD-aware: all N-tiles of one M-tile share an L2
  const int tn = t % tilesN, tm = xcd + 8 * (t / tilesN);
  if (tm >= tilesM) return;
  const int m0 = tm * XM, n0 = tn * XN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WGN, wn = wave % WGN;
  const int l31 = lane & 31, lh = lane >> 5;

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[AS], rb[BS];
  float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = make_float4(0.f, 0.f, 0.f, 0.f);
  const int lrow = tid >> 3, lc4 = (tid & 7) * 4;  // staging slot: row lrow + RPP*i, k offset lc4
  // 32-bit element offsets (host: M*lda, N*ldw < 2^32): scalar base + 32-bit offset addressing
  const unsigned oa = (unsigned)(m0 + lrow) * (unsigned)p.lda + (unsigned)lc4, sa = (unsigned)RPP * (unsigned)p.lda;
  const unsigned ow = (unsigned)(n0 + lrow) * (unsigned)p.ldw + (unsigned)lc4, sw = (unsigned)RPP * (unsigned)p.ldw;
  const float* pPsc = p_.p_scale;
  const float* pPsh = p_.p_shift;
  if (affine && p.p_group_rows > 0) {
    for (int r = m0 - p.p_group_rows; r >= 0; r -= p.p_group_rows) {  // a handful of groups: scalar subtractions instead of a division
      pPsc += p.K;
      pPsh += p.K;
    }
  }
  // LDS word offset of this thread's 8-byte staging slot (RPP is a multiple of 16: bits 2-3 of the row are lrow's for every slot):
  // k-quad lc4/4 lies in 16-byte unit lc4/8, half lc4/4 % 2 of it; unit u is stored at position u ^ ((row / 4) % 4)
  const int st_off = lrow * HLW + 4 * ((lc4 >> 3) ^ ((lrow >> 2) & 3)) + ((lc4 >> 2) & 1) * 2;

  auto load_a = [&](int i, int k0) { ra[i] = *reinterpret_cast<const float4*>(p.A + (oa + (unsigned)i * sa + (unsigned)k0)); };
  auto load_b = [&](int i, int k0) { rb[i] = *reinterpret_cast<const float4*>(p.W + (ow + (unsigned)i * sw + (unsigned)k0)); };
  auto load_pro = [&](int k0) {
    if (affine) {
      psc = *reinterpret_cast<const float4*>(pPsc + k0 + lc4);
      psh = *reinterpret_cast<const float4*>(pPsh + k0 + lc4);
    }
  };
  const float sl = affine ? p.p_slope : 1.f;     // host: 0 <= p_slope <= 1, so that lrelu(v) = max(v, v*slope)
  // the slot's values pass through an empty volatile asm: the piece's arithmetic cannot be hoisted above the sched_barrier in front of its tile row
  auto pin4 = [](float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); };
  auto st4h = [](uint32_t* q, float4 v) {
    const f32x4w f = {v.x, v.y, v.z, v.w};
    *reinterpret_cast<f16x4*>(q) = __builtin_convertvector(f, f16x4);
  };
  auto piece_a = [&](int i, int buf) {
    float4 v = ra[i];
    if (affine) {
      v.x = fmaf(v.x, psc.x, psh.x); v.y = fmaf(v.y, psc.y, psh.y); v.z = fmaf(v.z, psc.z, psh.z); v.w = fmaf(v.w, psc.w, psh.w);
      v.x = fmaxf(v.x, v.x * sl); v.y = fmaxf(v.y, v.y * sl); v.z = fmaxf(v.z, v.z * sl); v.w = fmaxf(v.w, v.w * sl);
    }
    st4h(smem16 + buf * X::BUF + st_off + i * RPP * HLW, v);
  };
  auto piece_b = [&](int i, int buf) { st4h(smem16 + buf * X::BUF + X::TILE_A + st_off + i * RPP * HLW, rb[i]); };

  // ---- fragments: lane (l31, lh) reads, per k-step ks, the 16-byte unit 2*ks + lh of its row (rows 32*i further have the same bits 2-3)
  const int sw_r = (l31 >> 2) & 3;
  const int fa_off = (wm * 128 + l31) * HLW, fb_off = X::TILE_A + (wn * 64 + l31) * HLW;
  const int u0 = 4 * ((0 + lh) ^ sw_r), u1 = 4 * ((2 + lh) ^ sw_r);      // word offsets of the two k-steps' units
  f16x8 aq[2][2];          // [i & 1][k-step]: row i+1 is read while row i multiplies
  constexpr bool XPF = X::THREADS == 512;   // one workgroup per CU: the next k-tile's first fragments are read across the barrier, under tile row 3
  f16x8 bq[XPF ? 2 : 1][2][TJ];      // [current / (XPF) next k-tile][k-step][j]
  auto read_a = [&](f16x8 (&dst)[2], int buf, int i) {
    dst[0] = *reinterpret_cast<const f16x8*>(smem16 + buf * X::BUF + fa_off + i * 32 * HLW + u0);
    dst[1] = *reinterpret_cast<const f16x8*>(smem16 + buf * X::BUF + fa_off + i * 32 * HLW + u1);
  };
  auto read_b = [&](f16x8 (&dst)[2][TJ], int buf) {
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      dst[0][j] = *reinterpret_cast<const f16x8*>(smem16 + buf * X::BUF + fb_off + j * 32 * HLW + u0);
      dst[1][j] = *reinterpret_cast<const f16x8*>(smem16 + buf * X::BUF + fb_off + j * 32 * HLW + u1);
    }
  };
  auto mfma_row = [&](int i, const f16x8 (&a)[2], const f16x8 (&b)[2][TJ]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], b[ks][j], acc[i][j], 0, 0, 0);
  };
  // issue order inside one tile row (4 MFMAs): every MFMA gap carries `valu` VALU instructions, one LDS and one VMEM instruction
  auto row_schedule = [&](int valu) {
#pragma unroll
    for (int g = 0; g < 2 * TJ; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
      if (valu >= 12) __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
      else if (valu >= 6) __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
      else if (valu > 0) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);   // DS
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
      __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);   // DS
    }
  };

  auto ktile = [&](int kt, auto st, auto ld, auto cur) {
    constexpr bool ST = decltype(st)::value, LD = decltype(ld)::value;
    constexpr int C = decltype(cur)::value;
    const int buf = kt & 1, nb = buf ^ 1, k2 = (kt + 2) * HK;
    constexpr int AH = AS / 2;
    // row 0
#pragma unroll
    for (int i = 0; i < AH; ++i) if (ST) pin4(ra[i]);
    read_a(aq[1], buf, 1);
    mfma_row(0, aq[0], bq[C]);
#pragma unroll
    for (int i = 0; i < AH; ++i) {
      if (ST) piece_a(i, nb);
      if (LD) load_a(i, k2);
    }
    row_schedule(ST ? AH * 4 : 0);
    __builtin_amdgcn_sched_barrier(0);
    // row 1
#pragma unroll
    for (int i = AH; i < AS; ++i) if (ST) pin4(ra[i]);
    read_a(aq[0], buf, 2);
    mfma_row(1, aq[1], bq[C]);
#pragma unroll
    for (int i = AH; i < AS; ++i) {
      if (ST) piece_a(i, nb);
      if (LD) load_a(i, k2);
    }
    if (LD) load_pro(k2);
    row_schedule(ST ? AH * 4 : 0);
    __builtin_amdgcn_sched_barrier(0);
    // row 2
#pragma unroll
    for (int i = 0; i < BS; ++i) if (ST) pin4(rb[i]);
    read_a(aq[1], buf, 3);
    mfma_row(2, aq[0], bq[C]);
#pragma unroll
    for (int i = 0; i < BS; ++i) {
      if (ST) piece_b(i, nb);
      if (LD) load_b(i, k2);
    }
    row_schedule(ST ? 3 : 0);
    __builtin_amdgcn_sched_barrier(0);
    if (XPF) {
      __syncthreads();
      // row 3: the first fragments of the next k-tile are read under it
      if (ST) {
        read_a(aq[0], nb, 0);
        read_b(bq[C ^ 1], nb);
      }
      mfma_row(3, aq[1], bq[C]);
      row_schedule(0);
    } else {
      // row 3, then the barrier: the next tile's first fragments are read behind it (the CU's other workgroup covers the latency)
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
  const int nk = p.K / HK;   // even, >= 2 (host: K % 64 == 0)
#pragma unroll
  for (int i = 0; i < AS; ++i) load_a(i, 0);
  load_pro(0);
#pragma unroll
  for (int i = 0; i < BS; ++i) load_b(i, 0);
#pragma unroll
  for (int i = 0; i < AS; ++i) {
    piece_a(i, 0);
    load_a(i, HK);
  }
  load_pro(HK);
#pragma unroll
  for (int i = 0; i < BS; ++i) {
    piece_b(i, 0);
    load_b(i, HK);
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

  wide_epilogue<EPI, TI, TJ, true>(p, acc, m0, n0, wm, wn, l31, lh);
}

template <int AMODE, int EPI, int WGN>
int launch_cfg(const spgan_gemm_nt_args& a, hipStream_t s) {
  using X = X16<WGN>;
  static LdsOptIn opt;
  opt.ensure(reinterpret_cast<const void*>(&gemm_nt_wide16_kernel<AMODE, EPI, WGN>), (int)X::LDS);
  const int tm8 = cdiv(a.M / X::XM, 8) * 8;
  hipLaunchKernelGGL((gemm_nt_wide16_kernel<AMODE, EPI, WGN>), dim3(tm8 * (a.N / X::XN)), dim3(X::THREADS), X::LDS, s, a);
  return spgan_launch_status();
}

template <int AMODE, int EPI>
int launch(const spgan_gemm_nt_args& a, hipStream_t s) {
  return launch_cfg<AMODE, EPI, 4>(a, s);
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

bool spgan_nt_wide16_eligible(const spgan_gemm_nt_args& a) {
  if (a.mfma_f16 != 1) return false;
  if (a.M < 256 || a.M % 256 || a.N % 256 || a.K % (2 * HK)) return false;
  if (a.tail.enabled || a.batch > 1 || a.A2 || a.a_half || a.sp_val) return false;
  if (a.a_mode == SPGAN_A_EDGE || a.epi_mode == SPGAN_EPI_EDGE_BNBWD) return false;
  if (a.lda % 4 || a.ldw % 4 || !al16(a.A) || !al16(a.W)) return false;
  if (a.a_mode != SPGAN_A_PLAIN && (!al16(a.p_scale) || !al16(a.p_shift) || !(a.p_slope >= 0.f && a.p_slope <= 1.f))) return false;
  if (a.p_group_rows > 0 && a.p_group_rows % 256) return false;
  if (a.epi_mode == SPGAN_EPI_LINEAR) {
    if (a.rowbias && a.rows_per_group != 1 && a.rows_per_group % 256) return false;
    if (!a.Y && !a.stats && !a.pool_val) return false;
  } else {
    if (a.y_bf16 || a.y_half) return false;
    if (a.epi_mode == SPGAN_EPI_MASK_OUT && a.a_mode != SPGAN_A_PLAIN) return false;
    if (a.epi_mode == SPGAN_EPI_BNBWD && a.rowbias && a.rows_per_group != 1) return false;
    if (a.epi_mode == SPGAN_EPI_MASK_OUT && (a.bias || a.rowbias)) return false;
  }
  return true;
}

// 256 x 256 tiles only: the 256 x 128 form (eight A slots per thread) spills and measured 2-3 x slower (profiles/r06_nt16_bench.txt)
int spgan_nt_wide16_tile_n(const spgan_gemm_nt_args&) { return 256; }

bool spgan_nt_wide16_selected(const spgan_gemm_nt_args& a) {
  static const bool off = getenv("SPGAN_NT_WIDE16") && atoi(getenv("SPGAN_NT_WIDE16")) == 0;
  if (off || a.tile_hint == 1 || !spgan_nt_wide16_eligible(a)) return false;
  if (a.tile_hint == 2) return true;
  return (long)(a.M / 256) * (a.N / 256) >= 256 && a.K >= 128;   // as spgan_nt_wide_pays: the tiles fill the chip, a k-loop worth its ramp
}

int spgan_launch_nt_wide16(const spgan_gemm_nt_args& a, hipStream_t s) {
  if (a.a_mode == SPGAN_A_PLAIN) return launch_epi<SPGAN_A_PLAIN>(a, s);
  return launch_epi<SPGAN_A_AFFINE_LRELU>(a, s);
}
