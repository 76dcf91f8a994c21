// gemm_nt with split-bf16 operands ("bf16x3", spgan_gemm_nt_args.mfma_f16 == 2) on 256-row tiles: the fp32-equivalent product on the
// bf16 matrix pipe for the large aligned products of the train step.
//
// Every fp32 operand value is split exactly into three bfloat16 terms v = hi + mid + lo (round-to-nearest at each level: the residuals are
// exactly representable, |mid| <= 2^-9 |v|, |lo| <= 2^-18 |v|) and a*b is evaluated as the six leading cross terms
//   lo*hi + hi*lo + mid*mid + mid*hi + hi*mid + hi*hi        (dropped: mid*lo + lo*mid + lo*lo <= 2^-26 |a*b|; every bf16 x bf16 product is
// exact in fp32) with v_mfma_f32_32x32x16_bf16, fp32 accumulation -- 6 MFMAs of 32 cycles per 16 k against 8 fp32 MFMAs of 64 cycles:
// 2.67 x the fp32 matrix pipe (2500 / 6 = 416.7 TFLOP/s fp32-equivalent peak).
//
// Geometry.  A workgroup owns 256 rows x (WGN*64) columns; its 2 x WGN waves own 128 x 64 each (4 x 2 accumulator tiles of 32x32: the
// statistics / pooling record tile of gemm_wide_epi.hpp).  WGN = 4: 256 x 256 tiles, 512 threads, one workgroup per CU; WGN = 2: 256 x 128
// tiles, 256 threads, two workgroups per CU (N % 256 != 0, or too few 256 x 256 tiles to fill the chip).
// k-tiles of 16 = ONE MFMA k-step, double-buffered in LDS (one barrier per k-tile); an LDS tile holds, per operand, three planes of
// [rows][16 bf16] without padding: a lane's fragment is the 16 bytes (row, k-half), stored at half index (k-half ^ bit 3 of the row) so that
// the 16 lanes of a ds_read_b128 group hit 16 different 16-byte bank groups, and so that the 8-byte staging stores of 16 consecutive lanes
// (4 rows x 4 k-quads) cover 128 contiguous bytes.  18 fragment reads feed 48 MFMAs per wave and k-tile.
// The split runs where the prologue runs: between the global load and the LDS store of the NEXT k-tile, under the current tile's MFMAs
// (11 VALU per pair of values: 3 v_cvt_pk_bf16_f32, 4 unpack, 4 subtract).  Loads are issued a whole k-tile ahead of their LDS store.
#include <type_traits>
#include "gemm_wide.hpp"
#include "gemm_wide_epi.hpp"

#include "split_bf16.hpp"

namespace {

constexpr int XK = 16;    // k-tile
constexpr int PLW = 8;    // 4-byte words per row and plane (16 bf16)
constexpr int TI = 4, TJ = 2;

// WGM x WGN waves of 128 x 64: (2,4) 256 x 256 tiles, 512 threads, one workgroup per CU; (2,2) 256 x 128 and (1,4) 128 x 256: 256 threads, two
// workgroups per CU (one's ramp and epilogue under the other's MFMAs).
template <int WGM, int WGN>
struct X3 {
  static constexpr int XM = WGM * 128, XN = WGN * 64, THREADS = WGM * WGN * 64;
  static constexpr int RPP = THREADS / 4;                                       // rows staged by one pass of the workgroup (4 float4 per row)
  static constexpr int A_SLOTS = XM / RPP, B_SLOTS = XN / RPP;                  // float4 staging slots per thread, operand and k-tile (fp32 rows)
  static constexpr int SPP = XN * 2 / THREADS;                                  // 16-byte image slots per thread and plane (pre-split B)
  static constexpr int PLANE_A = XM * PLW, PLANE_B = XN * PLW;                  // words per plane
  static constexpr int BUF = 3 * (PLANE_A + PLANE_B);                           // words per buffer
  static constexpr size_t LDS = (size_t)2 * BUF * sizeof(uint32_t);             // 96 KB (2,4) / 72 KB (2,2), (1,4)
};

template <int AMODE, int EPI, int WGM, int WGN, int BPRE>
__global__ __launch_bounds__((X3<WGM, WGN>::THREADS), 2) void gemm_nt_wide3_kernel(const spgan_gemm_nt_args p_) {
  using X = X3<WGM, WGN>;
  const spgan_gemm_nt_args& p = p_;  // stays in the kernarg segment (scalar loads)
  constexpr bool affine = AMODE != SPGAN_A_PLAIN;
  constexpr bool sparse = AMODE == SPGAN_WIDE_A_SPARSE;
  constexpr int XM = X::XM, XN = X::XN, AS = X::A_SLOTS, BS = X::B_SLOTS, RPP = X::RPP, SPP = X::SPP;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem3[];

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

  // B staging: BPRE = 0: fp32 rows of W, split here like A; BPRE = 1: 16-byte pieces of W's split image (spgan_split_bf16x3_image: the LDS tile
  // itself, plane by plane), copied as they are -- slot s of a thread is 16-byte unit (s % SPP)*THREADS + tid of plane s / SPP
  constexpr int BSL = BPRE ? 3 * SPP : BS;
  float4 ra[AS], rb[BSL];
  float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = make_float4(0.f, 0.f, 0.f, 0.f);
  int4 spa = make_int4(-1, -1, -1, -1);
  float4 spv = make_float4(0.f, 0.f, 0.f, 0.f);
  // Staging slots.  A thread stages, per operand and k-tile, one float4 (k offset lc4) of SLOTS rows.  Slot i of thread row-index lrow = tid / 4
  // takes tile row  slot_row(i) + 64 * (lrow / 32) + lrow % 32:  32-row tile 2*(lrow/32 + (RPP/32)*(i/2)) + (i & 1) -- every ODD slot stages rows of
  // an odd 32-row tile, every even slot rows of an even one (what the sign checkerboard below keys on: a compile-time property of the slot).
  const int lrow = tid >> 2, lc4 = (tid & 3) * 4;
  const int lrow2 = 64 * (lrow >> 5) + (lrow & 31);
  auto slot_row = [](int i) { return 32 * (i & 1) + 2 * RPP * (i >> 1); };
  static_assert(AS % 2 == 0 && (BPRE || BS % 2 == 0), "slots come in (even tile, odd tile) pairs");
  // 32-bit element offsets (host: M*lda, N*ldw < 2^32): scalar base + 32-bit offset addressing
  const unsigned oa = (unsigned)(m0 + lrow2) * (unsigned)p.lda + (unsigned)lc4, sa = (unsigned)p.lda;
  const unsigned ow = (unsigned)(n0 + lrow2) * (unsigned)p.ldw + (unsigned)lc4, sw = (unsigned)p.ldw;
  const int sp_b = sparse ? m0 / p.sp_rows : 0;  // the whole tile lies in one shape (host: sp_rows % 256 == 0)
  const float* pPsc = p_.p_scale;
  const float* pPsh = p_.p_shift;
  if (affine && p.p_group_rows > 0) {
    for (int r = m0 - p.p_group_rows; r >= 0; r -= p.p_group_rows) {  // a handful of groups: scalar subtractions instead of a division
      pPsc += p.K;
      pPsh += p.K;
    }
  }
  // LDS word offset of this thread's 8-byte staging slot inside a plane (row lrow; RPP is a multiple of 16, so bit 3 of the row is the same for every slot)
  const int st_off = lrow2 * PLW + 4 * ((lc4 >> 3) ^ ((lrow >> 3) & 1)) + ((lc4 >> 2) & 1) * 2;      // (slot_row(i) is a multiple of 16: bit 3 of the row is lrow's)

  // ---- staging, one float4 slot at a time ("piece"): global load -> (one k-tile later) prologue transform + split + three 8-byte LDS stores
  auto load_a = [&](int i, int k0) { ra[i] = *reinterpret_cast<const float4*>(p.A + (oa + (unsigned)slot_row(i) * sa + (unsigned)k0)); };
  const float4* wimg = reinterpret_cast<const float4*>(p_.w_image) + (size_t)n0 * 2 + tid;      // 16-byte units: [k-tile][plane][N][2]
  const unsigned img_plane = (unsigned)p.N * 2u, img_ktile = 3u * img_plane;
  auto load_b = [&](int i, int k0) {
    if (BPRE) rb[i] = wimg[(unsigned)(k0 / XK) * img_ktile + (unsigned)(i / SPP) * img_plane + (unsigned)((i % SPP) * X::THREADS)];
    else rb[i] = *reinterpret_cast<const float4*>(p.W + (ow + (unsigned)slot_row(i) * sw + (unsigned)k0));
  };
  auto load_pro = [&](int k0) {
    if (affine) {
      psc = *reinterpret_cast<const float4*>(pPsc + k0 + lc4);
      psh = *reinterpret_cast<const float4*>(pPsh + k0 + lc4);
    }
    if (sparse) {
      const size_t off = (size_t)sp_b * p.K + k0 + lc4;
      spa = *reinterpret_cast<const int4*>(p.sp_arg + off);
      spv = *reinterpret_cast<const float4*>(p.sp_val + off);
    }
  };
  const float sl = affine ? p.p_slope : 1.f;     // host: 0 <= p_slope <= 1, so that lrelu(v) = max(v, v*slope) (same bits as lrelu_f, one VALU less)
  // pin4: the slot's values pass through an empty volatile asm, i.e. through a point of the side-effect order -- the piece's arithmetic cannot be
  // hoisted above the sched_barrier in front of its tile row (pure VALU code is otherwise free to cross those fences at instruction selection)
  auto pin4 = [](float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); };
  // Sign checkerboard.  The bf16 MFMA's accumulation truncates the bits it shifts out toward -infinity (measured: every output carries a bias of
  // about -0.2 .. -0.6 * 2^-24 of sum |a||b|, where the fp32 MFMA's fmaf chain is unbiased) -- harmless per element, but BatchNorm statistics
  // and weight gradients SUM 10^5 of these outputs and the bias does not average out.  Rows (of A and of W) whose 32-row tile index is odd are
  // staged NEGATED (exact: the split of -v is minus the split of v), so accumulator tile (i, j) holds (-1)^(i+j) times its block and is negated
  // back in front of the epilogue: the bias alternates in sign from tile to tile, along the rows and along the columns, and cancels in sums.
  auto neg4 = [](float4& v) { v = make_float4(-v.x, -v.y, -v.z, -v.w); };      // odd slots = rows of odd 32-row tiles (slot_row): a compile-time property
  auto piece_a = [&](int i, int buf) {
    float4 v = ra[i];
    if (affine) {
      v.x = fmaf(v.x, psc.x, psh.x); v.y = fmaf(v.y, psc.y, psh.y); v.z = fmaf(v.z, psc.z, psh.z); v.w = fmaf(v.w, psc.w, psh.w);
      v.x = fmaxf(v.x, v.x * sl); v.y = fmaxf(v.y, v.y * sl); v.z = fmaxf(v.z, v.z * sl); v.w = fmaxf(v.w, v.w * sl);
    }
    if (sparse) {
      const int m = m0 + lrow2 + slot_row(i);
      v.x += (spa.x == m) ? spv.x : 0.f;
      v.y += (spa.y == m) ? spv.y : 0.f;
      v.z += (spa.z == m) ? spv.z : 0.f;
      v.w += (spa.w == m) ? spv.w : 0.f;
    }
    if (i & 1) neg4(v);
    st_split4(smem3 + buf * X::BUF + st_off + slot_row(i) * PLW, X::PLANE_A, v);
  };
  auto piece_b = [&](int i, int buf) {
    float4 v = rb[i];
    if (BPRE) *reinterpret_cast<float4*>(smem3 + buf * X::BUF + 3 * X::PLANE_A + (i / SPP) * X::PLANE_B + ((i % SPP) * X::THREADS + tid) * 4) = v;   // (the image holds the signs)
    else if (i & 1) neg4(v);
    if (!BPRE) st_split4(smem3 + buf * X::BUF + 3 * X::PLANE_A + st_off + slot_row(i) * PLW, X::PLANE_B, v);
  };

  // ---- fragments: lane (l31, lh) reads the 16 bytes (row, k-half lh) of each plane; rows 32*i further have the same bit 3
  const int fa_off = (wm * 128 + l31) * PLW + 4 * (lh ^ ((l31 >> 3) & 1));
  const int fb_off = 3 * X::PLANE_A + (wn * 64 + l31) * PLW + 4 * (lh ^ ((l31 >> 3) & 1));
  bf16x8 aq[2][3];          // the A fragments of tile row i live in aq[i & 1]: row i+1 is read while row i multiplies
  constexpr bool XPF = X::THREADS == 512;   // one workgroup per CU: the next k-tile's first fragments are read across the barrier, under tile row 3
  bf16x8 bq[XPF ? 2 : 1][3][TJ];   // the B fragments of the current k-tile and (XPF), from its last tile row on, of the next one
  auto read_a = [&](bf16x8 (&dst)[3], int buf, int i) {
#pragma unroll
    for (int q = 0; q < 3; ++q) dst[q] = *reinterpret_cast<const bf16x8*>(smem3 + buf * X::BUF + fa_off + q * X::PLANE_A + i * 32 * PLW);
  };
  auto read_b = [&](bf16x8 (&dst)[3][TJ], int buf) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int j = 0; j < TJ; ++j) dst[q][j] = *reinterpret_cast<const bf16x8*>(smem3 + buf * X::BUF + fb_off + q * X::PLANE_B + j * 32 * PLW);
  };
  // the six cross terms of one tile row, small ones first; the two accumulators of the row alternate
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
  // issue order inside one tile row (12 MFMAs): every MFMA gap carries `valu` VALU instructions and, every other gap, one LDS instruction
  auto row_schedule = [&](int valu) {
#pragma unroll
    for (int g = 0; g < 2 * 3 * TJ; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
      if (valu == 2) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      else if (valu == 3) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
      else if (valu == 4) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      else if (valu >= 5) __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
      if (g & 1) __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);   // DS
      else __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);         // VMEM read
    }
  };

  // One k-tile = four tile rows of 12 MFMAs.  Rows 0-2 carry, in their MFMA gaps, the staging of tile kt+1 into the other LDS buffer (slot by
  // slot: transform + split + store, then the slot's registers are reloaded with tile kt+2) and the fragment reads of the next row; the barrier
  // sits in front of row 3, whose gaps carry the first fragment reads of tile kt+1 -- no LDS latency is exposed behind the barrier.
  auto ktile = [&](int kt, auto st, auto ld, auto cur) {
    constexpr bool ST = decltype(st)::value, LD = decltype(ld)::value;
    constexpr int C = decltype(cur)::value;
    const int buf = kt & 1, nb = buf ^ 1, k2 = (kt + 2) * XK;
    constexpr int AH = (AS + 1) / 2;
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
    for (int i = 0; i < BSL; ++i) if (ST) pin4(rb[i]);
    read_a(aq[1], buf, 3);
    mfma_row(2, aq[0], bq[C]);
#pragma unroll
    for (int i = 0; i < BSL; ++i) {
      if (ST) piece_b(i, nb);
      if (LD) load_b(i, k2);
    }
    row_schedule(ST ? (BPRE ? 1 : 4) : 0);
    __builtin_amdgcn_sched_barrier(0);
    if (XPF) {
      __syncthreads();
      // row 3
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
  const int nk = p.K / XK;   // even, >= 2 (host: K % 32 == 0)
#pragma unroll
  for (int i = 0; i < AS; ++i) load_a(i, 0);
  load_pro(0);
#pragma unroll
  for (int i = 0; i < BSL; ++i) load_b(i, 0);
#pragma unroll
  for (int i = 0; i < AS; ++i) {
    piece_a(i, 0);
    load_a(i, XK);
  }
  load_pro(XK);
#pragma unroll
  for (int i = 0; i < BSL; ++i) {
    piece_b(i, 0);
    load_b(i, XK);
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

#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
      if ((i + j) & 1) acc[i][j] = -acc[i][j];     // undo the sign checkerboard
  wide_epilogue<EPI, TI, TJ, false>(p, acc, m0, n0, wm, wn, l31, lh);
}

template <int AMODE, int EPI, int WGM, int WGN, int BPRE>
int launch_cfg(const spgan_gemm_nt_args& a, hipStream_t s) {
  using X = X3<WGM, WGN>;
  static LdsOptIn opt;  // > 64 KB of dynamic LDS: once per kernel and device
  opt.ensure(reinterpret_cast<const void*>(&gemm_nt_wide3_kernel<AMODE, EPI, WGM, WGN, BPRE>), (int)X::LDS);
  const int tm8 = cdiv(a.M / X::XM, 8) * 8;
  hipLaunchKernelGGL((gemm_nt_wide3_kernel<AMODE, EPI, WGM, WGN, BPRE>), dim3(tm8 * (a.N / X::XN)), dim3(X::THREADS), X::LDS, s, a);
  return spgan_launch_status();
}

template <int AMODE, int EPI>
int launch(const spgan_gemm_nt_args& a, hipStream_t s) {
  if constexpr (AMODE == SPGAN_WIDE_A_SPARSE) {
    return launch_cfg<AMODE, EPI, 2, 4, 0>(a, s);   // the config rule's only form for the sparse addend
  } else {
    switch (spgan_nt_wide3_config(a)) {
      case 24: return a.w_image ? launch_cfg<AMODE, EPI, 2, 4, 1>(a, s) : launch_cfg<AMODE, EPI, 2, 4, 0>(a, s);
      case 14: return launch_cfg<AMODE, EPI, 1, 4, 1>(a, s);   // with the image only (config rule)
      default: return a.w_image ? launch_cfg<AMODE, EPI, 2, 2, 1>(a, s) : launch_cfg<AMODE, EPI, 2, 2, 0>(a, s);
    }
  }
}

template <int AMODE>
int launch_epi(const spgan_gemm_nt_args& a, hipStream_t s) {
  switch (a.epi_mode) {
    case SPGAN_EPI_LINEAR: return launch<AMODE, SPGAN_EPI_LINEAR>(a, s);
    case SPGAN_EPI_MASK_OUT: return AMODE == SPGAN_A_PLAIN ? launch<SPGAN_A_PLAIN, SPGAN_EPI_MASK_OUT>(a, s) : SPGAN_EINVAL;
    case SPGAN_EPI_BNBWD:
      if (a.gout_add) {   // stored tile gout_add + gout_scale * g: the activation operand only (eligibility rule)
        if constexpr (AMODE == SPGAN_A_AFFINE_LRELU) return launch<AMODE, SPGAN_WIDE_EPI_BNBWD_GOUT>(a, s);
        return SPGAN_EINVAL;
      }
      return launch<AMODE, SPGAN_EPI_BNBWD>(a, s);
  }
  return SPGAN_EINVAL;
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

// One thread per (row n, k-quad): 4 values -> 8 bytes in each plane of the image [k/16][plane][N][16 bf16], halves swapped where bit 3 of n is set.
__global__ __launch_bounds__(256) void split_image_kernel(const float* __restrict__ W, int ldw, int N, int K, uint32_t* __restrict__ img) {
  const int kq = K >> 2;
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= (long)N * kq) return;
  const int n = (int)(id / kq), k = (int)(id % kq) * 4;
  float4 v = *reinterpret_cast<const float4*>(W + (size_t)n * ldw + k);
  if ((n >> 5) & 1) v = make_float4(-v.x, -v.y, -v.z, -v.w);      // the kernel's sign checkerboard
  const int kt = k >> 4, c4 = (k >> 2) & 3;
  uint32_t* p = img + ((size_t)kt * 3 * N + n) * PLW + 4 * ((c4 >> 1) ^ ((n >> 3) & 1)) + (c4 & 1) * 2;
  st_split4(p, N * PLW, v);
}

extern "C" size_t spgan_split_bf16x3_image_bytes(int N, int K) { return (N > 0 && K > 0) ? (size_t)6 * N * K : 0; }

extern "C" int spgan_split_bf16x3_image(const float* W, int ldw, int N, int K, void* image, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(W && image && N > 0 && K > 0 && N % 128 == 0 && K % XK == 0 && ldw >= K && ldw % 4 == 0);
  SPGAN_CHECK_ARG((reinterpret_cast<uintptr_t>(W) & 15) == 0 && (reinterpret_cast<uintptr_t>(image) & 15) == 0);
  const long n = (long)N * (K / 4);
  hipLaunchKernelGGL(split_image_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)s_, W, ldw, N, K, reinterpret_cast<uint32_t*>(image));
  return spgan_launch_status();
}

extern "C" int spgan_gemm_nt_uses_w_image(const spgan_gemm_nt_args* a) {
  if (!a || a->a_mode == SPGAN_A_EDGE || a->A2 || a->epi_mode == SPGAN_EPI_EDGE_BNBWD) return 0;
  return spgan_nt_wide3_selected(*a) ? 1 : 0;
}

bool spgan_nt_wide3_eligible(const spgan_gemm_nt_args& a) {
  if (a.mfma_f16 != 2) return false;
  if (a.M < 256 || a.M % 256 || a.N % 128 || a.K % (2 * XK)) return false;
  if (a.tail.enabled || a.batch > 1 || a.A2 || a.a_half || a.y_bf16 || a.y_half) return false;
  if (a.a_mode == SPGAN_A_EDGE || a.epi_mode == SPGAN_EPI_EDGE_BNBWD) return false;
  if (a.lda % 4 || a.ldw % 4 || !al16(a.A) || !al16(a.W) || !al16(a.w_image)) return false;
  if (a.a_mode != SPGAN_A_PLAIN && (!al16(a.p_scale) || !al16(a.p_shift) || !(a.p_slope >= 0.f && a.p_slope <= 1.f))) return false;
  if (a.p_group_rows > 0 && a.p_group_rows % 256) return false;
  if (a.sp_val) {
    if (a.a_mode != SPGAN_A_AFFINE_LRELU || a.sp_rows % 256 || a.N % 256 || !al16(a.sp_val) || !al16(a.sp_arg)) return false;
  }
  if (a.epi_mode == SPGAN_EPI_LINEAR) {
    if (a.rowbias && a.rows_per_group != 1 && a.rows_per_group % 256) return false;
    if (!a.Y && !a.stats && !a.pool_val) return false;
  } else {
    if (a.epi_mode == SPGAN_EPI_MASK_OUT && a.a_mode != SPGAN_A_PLAIN) return false;
    if (a.epi_mode == SPGAN_EPI_BNBWD && a.rowbias && a.rows_per_group != 1) return false;
    if (a.epi_mode == SPGAN_EPI_MASK_OUT && (a.bias || a.rowbias)) return false;
  }
  if (a.gout_add && (a.epi_mode != SPGAN_EPI_BNBWD || a.a_mode != SPGAN_A_AFFINE_LRELU || a.sp_val)) return false;
  return true;
}

// Tile configuration (WGM*10 + WGN) of an eligible problem, from measurements on MI355X (tools/nt3_bench.py, profiles/r06_nt3_*):
//   N % 256 != 0 -> 256 x 128 tiles (22).  Short K (<= 128): two workgroups per CU pay (one's ramp and epilogue under the other's MFMAs; the k-loop
//   is 4-8 tiles long): 128 x 256 (14) with W's pre-split image, 256 x 128 (22) without (the B rows are then split in the workgroup, which a
//   128-row tile would do twice as often).  Longer K: 256 x 256 tiles (24) when they fill the chip at least once (decided from the rows of ONE
//   group, as spgan_nt_wide_pays does: a grouped launch runs the kernel its passes would run alone), else the two-workgroup forms.
//   The sparse addend keeps to (24) without image: its extra registers spill in the other forms.
int spgan_nt_wide3_config(const spgan_gemm_nt_args& a) {
  static const int force = getenv("SPGAN_NT3_CFG") ? atoi(getenv("SPGAN_NT3_CFG")) : 0;   // A/B measurements: 24 / 22 / 14
  if (a.N % 256) return 22;
  if (a.sp_val) return 24;
  if (force == 22 || force == 24 || (force == 14 && a.w_image)) return force;
  const long tiles = (long)((a.p_group_rows > 0 ? a.p_group_rows : a.M) / 256) * (a.N / 256);
  if (a.K <= 128 || tiles < 256) return a.w_image ? 14 : 22;
  return 24;
}

int spgan_nt_wide3_tile_n(const spgan_gemm_nt_args& a) { return spgan_nt_wide3_config(a) == 22 ? 128 : 256; }

bool spgan_nt_wide3_selected(const spgan_gemm_nt_args& a) {
  static const bool off = getenv("SPGAN_NT_WIDE3") && atoi(getenv("SPGAN_NT_WIDE3")) == 0;
  if (off || a.tile_hint == 1 || !spgan_nt_wide3_eligible(a)) return false;
  if (a.tile_hint == 2) return true;
  // a workgroup per CU at least, and a k-loop worth its ramp (the 128-row split kernel serves the rest)
  return (long)(a.M / 256) * (a.N / 128) >= 128 && a.K >= 64;
}

int spgan_launch_nt_wide3(const spgan_gemm_nt_args& a, hipStream_t s) {
  if (a.a_mode == SPGAN_A_PLAIN) return launch_epi<SPGAN_A_PLAIN>(a, s);
  if (a.sp_val) return launch_epi<SPGAN_WIDE_A_SPARSE>(a, s);
  return launch_epi<SPGAN_A_AFFINE_LRELU>(a, s);
}
