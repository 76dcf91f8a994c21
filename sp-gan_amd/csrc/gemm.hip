// Shared-MLP contractions on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32
// accumulate, bit-exact fmaf chains; 157.3 TFLOP/s peak).  Two kernels:
//
//   gemm_nt  Y[M,N] = epi( pro(A)[M,K] . W[N,K]^T )   forward 1x1 convs / linears and input-gradients
//   gemm_tn  C[Na,Nb] = sum_m A[m,Na]^T . pro(B)[m,Nb] weight-gradients (deterministic split over m)
//
// Tiling (wave64, 4 waves / workgroup, 128-row output tiles):
//   gemm_nt: 128 x BN tile, BK = 32; BN = 128 (waves 2x2, 2x2 MFMA tiles each), 64 (2x2 waves, 2x1 tiles)
//            or 32 (4x1 waves, 1x1 tile).  LDS tiles are row-major [row][BK+2]: the leading dimension
//            34 == 2 (mod 64) makes the 32 rows x one k-pair that a half-wave reads with ds_read_b64 land on
//            64 distinct banks.  One lane's b64 read feeds the k-operands of TWO consecutive MFMAs (any
//            permutation of k inside a tile is legal as long as A and B agree).  For long K the LDS tiles
//            are double-buffered (one barrier per k-tile); the next tile's global loads always fly under
//            the current tile's MFMAs.  Measured on MI355X (tools/exp): 104-117 TFLOP/s on the big shapes vs
//            88-100 for the earlier 16x16x4 / single-buffer structure.
//   gemm_tn: both operands are staged [m][cols] exactly as they lie in memory (no transpose); a fragment
//            read is 32 lanes along the columns of one m-row (b32, conflict-free), double-buffered.
//   Workgroup -> tile map is XCD-aware: the 8 XCDs each have a private L2 and block b lands on XCD b%8,
//   so all N-tiles of one M-tile get ids with the same (b % 8): the A rows they share are fetched into one L2.
//
// Fused prologues/epilogues (see spgan_hip.h): BatchNorm-apply + LeakyReLU on the operand load, the
// EdgeBlock per-edge difference gather, bias / per-shape bias / activation, per-tile column statistics
// for the following train-mode BatchNorm, and the LeakyReLU/BatchNorm backward masks with their
// column sums.
#include <type_traits>

#include "common.hpp"
#include "gemm_wide.hpp"
#include "gemm_tn_wide3.hpp"
#include "sparse_rows.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

namespace {

// internal operand mode: A_AFFINE_LRELU with the sparse addend (sp_val != NULL) -- its own instantiation, so the plain
// affine kernels do not carry the addend's registers and LDS-patch code
constexpr int A_AFFINE_SPARSE = 3;
// internal operand mode: a = A*p_scale[k] + A2*p_scale2[k] + p_shift[k] (two tensors, no activation): the BatchNorm-backward operand
// dy = p*g + q*y + r evaluated on the operand load instead of by a pass of its own (spgan_gemm_nt_args.A2)
constexpr int A_AFFINE2 = 4;

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDT = BK + 2;  // 34

struct Tile {
  int tm, tn;
};

// XCD-aware bijection from the 1-D block id to (tileM, tileN); ids with tileM >= tilesM are idle.
__device__ __forceinline__ Tile map_tile(int tilesN) {
  const int id = blockIdx.x;
  const int xcd = id & 7;
  const int t = id >> 3;
  Tile r;
  r.tn = t % tilesN;
  r.tm = xcd + 8 * (t / tilesN);
  return r;
}

__device__ __forceinline__ float4 ld4(const float* p, bool vec, int k, int K) {
  // p points at element k of a row; K is the row's logical length.
  if (vec && k + 3 < K) return *reinterpret_cast<const float4*>(p);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (k < K) v.x = p[0];
  if (k + 1 < K) v.y = p[1];
  if (k + 2 < K) v.z = p[2];
  if (k + 3 < K) v.w = p[3];
  return v;
}

__device__ __forceinline__ float4 affine_lrelu4(float4 v, float4 sc, float4 sh, float slope) {
  v.x = lrelu_f(fmaf(v.x, sc.x, sh.x), slope);
  v.y = lrelu_f(fmaf(v.y, sc.y, sh.y), slope);
  v.z = lrelu_f(fmaf(v.z, sc.z, sh.z), slope);
  v.w = lrelu_f(fmaf(v.w, sc.w, sh.w), slope);
  return v;
}

// Zero the k >= K tail of a transformed operand (affine of a padded zero is not zero).
__device__ __forceinline__ float4 mask_tail(float4 v, int k, int K) {
  if (k >= K) v.x = 0.f;
  if (k + 1 >= K) v.y = 0.f;
  if (k + 2 >= K) v.z = 0.f;
  if (k + 3 >= K) v.w = 0.f;
  return v;
}

// m / d for 0 <= m < 2^24 without the ~40-instruction integer division (float estimate + one fix-up each way)
__device__ __forceinline__ int fast_div(int m, int d) {
  int q = (int)((float)m * (1.0f / (float)d));
  q -= (q * d > m) ? 1 : 0;
  q += ((q + 1) * d <= m) ? 1 : 0;
  return q;
}

// v[q] += val[off+q] where arg[off+q] == m   (q < n valid elements): the sparse gradient behind a global max-pool
__device__ __forceinline__ void sparse_add4(float4& v, const float* __restrict__ val, const int32_t* __restrict__ arg, size_t off, int m, int n) {
  if (n > 0 && arg[off] == m) v.x += val[off];
  if (n > 1 && arg[off + 1] == m) v.y += val[off + 1];
  if (n > 2 && arg[off + 2] == m) v.z += val[off + 2];
  if (n > 3 && arg[off + 3] == m) v.w += val[off + 3];
}

// FAST (template flag below): every operand pointer is 16-byte aligned, leading dimensions and K are multiples of 4 ->
// unconditional float4 loads (no divergent scalar tail path; the loads of a k-tile issue back to back).
// 4 consecutive k-values as fp16 (round to nearest) into 2 LDS words
__device__ __forceinline__ void st_row4h(float* p, float4 v) {
  f32x4v f = {v.x, v.y, v.z, v.w};
  const f16x4 h = __builtin_convertvector(f, f16x4);
  *reinterpret_cast<f16x4*>(p) = h;
}

// F16 = 2 ("bf16x3"): every fp32 operand value v is split EXACTLY into three bfloat16 terms v = hi + mid + lo (8 significand bits
// each: 24 together = fp32's significand; each residual is exactly representable in fp32, so the subtractions are exact) and the
// product a*b is evaluated as the six leading cross terms on the bf16 matrix pipe (16x the fp32 MFMA rate), accumulated in fp32:
//   a*b ~ ah*bh + (ah*bm + am*bh) + (ah*bl + al*bh + am*bm),  dropped terms <= 3 * 2^-24 |a*b|  (one fp32 ulp is 2^-23 |a*b|)
// -- fp32-equivalent products (each bf16 x bf16 partial product is exact in fp32) at 6/16 of the fp32-MFMA time.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int LDX3 = 52;  // LDS row: three planes of 32 bf16 (16 words each) + 4 words of padding; 52 = 4*13 keeps the 16-byte
                          // fragment reads of 16 consecutive rows on disjoint 4-bank groups (conflict-free ds_read_b128)

__device__ __forceinline__ void st_row4b3(float* p, float4 v) {  // p: word address of this k-quad inside the hi plane
  f32x4v f = {v.x, v.y, v.z, v.w};
  const bf16x4 hi = __builtin_convertvector(f, bf16x4);
  const f32x4v r1 = f - __builtin_convertvector(hi, f32x4v);
  const bf16x4 mid = __builtin_convertvector(r1, bf16x4);
  const f32x4v r2 = r1 - __builtin_convertvector(mid, f32x4v);
  const bf16x4 lo = __builtin_convertvector(r2, bf16x4);
  *reinterpret_cast<bf16x4*>(p) = hi;
  *reinterpret_cast<bf16x4*>(p + 16) = mid;
  *reinterpret_cast<bf16x4*>(p + 32) = lo;
}

__device__ __forceinline__ void st_row4(float* p, float4 v) {  // rows are 8-byte aligned (LDT even)
  *reinterpret_cast<float2*>(p) = make_float2(v.x, v.y);
  *reinterpret_cast<float2*>(p + 2) = make_float2(v.z, v.w);
}

// Geometry of one workgroup: WGM x WGN waves, each owning TI x TJ MFMA tiles of 32x32.
template <int CFG> struct Geo;
template <> struct Geo<0> { static constexpr int WGM = 2, WGN = 2, TI = 2, TJ = 2; };  // BN = 128
template <> struct Geo<1> { static constexpr int WGM = 2, WGN = 2, TI = 2, TJ = 1; };  // BN = 64
template <> struct Geo<2> { static constexpr int WGM = 4, WGN = 1, TI = 1, TJ = 1; };  // BN = 32

// Sum per-lane column partials over the two row halves of a wave (lanes l, l^32) and over the
// WGM M-waves of the workgroup.  `red` is [WGM][BN] floats of LDS.
template <int CFG>
__device__ __forceinline__ void col_reduce(float (&part)[Geo<CFG>::TJ], float* red, int wm, int wn, int lane) {
  using G = Geo<CFG>;
  constexpr int BN = G::WGN * G::TJ * 32;
#pragma unroll
  for (int j = 0; j < G::TJ; ++j) {
    float v = part[j];
    v += __shfl_xor(v, 32);
    if (lane < 32) red[wm * BN + (wn * G::TJ + j) * 32 + lane] = v;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < G::TJ; ++j) {
    const int c = (wn * G::TJ + j) * 32 + (lane & 31);
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < G::WGM; ++w) s += red[w * BN + c];
    part[j] = s;
  }
  __syncthreads();
}

// Two sets of column partials in ONE exchange (one barrier, none behind it: for the end of a kernel, `buf` = [2][WGM][BN] floats
// that nobody touches afterwards).
template <int CFG>
__device__ __forceinline__ void col_reduce2(float (&pa)[Geo<CFG>::TJ], float (&pb)[Geo<CFG>::TJ], float* buf, int wm, int wn, int lane) {
  using G = Geo<CFG>;
  constexpr int BN = G::WGN * G::TJ * 32;
#pragma unroll
  for (int j = 0; j < G::TJ; ++j) {
    const float va = pa[j] + __shfl_xor(pa[j], 32), vb = pb[j] + __shfl_xor(pb[j], 32);
    if (lane < 32) {
      buf[wm * BN + (wn * G::TJ + j) * 32 + lane] = va;
      buf[(G::WGM + wm) * BN + (wn * G::TJ + j) * 32 + lane] = vb;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < G::TJ; ++j) {
    const int c = (wn * G::TJ + j) * 32 + (lane & 31);
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int w = 0; w < G::WGM; ++w) {
      sa += buf[w * BN + c];
      sb += buf[(G::WGM + w) * BN + c];
    }
    pa[j] = sa;
    pb[j] = sb;
  }
}

// F16 = 1: the operands are rounded to fp16 (round-to-nearest) when they are staged into LDS and multiplied with
// v_mfma_f32_32x32x8_f16 (fp32 accumulate): BASELINE configs[4] "fp16 MFMA MLPs".  Global loads, prologues and epilogues stay
// fp32; an LDS row holds the 32 k-values of the tile as 16 words + 2 words of padding (stride 18 == 2 mod 16: the 32 rows x
// 8-byte fragment reads of a half-wave are conflict-free), a fragment read (4 halfs) feeds ONE MFMA of k = 8.
// AH = 1 (F16 = 1, straight-line loads only; spgan_gemm_nt_args.a_half): 16-bit operand storage.  Plain A: the operand lies in memory
// as fp16 already (the EdgeBlock's T, written by spgan_edge_attend_fwd_h): a staging slot is one 8-byte load that goes to LDS as it is.
// Two-tensor A (the EdgeBlock's lazy BatchNorm-backward operand p*g + q*y + r): g is stored as bfloat16 (spgan_edge_attend_bwd_b), y as
// fp16 (h2pre: spgan_gemm_nt_args.y_half); both 8-byte loads, converted where the operand is evaluated.
template <int AMODE, int EPI, int CFG, int DB, int FAST, int F16 = 0, int AH = 0>
__global__ __launch_bounds__(256, 3) void gemm_nt_kernel(const spgan_gemm_nt_args p_) {  // <= 168 VGPRs: 3 waves/SIMD
  // The argument block stays in the kernarg segment (scalar loads): it is never copied or modified -- a modified copy of a
  // struct this size lands in scratch.  Only the three operand pointers of a batched product are adjusted, as locals.
  const spgan_gemm_nt_args& p = p_;
  const float* pA = p_.A;
  const float* pW = p_.W;
  float* pY = p_.Y;
  if constexpr (AMODE == SPGAN_A_PLAIN && EPI == SPGAN_EPI_LINEAR) {
    if (blockIdx.y != 0) {  // batched product: blockIdx.y selects the (A, W, Y) triple (scalar pointer arithmetic)
      pA += (size_t)blockIdx.y * (size_t)p.batch_stride_a;
      pW += (size_t)blockIdx.y * (size_t)p.batch_stride_w;
      pY += (size_t)blockIdx.y * (size_t)p.batch_stride_y;
    }
  }
  using G = Geo<CFG>;
  constexpr int TI = G::TI, TJ = G::TJ;
  constexpr int BN = G::WGN * TJ * 32;
  constexpr int NB = DB + 1;
  constexpr int BSLOT = BN / 32;  // float4 staging slots per thread for the weight tile
  constexpr int LDX = F16 == 2 ? LDX3 : (F16 ? 18 : LDT);   // LDS row stride in 4-byte words
  constexpr int KK = F16 == 2 ? BK / 16 : (F16 ? BK / 8 : BK / 4);  // fragment reads per k-tile
  static_assert(!(F16 && AMODE == A_AFFINE_SPARSE), "the LDS patch path of the sparse addend is fp32 only");
  static_assert(!AH || (F16 == 1 && FAST && (AMODE == SPGAN_A_PLAIN || AMODE == A_AFFINE2)), "16-bit stored A: plain / two-tensor operand, fp16 MFMA, aligned problems");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                  // [NB][BM*LDX]
  float* Bs = smem + NB * BM * LDX;  // [NB][BN*LDX]
  float* red = Bs + NB * BN * LDX;   // [WGM][BN]
  float* pool = red + G::WGM * BN;   // [4][WGM][BN]: column max / arg-max / min / arg-min exchange of the pooling epilogue

  const int tilesN = (p.N + BN - 1) / BN;
  const int tilesM = (p.M + BM - 1) / BM;
  const Tile t = map_tile(tilesN);
  if (t.tm >= tilesM) return;
  const int m0 = t.tm * BM, n0 = t.tn * BN;
  // prologue vectors: one pair for all rows, or one pair per group of p_group_rows rows (a tile never straddles two groups)
  const float* pPsc = AMODE != SPGAN_A_PLAIN ? p_.p_scale : nullptr;
  const float* pPsh = AMODE != SPGAN_A_PLAIN ? p_.p_shift : nullptr;
  if (AMODE != SPGAN_A_PLAIN && p.p_group_rows > 0) {
    // a handful of groups: walked with scalar subtractions (a division would cost vector registers right where the kernel is tightest)
    for (int r = m0 - p.p_group_rows; r >= 0; r -= p.p_group_rows) {
      pPsc += p.K;
      pPsh += p.K;
    }
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / G::WGN, wn = wave % G::WGN;
  const int l31 = lane & 31, lh = lane >> 5;

  const bool vecA = ((p.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(pA) & 15) == 0);
  const bool vecW = ((p.ldw & 3) == 0) && ((reinterpret_cast<uintptr_t>(pW) & 15) == 0);

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Staging registers hold the RAW loads of the tile in flight; every prologue transform (affine + LeakyReLU, the edge
  // difference, the sparse addend) is applied in sstore, i.e. after this tile's MFMAs were issued.  Transforming at load
  // time would put a vmcnt wait in front of the MFMAs and expose the whole HBM/L2 latency once per k-tile.
  float4 ra[4], rb[BSLOT];
  float4 ra2[(AMODE == SPGAN_A_EDGE || AMODE == A_AFFINE2) ? 4 : 1];  // EDGE: the centre rows; AFFINE2: the rows of the second tensor
  float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = make_float4(0.f, 0.f, 0.f, 0.f), peb = psh;
  const int lrow = tid >> 3, lc4 = (tid & 7) * 4;  // staging slot: row (tid/8 + 32*i), k offset 4*(tid%8)
  // Element offset of the operand row (EDGE: neighbour j) and centre row i of each staging slot, 32-bit (checked on the
  // host: M*lda, N*ldw < 2^32) so the loads use the scalar-base + 32-bit-offset addressing form.  Rows >= M are clamped
  // to M-1: what they stage only reaches accumulator rows >= M, which no epilogue reads -- so the aligned path loads
  // unconditionally.  Likewise weight rows >= N only feed output columns >= N.
  unsigned offA[4], offC[(AMODE == SPGAN_A_EDGE || AMODE == A_AFFINE2) ? 4 : 1], offW[BSLOT];
  const float* pA2 = AMODE == A_AFFINE2 ? p_.A2 : pA;   // second operand tensor (EDGE: the same tensor, centre rows)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = min(m0 + lrow + 32 * i, p.M - 1);
    offA[i] = (unsigned)((AMODE == SPGAN_A_EDGE) ? p.e_idx[m] : m) * (unsigned)p.lda;
    if (AMODE == SPGAN_A_EDGE) offC[i] = (unsigned)fast_div(m, p.e_k) * (unsigned)p.lda;
    if (AMODE == A_AFFINE2) offC[i] = (unsigned)m * (unsigned)p.lda2;
  }
#pragma unroll
  for (int i = 0; i < BSLOT; ++i) offW[i] = (unsigned)min(n0 + lrow + 32 * i, p.N - 1) * (unsigned)p.ldw;

  // Sparse addend of the A operand (sp_val/sp_arg, one hit per (shape, column)).  Fast form: when every row of this
  // tile lies in ONE shape and the loads are 16-byte aligned, each thread fetches the (arg, val) quads of its 4 staging
  // columns together with the operand loads and patches its own 4x4 values before the LDS store.  Otherwise the staged
  // LDS tile is patched (sfix below).
  constexpr bool sparse = AMODE == A_AFFINE_SPARSE;
  const int sp_b = sparse ? fast_div(m0, p.sp_rows) : 0;
  const bool sp_reg = sparse && FAST && (fast_div(min(m0 + BM, p.M) - 1, p.sp_rows) == sp_b);
  int4 spa = make_int4(-1, -1, -1, -1);
  float4 spv = make_float4(0.f, 0.f, 0.f, 0.f);

  auto ldrow = [&](const float* base, unsigned off, int k, bool vec) -> float4 {
    if (FAST) return *reinterpret_cast<const float4*>(base + (off + (unsigned)k));
    return ld4(base + (off + (unsigned)k), vec, k, p.K);
  };
  auto ldpar = [&](const float* q, int k) -> float4 {
    if (FAST) return *reinterpret_cast<const float4*>(q + k);
    return ld4(q + k, false, k, p.K);
  };
  auto gload = [&](int k0) {
    const int k = k0 + lc4;
    const bool kok = k < p.K;
    if (FAST) {  // straight-line: every load of the tile issues back to back; k >= K slots are zeroed in sstore
      const int kc = kok ? k : 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (AH) {   // 4 values of 16 bits: fp16 (plain A) or bfloat16 (two-tensor A), kept raw until sstore
          const float2 h = *reinterpret_cast<const float2*>(reinterpret_cast<const uint16_t*>(pA) + (offA[i] + (unsigned)kc));
          ra[i].x = h.x;
          ra[i].y = h.y;
        } else
        ra[i] = ldrow(pA, offA[i], kc, true);
        if (AH && AMODE == A_AFFINE2) {
          const float2 h = *reinterpret_cast<const float2*>(reinterpret_cast<const uint16_t*>(pA2) + (offC[i] + (unsigned)kc));
          ra2[i].x = h.x;
          ra2[i].y = h.y;
        } else
        if (AMODE == SPGAN_A_EDGE || AMODE == A_AFFINE2) ra2[i] = ldrow(pA2, offC[i], kc, true);
      }
      if (AMODE != SPGAN_A_PLAIN) {
        psc = ldpar(pPsc, kc);
        psh = ldpar(pPsh, kc);
        if (AMODE == SPGAN_A_EDGE) peb = ldpar(p.e_bias, kc);
        if (AMODE == A_AFFINE2) peb = ldpar(p.p_scale2, kc);
      }
      if (sp_reg) {
        const size_t off = (size_t)sp_b * p.K + kc;
        spa = *reinterpret_cast<const int4*>(p.sp_arg + off);
        spv = *reinterpret_cast<const float4*>(p.sp_val + off);
      }
#pragma unroll
      for (int i = 0; i < BSLOT; ++i) rb[i] = ldrow(pW, offW[i], kc, true);
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (AMODE == SPGAN_A_EDGE || AMODE == A_AFFINE2) ra2[i] = ra[i];
      if (kok) {
        ra[i] = ldrow(pA, offA[i], k, vecA);
        if (AMODE == SPGAN_A_EDGE) ra2[i] = ldrow(pA, offC[i], k, vecA);
        if (AMODE == A_AFFINE2) ra2[i] = ldrow(pA2, offC[i], k, ((p.lda2 & 3) == 0) && ((reinterpret_cast<uintptr_t>(pA2) & 15) == 0));
      }
    }
    if (AMODE != SPGAN_A_PLAIN && kok) {
      psc = ldpar(pPsc, k);
      psh = ldpar(pPsh, k);
      if (AMODE == SPGAN_A_EDGE) peb = ldpar(p.e_bias, k);
      if (AMODE == A_AFFINE2) peb = ldpar(p.p_scale2, k);
    }
#pragma unroll
    for (int i = 0; i < BSLOT; ++i) rb[i] = kok ? ldrow(pW, offW[i], k, vecW) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto sstore = [&](int buf, int k0) {
    float* a = As + buf * BM * LDX;
    float* b = Bs + buf * BN * LDX;
    const int k = k0 + lc4;
    const bool kok = k < p.K;
    if (AMODE != SPGAN_A_PLAIN) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float4 v = ra[i];
        if (AMODE == SPGAN_A_EDGE) {
          v.x = (v.x - ra2[i].x) + peb.x;
          v.y = (v.y - ra2[i].y) + peb.y;
          v.z = (v.z - ra2[i].z) + peb.z;
          v.w = (v.w - ra2[i].w) + peb.w;
        }
        if (AH && AMODE == A_AFFINE2) {   // g: 4 bfloat16, y: 4 fp16
          const float2 rg = make_float2(ra[i].x, ra[i].y), ry = make_float2(ra2[i].x, ra2[i].y);
          const bf16x4 gb = *reinterpret_cast<const bf16x4*>(&rg);
          const f16x4 yh = *reinterpret_cast<const f16x4*>(&ry);
          v = make_float4((float)gb[0], (float)gb[1], (float)gb[2], (float)gb[3]);
          ra2[i] = make_float4((float)yh[0], (float)yh[1], (float)yh[2], (float)yh[3]);
        }
        if (AMODE == A_AFFINE2) {  // p*g + (q*y + r): the same expression as bn_bwd_coef_apply in tests/kernel_model.py
          v.x = fmaf(v.x, psc.x, fmaf(ra2[i].x, peb.x, psh.x));
          v.y = fmaf(v.y, psc.y, fmaf(ra2[i].y, peb.y, psh.y));
          v.z = fmaf(v.z, psc.z, fmaf(ra2[i].z, peb.z, psh.z));
          v.w = fmaf(v.w, psc.w, fmaf(ra2[i].w, peb.w, psh.w));
        } else
        v = affine_lrelu4(v, psc, psh, p.p_slope);
        if (!FAST) v = mask_tail(v, k, p.K);
        ra[i] = v;
      }
    }
    if (sp_reg) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + lrow + 32 * i;
        ra[i].x += (spa.x == m) ? spv.x : 0.f;
        ra[i].y += (spa.y == m) ? spv.y : 0.f;
        ra[i].z += (spa.z == m) ? spv.z : 0.f;
        ra[i].w += (spa.w == m) ? spv.w : 0.f;
      }
    }
    if (FAST && !kok) {  // K % 32 != 0: the tail slots of the last k-tile
#pragma unroll
      for (int i = 0; i < 4; ++i) ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < BSLOT; ++i) rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (F16 == 2) {
      // sign checkerboard (see csrc/gemm_wide3.hip: the bf16 MFMA's accumulation is biased toward -infinity; staging the rows of odd 32-row
      // tiles negated makes the bias alternate in sign from accumulator tile to accumulator tile, so that it cancels in sums over outputs)
      auto neg4 = [](float4 v) { return make_float4(-v.x, -v.y, -v.z, -v.w); };
#pragma unroll
      for (int i = 0; i < 4; ++i) st_row4b3(&a[(lrow + 32 * i) * LDX + (lc4 >> 1)], (i & 1) ? neg4(ra[i]) : ra[i]);
#pragma unroll
      for (int i = 0; i < BSLOT; ++i) st_row4b3(&b[(lrow + 32 * i) * LDX + (lc4 >> 1)], (i & 1) ? neg4(rb[i]) : rb[i]);
    } else if (F16) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (AH && AMODE == SPGAN_A_PLAIN) *reinterpret_cast<float2*>(&a[(lrow + 32 * i) * LDX + (lc4 >> 1)]) = make_float2(ra[i].x, ra[i].y);
        else st_row4h(&a[(lrow + 32 * i) * LDX + (lc4 >> 1)], ra[i]);
      }
#pragma unroll
      for (int i = 0; i < BSLOT; ++i) st_row4h(&b[(lrow + 32 * i) * LDX + (lc4 >> 1)], rb[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) st_row4(&a[(lrow + 32 * i) * LDX + lc4], ra[i]);
#pragma unroll
      for (int i = 0; i < BSLOT; ++i) st_row4(&b[(lrow + 32 * i) * LDX + lc4], rb[i]);
    }
  };
  auto compute = [&](int buf, int kk0, int kk1) {
    const float* a = As + buf * BM * LDX + (wm * TI * 32 + l31) * LDX + (F16 == 2 ? 4 : 2) * lh;
    const float* b = Bs + buf * BN * LDX + (wn * TJ * 32 + l31) * LDX + (F16 == 2 ? 4 : 2) * lh;
#pragma unroll
    for (int kk = kk0; kk < kk1; ++kk) {
      if (F16 == 2) {
        // k-step of 16: lane half lh holds k = 8*lh .. 8*lh+7 of each plane (16 bytes); six cross terms, small ones first
        bf16x8 ap[3][TI], bp[3][TJ];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
          for (int i = 0; i < TI; ++i) ap[q][i] = *reinterpret_cast<const bf16x8*>(a + i * 32 * LDX + q * 16 + kk * 8);
#pragma unroll
          for (int j = 0; j < TJ; ++j) bp[q][j] = *reinterpret_cast<const bf16x8*>(b + j * 32 * LDX + q * 16 + kk * 8);
        }
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) {
            f32x16 c = acc[i][j];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[2][i], bp[0][j], c, 0, 0, 0);  // lo * hi
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0][i], bp[2][j], c, 0, 0, 0);  // hi * lo
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1][i], bp[1][j], c, 0, 0, 0);  // mid * mid
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1][i], bp[0][j], c, 0, 0, 0);  // mid * hi
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0][i], bp[1][j], c, 0, 0, 0);  // hi * mid
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0][i], bp[0][j], c, 0, 0, 0);  // hi * hi
            acc[i][j] = c;
          }
      } else if (F16) {
        // two 8-byte fragment reads (k-groups kk and kk+1 of this lane half) feed ONE v_mfma_f32_32x32x16_f16: gfx950's k = 16 form
        // issues in the cycles the k = 8 one takes (any assignment of the tile's k values to MFMA k-slots is legal as long as both
        // operands use the same one); kk0/kk1 are even (KK = 4).
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
      } else {
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
    }
  };

  // Sparse addend of the A operand (sp_val/sp_arg): per column k at most one row of a shape carries it, so it is
  // patched into the staged LDS tile by BK threads per k-tile instead of being tested on every operand load.
  const bool sp_lds = sparse && !sp_reg;
  auto sfix = [&](int buf, int k0) {
    if (tid < BK && k0 + tid < p.K) {
      float* a = As + buf * BM * LDT;
      const int b_lo = fast_div(m0, p.sp_rows), b_hi = fast_div(min(m0 + BM, p.M) - 1, p.sp_rows);
      for (int b = b_lo; b <= b_hi; ++b) {
        const size_t off = (size_t)b * p.K + k0 + tid;
        const int r = p.sp_arg[off] - m0;
        if (r >= 0 && r < BM) a[r * LDT + tid] += p.sp_val[off];
      }
    }
  };

  const int nk = (p.K + BK - 1) / BK;
#ifdef SPGAN_TRACE
  unsigned long long* trc = (AMODE != SPGAN_A_EDGE && EPI == SPGAN_EPI_LINEAR && p.e_bias2) ? ((unsigned long long*)p.e_bias2) + (size_t)blockIdx.x * 8 : nullptr;
#define TRC(i) do { if (trc && threadIdx.x == 0) trc[i] = __builtin_amdgcn_s_memtime(); } while (0)
  TRC(0);
#else
#define TRC(i)
#endif
  gload(0);
  TRC(1);
  sstore(0, 0);
  __syncthreads();
  TRC(2);
  if (sp_lds) {
    sfix(0, 0);
    __syncthreads();
  }
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload((kt + 1) * BK);  // next tile's HBM/L2 loads fly under this tile's MFMAs
    if (DB) {
      // the next tile's LDS stores go between the two halves of this tile's MFMAs (other buffer: its last readers passed the
      // previous barrier): they overlap with the second half instead of sitting between the last MFMA and the barrier
      compute(kt & 1, 0, KK / 2);
      if (kt + 1 < nk) sstore((kt + 1) & 1, (kt + 1) * BK);
      compute(kt & 1, KK / 2, KK);
      __syncthreads();
      if (sp_lds && kt + 1 < nk) {
        sfix((kt + 1) & 1, (kt + 1) * BK);
        __syncthreads();
      }
    } else {
      compute(0, 0, KK);
      __syncthreads();
      if (kt + 1 < nk) {
        sstore(0, (kt + 1) * BK);
        __syncthreads();
        if (sp_lds) {
          sfix(0, (kt + 1) * BK);
          __syncthreads();
        }
      }
    }
  }

  TRC(3);
  if (F16 == 2) {   // undo the sign checkerboard: accumulator tile (i, j) of this wave holds (-1)^(row tile + column tile) times its block
    const uint32_t ws = (uint32_t)((wm * TI + wn * TJ) & 1) << 31;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const uint32_t sg = ws ^ (((i + j) & 1) ? 0x80000000u : 0u);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = __uint_as_float(__float_as_uint(acc[i][j][r]) ^ sg);
      }
  }
  // ---------------------------------------------------------------- epilogue
  // C/D layout of v_mfma_f32_32x32x2_f32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  const int rbase = m0 + wm * TI * 32 + 4 * lh;
  const int cbase = n0 + wn * TJ * 32 + l31;
  const int rows_valid = min(BM, p.M - m0);
#define ROW_OF(i, r) (rbase + (i) * 32 + ((r) & 3) + 8 * ((r) >> 2))

  // A workgroup whose tile lies completely inside the output (the common case: M % 128 == 0, N % BN == 0) runs straight-line
  // epilogues: no per-element bounds tests, the uniform switches (activation, optional operands) hoisted out of the element loops
  // and the addresses as one per-lane base plus offsets that are uniform over the wave.  Measured with s_memtime: the generic
  // per-element code below took 25-33 % of a workgroup's lifetime on the K <= 256 layers.
  const bool full = (rows_valid == BM) && (n0 + BN <= p.N);
#define ROFF(r) (((r) & 3) + 8 * ((r) >> 2))

  if (EPI == SPGAN_EPI_LINEAR) {
    float csum[TJ];
    // per-group bias rows: usable on the straight-line path when the whole tile lies in one group (per-shape bias, N % 128 == 0) or
    // when the "group" is a single row (a dense [M,N] addend)
    const int rb_g0 = p.rowbias ? fast_div(m0, p.rows_per_group) : 0;
    const bool rb_uniform = p.rowbias && p.rows_per_group > 1 && fast_div(m0 + BM - 1, p.rows_per_group) == rb_g0;
    const bool rb_dense = p.rowbias && p.rows_per_group == 1;
    if (full && (!p.rowbias || rb_uniform || rb_dense)) {
      const float* ab = rb_dense ? p.rowbias + (size_t)rbase * p.ld_rowbias + cbase : nullptr;
      const unsigned lda2 = (unsigned)p.ld_rowbias;
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        float b = p.bias ? p.bias[cbase + j * 32] : 0.f;
        if (rb_uniform) b += p.rowbias[(size_t)rb_g0 * p.ld_rowbias + cbase + j * 32];
        csum[j] = 0.f;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          if (ab) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += ab[(size_t)((unsigned)(i * 32 + ROFF(r)) * lda2 + (unsigned)(j * 32))];
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[i][j][r] + b;
            acc[i][j][r] = v;  // keep the pre-activation value for the statistics / pooling passes
            csum[j] += v;
          }
        }
      }
      if (F16 == 1 && pY && p.y_bf16) {   // 16-bit result storage (no activation: checked on the host): bfloat16 for the EdgeBlock's dT (a gradient)
        __bf16* yb = reinterpret_cast<__bf16*>(pY) + (size_t)rbase * p.ldy + cbase;
        const unsigned ldy = (unsigned)p.ldy;
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
          for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) yb[(size_t)((unsigned)(i * 32 + ROFF(r)) * ldy + (unsigned)(j * 32))] = (__bf16)acc[i][j][r];
      } else if (F16 == 1 && pY && p.y_half) {   // ... fp16 for its h2pre (a pre-BatchNorm activation)
        _Float16* yb = reinterpret_cast<_Float16*>(pY) + (size_t)rbase * p.ldy + cbase;
        const unsigned ldy = (unsigned)p.ldy;
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
          for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) yb[(size_t)((unsigned)(i * 32 + ROFF(r)) * ldy + (unsigned)(j * 32))] = (_Float16)acc[i][j][r];
      } else if (pY) {
        float* yb = pY + (size_t)rbase * p.ldy + cbase;
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
    } else {
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int col = cbase + j * 32;
      const bool cok = col < p.N;
      const float b = (cok && p.bias) ? p.bias[col] : 0.f;
      csum[j] = 0.f;
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = ROW_OF(i, r);
          float v = acc[i][j][r] + b;
          if (p.rowbias && cok && row < p.M) v += p.rowbias[(size_t)fast_div(row, p.rows_per_group) * p.ld_rowbias + col];
          acc[i][j][r] = v;  // keep the pre-activation value for the statistics pass
          if (row < p.M) csum[j] += v;
          if (cok && row < p.M && pY) {
            float o = v;
            if (p.act == SPGAN_ACT_LRELU) o = lrelu_f(v, p.act_slope);
            else if (p.act == SPGAN_ACT_TANH) o = tanhf(v);
            if (F16 == 1 && p.y_bf16) reinterpret_cast<__bf16*>(pY)[(size_t)row * p.ldy + col] = (__bf16)o;
            else if (F16 == 1 && p.y_half) reinterpret_cast<_Float16*>(pY)[(size_t)row * p.ldy + col] = (_Float16)o;
            else
            pY[(size_t)row * p.ldy + col] = o;
          }
        }
    }
    }
    if (full && (p.stats || p.pool_val)) {
      // One LDS exchange for everything (the operand tiles are dead: every wave passed the k-loop's last barrier).  Statistics:
      // every lane holds NL = 16*TI rows of a column; (sum, M2 about the lane's own mean) pairs are merged with Chan's formula --
      // lane halves by shuffle, the WGM waves through LDS in ascending order -- which gives the tile's (sum, centred M2) without
      // a second pass that needs the tile mean first (that cost two more exchanges).
      constexpr int WGM = G::WGM;
      constexpr float NL = (float)(16 * TI);
      float* xs = As;
      float* xm = As + WGM * BN;
      float* xvx = As + 2 * WGM * BN;
      int* xax = reinterpret_cast<int*>(As + 3 * WGM * BN);
      float* xvn = As + 4 * WGM * BN;
      int* xan = reinterpret_cast<int*>(As + 5 * WGM * BN);
      const bool do_pool = p.pool_val != nullptr;
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const float sl = csum[j], mean = sl * (1.f / NL);
        float m2 = 0.f;
        float vx = -INFINITY, vn = INFINITY;
        int ax = 0x7fffffff, an = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[i][j][r];
            const float d = v - mean;
            m2 = fmaf(d, d, m2);
            if (do_pool) {  // rows ascend with (i, r): strict compares keep the first
              const int row = ROW_OF(i, r);
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
          const int c = wm * BN + (wn * TJ + j) * 32 + l31;
          xs[c] = S; xm[c] = M2;
          if (do_pool) { xvx[c] = vx; xax[c] = ax; xvn[c] = vn; xan[c] = an; }
        }
      }
      __syncthreads();
      if (wm == 0 && lh == 0) {
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          const int c = (wn * TJ + j) * 32 + l31, col = cbase + j * 32;
          float S = xs[c], M2 = xm[c], n = 2.f * NL;
#pragma unroll
          for (int w = 1; w < WGM; ++w) {
            const float Sb = xs[w * BN + c], nb = 2.f * NL;
            const float dl = Sb / nb - S / n;
            M2 = (M2 + xm[w * BN + c]) + dl * dl * (n * nb / (n + nb));
            S += Sb;
            n += nb;
          }
          if (p.stats) {
            float* o = p.stats + ((size_t)t.tm * p.N + col) * 2;
            o[0] = S;
            o[1] = M2;
          }
          if (do_pool) {
            float vx = xvx[c], vn = xvn[c];
            int ax = xax[c], an = xan[c];
#pragma unroll
            for (int w = 1; w < WGM; ++w) {  // wave w holds higher rows than wave w-1: strict compares keep the first
              if (xvx[w * BN + c] > vx) { vx = xvx[w * BN + c]; ax = xax[w * BN + c]; }
              if (xvn[w * BN + c] < vn) { vn = xvn[w * BN + c]; an = xan[w * BN + c]; }
            }
            const size_t o = ((size_t)t.tm * p.N + col) * 2;
            p.pool_val[o] = vx; p.pool_val[o + 1] = vn;
            p.pool_arg[o] = ax; p.pool_arg[o + 1] = an;
          }
        }
      }
    } else {
    if (p.pool_val) {
      // Per-tile column max / min of the pre-activation output with their rows (first row on ties): a global max-pool behind a
      // per-channel monotone map (BatchNorm affine of either sign + LeakyReLU) is finished from these by spgan_pool_finalize
      // once the batch statistics are known -- the [M,N] output itself need not be stored (Y == NULL).
      constexpr int WGM = G::WGM;
      float* pvx = pool;
      int* pax = reinterpret_cast<int*>(pool + WGM * BN);
      float* pvn = pool + 2 * WGM * BN;
      int* pan = reinterpret_cast<int*>(pool + 3 * WGM * BN);
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        float vx = -INFINITY, vn = INFINITY;
        int ax = 0x7fffffff, an = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {  // rows ascend with (i, r): strict compares keep the first
            const int row = ROW_OF(i, r);
            const float v = acc[i][j][r];
            if ((full || row < p.M) && v > vx) { vx = v; ax = row; }
            if ((full || row < p.M) && v < vn) { vn = v; an = row; }
          }
        const float ovx = __shfl_xor(vx, 32), ovn = __shfl_xor(vn, 32);
        const int oax = __shfl_xor(ax, 32), oan = __shfl_xor(an, 32);
        if (ovx > vx || (ovx == vx && oax < ax)) { vx = ovx; ax = oax; }
        if (ovn < vn || (ovn == vn && oan < an)) { vn = ovn; an = oan; }
        if (lh == 0) {
          const int c = wm * BN + (wn * TJ + j) * 32 + l31;
          pvx[c] = vx; pax[c] = ax; pvn[c] = vn; pan[c] = an;
        }
      }
      __syncthreads();
      if (wm == 0 && lh == 0) {
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          const int c = (wn * TJ + j) * 32 + l31, col = cbase + j * 32;
          float vx = pvx[c], vn = pvn[c];
          int ax = pax[c], an = pan[c];
#pragma unroll
          for (int w = 1; w < WGM; ++w) {  // wave w holds higher rows than wave w-1: strict compares keep the first
            if (pvx[w * BN + c] > vx) { vx = pvx[w * BN + c]; ax = pax[w * BN + c]; }
            if (pvn[w * BN + c] < vn) { vn = pvn[w * BN + c]; an = pan[w * BN + c]; }
          }
          if (col < p.N) {
            const size_t o = ((size_t)t.tm * p.N + col) * 2;
            p.pool_val[o] = vx; p.pool_val[o + 1] = vn;
            p.pool_arg[o] = ax; p.pool_arg[o + 1] = an;
          }
        }
      }
      __syncthreads();
    }
    if (p.stats) {
      // per-tile (sum, centred M2): combined later with Chan's formula -> no E[x^2]-E[x]^2 cancellation
      col_reduce<CFG>(csum, red, wm, wn, lane);
      float m2[TJ];
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const float mean = csum[j] / (float)rows_valid;
        m2[j] = 0.f;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float d = acc[i][j][r] - mean;
            if (full || ROW_OF(i, r) < p.M) m2[j] = fmaf(d, d, m2[j]);
          }
      }
      col_reduce<CFG>(m2, red, wm, wn, lane);
      if (wm == 0 && lh == 0) {
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          const int col = cbase + j * 32;
          if (col < p.N) {
            float* o = p.stats + ((size_t)t.tm * p.N + col) * 2;
            o[0] = csum[j];
            o[1] = m2[j];
          }
        }
      }
    }
    }
  } else if (EPI == SPGAN_EPI_MASK_OUT) {
    if (full) {
      const float* rb = p.ref + (size_t)rbase * p.ld_ref + cbase;
      float* yb = pY + (size_t)rbase * p.ldy + cbase;
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
    } else
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int col = cbase + j * 32;
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = ROW_OF(i, r);
          if (col < p.N && row < p.M) {
            const float ref = p.ref[(size_t)row * p.ld_ref + col];
            pY[(size_t)row * p.ldy + col] = acc[i][j][r] * lrelu_mask(ref, p.b_slope);
          }
        }
    }
  } else {  // BNBWD / EDGE_BNBWD
    float s0[TJ], s1[TJ];
    if (EPI == SPGAN_EPI_BNBWD && full && (!p.rowbias || p.rows_per_group == 1)) {
      const float* rb = p.ref + (size_t)rbase * p.ld_ref + cbase;
      float* yb = pY + (size_t)rbase * p.ldy + cbase;
      const unsigned ldr = (unsigned)p.ld_ref, ldy = (unsigned)p.ldy;
      const float sl = p.b_slope;
      // dense [M,N] addend (rows_per_group == 1: the S.W rows of the collapsed 256->1024 backward): one more straight-line load
      const float* ab = p.rowbias ? p.rowbias + (size_t)rbase * p.ld_rowbias + cbase : nullptr;
      const unsigned lda2 = (unsigned)p.ld_rowbias;
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int col = cbase + j * 32;
        const float sc = p.b_scale[col], sh = p.b_shift[col], mu = p.b_mean[col], inv = p.b_invstd[col];
        const float bia = p.bias ? p.bias[col] : 0.f;
        s0[j] = 0.f;
        s1[j] = 0.f;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          float yv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) yv[r] = rb[(size_t)((unsigned)(i * 32 + ROFF(r)) * ldr + (unsigned)(j * 32))];
          if (ab) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += ab[(size_t)((unsigned)(i * 32 + ROFF(r)) * lda2 + (unsigned)(j * 32))];
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float y = yv[r];
            const float z = fmaf(y, sc, sh);
            const float g = (acc[i][j][r] + bia) * lrelu_mask(z, sl);
            const float xh = (y - mu) * inv;
            yb[(size_t)((unsigned)(i * 32 + ROFF(r)) * ldy + (unsigned)(j * 32))] = g;
            s0[j] += g;
            s1[j] = fmaf(g, xh, s1[j]);
          }
        }
      }
    } else if (EPI == SPGAN_EPI_EDGE_BNBWD && full && !p.rowbias) {
      // rows are edges e = (i, r): ref[e,n] = (P[idx[e],n] - P[i,n]) + e_bias2[n]; the two row offsets of an accumulator row are
      // uniform over the 32 lanes of a half-wave, the gathers of a 16-row block issue together
      float* yb = pY + (size_t)rbase * p.ldy + cbase;
      const unsigned ldr = (unsigned)p.ld_ref, ldy = (unsigned)p.ldy;
      const float sl = p.b_slope;
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int col = cbase + j * 32;
        const float sc = p.b_scale[col], sh = p.b_shift[col], mu = p.b_mean[col], inv = p.b_invstd[col], eb = p.e_bias2[col];
        const float bia = p.bias ? p.bias[col] : 0.f;
        const float* pc = p.ref + col;
        s0[j] = 0.f;
        s1[j] = 0.f;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          float yv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = rbase + i * 32 + ROFF(r);
            const unsigned oj = (unsigned)p.e_idx[row] * ldr, oi = (unsigned)fast_div(row, p.e_k) * ldr;
            yv[r] = (pc[oj] - pc[oi]) + eb;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float y = yv[r];
            const float z = fmaf(y, sc, sh);
            const float g = (acc[i][j][r] + bia) * lrelu_mask(z, sl);
            const float xh = (y - mu) * inv;
            yb[(size_t)((unsigned)(i * 32 + ROFF(r)) * ldy + (unsigned)(j * 32))] = g;
            s0[j] += g;
            s1[j] = fmaf(g, xh, s1[j]);
          }
        }
      }
    } else
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int col = cbase + j * 32;
      const bool cok = col < p.N;
      const float sc = cok ? p.b_scale[col] : 0.f, sh = cok ? p.b_shift[col] : 0.f;
      const float mu = cok ? p.b_mean[col] : 0.f, inv = cok ? p.b_invstd[col] : 0.f;
      const float eb = (EPI == SPGAN_EPI_EDGE_BNBWD && cok) ? p.e_bias2[col] : 0.f;
      const float bia = (cok && p.bias) ? p.bias[col] : 0.f;
      s0[j] = 0.f;
      s1[j] = 0.f;
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = ROW_OF(i, r);
          if (cok && row < p.M) {
            float y;
            if (EPI == SPGAN_EPI_EDGE_BNBWD) {
              const int pi = fast_div(row, p.e_k), pj = p.e_idx[row];
              y = (p.ref[(size_t)pj * p.ld_ref + col] - p.ref[(size_t)pi * p.ld_ref + col]) + eb;
            } else {
              y = p.ref[(size_t)row * p.ld_ref + col];
            }
            const float z = fmaf(y, sc, sh);
            float a = acc[i][j][r] + bia;
            if (p.rowbias) a += p.rowbias[(size_t)fast_div(row, p.rows_per_group) * p.ld_rowbias + col];
            const float g = a * lrelu_mask(z, p.b_slope);
            const float xh = (y - mu) * inv;
            pY[(size_t)row * p.ldy + col] = g;
            s0[j] += g;
            s1[j] = fmaf(g, xh, s1[j]);
          }
        }
    }
    if (p.stats) {
      col_reduce2<CFG>(s0, s1, As, wm, wn, lane);  // the operand tiles are dead: every wave passed the k-loop's last barrier
      if (wm == 0 && lh == 0) {
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          const int col = cbase + j * 32;
          if (col < p.N) {
            float* o = p.stats + ((size_t)t.tm * p.N + col) * 2;
            o[0] = s0[j];
            o[1] = s1[j];
          }
        }
      }
    }
  }
#undef ROW_OF
#undef ROFF
  TRC(4);
}

template <int CFG, int DB, int F16 = 0>
constexpr size_t nt_lds_bytes() {
  return (size_t)((DB + 1) * (BM + Geo<CFG>::WGN * Geo<CFG>::TJ * 32) * (F16 == 2 ? LDX3 : (F16 ? 18 : LDT)) + 5 * Geo<CFG>::WGM * Geo<CFG>::WGN * Geo<CFG>::TJ * 32) *
         sizeof(float);
}

template <int AMODE, int EPI, int CFG, int DB, int FAST, int F16 = 0, int AH = 0>
void launch_nt_cfg(const spgan_gemm_nt_args& a, hipStream_t s) {
  constexpr int BN = Geo<CFG>::WGN * Geo<CFG>::TJ * 32;
  constexpr size_t lds = nt_lds_bytes<CFG, DB, F16>();
  static LdsOptIn opt;  // > 64 KB of dynamic LDS: once per kernel and device
  if (lds > 64 * 1024) opt.ensure(reinterpret_cast<const void*>(&gemm_nt_kernel<AMODE, EPI, CFG, DB, FAST, F16, AH>), (int)lds);
  const int tm8 = cdiv(cdiv(a.M, BM), 8) * 8;
  const int batch = (AMODE == SPGAN_A_PLAIN && EPI == SPGAN_EPI_LINEAR && a.batch > 1) ? a.batch : 1;
  hipLaunchKernelGGL((gemm_nt_kernel<AMODE, EPI, CFG, DB, FAST, F16, AH>), dim3(tm8 * cdiv(a.N, BN), batch), dim3(256), lds, s, a);
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// ------------------------------------------------------------------------------------------ gemm_nt, M <= 64
// The per-shape linears (D's fc head, G's global_conv: M = batch size, Discriminator.py:83-95, Generator.py:119-126)
// have no use for 128-row MFMA tiles: one workgroup would walk all of K alone (87 us for 32x512x1024).  Here a
// workgroup owns SC = 4 output columns, its 256 threads split K (one float4 per thread and step), the rows go
// through the registers in chunks of SR = 16, and the 64 partial dot products of a thread are summed over the
// workgroup in a fixed order (a butterfly that halves the value count per step, then the 4 waves through LDS).
constexpr int SR = 16, SC = 4;

template <int AMODE, int EPI>
__global__ __launch_bounds__(256) void gemm_nt_small_kernel(const spgan_gemm_nt_args p) {
  __shared__ float red[4][SR * SC];
  __shared__ float colv[2][64][SC];  // per (row, column) values of the statistics epilogues (M <= 64)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * SC;
  // gridDim.y == 1 with more than SR rows: this workgroup walks all row chunks of its SC columns (column statistics: LINEAR + stats,
  // BNBWD); otherwise one SR-row chunk per workgroup.
  const bool whole = gridDim.y == 1;
  const int chunks = whole ? (p.M + SR - 1) / SR : 1;
  for (int ch = 0; ch < chunks; ++ch) {
    const int mc = (whole ? ch : (int)blockIdx.y) * SR;
    float v[SR * SC];
#pragma unroll
    for (int i = 0; i < SR * SC; ++i) v[i] = 0.f;
    for (int k = tid * 4; k < p.K; k += 1024) {
      float4 w[SC];
#pragma unroll
      for (int c = 0; c < SC; ++c)
        w[c] = (n0 + c < p.N) ? *reinterpret_cast<const float4*>(p.W + (size_t)(n0 + c) * p.ldw + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
      if (AMODE == SPGAN_A_AFFINE_LRELU) {
        sc = *reinterpret_cast<const float4*>(p.p_scale + k);
        sh = *reinterpret_cast<const float4*>(p.p_shift + k);
      }
      float4 av[SR];  // rows >= M are clamped (their sums are never stored): all loads issue together, no branches
#pragma unroll
      for (int r = 0; r < SR; ++r) av[r] = *reinterpret_cast<const float4*>(p.A + (size_t)min(mc + r, p.M - 1) * p.lda + k);
#pragma unroll
      for (int r = 0; r < SR; ++r) {
        float4 a = av[r];
        if (AMODE == SPGAN_A_AFFINE_LRELU) a = affine_lrelu4(a, sc, sh, p.p_slope);
#pragma unroll
        for (int c = 0; c < SC; ++c)
          v[r * SC + c] = fmaf(a.w, w[c].w, fmaf(a.z, w[c].z, fmaf(a.y, w[c].y, fmaf(a.x, w[c].x, v[r * SC + c]))));
      }
    }
    // wave total of value i ends up in lane i
#pragma unroll
    for (int half = SR * SC / 2; half >= 1; half >>= 1) {
      const bool up = (lane & half) != 0;
#pragma unroll
      for (int i = 0; i < half; ++i) {
        const float keep = up ? v[i + half] : v[i];
        const float send = up ? v[i] : v[i + half];
        v[i] = keep + __shfl_xor(send, half);
      }
    }
    __syncthreads();  // red is reused from chunk to chunk
    red[wave][lane] = v[0];
    __syncthreads();
    if (tid < SR * SC) {
      const float acc = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
      const int row = mc + tid / SC, col = n0 + tid % SC;
      float c0 = 0.f, c1 = 0.f;
      if (row < p.M && col < p.N) {
        float o;
        if (EPI == SPGAN_EPI_LINEAR) {
          o = acc + (p.bias ? p.bias[col] : 0.f);
          if (p.rowbias) o += p.rowbias[(size_t)(row / p.rows_per_group) * p.ld_rowbias + col];
          c0 = o;  // pre-activation value for the statistics
          if (p.act == SPGAN_ACT_LRELU) o = lrelu_f(o, p.act_slope);
          else if (p.act == SPGAN_ACT_TANH) o = tanhf(o);
        } else if (EPI == SPGAN_EPI_MASK_OUT) {
          o = acc * lrelu_mask(p.ref[(size_t)row * p.ld_ref + col], p.b_slope);
          c0 = o;  // column sums of the masked product (the bias gradient of the layer below), when statistics are asked for
        } else {  // BNBWD
          const float y = p.ref[(size_t)row * p.ld_ref + col];
          const float z = fmaf(y, p.b_scale[col], p.b_shift[col]);
          float a = acc + (p.bias ? p.bias[col] : 0.f);
          if (p.rowbias) a += p.rowbias[(size_t)(row / p.rows_per_group) * p.ld_rowbias + col];
          o = a * lrelu_mask(z, p.b_slope);
          c0 = o;
          c1 = o * ((y - p.b_mean[col]) * p.b_invstd[col]);
        }
        if (p.Y) p.Y[(size_t)row * p.ldy + col] = o;
      }
      if (whole && row < 64) {
        colv[0][row][tid % SC] = c0;
        colv[1][row][tid % SC] = c1;
      }
    }
  }
  if (whole && p.stats) {
    __syncthreads();
    if (tid < SC && n0 + tid < p.N) {  // one thread per column, rows in ascending order
      float s0 = 0.f, s1 = 0.f;
      for (int r = 0; r < p.M; ++r) s0 += colv[0][r][tid];
      if (EPI == SPGAN_EPI_LINEAR) {
        const float mean = s0 / (float)p.M;
        for (int r = 0; r < p.M; ++r) {
          const float d = colv[0][r][tid] - mean;
          s1 = fmaf(d, d, s1);
        }
      } else {
        for (int r = 0; r < p.M; ++r) s1 += colv[1][r][tid];
      }
      float* o = p.stats + (size_t)(n0 + tid) * 2;  // a single 128-row tile: partials [1, N, 2]
      o[0] = s0;
      o[1] = s1;
      if (p.tail.enabled) {  // this workgroup owns its columns entirely: finish them here (spgan_coltail)
        const spgan_coltail& f = p.tail;
        const int c = n0 + tid;
        if (f.mode != 0) {
          f.out0[c] = s0;
          f.out1[c] = s1;
        } else {
          const float mean = s0 / (float)p.M, var = s1 / (float)p.M;
          if (f.out0) f.out0[c] = mean;
          if (f.out1) f.out1[c] = var;
          if (f.scale) {
            if (f.rmean) {
              const float cnt = (float)p.M * (float)max(f.count_rep, 1);
              const float unb = cnt > 1.f ? var * (cnt / (cnt - 1.f)) : var;
              f.rmean[c] = (1.f - f.momentum) * f.rmean[c] + f.momentum * mean;
              f.rvar[c] = (1.f - f.momentum) * f.rvar[c] + f.momentum * unb;
            }
            const float inv = 1.0f / sqrtf(var + f.eps);
            const float ga = f.gamma ? f.gamma[c] : 1.f, be = f.beta ? f.beta[c] : 0.f;
            const float sc = ga * inv;
            f.scale[c] = sc;
            f.shift[c] = be - mean * sc;
            f.invstd[c] = inv;
            f.mean_out[c] = mean;
          }
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------ gemm_nt, K <= 4
// The products against the 3 coordinate columns (D.conv1, G's first conv, the EdgeBlocks' P|Q|R of the points: K = 3) have nothing for
// the matrix cores to do: 128 x 64 MFMA tiles ran them at 1.1 TB/s of the result stream.  A streaming kernel instead (fp32 FMAs in every
// operand mode, in the MFMA kernel's summation order: bit-identical results): a workgroup owns 128 rows (= one statistics record); a
// thread owns 4 fixed output columns (its 4 x K weights in registers) and walks the rows in steps of 256/(N/4): float4 stores,
// consecutive lanes on consecutive columns.  LINEAR (bias, per-group bias rows, activation; products that also want column statistics
// stay on the MFMA kernel: the epilogue's reduction tree is part of the BatchNorm bits, and a gradient-penalty golden sits on a LeakyReLU
// kink that a last-bit change of those statistics moves) and MASK_OUT epilogues.  (A butterfly kernel for the N <= 4 products -- the input gradients of those layers, G's last conv -- was tried:
// 14.7 -> 11.6 us per launch, but it sums K in another order than the MFMA kernel and moved a LeakyReLU kink of the gradient-penalty
// golden; not kept.)
template <int EPI>
__global__ __launch_bounds__(256) void gemm_nt_k4_kernel(const spgan_gemm_nt_args p, int N4) {
  __shared__ float4 As[BM];
  const int tid = threadIdx.x;
  const int m0 = blockIdx.x * BM;
  const int rows = min(BM, p.M - m0);
  if (tid < BM) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < rows) {
      const float* ar = p.A + (size_t)(m0 + tid) * p.lda;
      a.x = ar[0];
      if (p.K > 1) a.y = ar[1];
      if (p.K > 2) a.z = ar[2];
      if (p.K > 3) a.w = ar[3];
    }
    As[tid] = a;
  }
  const int c4 = tid % N4, r0 = tid / N4, rpp = 256 / N4;   // N4 = N/4 is a power of two in 2 .. 128
  const int col = c4 * 4;
  float4 w[4];
  float bias[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float* wr = p.W + (size_t)(col + q) * p.ldw;
    w[q] = make_float4(wr[0], p.K > 1 ? wr[1] : 0.f, p.K > 2 ? wr[2] : 0.f, p.K > 3 ? wr[3] : 0.f);
    bias[q] = (EPI == SPGAN_EPI_LINEAR && p.bias) ? p.bias[col + q] : 0.f;
  }
  __syncthreads();
  const bool rbias = EPI == SPGAN_EPI_LINEAR && p.rowbias != nullptr;   // per-group rows of a [groups, N] addend (the generator's per-shape latent part)
  auto value = [&](int r, int q) -> float {
    const float4 a = As[r];
    // k order 0, 2, 1, 3: the order the 128-row MFMA kernel's LDS layout feeds v_mfma_f32_32x32x2_f32 (lane half 0 holds k = 0, 1, lane
    // half 1 k = 2, 3 of a quad; the first MFMA takes the .x values = k 0 and 2) -- this kernel reproduces its results bit for bit
    float b = bias[q];
    if (rbias) b += p.rowbias[(size_t)fast_div(m0 + r, p.rows_per_group) * p.ld_rowbias + col + q];   // (bias + group row) first, like the MFMA epilogue
    return fmaf(a.w, w[q].w, fmaf(a.y, w[q].y, fmaf(a.z, w[q].z, a.x * w[q].x))) + b;
  };
  for (int r = r0; r < rows; r += rpp) {
    float v[4], o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = value(r, q);
    if (EPI == SPGAN_EPI_LINEAR) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        o[q] = p.act == SPGAN_ACT_LRELU ? lrelu_f(v[q], p.act_slope) : (p.act == SPGAN_ACT_TANH ? tanhf(v[q]) : v[q]);
      }
    } else {  // MASK_OUT
      const float4 rv = *reinterpret_cast<const float4*>(p.ref + (size_t)(m0 + r) * p.ld_ref + col);
      o[0] = v[0] * lrelu_mask(rv.x, p.b_slope); o[1] = v[1] * lrelu_mask(rv.y, p.b_slope);
      o[2] = v[2] * lrelu_mask(rv.z, p.b_slope); o[3] = v[3] * lrelu_mask(rv.w, p.b_slope);
    }
    if (p.Y) *reinterpret_cast<float4*>(p.Y + (size_t)(m0 + r) * p.ldy + col) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// which problems the streaming kernel takes (nothing on the host mirrors this: neither results nor layouts depend on it)
inline bool nt_skinny_on(const spgan_gemm_nt_args& a) {
  return a.tile_hint != 1 && a.M > 64 && a.batch <= 1 && !a.tail.enabled && !a.pool_val && !a.sp_val && !a.A2 && !a.a_half && !a.y_bf16 && !a.y_half;
}
inline bool nt_k4_ok(const spgan_gemm_nt_args& a) {
  if (!nt_skinny_on(a) || a.K > 4 || a.a_mode != SPGAN_A_PLAIN || a.N % 4 || a.N < 8 || a.N > 512) return false;
  const int n4 = a.N / 4;
  if (n4 & (n4 - 1)) return false;
  if (a.ldy % 4 || !al16(a.Y)) return false;
  if (a.stats) return false;   // column statistics stay with the MFMA epilogue: its reduction tree is part of the BatchNorm bits
  // per-group bias rows: only where the MFMA kernel would take its whole-tile path (bias + group row added first: same bits)
  if (a.rowbias && (a.rows_per_group <= 0 || a.rows_per_group % BM || a.M % BM || a.N % 64)) return false;
  if (a.epi_mode == SPGAN_EPI_MASK_OUT) return !a.bias && !a.rowbias && a.ld_ref % 4 == 0 && al16(a.ref);
  return a.epi_mode == SPGAN_EPI_LINEAR;
}

template <int AMODE, int EPI>
int launch_nt(const spgan_gemm_nt_args& a, hipStream_t s) {
  bool fast = (a.K % 4 == 0) && (a.lda % 4 == 0) && (a.ldw % 4 == 0) && al16(a.A) && al16(a.W);
  if (AMODE != SPGAN_A_PLAIN) fast = fast && al16(a.p_scale) && al16(a.p_shift);
  if (AMODE == SPGAN_A_EDGE) fast = fast && al16(a.e_bias);
  if (AMODE == A_AFFINE_SPARSE) fast = fast && al16(a.sp_val) && al16(a.sp_arg);
  if (AMODE == A_AFFINE2) fast = fast && al16(a.A2) && (a.lda2 % 4 == 0) && al16(a.p_scale2);
  if constexpr (AMODE == SPGAN_A_PLAIN && (EPI == SPGAN_EPI_LINEAR || EPI == SPGAN_EPI_MASK_OUT)) {
    if (nt_k4_ok(a)) {  // 3 coordinate columns in: the streaming kernel
      hipLaunchKernelGGL((gemm_nt_k4_kernel<EPI>), dim3(cdiv(a.M, BM)), dim3(256), 0, s, a, a.N / 4);
      return spgan_launch_status();
    }
  }
  if constexpr (AMODE == A_AFFINE2 && EPI == SPGAN_EPI_EDGE_BNBWD) {
    if (a.a_half) {  // bfloat16 g / fp16 y (validated in spgan_gemm_nt): the EdgeBlock's lazy BatchNorm-backward operand in 16-bit storage
      if (a.N > 64 && a.K >= 512) launch_nt_cfg<AMODE, EPI, 0, 1, 1, 1, 1>(a, s);
      else launch_nt_cfg<AMODE, EPI, 1, 0, 1, 1, 1>(a, s);
      return spgan_launch_status();
    }
  }
  if constexpr (AMODE == SPGAN_A_PLAIN && EPI == SPGAN_EPI_LINEAR) {
    if (a.a_half) {  // fp16-stored A (validated in spgan_gemm_nt: fp16 mode, aligned, N > 32, M > 64, one product): the 128-row kernels only
      if (a.N > 64 && a.K >= 512) launch_nt_cfg<AMODE, EPI, 0, 1, 1, 1, 1>(a, s);
      else launch_nt_cfg<AMODE, EPI, 1, 0, 1, 1, 1>(a, s);
      return spgan_launch_status();
    }
  }
  if constexpr (AMODE != SPGAN_A_EDGE && AMODE != A_AFFINE2 && EPI != SPGAN_EPI_EDGE_BNBWD) {
    if (spgan_nt_wide16_selected(a)) return spgan_launch_nt_wide16(a, s);  // fp16 operands, large aligned products: row-pipelined 256-row tiles (gemm_wide16.hip)
    if (spgan_nt_wide_selected(a)) return spgan_launch_nt_wide(a, s);  // large aligned products: 256 x 256 tiles (gemm_wide.hip)
    if (spgan_nt_wide3_selected(a)) return spgan_launch_nt_wide3(a, s);  // ... with split-bf16 operands: 256-row tiles (gemm_wide3.hip)
  }
  if constexpr (AMODE != SPGAN_A_EDGE && AMODE != A_AFFINE_SPARSE && AMODE != A_AFFINE2 && EPI != SPGAN_EPI_EDGE_BNBWD) {
    if (a.M <= 64 && fast && !a.sp_val && a.batch <= 1 && !a.pool_val) {
      const bool whole = a.stats != nullptr || EPI == SPGAN_EPI_BNBWD;  // column statistics: one workgroup walks all rows of its columns
      hipLaunchKernelGGL((gemm_nt_small_kernel<AMODE, EPI>), dim3(cdiv(a.N, SC), (whole || a.M <= SR) ? 1 : cdiv(a.M, SR)), dim3(256), 0, s, a);
      return spgan_launch_status();
    }
  }
  if (a.tail.enabled) return SPGAN_EINVAL;  // only the M <= 64 kernel above finishes its columns in the launch (spgan_gemm_nt_owns_columns)
  if constexpr (AMODE != A_AFFINE_SPARSE) {
    if (a.mfma_f16 == 2 && fast && a.N > 32) {  // fp32 operands split into three bf16 terms: 128x64 tiles (three operand planes in LDS)
      launch_nt_cfg<AMODE, EPI, 1, 0, 1, 2>(a, s);
      return spgan_launch_status();
    }
    if (a.mfma_f16 == 1 && fast && a.N > 32) {  // fp16 operands (fp32 accumulate): same tiling rules
      if (a.N > 64 && a.K >= 512) launch_nt_cfg<AMODE, EPI, 0, 1, 1, 1>(a, s);
      else launch_nt_cfg<AMODE, EPI, 1, 0, 1, 1>(a, s);
      return spgan_launch_status();
    }
  }
  if (a.N > 64 && a.K >= 512) {  // long K: 128x128 tiles, double-buffered LDS, one barrier per k-tile
    if (fast) launch_nt_cfg<AMODE, EPI, 0, 1, 1>(a, s);
    else launch_nt_cfg<AMODE, EPI, 0, 1, 0>(a, s);
  } else if (a.N > 32) {
    // short K: 128x64 tiles.  With 2-8 k-tiles per workgroup the fixed load/epilogue latency dominates; the narrower tile
    // doubles the workgroups (2048 instead of 1024 at M=65536, N=256: finer quantisation over the 256 CUs, 5 resident per
    // CU instead of 3) -- measured 15-30 % faster than 128x128 on every N <= 1280, K <= 256 shape of the step.
    if (fast) launch_nt_cfg<AMODE, EPI, 1, 0, 1>(a, s);
    else launch_nt_cfg<AMODE, EPI, 1, 0, 0>(a, s);
  } else {
    launch_nt_cfg<AMODE, EPI, 2, 0, 0>(a, s);
  }
  return spgan_launch_status();
}

// ------------------------------------------------------------------------------------------ gemm_tn
constexpr int TKM = 32;  // m-rows per staging step (= MFMA k-steps * 2 between two barriers)
constexpr int TA = 128;  // output rows (columns of A) per workgroup

// Column-constant prologue parameters of this thread's staging slot (hoisted out of the m loop).
struct ColPro {
  float4 sc, sh, eb;
};

// grid: (tilesA * tilesB, splits).  Each workgroup reduces `rows_per_split` m-rows into one
// TA x TB partial tile written to ws[split][Na][Nb].  CFG as in gemm_nt: TB = 128 / 64 / 32.
constexpr int TKF = TKM;  // fp32 kernel: m-rows per staging step (16 measured slower: 90.8 -> 100.2 us on 65536 x 256 x 256 although three workgroups fit a CU)
template <int BMODE, int CFG, int FAST>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const spgan_gemm_tn_args p, int rows_per_split) {
  using G = Geo<CFG>;
  constexpr int TI = G::TI, TJ = G::TJ;
  constexpr int TB = G::WGN * TJ * 32;
  constexpr int LDA_ = TA, LDB_ = TB;  // fragment reads are 32 consecutive floats of one row: conflict-free as is
  __shared__ __attribute__((aligned(16))) float smem[2 * TKF * (LDA_ + LDB_)];
  float* As = smem;                   // [2][TKF*LDA_]
  float* Bs = smem + 2 * TKF * LDA_;  // [2][TKF*LDB_]

  const int tilesB = (p.Nb + TB - 1) / TB;
  const int ta = blockIdx.x / tilesB, tb = blockIdx.x % tilesB;
  const int a0 = ta * TA, b0 = tb * TB;
  const int split = blockIdx.y;
  const int mbeg = split * rows_per_split;
  const int mend = min(p.M, mbeg + rows_per_split);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / G::WGN, wn = wave % G::WGN;  // wave owns A-cols [wm*TI*32, +TI*32), B-cols [wn*TJ*32, +TJ*32)
  const int l31 = lane & 31, lh = lane >> 5;
  const bool vecA = ((p.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.A) & 15) == 0);
  const bool vecB = ((p.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.B) & 15) == 0);

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging: A tile TKF x 128 floats -> ASLOTS float4 per thread; B tile TKF x TB -> BSLOTS per thread
  constexpr int ASLOTS = TKF * TA / 4 / 256;
  constexpr int BSLOTS = (TKF * TB / 4 + 255) / 256;
  float4 ra[ASLOTS], rb[BSLOTS];
  ColPro cp[BSLOTS];
#pragma unroll
  for (int i = 0; i < BSLOTS; ++i) {
    const int s = tid + 256 * i;
    const int c = b0 + (s % (TB / 4)) * 4;
    cp[i].sc = cp[i].sh = cp[i].eb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (BMODE != SPGAN_A_PLAIN && c < p.Nb) {
      cp[i].sc = ld4(p.p_scale + c, false, c, p.Nb);
      cp[i].sh = ld4(p.p_shift + c, false, c, p.Nb);
      if (BMODE == SPGAN_A_EDGE) cp[i].eb = ld4(p.e_bias + c, false, c, p.Nb);
    }
  }
  // optional A-side prologue (per column of A; the column of a staging slot is fixed -> parameters hoisted)
  const bool apro = p.a_scale != nullptr;
  const bool a2 = p.A2 != nullptr;  // two-tensor A operand: a = A*a_scale + A2*a_scale2 + a_shift (host: with a_scale, without a_sp_val)
  float4 asc = make_float4(0.f, 0.f, 0.f, 0.f), ash = asc, asc2 = asc;  // slot i covers column (tid + 256 i) & 31: the same for every i
  {
    const int col = a0 + (tid & 31) * 4;
    if (apro && col < p.Na) {
      asc = ld4(p.a_scale + col, false, col, p.Na);
      ash = ld4(p.a_shift + col, false, col, p.Na);
      if (a2) asc2 = ld4(p.a_scale2 + col, false, col, p.Na);
    }
  }
  float4 ra2[ASLOTS];
  // column sums of the (transformed) A operand: the bias gradient next to this weight gradient, for free (a_colsum_ws)
  const bool acs = p.a_colsum_ws != nullptr && tb == 0;
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
  // As in gemm_nt the staging registers keep RAW loads; the prologues run in sstore, after the MFMAs of the current tile.
  float4 rb2[BMODE == SPGAN_A_EDGE ? BSLOTS : 1];
  // FAST (16-byte aligned operands, Na/Nb/lda/ldb multiples of 4): straight-line float4 loads from clamped addresses --
  // no divergent tail handling, all loads of a tile issue back to back; out-of-range slots are zeroed in sstore.
  auto gload = [&](int mb) {
#pragma unroll
    for (int i = 0; i < ASLOTS; ++i) {
      const int s = tid + 256 * i, r = s >> 5, c = (s & 31) * 4;
      const int m = mb + r, col = a0 + c;
      if (FAST) ra[i] = *reinterpret_cast<const float4*>(p.A + (size_t)min(m, p.M - 1) * p.lda + (col < p.Na ? col : 0));
      else ra[i] = (m < mend && col < p.Na) ? ld4(p.A + (size_t)m * p.lda + col, vecA, col, p.Na) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (a2) {
        if (FAST) ra2[i] = *reinterpret_cast<const float4*>(p.A2 + (size_t)min(m, p.M - 1) * p.lda2 + (col < p.Na ? col : 0));
        else ra2[i] = (m < mend && col < p.Na) ? ld4(p.A2 + (size_t)m * p.lda2 + col, false, col, p.Na) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int i = 0; i < BSLOTS; ++i) {
      const int s = tid + 256 * i;
      const int r = s / (TB / 4), c = b0 + (s % (TB / 4)) * 4;
      const int m = mb + r;
      if (FAST) {
        if (r < TKF) {
          const int mc = min(m, p.M - 1), cc = c < p.Nb ? c : 0;
          if (BMODE == SPGAN_A_EDGE) {
            rb[i] = *reinterpret_cast<const float4*>(p.B + (size_t)p.e_idx[mc] * p.ldb + cc);
            rb2[i] = *reinterpret_cast<const float4*>(p.B + (size_t)fast_div(mc, p.e_k) * p.ldb + cc);
          } else {
            rb[i] = *reinterpret_cast<const float4*>(p.B + (size_t)mc * p.ldb + cc);
          }
        }
      } else {
        rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (BMODE == SPGAN_A_EDGE) rb2[i] = rb[i];
        if (r < TKF && m < mend && c < p.Nb) {
          if (BMODE == SPGAN_A_EDGE) {
            rb[i] = ld4(p.B + (size_t)p.e_idx[m] * p.ldb + c, vecB, c, p.Nb);
            rb2[i] = ld4(p.B + (size_t)fast_div(m, p.e_k) * p.ldb + c, vecB, c, p.Nb);
          } else {
            rb[i] = ld4(p.B + (size_t)m * p.ldb + c, vecB, c, p.Nb);
          }
        }
      }
    }
  };
  auto sstore = [&](int buf, int mb) {
    float* a = As + buf * TKF * LDA_;
    float* b = Bs + buf * TKF * LDB_;
#pragma unroll
    for (int i = 0; i < ASLOTS; ++i) {
      const int s = tid + 256 * i, r = s >> 5, c = (s & 31) * 4;
      float4 v = ra[i];
      const bool ok = mb + r < mend && a0 + c < p.Na;
      if (a2 && ok) {
        v.x = fmaf(v.x, asc.x, fmaf(ra2[i].x, asc2.x, ash.x));
        v.y = fmaf(v.y, asc.y, fmaf(ra2[i].y, asc2.y, ash.y));
        v.z = fmaf(v.z, asc.z, fmaf(ra2[i].z, asc2.z, ash.z));
        v.w = fmaf(v.w, asc.w, fmaf(ra2[i].w, asc2.w, ash.w));
        v = mask_tail(v, a0 + c, p.Na);
      } else if (apro && ok) v = mask_tail(affine_lrelu4(v, asc, ash, p.a_lrelu ? p.a_slope : 1.0f), a0 + c, p.Na);
      if (FAST && !ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (acs) { cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w; }   // rows in ascending order per thread: deterministic
      *reinterpret_cast<float4*>(&a[r * LDA_ + c]) = v;
    }
#pragma unroll
    for (int i = 0; i < BSLOTS; ++i) {
      const int s = tid + 256 * i;
      const int r = s / (TB / 4), c = (s % (TB / 4)) * 4;
      if (r < TKF) {
        float4 v = rb[i];
        const bool ok = mb + r < mend && b0 + c < p.Nb;
        if (FAST && !ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (BMODE != SPGAN_A_PLAIN && ok) {
          if (BMODE == SPGAN_A_EDGE) {
            v.x = (v.x - rb2[i].x) + cp[i].eb.x;
            v.y = (v.y - rb2[i].y) + cp[i].eb.y;
            v.z = (v.z - rb2[i].z) + cp[i].eb.z;
            v.w = (v.w - rb2[i].w) + cp[i].eb.w;
          }
          v = mask_tail(affine_lrelu4(v, cp[i].sc, cp[i].sh, p.p_slope), b0 + c, p.Nb);
        }
        *reinterpret_cast<float4*>(&b[r * LDB_ + c]) = v;
      }
    }
  };

  if (mbeg < mend) {
    gload(mbeg);
    sstore(0, mbeg);
    __syncthreads();
    int buf = 0;
    for (int mb = mbeg; mb < mend; mb += TKF, buf ^= 1) {
      const bool more = mb + TKF < mend;
      if (more) gload(mb + TKF);
      const float* a = As + buf * TKF * LDA_ + wm * TI * 32 + l31;
      const float* b = Bs + buf * TKF * LDB_ + wn * TJ * 32 + l31;
#pragma unroll
      for (int kq = 0; kq < TKF / 2; ++kq) {  // MFMA k = 2 m-rows: lane half lh takes row 2*kq + lh
        float af[TI], bf[TJ];
#pragma unroll
        for (int i = 0; i < TI; ++i) af[i] = a[(kq * 2 + lh) * LDA_ + i * 32];
#pragma unroll
        for (int j = 0; j < TJ; ++j) bf[j] = b[(kq * 2 + lh) * LDB_ + j * 32];
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
      if (more) sstore(buf ^ 1, mb + TKF);
      __syncthreads();
    }
  }
  if (p.a_colsum_ws != nullptr && tb == 0) {  // 8 row groups per column quad -> one partial per (split, column); the LDS tiles are dead
    __syncthreads();
    *reinterpret_cast<float4*>(&As[(tid >> 5) * TA + (tid & 31) * 4]) = cs;
    __syncthreads();
    if (tid < TA && a0 + tid < p.Na) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) t += As[g * TA + tid];
      p.a_colsum_ws[(size_t)split * p.Na + a0 + tid] = t;
    }
  }
  // partial tile -> ws[split][Na][Nb]   (D: row = A-col index, col = B-col index)
  float* out = p.ws + (size_t)split * p.Na * p.Nb;
  if (a0 + G::WGM * TI * 32 <= p.Na && b0 + G::WGN * TJ * 32 <= p.Nb) {  // tile inside the output: straight-line stores, uniform offsets
    float* ob = out + (size_t)(a0 + wm * TI * 32 + 4 * lh) * p.Nb + (b0 + wn * TJ * 32 + l31);
    const unsigned ldo = (unsigned)p.Nb;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) ob[(size_t)((unsigned)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldo + (unsigned)(j * 32))] = acc[i][j][r];
    return;
  }
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = a0 + (wm * TI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int col = b0 + (wn * TJ + j) * 32 + l31;
        if (row < p.Na && col < p.Nb) out[(size_t)row * p.Nb + col] = acc[i][j][r];
      }
}

// gemm_tn with bfloat16 operands (spgan_gemm_tn_args.mfma_lp == 1; BASELINE configs[4], "fp16 MFMA MLPs"): both operands are
// rounded to bf16 (round to nearest even; bf16 keeps fp32's exponent range -- weight gradients multiply activations with
// per-point gradients of magnitude 1e-5..1e-8, which fp16 would flush) when they are staged into LDS, AFTER the fp32 prologues,
// and multiplied with v_mfma_f32_32x32x16_bf16; accumulation, split-K partials and their fixed-order sum stay fp32.
// The MFMA wants, per lane, 8 consecutive k (= m-rows) of one operand column, while the operands lie [m][column] in memory: a
// thread therefore loads RP consecutive m-rows of 4 columns (RP = 4 / 2 / 1 for 128 / 64 / 32 staged columns per row), transposes
// that block in registers and writes RP packed values per column: the LDS tiles are [column][32 m-values as bf16] with a row
// stride of 20 words (16-byte aligned rows; 16 consecutive rows land on 16 distinct 4-bank groups: conflict-free b128 reads).
// Same split plan, workspace layout, epilogue and reduction as gemm_tn_kernel.  Aligned operands only (the FAST path).
constexpr int LDH = 20;  // words per LDS row: 32 bf16 = 16 words + 4 of padding
template <int BMODE, int CFG>
__global__ __launch_bounds__(256) void gemm_tn_lp_kernel(const spgan_gemm_tn_args p, int rows_per_split) {
  using G = Geo<CFG>;
  constexpr int TI = G::TI, TJ = G::TJ;
  constexpr int TB = G::WGN * TJ * 32;
  __shared__ __attribute__((aligned(16))) float smem[2 * (TA + TB) * LDH];
  float* As = smem;                 // [2][TA*LDH]
  float* Bs = smem + 2 * TA * LDH;  // [2][TB*LDH]

  const int tilesB = (p.Nb + TB - 1) / TB;
  const int ta = blockIdx.x / tilesB, tb = blockIdx.x % tilesB;
  const int a0 = ta * TA, b0 = tb * TB;
  const int split = blockIdx.y;
  const int mbeg = split * rows_per_split;
  const int mend = min(p.M, mbeg + rows_per_split);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / G::WGN, wn = wave % G::WGN;
  const int l31 = lane & 31, lh = lane >> 5;

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // A: 32 float4 per m-row, thread (tid & 31) owns 4 columns, (tid >> 5) the m-rows 4*(tid>>5) .. +3
  // B: QB = TB/4 float4 per m-row, 256/QB row groups of RP = 32*QB/256 rows
  constexpr int QB = TB / 4, RP = TKM * QB / 256;
  const int ac = (tid & 31) * 4, ar = 4 * (tid >> 5);
  const int bc = (tid % QB) * 4, br = RP * (tid / QB);
  const bool aok = a0 + ac < p.Na, bok = b0 + bc < p.Nb;
  float4 ra[4], rb[RP];
  float4 rb2[BMODE == SPGAN_A_EDGE ? RP : 1];
  ColPro cp;
  cp.sc = cp.sh = cp.eb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (BMODE != SPGAN_A_PLAIN && bok) {
    cp.sc = *reinterpret_cast<const float4*>(p.p_scale + b0 + bc);
    cp.sh = *reinterpret_cast<const float4*>(p.p_shift + b0 + bc);
    if (BMODE == SPGAN_A_EDGE) cp.eb = *reinterpret_cast<const float4*>(p.e_bias + b0 + bc);
  }
  const bool apro = p.a_scale != nullptr;
  const bool a2 = p.A2 != nullptr;
  float4 asc = make_float4(0.f, 0.f, 0.f, 0.f), ash = asc, asc2 = asc;
  if (apro && aok) {
    asc = *reinterpret_cast<const float4*>(p.a_scale + a0 + ac);
    ash = *reinterpret_cast<const float4*>(p.a_shift + a0 + ac);
    if (a2) asc2 = *reinterpret_cast<const float4*>(p.a_scale2 + a0 + ac);
  }
  float4 ra2[4];
  const bool a16 = p.a_half != 0;                          // A lies in memory as bfloat16, A2 (when given) as fp16: the EdgeBlock's 16-bit lazy operand
  const bool bh = BMODE == SPGAN_A_PLAIN && p.b_half;     // B lies in memory as fp16 (the EdgeBlock's T: spgan_edge_attend_fwd_h)
  const bool acs = p.a_colsum_ws != nullptr && tb == 0;   // fp32 column sums of the transformed A operand (before the bf16 rounding)
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
  auto gload = [&](int mb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (a16) {   // A stored as bfloat16 (+ A2 as fp16): 8-byte loads, kept raw until sstore
        const float2 h = *reinterpret_cast<const float2*>(reinterpret_cast<const uint16_t*>(p.A) + (size_t)min(mb + ar + i, p.M - 1) * p.lda + (aok ? a0 + ac : 0));
        ra[i].x = h.x;
        ra[i].y = h.y;
      } else
      ra[i] = *reinterpret_cast<const float4*>(p.A + (size_t)min(mb + ar + i, p.M - 1) * p.lda + (aok ? a0 + ac : 0));
    }
    if (a2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (a16) {
          const float2 h = *reinterpret_cast<const float2*>(reinterpret_cast<const uint16_t*>(p.A2) + (size_t)min(mb + ar + i, p.M - 1) * p.lda2 + (aok ? a0 + ac : 0));
          ra2[i].x = h.x;
          ra2[i].y = h.y;
        } else
        ra2[i] = *reinterpret_cast<const float4*>(p.A2 + (size_t)min(mb + ar + i, p.M - 1) * p.lda2 + (aok ? a0 + ac : 0));
      }
    }
#pragma unroll
    for (int i = 0; i < RP; ++i) {
      const int mc = min(mb + br + i, p.M - 1), cc = bok ? b0 + bc : 0;
      if (BMODE == SPGAN_A_EDGE) {
        rb[i] = *reinterpret_cast<const float4*>(p.B + (size_t)p.e_idx[mc] * p.ldb + cc);
        rb2[i] = *reinterpret_cast<const float4*>(p.B + (size_t)fast_div(mc, p.e_k) * p.ldb + cc);
      } else if (bh) {   // fp16-stored B: 4 values = 8 bytes, kept raw until sstore
        const float2 h = *reinterpret_cast<const float2*>(reinterpret_cast<const _Float16*>(p.B) + (size_t)mc * p.ldb + cc);
        rb[i].x = h.x;
        rb[i].y = h.y;
      } else {
        rb[i] = *reinterpret_cast<const float4*>(p.B + (size_t)mc * p.ldb + cc);
      }
    }
  };
  auto sstore = [&](int buf, int mb) {
    __bf16* a = reinterpret_cast<__bf16*>(As + buf * TA * LDH);
    __bf16* b = reinterpret_cast<__bf16*>(Bs + buf * TB * LDH);
    float va[4][4];  // [m-row][column]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = ra[i];
      if (a16) {
        const float2 rg = make_float2(ra[i].x, ra[i].y), ry = make_float2(ra2[i].x, ra2[i].y);
        const bf16x4 gb = *reinterpret_cast<const bf16x4*>(&rg);
        const f16x4 yh = *reinterpret_cast<const f16x4*>(&ry);
        v = make_float4((float)gb[0], (float)gb[1], (float)gb[2], (float)gb[3]);
        if (a2) ra2[i] = make_float4((float)yh[0], (float)yh[1], (float)yh[2], (float)yh[3]);
      }
      const bool ok = mb + ar + i < mend && aok;
      if (a2 && ok) {
        v.x = fmaf(v.x, asc.x, fmaf(ra2[i].x, asc2.x, ash.x));
        v.y = fmaf(v.y, asc.y, fmaf(ra2[i].y, asc2.y, ash.y));
        v.z = fmaf(v.z, asc.z, fmaf(ra2[i].z, asc2.z, ash.z));
        v.w = fmaf(v.w, asc.w, fmaf(ra2[i].w, asc2.w, ash.w));
      } else if (apro && ok) v = affine_lrelu4(v, asc, ash, p.a_lrelu ? p.a_slope : 1.0f);
      if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (acs) { cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w; }
      va[i][0] = v.x; va[i][1] = v.y; va[i][2] = v.z; va[i][3] = v.w;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // column ac+q: m-rows ar..ar+3 as 4 bf16 = one 8-byte store
      f32x4v f = {va[0][q], va[1][q], va[2][q], va[3][q]};
      *reinterpret_cast<bf16x4*>(a + (ac + q) * (2 * LDH) + ar) = __builtin_convertvector(f, bf16x4);
    }
    float vb[RP][4];
#pragma unroll
    for (int i = 0; i < RP; ++i) {
      float4 v = rb[i];
      if (bh) {
        const float2 raw = make_float2(rb[i].x, rb[i].y);
        const f16x4 h = *reinterpret_cast<const f16x4*>(&raw);
        v = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
      }
      const bool ok = mb + br + i < mend && bok;
      if (BMODE != SPGAN_A_PLAIN && ok) {
        if (BMODE == SPGAN_A_EDGE) {
          v.x = (v.x - rb2[i].x) + cp.eb.x;
          v.y = (v.y - rb2[i].y) + cp.eb.y;
          v.z = (v.z - rb2[i].z) + cp.eb.z;
          v.w = (v.w - rb2[i].w) + cp.eb.w;
        }
        v = affine_lrelu4(v, cp.sc, cp.sh, p.p_slope);
      }
      if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
      vb[i][0] = v.x; vb[i][1] = v.y; vb[i][2] = v.z; vb[i][3] = v.w;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      __bf16* d = b + (bc + q) * (2 * LDH) + br;
      if constexpr (RP == 4) {
        f32x4v f = {vb[0][q], vb[1][q], vb[2][q], vb[3][q]};
        *reinterpret_cast<bf16x4*>(d) = __builtin_convertvector(f, bf16x4);
      } else {
#pragma unroll
        for (int i = 0; i < RP; ++i) d[i] = (__bf16)vb[i][q];
      }
    }
  };

  if (mbeg < mend) {
    gload(mbeg);
    sstore(0, mbeg);
    __syncthreads();
    int buf = 0;
    for (int mb = mbeg; mb < mend; mb += TKM, buf ^= 1) {
      const bool more = mb + TKM < mend;
      if (more) gload(mb + TKM);
      const __bf16* a = reinterpret_cast<const __bf16*>(As + buf * TA * LDH) + (wm * TI * 32 + l31) * (2 * LDH) + 8 * lh;
      const __bf16* b = reinterpret_cast<const __bf16*>(Bs + buf * TB * LDH) + (wn * TJ * 32 + l31) * (2 * LDH) + 8 * lh;
#pragma unroll
      for (int kk = 0; kk < TKM / 16; ++kk) {  // MFMA k = 16 m-rows: lane half lh takes rows 16*kk + 8*lh .. +7
        bf16x8 af[TI], bf[TJ];
#pragma unroll
        for (int i = 0; i < TI; ++i) af[i] = *reinterpret_cast<const bf16x8*>(a + i * 32 * (2 * LDH) + 16 * kk);
#pragma unroll
        for (int j = 0; j < TJ; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(b + j * 32 * (2 * LDH) + 16 * kk);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
      if (more) sstore(buf ^ 1, mb + TKM);
      __syncthreads();
    }
  }
  if (p.a_colsum_ws != nullptr && tb == 0) {
    __syncthreads();
    *reinterpret_cast<float4*>(&As[(tid >> 5) * TA + ac]) = cs;      // [8 row groups][128 columns] floats: 4 KB of the dead A tiles
    __syncthreads();
    if (tid < TA && a0 + tid < p.Na) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) t += As[g * TA + tid];
      p.a_colsum_ws[(size_t)split * p.Na + a0 + tid] = t;
    }
  }
  float* out = p.ws + (size_t)split * p.Na * p.Nb;
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = a0 + (wm * TI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int col = b0 + (wn * TJ + j) * 32 + l31;
        if (row < p.Na && col < p.Nb) out[(size_t)row * p.Nb + col] = acc[i][j][r];
      }
}

// Sparse addend of A (a_sp_val/a_sp_arg): each (shape b, A-column a) contributes val[b,a] * pro(B)[arg[b,a], :] to output
// row a.  That is a gather of one B row per (b, a) -- done here after the dense reduction, one workgroup per A-column,
// shapes in ascending order (deterministic), the gathers of a chunk of shapes all in flight together.
template <int BMODE>
__global__ __launch_bounds__(256) void tn_sparse_rows_kernel(const spgan_gemm_tn_args p) {
  __shared__ int sarg[256];
  __shared__ float sval[256];
  const int a = blockIdx.x, tid = threadIdx.x;
  const int shapes = (p.M + p.a_sp_rows - 1) / p.a_sp_rows;
  for (int c0 = 0; c0 < p.Nb; c0 += 256) {
    const int col = c0 + tid;
    const bool cok = col < p.Nb;
    float sc = 1.f, sh = 0.f;
    if (BMODE == SPGAN_A_AFFINE_LRELU && cok) { sc = p.p_scale[col]; sh = p.p_shift[col]; }
    float add = 0.f;
    for (int s0 = 0; s0 < shapes; s0 += 256) {
      __syncthreads();
      if (s0 + tid < shapes) {
        const int r = p.a_sp_arg[(size_t)(s0 + tid) * p.Na + a];
        sarg[tid] = (r >= 0 && r < p.M) ? r : -1;
        sval[tid] = p.a_sp_val[(size_t)(s0 + tid) * p.Na + a];
      }
      __syncthreads();
      const int n = min(256, shapes - s0);
      if (cok) {
#pragma unroll 8
        for (int b = 0; b < n; ++b) {
          const int r = sarg[b];
          float bv = (r >= 0) ? p.B[(size_t)r * p.ldb + col] : 0.f;
          if (BMODE == SPGAN_A_AFFINE_LRELU) bv = lrelu_f(fmaf(bv, sc, sh), p.p_slope);
          add = (r >= 0) ? fmaf(sval[b], bv, add) : add;
        }
      }
    }
    if (cok) p.C[(size_t)a * p.ldc + col] += add;
  }
}

// E[m, :] = sum over the channels c whose arg[b,c] == m of val[b,c] * W[c, :]: sparse_rows.hpp
__global__ __launch_bounds__(256) void sparse_rows_nt_kernel(const float* __restrict__ val, const int32_t* __restrict__ arg, int rows, int Cs,
                                                             const float* __restrict__ W, int ldw, int N, float* __restrict__ E, int lde,
                                                             int RB) {
  extern __shared__ unsigned sm_u[];
  sparse_rows_nt_body(blockIdx.x, blockIdx.y, sm_u, val, arg, rows, Cs, W, ldw, N, E, lde, RB);
}

// Fixed-order sum over the split partials: 64 consecutive outputs x 4 split-slices per workgroup.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int Na, int Nb, float* __restrict__ C,
                                                            int ldc, float beta) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const size_t stride = (size_t)Na * Nb;
  float s0 = 0.f, s1 = 0.f;
  if (i < Na * Nb) {
    int k = sl;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (; k + 28 < splits; k += 32) {  // eight partials in flight per thread (a long split list is latency-bound otherwise)
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += ws[(size_t)(k + 4 * u) * stride + i];
    }
    s0 = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    for (; k + 4 < splits; k += 8) {
      s0 += ws[(size_t)k * stride + i];
      s1 += ws[(size_t)(k + 4) * stride + i];
    }
    if (k < splits) s0 += ws[(size_t)k * stride + i];
  }
  red[sl][lane] = s0 + s1;
  __syncthreads();
  if (sl == 0 && i < Na * Nb) {
    const float s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    const int r = i / Nb, c = i % Nb;
    float* o = C + (size_t)r * ldc + c;
    *o = (beta == 0.f) ? s : fmaf(beta, *o, s);
  }
}

// Many partials of few outputs (the skinny products: 512 x [64,3]): 64 outputs x 4
// slices per workgroup would leave three workgroups walking 128 partials per thread -- here one WAVE owns an output, its lanes
// take the partials k = lane, lane+64, ... and a fixed butterfly adds them up.
inline bool reduce_by_wave(int splits, long n) { return splits >= 32 && n <= 1024; }  // beyond ~1k outputs the k-strided lane reads thrash (15 us for 8192 outputs)
inline int reduce_blocks(int splits, long n) { return reduce_by_wave(splits, n) ? cdiv(n, 4) : cdiv(n, 64); }

__device__ __forceinline__ void splitk_reduce_wave(const float* __restrict__ ws, int splits, int Na, int Nb, float* __restrict__ C, int ldc,
                                                   float beta, int block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = block * 4 + wave;
  if (i >= Na * Nb) return;
  const size_t stride = (size_t)Na * Nb;
  float s = 0.f;
  for (int k = lane; k < splits; k += 64) s += ws[(size_t)k * stride + i];
  s = wave_sum(s);
  if (lane == 0) {
    float* o = C + (size_t)(i / Nb) * ldc + (i % Nb);
    *o = (beta == 0.f) ? s : fmaf(beta, *o, s);
  }
}

__global__ __launch_bounds__(256) void splitk_reduce_wave_kernel(const float* __restrict__ ws, int splits, int Na, int Nb, float* __restrict__ C,
                                                                 int ldc, float beta) {
  splitk_reduce_wave(ws, splits, Na, Nb, C, ldc, beta, blockIdx.x);
}

// The same reduction for up to SPGAN_MULTI_MAX pending products in one launch (spgan_splitk_reduce_multi): block b serves
// entry e with start[e] <= b < start[e+1].
__global__ __launch_bounds__(256) void splitk_reduce_multi_kernel(const spgan_splitk_multi_args a) {
  __shared__ float red[4][64];
  int e = 0;
  while (e + 1 < a.count && (int)blockIdx.x >= a.block_start[e + 1]) ++e;
  const float* __restrict__ ws = a.ws[e];
  const int splits = a.splits[e], Na = a.Na[e], Nb = a.Nb[e], ldc = a.ldc[e];
  const float beta = a.beta[e];
  if (splits >= 32 && (long)Na * Nb <= 1024) {  // reduce_by_wave (uniform per workgroup; no barrier on this path)
    splitk_reduce_wave(ws, splits, Na, Nb, a.C[e], ldc, beta, (int)blockIdx.x - a.block_start[e]);
    return;
  }
  const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int i = ((int)blockIdx.x - a.block_start[e]) * 64 + lane;
  const size_t stride = (size_t)Na * Nb;
  float s0 = 0.f, s1 = 0.f;
  if (i < Na * Nb) {
    int k = sl;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (; k + 28 < splits; k += 32) {  // eight partials in flight per thread (a long split list is latency-bound otherwise)
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += ws[(size_t)(k + 4 * u) * stride + i];
    }
    s0 = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    for (; k + 4 < splits; k += 8) {
      s0 += ws[(size_t)k * stride + i];
      s1 += ws[(size_t)(k + 4) * stride + i];
    }
    if (k < splits) s0 += ws[(size_t)k * stride + i];
  }
  red[sl][lane] = s0 + s1;
  __syncthreads();
  if (sl == 0 && i < Na * Nb) {
    const float s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    const int r = i / Nb, c = i % Nb;
    float* o = a.C[e] + (size_t)r * ldc + c;
    *o = (beta == 0.f) ? s : fmaf(beta, *o, s);
  }
}

inline int tn_tb(int Nb) { return Nb > 64 ? 128 : (Nb > 32 ? 64 : 32); }

// Split choice: about two workgroups per CU in total, at least 256 m-rows each.
// One side of the product has <= 4 columns (the weight gradients of the 3-channel layers: D's first conv, the generator's tail.4
// and pc inputs): nothing for the matrix cores to do -- gemm_tn_skinny_kernel streams the wide operand once.
inline bool tn_skinny(int Na, int Nb) { return (Na <= 4 || Nb <= 4) && Na <= 2048 && Nb <= 2048; }

inline void tn_plan(int M, int Na, int Nb, int* splits, int* rows) {
  if (tn_skinny(Na, Nb)) {
    int r = cdiv(cdiv(M, 1024), TKM) * TKM;  // <= 1024 partials, each over a multiple of TKM rows (the MFMA kernel may have to serve the plan)
    if (r < 128) r = 128;
    *rows = r;
    *splits = cdiv(M, r);
    return;
  }
  const int tiles = cdiv(Na, TA) * cdiv(Nb, tn_tb(Nb));
  int want = cdiv(512, tiles);
  int r = cdiv(M, want);
  if (r < 64) r = 64;  // two staging steps: short reductions (M ~ 1000: the K x K products of the collapsed backward) are latency-bound, spread them
  r = cdiv(r, TKM) * TKM;
  *rows = r;
  *splits = cdiv(M, r);
}

// out[split][Na][Nb] partials of A^T B when A or B has <= 4 columns: 64 columns of the wide operand x 4 row-lanes per workgroup,
// the narrow operand's row is a broadcast load; the four row-lanes are summed in a fixed order.
template <bool NARROW_B>
__device__ __forceinline__ void gemm_tn_skinny_body(const spgan_gemm_tn_args& p, int rows_per_split, int bx, int split) {
  __shared__ float red[4][4][64];
  const int mbeg = split * rows_per_split, mend = min(p.M, mbeg + rows_per_split);
  const float* __restrict__ Lm = NARROW_B ? p.A : p.B;
  const float* __restrict__ Sm = NARROW_B ? p.B : p.A;
  const int ldl = NARROW_B ? p.lda : p.ldb, lds_ = NARROW_B ? p.ldb : p.lda;
  const int Ln = NARROW_B ? p.Na : p.Nb, Sn = NARROW_B ? p.Nb : p.Na;
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int l = bx * 64 + cl;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  // two-tensor A operand (spgan_gemm_tn_args.A2) on the wide side: a = A*a_scale[l] + A2*a_scale2[l] + a_shift[l], column l fixed per thread
  const bool a2 = NARROW_B && p.A2 != nullptr;
  float cp = 1.f, cq = 0.f, cr = 0.f;
  if (a2 && l < Ln) { cp = p.a_scale[l]; cq = p.a_scale2[l]; cr = p.a_shift[l]; }
  if (l < Ln) {
    int m = mbeg + rl;
    for (; m + 28 < mend; m += 32) {  // 8 rows in flight per thread
      float v[8], sv[8][4];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v[u] = Lm[(size_t)(m + 4 * u) * ldl + l];
        if (a2) v[u] = fmaf(v[u], cp, fmaf(p.A2[(size_t)(m + 4 * u) * p.lda2 + l], cq, cr));
        const float* srow = Sm + (size_t)(m + 4 * u) * lds_;
#pragma unroll
        for (int q = 0; q < 4; ++q) sv[u][q] = (q < Sn) ? srow[q] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = fmaf(v[u], sv[u][q], acc[q]);
    }
    for (; m < mend; m += 4) {
      float v = Lm[(size_t)m * ldl + l];
      if (a2) v = fmaf(v, cp, fmaf(p.A2[(size_t)m * p.lda2 + l], cq, cr));
      const float* srow = Sm + (size_t)m * lds_;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q < Sn) acc[q] = fmaf(v, srow[q], acc[q]);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) red[rl][q][cl] = acc[q];
  __syncthreads();
  if (rl == 0 && l < Ln) {
    float* out = p.ws + (size_t)split * p.Na * p.Nb;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < Sn) {
        const float v = (red[0][q][cl] + red[1][q][cl]) + (red[2][q][cl] + red[3][q][cl]);
        if (NARROW_B) out[(size_t)l * p.Nb + q] = v;
        else out[(size_t)q * p.Nb + l] = v;
      }
  }
}

template <bool NARROW_B>
__global__ __launch_bounds__(256) void gemm_tn_skinny_kernel(const spgan_gemm_tn_args p, int rows_per_split) {
  gemm_tn_skinny_body<NARROW_B>(p, rows_per_split, blockIdx.x, blockIdx.y);
}

// spgan_gemm_tn_skinny_multi: the 3-column weight gradient of D's first conv for several passes (blockIdx.z) as one launch
struct TnSkinnyMulti {
  spgan_gemm_tn_args a[SPGAN_GROUP_MAX];
};
__global__ __launch_bounds__(256) void gemm_tn_skinny_multi_kernel(const TnSkinnyMulti m, int rows_per_split) {
  gemm_tn_skinny_body<true>(m.a[blockIdx.z], rows_per_split, blockIdx.x, blockIdx.y);
}

inline void launch_reduce(const spgan_gemm_tn_args& a, int splits, hipStream_t s) {
  const long n = (long)a.Na * a.Nb;
  if (reduce_by_wave(splits, n))
    hipLaunchKernelGGL(splitk_reduce_wave_kernel, dim3(cdiv(n, 4)), dim3(256), 0, s, a.ws, splits, a.Na, a.Nb, a.C, a.ldc, a.beta);
  else
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(n, 64)), dim3(256), 0, s, a.ws, splits, a.Na, a.Nb, a.C, a.ldc, a.beta);
}

// The plan of a launch with operand mode lp: the split-bf16 kernel (lp == 2, gemm_tn_wide3.hip) has larger output tiles, hence its own plan for
// the shapes it takes (from the shape alone: a problem it then cannot run -- alignment -- runs the plan on the kernels below)
inline void tn_plan_lp(int M, int Na, int Nb, int lp, int* splits, int* rows) {
  if (lp == 2 && !tn_skinny(Na, Nb) && spgan_tn_wide3_config(M, Na, Nb)) spgan_tn_wide3_plan(M, Na, Nb, splits, rows);
  else tn_plan(M, Na, Nb, splits, rows);
}

template <int BMODE>
int launch_tn(const spgan_gemm_tn_args& a, hipStream_t s) {
  int splits, rows;
  tn_plan_lp(a.M, a.Na, a.Nb, a.mfma_lp, &splits, &rows);
  if constexpr (BMODE != SPGAN_A_EDGE) {
    if (!tn_skinny(a.Na, a.Nb) && spgan_tn_wide3_eligible(a)) {   // split-bf16 operands: the fp32-equivalent product on the bf16 matrix pipe
      const int rc = spgan_launch_tn_wide3(a, splits, rows, s);
      if (rc != SPGAN_OK) return rc;
      if (!a.defer_reduce) launch_reduce(a, splits, s);
      if (a.a_sp_val) hipLaunchKernelGGL((tn_sparse_rows_kernel<BMODE>), dim3(a.Na), dim3(256), 0, s, a);
      return spgan_launch_status();
    }
  }
  // the streaming kernels: no prologues -- except the two-tensor A operand on the wide side (a lazy BatchNorm-backward tensor against 3 input columns)
  if (BMODE == SPGAN_A_PLAIN && tn_skinny(a.Na, a.Nb) && !a.a_colsum_ws && !a.b_half &&
      (!a.a_scale || (a.A2 && a.Nb <= 4 && !a.a_lrelu && !a.a_sp_val))) {
    const bool narrow_b = a.Nb <= 4;
    const dim3 g(cdiv(narrow_b ? a.Na : a.Nb, 64), splits);
    if (narrow_b) hipLaunchKernelGGL((gemm_tn_skinny_kernel<true>), g, dim3(256), 0, s, a, rows);
    else hipLaunchKernelGGL((gemm_tn_skinny_kernel<false>), g, dim3(256), 0, s, a, rows);
    if (!a.defer_reduce) launch_reduce(a, splits, s);
    return spgan_launch_status();
  }
  if (a.b_half && (BMODE != SPGAN_A_PLAIN || a.mfma_lp != 1 || a.a_sp_val)) return SPGAN_EINVAL;  // fp16-stored B: the bf16 kernel only
  if (a.a_half && (a.mfma_lp != 1 || a.a_sp_val || tn_skinny(a.Na, a.Nb))) return SPGAN_EINVAL;    // 16-bit stored A (/A2): likewise
  const int TB = tn_tb(a.Nb);
  const dim3 grid(cdiv(a.Na, TA) * cdiv(a.Nb, TB), splits);
  const bool fast = (a.Na % 4 == 0) && (a.Nb % 4 == 0) && (a.lda % 4 == 0) && (a.ldb % 4 == 0) && al16(a.A) && al16(a.B) &&
                    (!a.A2 || (al16(a.A2) && a.lda2 % 4 == 0));
  if ((a.b_half || a.a_half) && !fast) return SPGAN_EINVAL;
  if (fast && a.mfma_lp == 1) {  // bf16 operands (aligned problems only; others keep the fp32 kernel)
    if (TB == 128) hipLaunchKernelGGL((gemm_tn_lp_kernel<BMODE, 0>), grid, dim3(256), 0, s, a, rows);
    else if (TB == 64) hipLaunchKernelGGL((gemm_tn_lp_kernel<BMODE, 1>), grid, dim3(256), 0, s, a, rows);
    else hipLaunchKernelGGL((gemm_tn_lp_kernel<BMODE, 2>), grid, dim3(256), 0, s, a, rows);
  } else if (fast) {
    if (TB == 128) hipLaunchKernelGGL((gemm_tn_kernel<BMODE, 0, 1>), grid, dim3(256), 0, s, a, rows);
    else if (TB == 64) hipLaunchKernelGGL((gemm_tn_kernel<BMODE, 1, 1>), grid, dim3(256), 0, s, a, rows);
    else hipLaunchKernelGGL((gemm_tn_kernel<BMODE, 2, 1>), grid, dim3(256), 0, s, a, rows);
  } else {
    if (TB == 128) hipLaunchKernelGGL((gemm_tn_kernel<BMODE, 0, 0>), grid, dim3(256), 0, s, a, rows);
    else if (TB == 64) hipLaunchKernelGGL((gemm_tn_kernel<BMODE, 1, 0>), grid, dim3(256), 0, s, a, rows);
    else hipLaunchKernelGGL((gemm_tn_kernel<BMODE, 2, 0>), grid, dim3(256), 0, s, a, rows);
  }
  if (!a.defer_reduce) launch_reduce(a, splits, s);  // deferred: the caller sums the partials later, batched with others (spgan_splitk_reduce_multi)
  if constexpr (BMODE != SPGAN_A_EDGE) {
    if (a.a_sp_val) hipLaunchKernelGGL((tn_sparse_rows_kernel<BMODE>), dim3(a.Na), dim3(256), 0, s, a);
  }
  return spgan_launch_status();
}

}  // namespace

// N-tile width launch_nt picks for this problem (must mirror launch_nt)
static int nt_tile_n(const spgan_gemm_nt_args& a) {
  if (a.a_mode != SPGAN_A_EDGE && !a.A2 && a.epi_mode != SPGAN_EPI_EDGE_BNBWD && spgan_nt_wide16_selected(a)) return spgan_nt_wide16_tile_n(a);
  if (spgan_nt_wide_selected(a)) return 256;
  if (a.a_mode != SPGAN_A_EDGE && !a.A2 && a.epi_mode != SPGAN_EPI_EDGE_BNBWD && spgan_nt_wide3_selected(a)) return spgan_nt_wide3_tile_n(a);
  const bool fast = (a.K % 4 == 0) && (a.lda % 4 == 0) && (a.ldw % 4 == 0) && al16(a.A) && al16(a.W);
  if (a.mfma_f16 == 2 && fast && a.N > 32 && !a.sp_val) return 64;
  if (a.mfma_f16 == 1 && fast && a.N > 32 && !a.sp_val) return (a.N > 64 && a.K >= 512) ? 128 : 64;
  if (a.N > 64 && a.K >= 512) return 128;
  return a.N > 32 ? 64 : 32;
}

extern "C" int spgan_gemm_nt_col_blocks(const spgan_gemm_nt_args* a) {
  if (!a || a->N <= 0) return 0;
  return cdiv(a->N, nt_tile_n(*a));
}

extern "C" int spgan_gemm_nt_owns_columns(const spgan_gemm_nt_args* a) {
  // mirrors launch_nt: the M <= 64 kernel (one workgroup walks all rows of its columns) takes aligned, unbatched problems
  if (!a || a->M <= 0 || a->M > 64 || a->sp_val || a->batch > 1 || a->pool_val || a->A2) return 0;
  if (a->a_mode == SPGAN_A_EDGE || a->epi_mode == SPGAN_EPI_EDGE_BNBWD) return 0;
  bool fast = (a->K % 4 == 0) && (a->lda % 4 == 0) && (a->ldw % 4 == 0) && al16(a->A) && al16(a->W);
  if (a->a_mode != SPGAN_A_PLAIN) fast = fast && al16(a->p_scale) && al16(a->p_shift);
  return fast ? 1 : 0;
}

// y_bf16 / y_half are honoured by the fp16-operand kernels' LINEAR epilogues (128-row and 256 x 256): everything fp16 mode sends there
extern "C" int spgan_gemm_nt_y16_ok(const spgan_gemm_nt_args* a) {
  if (!a || a->mfma_f16 != 1 || a->epi_mode != SPGAN_EPI_LINEAR || a->act != SPGAN_ACT_NONE || a->batch > 1 || a->M <= 64 || a->N <= 32) return 0;
  if (a->sp_val) return 0;   // the sparse-addend kernels have no fp16 form
  bool fast = (a->K % 4 == 0) && (a->lda % 4 == 0) && (a->ldw % 4 == 0) && al16(a->A) && al16(a->W);
  if (a->a_mode != SPGAN_A_PLAIN) fast = fast && al16(a->p_scale) && al16(a->p_shift);
  if (a->a_mode == SPGAN_A_EDGE) fast = fast && al16(a->e_bias);
  if (a->A2) fast = fast && al16(a->A2) && (a->lda2 % 4 == 0) && al16(a->p_scale2);
  return fast ? 1 : 0;
}

extern "C" int spgan_gemm_nt(const spgan_gemm_nt_args* a, spgan_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  SPGAN_CHECK_ARG(a && a->A && a->W && a->M > 0 && a->N > 0 && a->K > 0);
  SPGAN_CHECK_ARG(a->Y || (a->pool_val && a->epi_mode == SPGAN_EPI_LINEAR));
  if (a->pool_val) SPGAN_CHECK_ARG(a->pool_arg && a->epi_mode == SPGAN_EPI_LINEAR && a->M > 64);
  SPGAN_CHECK_ARG(a->lda >= a->K && a->ldw >= a->K && a->ldy >= a->N);
  SPGAN_CHECK_ARG((uint64_t)a->M * (uint64_t)a->lda < (1ull << 32) && (uint64_t)a->N * (uint64_t)a->ldw < (1ull << 32));  // 32-bit operand offsets
  if (a->a_mode != SPGAN_A_PLAIN) SPGAN_CHECK_ARG(a->p_scale && a->p_shift);
  if (a->p_group_rows != 0)  // row groups (per-group prologue vectors when there is a prologue): whole 128-row tiles per group, not for the M <= 64 kernel
    SPGAN_CHECK_ARG(a->p_group_rows > 0 && a->p_group_rows % BM == 0 && a->M % a->p_group_rows == 0 && a->M > 64);
  if (a->a_mode == SPGAN_A_EDGE) SPGAN_CHECK_ARG(a->e_idx && a->e_bias && a->e_k > 0);
  if (a->rowbias) SPGAN_CHECK_ARG(a->rows_per_group > 0 && a->ld_rowbias >= a->N);
  if (a->sp_val) SPGAN_CHECK_ARG(a->a_mode == SPGAN_A_AFFINE_LRELU && a->sp_arg && a->sp_rows > 0);
  if (a->A2)  // two-tensor operand a = A*p_scale + A2*p_scale2 + p_shift (no activation, no sparse addend, one vector set for all rows)
    SPGAN_CHECK_ARG(a->a_mode == SPGAN_A_AFFINE_LRELU && !a->sp_val && a->p_scale2 && a->lda2 >= a->K && a->p_group_rows == 0 && a->batch <= 1 &&
                    a->M > 64 && (uint64_t)a->M * (uint64_t)a->lda2 < (1ull << 32) && a->epi_mode != SPGAN_EPI_MASK_OUT);
  if (a->batch > 1)
    SPGAN_CHECK_ARG(a->a_mode == SPGAN_A_PLAIN && a->epi_mode == SPGAN_EPI_LINEAR && a->Y && !a->stats && !a->rowbias && !a->pool_val &&
                    a->batch <= 65535 && a->batch_stride_a >= 0 && a->batch_stride_w >= 0 && a->batch_stride_y > 0);
  if (a->gout_add)   // the stored tile gout_add + gout_scale * g: an epilogue of the split-bf16 256-row-tile kernel only (its one caller's route)
    SPGAN_CHECK_ARG(a->epi_mode == SPGAN_EPI_BNBWD && a->gout_scale && a->ld_gout_add >= a->N && a->Y && spgan_nt_wide3_selected(*a) &&
                    (uint64_t)a->M * (uint64_t)a->ld_gout_add < (1ull << 32));
  if (a->tail.enabled) {  // finished in the launch only where one workgroup owns its columns: the M <= 64 kernel
    const spgan_coltail& f = a->tail;
    SPGAN_CHECK_ARG(a->stats && a->M <= 64 && (f.mode == 0 || f.mode == 1));
    if (a->epi_mode == SPGAN_EPI_MASK_OUT) SPGAN_CHECK_ARG(f.mode == 1);  // column sums of a masked product
    if (f.mode == 1) SPGAN_CHECK_ARG(f.out0 && f.out1);
    if (f.scale) SPGAN_CHECK_ARG(f.mode == 0 && f.shift && f.invstd && f.mean_out && (!f.rmean || f.rvar));
  }
  if (a->a_half || a->y_bf16 || a->y_half) {  // 16-bit storage: the fp16-operand kernels, aligned problems
    SPGAN_CHECK_ARG(a->mfma_f16 == 1 && a->batch <= 1 && !a->tail.enabled && !a->sp_val &&
                    a->M > 64 && a->N > 32 && a->K % 4 == 0 && a->lda % 4 == 0 && a->ldw % 4 == 0 && al16(a->A) && al16(a->W));
    if (a->a_half && !a->A2) SPGAN_CHECK_ARG(a->a_mode == SPGAN_A_PLAIN && a->epi_mode == SPGAN_EPI_LINEAR);   // fp16 A of a plain linear product
    if (a->a_half && a->A2)   // bfloat16 A + fp16 A2: the EdgeBlock's lazy BatchNorm-backward operand into the edge BatchNorm-backward epilogue
      SPGAN_CHECK_ARG(a->epi_mode == SPGAN_EPI_EDGE_BNBWD && a->lda2 % 4 == 0 && al16(a->A2) && al16(a->p_scale) && al16(a->p_shift) && al16(a->p_scale2));
    if (a->y_bf16 || a->y_half) SPGAN_CHECK_ARG(a->Y && !(a->y_bf16 && a->y_half) && a->act == SPGAN_ACT_NONE && spgan_gemm_nt_y16_ok(a));
  }
  switch (a->epi_mode) {
    case SPGAN_EPI_LINEAR:
      if (a->a_mode == SPGAN_A_PLAIN) return launch_nt<SPGAN_A_PLAIN, SPGAN_EPI_LINEAR>(*a, s);
      if (a->A2) return launch_nt<A_AFFINE2, SPGAN_EPI_LINEAR>(*a, s);
      if (a->a_mode == SPGAN_A_AFFINE_LRELU)
        return a->sp_val ? launch_nt<A_AFFINE_SPARSE, SPGAN_EPI_LINEAR>(*a, s) : launch_nt<SPGAN_A_AFFINE_LRELU, SPGAN_EPI_LINEAR>(*a, s);
      if (a->a_mode == SPGAN_A_EDGE) return launch_nt<SPGAN_A_EDGE, SPGAN_EPI_LINEAR>(*a, s);
      return SPGAN_EINVAL;
    case SPGAN_EPI_MASK_OUT:
      SPGAN_CHECK_ARG(a->a_mode == SPGAN_A_PLAIN && a->ref && a->ld_ref >= a->N);
      return launch_nt<SPGAN_A_PLAIN, SPGAN_EPI_MASK_OUT>(*a, s);
    case SPGAN_EPI_BNBWD:
      SPGAN_CHECK_ARG(a->a_mode != SPGAN_A_EDGE && a->ref && a->ld_ref >= a->N && a->b_scale && a->b_shift && a->b_mean && a->b_invstd);
      if (a->A2) return launch_nt<A_AFFINE2, SPGAN_EPI_BNBWD>(*a, s);
      if (a->a_mode == SPGAN_A_AFFINE_LRELU)
        return a->sp_val ? launch_nt<A_AFFINE_SPARSE, SPGAN_EPI_BNBWD>(*a, s) : launch_nt<SPGAN_A_AFFINE_LRELU, SPGAN_EPI_BNBWD>(*a, s);
      return launch_nt<SPGAN_A_PLAIN, SPGAN_EPI_BNBWD>(*a, s);
    case SPGAN_EPI_EDGE_BNBWD:
      SPGAN_CHECK_ARG((a->a_mode == SPGAN_A_PLAIN || a->A2) && a->ref && a->ld_ref >= a->N && a->b_scale && a->b_shift && a->b_mean && a->b_invstd &&
                      a->e_idx && a->e_k > 0 && a->e_bias2);
      if (a->A2) return launch_nt<A_AFFINE2, SPGAN_EPI_EDGE_BNBWD>(*a, s);
      return launch_nt<SPGAN_A_PLAIN, SPGAN_EPI_EDGE_BNBWD>(*a, s);
    default:
      return SPGAN_EINVAL;
  }
}

extern "C" int spgan_sparse_rows_nt(const float* val, const int32_t* arg, int B, int rows, int Cs, const float* W, int ldw, int N, float* E,
                                    int lde, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(val && arg && W && E && B > 0 && rows > 0 && Cs > 0 && N > 0 && ldw >= N && lde >= N && Cs <= 8192);
  const int RB = sparse_rows_nt_rb(rows, Cs);
  const size_t lds = sparse_rows_nt_lds(RB, Cs);
  hipLaunchKernelGGL(sparse_rows_nt_kernel, dim3(cdiv(rows, RB), B), dim3(256), lds, (hipStream_t)s_, val, arg, rows, Cs, W, ldw, N, E, lde, RB);
  return spgan_launch_status();
}

extern "C" int spgan_sparse_rows_tn(const float* val, const int32_t* arg, int B, int rows, int Cs, const float* Bm, int ldb, int Nb,
                                    const float* p_scale, const float* p_shift, float p_slope, float* C, int ldc, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(val && arg && Bm && C && B > 0 && rows > 0 && Cs > 0 && Nb > 0 && ldb >= Nb && ldc >= Nb && (!p_scale == !p_shift));
  spgan_gemm_tn_args a = {};
  a.B = Bm; a.ldb = ldb; a.C = C; a.ldc = ldc;
  a.M = B * rows; a.Na = Cs; a.Nb = Nb;
  a.p_scale = p_scale; a.p_shift = p_shift; a.p_slope = p_slope;
  a.a_sp_val = val; a.a_sp_arg = arg; a.a_sp_rows = rows;
  if (p_scale) hipLaunchKernelGGL((tn_sparse_rows_kernel<SPGAN_A_AFFINE_LRELU>), dim3(Cs), dim3(256), 0, (hipStream_t)s_, a);
  else hipLaunchKernelGGL((tn_sparse_rows_kernel<SPGAN_A_PLAIN>), dim3(Cs), dim3(256), 0, (hipStream_t)s_, a);
  return spgan_launch_status();
}

extern "C" size_t spgan_gemm_tn_ws_bytes(int M, int Na, int Nb) {
  if (M <= 0 || Na <= 0 || Nb <= 0) return 0;
  int splits, rows;
  tn_plan(M, Na, Nb, &splits, &rows);
  return (size_t)splits * Na * Nb * sizeof(float);
}

extern "C" int spgan_gemm_tn_splits(int M, int Na, int Nb) {
  if (M <= 0 || Na <= 0 || Nb <= 0) return 0;
  int splits, rows;
  tn_plan(M, Na, Nb, &splits, &rows);
  return splits;
}

extern "C" size_t spgan_gemm_tn_ws_bytes_lp(int M, int Na, int Nb, int mfma_lp) {
  if (M <= 0 || Na <= 0 || Nb <= 0) return 0;
  int splits, rows;
  tn_plan_lp(M, Na, Nb, mfma_lp, &splits, &rows);
  return (size_t)splits * Na * Nb * sizeof(float);
}

extern "C" int spgan_gemm_tn_splits_lp(int M, int Na, int Nb, int mfma_lp) {
  if (M <= 0 || Na <= 0 || Nb <= 0) return 0;
  int splits, rows;
  tn_plan_lp(M, Na, Nb, mfma_lp, &splits, &rows);
  return splits;
}

extern "C" int spgan_splitk_reduce_blocks(int splits, int Na, int Nb) {
  if (splits <= 0 || Na <= 0 || Nb <= 0) return 0;
  return reduce_blocks(splits, (long)Na * Nb);
}

extern "C" int spgan_splitk_reduce_multi(const spgan_splitk_multi_args* a, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(a && a->count > 0 && a->count <= SPGAN_MULTI_MAX && a->block_start[0] == 0);
  for (int e = 0; e < a->count; ++e) {
    SPGAN_CHECK_ARG(a->ws[e] && a->C[e] && a->splits[e] > 0 && a->Na[e] > 0 && a->Nb[e] > 0 && a->ldc[e] >= a->Nb[e]);
    SPGAN_CHECK_ARG(a->block_start[e + 1] - a->block_start[e] == reduce_blocks(a->splits[e], (long)a->Na[e] * a->Nb[e]));
  }
  hipLaunchKernelGGL(splitk_reduce_multi_kernel, dim3(a->block_start[a->count]), dim3(256), 0, (hipStream_t)s_, *a);
  return spgan_launch_status();
}

// `count` streaming weight-gradient products A^T B with a narrow B (Nb <= 4: the three input coordinates) of ONE shape as one launch; partials only
// (defer_reduce: the caller sums them with spgan_splitk_reduce_multi, splits = spgan_gemm_tn_splits): the stand-alone kernel's body per problem.
extern "C" int spgan_gemm_tn_skinny_multi(const spgan_gemm_tn_args* a, int count, spgan_stream_t s_) {
  SPGAN_CHECK_ARG(a && count >= 1 && count <= SPGAN_GROUP_MAX);
  TnSkinnyMulti m;
  for (int g = 0; g < count; ++g) {
    const spgan_gemm_tn_args& q = a[g];
    SPGAN_CHECK_ARG(q.A && q.B && q.ws && q.M > 0 && q.M < (1 << 24) && q.Na > 0 && q.Nb > 0 && q.Nb <= 4 && tn_skinny(q.Na, q.Nb));
    SPGAN_CHECK_ARG(q.M == a[0].M && q.Na == a[0].Na && q.Nb == a[0].Nb && q.lda >= q.Na && q.ldb >= q.Nb);
    SPGAN_CHECK_ARG(q.b_mode == SPGAN_A_PLAIN && q.defer_reduce && !q.a_colsum_ws && !q.b_half && !q.a_half && !q.a_lrelu && !q.a_sp_val);
    SPGAN_CHECK_ARG(q.ws_bytes >= spgan_gemm_tn_ws_bytes(q.M, q.Na, q.Nb));
    if (q.A2) SPGAN_CHECK_ARG(q.a_scale && q.a_scale2 && q.a_shift && q.lda2 >= q.Na);
    else SPGAN_CHECK_ARG(!q.a_scale);
    m.a[g] = q;
  }
  int splits, rows;
  tn_plan(a->M, a->Na, a->Nb, &splits, &rows);
  hipLaunchKernelGGL(gemm_tn_skinny_multi_kernel, dim3(cdiv(a->Na, 64), splits, count), dim3(256), 0, (hipStream_t)s_, m, rows);
  return spgan_launch_status();
}

extern "C" int spgan_gemm_tn(const spgan_gemm_tn_args* a, spgan_stream_t s_) {
  hipStream_t s = (hipStream_t)s_;
  SPGAN_CHECK_ARG(a && !(a->defer_reduce && a->a_sp_val));
  SPGAN_CHECK_ARG(a && a->A && a->B && a->C && a->ws && a->M > 0 && a->Na > 0 && a->Nb > 0);
  SPGAN_CHECK_ARG(a->lda >= a->Na && a->ldb >= a->Nb && a->ldc >= a->Nb);
  SPGAN_CHECK_ARG(a->ws_bytes >= spgan_gemm_tn_ws_bytes_lp(a->M, a->Na, a->Nb, a->mfma_lp));
  if (a->b_mode != SPGAN_A_PLAIN) SPGAN_CHECK_ARG(a->p_scale && a->p_shift);
  if (a->a_scale) SPGAN_CHECK_ARG(a->a_shift && (!a->a_sp_val || (a->a_sp_arg && a->a_sp_rows > 0 && a->b_mode != SPGAN_A_EDGE)));
  if (a->A2) SPGAN_CHECK_ARG(a->a_scale && a->a_scale2 && !a->a_sp_val && a->lda2 >= a->Na);
  if (a->a_lrelu) SPGAN_CHECK_ARG(a->a_scale && !a->A2 && !a->a_sp_val);
  if (a->a_colsum_ws) SPGAN_CHECK_ARG(!a->a_sp_val);  // (a streaming-shaped problem with this by-product runs on the MFMA kernel)
  SPGAN_CHECK_ARG(a->M < (1 << 24));  // fast_div domain
  switch (a->b_mode) {
    case SPGAN_A_PLAIN: return launch_tn<SPGAN_A_PLAIN>(*a, s);
    case SPGAN_A_AFFINE_LRELU: return launch_tn<SPGAN_A_AFFINE_LRELU>(*a, s);
    case SPGAN_A_EDGE:
      SPGAN_CHECK_ARG(a->e_idx && a->e_bias && a->e_k > 0);
      return launch_tn<SPGAN_A_EDGE>(*a, s);
    default: return SPGAN_EINVAL;
  }
}
