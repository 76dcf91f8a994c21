// KEPT NEGATIVE RESULT (round 4), not compiled into libspgan_hip.so.  It was routed from launch_nt for plain / LINEAR problems with
// few rows and K <= 256.  Back to back it is faster than the 128 x 64-tile kernel (13.4 -> 9.7 us at 1024 x 256 x 256, 8.0 -> 6.0 at
// 2048 x 64 x 128; tools/exp/mid_gemm_ab.py of commit 68a6d64), but inside the replayed train step the six launches it served went
// 15.6 -> 14.5 us only (cold operands: the chain is one HBM round trip either way), and its summation order (K split over the four
// waves, tree sum) differs from the 128-row kernels' single chain: at the small golden size (B=4, N=256, where activations also
// have ~1000 rows) the 1e-7 changes flipped a near-tie of D's max-pool arg-max and moved the D-step gradients by 2-5 % -- outside the
// reference-golden tolerances that hold when the arg-max agrees (tools/exp/op_trace.py found the flip).  6 us per step was not worth that.
// gemm_nt for the weight-by-weight products of the train step: a few hundred to a few thousand rows, K <= 256 -- the Gram-matrix
// forms of D's collapsed 256 -> 1024 layer (Discriminator.py:77-81: W [1024,256] times [256,256] sums over the points, six per step)
// and their relatives.  The 128-row kernels of gemm.hip give such a problem 16-32 workgroups that each walk K in 32-wide steps:
// 15-18 us of load latency per launch for 0.13 GFLOP.  Here
//   * one workgroup per 32 x 32 output tile (256 workgroups for 1024 x 256): the chip is filled;
//   * both operand tiles [32, K] are staged ONCE (all loads of the launch in flight together: one memory round trip);
//   * the four waves split K, each runs K/8 v_mfma_f32_32x32x2_f32 on its slice (16-byte conflict-free fragment reads feed four MFMAs),
//     and the four partial tiles are summed in a fixed order through LDS -- fp32 operands, deterministic.
// Plain operands and the LINEAR epilogue (bias, row addend, activation) only; everything else stays with gemm.hip.
#include <math.h>
#include <stdlib.h>
#include "gemm_mid.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int MT = 32;          // output tile (rows and columns)
constexpr int MID_KMAX = 256;
constexpr int mid_ldk(int K) { return K + 4; }  // 4-word groups per row odd (K % 8 == 0): conflict-free 16-byte reads, aligned 16-byte stores
constexpr size_t MID_LDS = (size_t)2 * MT * mid_ldk(MID_KMAX) * sizeof(float);

__global__ __launch_bounds__(256) void gemm_nt_mid_kernel(const spgan_gemm_nt_args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int K = p.K, LDK = K + 4, K4 = K / 4;
  float* As = smem;                // [32][LDK]
  float* Ws = smem + MT * LDK;     // [32][LDK]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * MT, n0 = blockIdx.y * MT;
  // stage both tiles: 2 * 32 * K/4 float4, all loads issued before the first store
  const int per = MT * K4;                 // float4 per operand tile: a multiple of 256 (K % 32 == 0)
  float4 va[MID_KMAX / 32], vw[MID_KMAX / 32];
#pragma unroll
  for (int i = 0; i < MID_KMAX / 32; ++i) {
    const int e = tid + 256 * i;
    if (e < per) {
      const int r = e / K4, c = (e % K4) * 4;
      va[i] = *reinterpret_cast<const float4*>(p.A + (size_t)(m0 + r) * p.lda + c);
      vw[i] = *reinterpret_cast<const float4*>(p.W + (size_t)(n0 + r) * p.ldw + c);
    }
  }
#pragma unroll
  for (int i = 0; i < MID_KMAX / 32; ++i) {
    const int e = tid + 256 * i;
    if (e < per) {
      const int r = e / K4, c = (e % K4) * 4;
      *reinterpret_cast<float4*>(As + r * LDK + c) = va[i];
      *reinterpret_cast<float4*>(Ws + r * LDK + c) = vw[i];
    }
  }
  __syncthreads();
  // wave w: k in [w*K/4, (w+1)*K/4); a 16-byte read at k + 4*lh holds the lane's value for four MFMA steps (k-pairs (k+i, k+4+i))
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int ks = wave * (K / 4), ke = ks + K / 4;
  const float* ap = As + l31 * LDK + 4 * lh;
  const float* wp = Ws + l31 * LDK + 4 * lh;
  for (int k = ks; k < ke; k += 8) {
    const float4 a = *reinterpret_cast<const float4*>(ap + k), w = *reinterpret_cast<const float4*>(wp + k);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w.w, acc, 0, 0, 0);
  }
  __syncthreads();                 // the operand tiles are dead: their LDS holds the four partial tiles [wave][r][lane]
  float* red = smem;
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  // wave w finishes accumulator registers 4w .. 4w+3 of the tile: row = (r & 3) + 8*(r >> 2) + 4*lh, column = l31
  const int col = n0 + l31;
  const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int r = 4 * wave + u;
    const float s = (red[(0 * 16 + r) * 64 + lane] + red[(1 * 16 + r) * 64 + lane]) + (red[(2 * 16 + r) * 64 + lane] + red[(3 * 16 + r) * 64 + lane]);
    const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    float o = s + bias;
    if (p.rowbias) o += p.rowbias[(size_t)(row / p.rows_per_group) * p.ld_rowbias + col];
    if (p.act == SPGAN_ACT_LRELU) o = lrelu_f(o, p.act_slope);
    else if (p.act == SPGAN_ACT_TANH) o = tanhf(o);
    p.Y[(size_t)row * p.ldy + col] = o;
  }
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

bool spgan_nt_mid_selected(const spgan_gemm_nt_args& a) {
  static const bool off = getenv("SPGAN_NT_MID") && atoi(getenv("SPGAN_NT_MID")) == 0;   // A/B measurements of whole programs
  if (off) return false;
  if (a.a_mode != SPGAN_A_PLAIN || a.epi_mode != SPGAN_EPI_LINEAR || a.stats || a.pool_val || a.batch > 1 || a.tail.enabled || a.mfma_f16 != 0 ||
      a.a_half || a.y_bf16 || a.y_half || a.A2 || a.sp_val || a.tile_hint != 0 || !a.Y)
    return false;
  if (a.M <= 64 || a.M % MT || a.N % MT || a.K % 32 || a.K < 32 || a.K > MID_KMAX) return false;
  if ((a.lda % 4) || (a.ldw % 4) || !al16(a.A) || !al16(a.W)) return false;
  if (a.rowbias && a.rows_per_group <= 0) return false;
  // only where the 128 x 64 tiles underfill the chip: at most 64 of them
  const long tiles128 = (long)((a.M + 127) / 128) * ((a.N + 63) / 64);
  return tiles128 <= 64;
}

int spgan_launch_nt_mid(const spgan_gemm_nt_args& a, hipStream_t s) {
  static bool attr_set = false;  // > 64 KB of dynamic LDS must be opted into once per kernel
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_mid_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)MID_LDS);
    attr_set = true;
  }
  const size_t lds = (size_t)2 * MT * mid_ldk(a.K) * sizeof(float);
  const size_t need = lds > (size_t)4 * 16 * 64 * sizeof(float) ? lds : (size_t)4 * 16 * 64 * sizeof(float);
  hipLaunchKernelGGL(gemm_nt_mid_kernel, dim3(a.M / MT, a.N / MT), dim3(256), need, s, a);
  return spgan_launch_status();
}
