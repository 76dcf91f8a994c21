/*
 * spgan_hip.h -- C ABI of libspgan_hip.so: the MI355X (gfx950) kernels behind the SP-GAN
 * G+D train-step hot path.
 *
 * The reference (liruihui/SP-GAN) has no native plugin on its train path: Generator /
 * Discriminator bottom out in ATen ops.  The boundary below is therefore
 * new; it follows the calling convention of the reference's own extensions
 * (metrics/pointops/src/pointops_api.cpp:15-40, knnquery/knnquery_cuda.cpp:15-28: leading
 * integer sizes, then raw device pointers, outputs pre-allocated by the caller) with three
 * deliberate changes: every entry point takes the HIP stream explicitly, returns a status
 * (0 = ok, otherwise a hipError_t / negative argument-error code) instead of exit(-1)
 * (knnquery_cuda_kernel.cu:66-70), and never allocates, frees or synchronises.
 *
 * Layout conventions
 *   "pm"  point-major   float32 [M, C] row-major, M = B*N rows (one row per point)
 *   "cm"  channel-major float32 [B, C, N]  (the reference's nn.Module boundary layout)
 *   edge tensors        float32 [M*k, C]   row e = i*k + r  (point i, neighbour rank r)
 *   idx                 int32   [M, k]     GLOBAL row index (b*N + j) of the r-th neighbour
 *
 * Each function documents the reference lines whose arithmetic it carries.
 */
#ifndef SPGAN_HIP_H
#define SPGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* spgan_stream_t; /* hipStream_t */

#define SPGAN_OK 0
#define SPGAN_EINVAL (-22)

int spgan_version(void);
/* returns the compiled-for architecture string, e.g. "gfx950" */
const char* spgan_arch(void);

/* ------------------------------------------------------------------------------------------
 * Graph construction.  Generation/modules.py:683-725 (get_edge_features): pairwise distances
 * (695-699), full ascending sort and ranks 1..k (702-703), gather + concat (708-720).
 * ---------------------------------------------------------------------------------------- */

/* mode 0: d = (-2*<xi,xj> + |xi|^2) + |xj|^2 in fp32 (the reference's expanded form);
 * mode 1: d = sum_c (xi_c - xj_c)^2 in fp64 from the fp32 inputs (coordinate-space inputs, C<=8).
 * Writes the ranks 1..k of the ascending (distance, index) order of every row: rank 0 is dropped
 * positionally (modules.py:703), ties go to the lower index (stable sort).  k <= 32, k+1 <= N. */
int spgan_knn(const float* x_pm, int B, int N, int C, int k, int mode, int32_t* idx, spgan_stream_t s);

/* In-edge lists of the kNN graph (for deterministic gather-style backward instead of float
 * atomics; replaces the atomicAdd scatter of metrics/pointops/src/grouping/grouping_cuda_kernel.cu:28-45).
 * rowptr [M+1], src [M*k]: src[rowptr[j] .. rowptr[j+1]) = ascending edge ids e with idx[e] == j. */
int spgan_csr_build(const int32_t* idx, int B, int N, int k, int32_t* rowptr, int32_t* src, spgan_stream_t s);

/* ee[B,2C,N,k] = cat[x_i, x_j - x_i] from channel-major x and int64 idx [B, N*k] holding LOCAL
 * indices (the reference's return_idx format, modules.py:704,723). */
int spgan_edge_features_cm(const float* x_cm, const int64_t* idx_local, int B, int C, int N, int k,
                           float* ee, spgan_stream_t s);

/* int32 global [M,k]  <->  int64 local [B, N*k] */
int spgan_idx_to_local64(const int32_t* idx, int B, int N, int k, int64_t* out, spgan_stream_t s);
int spgan_idx_from_local64(const int64_t* idx_local, int B, int N, int k, int32_t* out, spgan_stream_t s);

/* ------------------------------------------------------------------------------------------
 * Layout helpers
 * ---------------------------------------------------------------------------------------- */
int spgan_cm_to_pm(const float* x_cm, int B, int C, int N, float* y_pm, spgan_stream_t s);
int spgan_pm_to_cm(const float* x_pm, int B, int C, int N, float* y_cm, spgan_stream_t s);
/* out[m, 0:Ca] = a[m,:], out[m, Ca:Ca+Cb] = b[m,:]   (torch.cat([x,z],-1), Generator.py:166) */
int spgan_concat2(const float* a, int Ca, const float* b, int Cb, int M, float* out, spgan_stream_t s);

/* ------------------------------------------------------------------------------------------
 * Shared-MLP contractions on the matrix cores (fp32-in / fp32-accumulate MFMA, exact f32).
 * Every 1x1 Conv1d/Conv2d/Linear of Generator.py:56-71,107-136 and Discriminator.py:55-95,
 * their input-gradients, and (spgan_gemm_tn) their weight-gradients.
 * ---------------------------------------------------------------------------------------- */

enum { SPGAN_A_PLAIN = 0, SPGAN_A_AFFINE_LRELU = 1, SPGAN_A_EDGE = 2 };
enum { SPGAN_EPI_LINEAR = 0, SPGAN_EPI_MASK_OUT = 1, SPGAN_EPI_BNBWD = 2, SPGAN_EPI_EDGE_BNBWD = 3 };
enum { SPGAN_ACT_NONE = 0, SPGAN_ACT_LRELU = 1, SPGAN_ACT_TANH = 2 };

typedef struct spgan_gemm_nt_args {
  /* Y[M,N] = epilogue( prologue(A)[M,K] . W[N,K]^T ) */
  const float* A; int lda;
  const float* W; int ldw;
  float* Y; int ldy;
  int M, N, K;
  int a_mode;                 /* SPGAN_A_* */
  /* A_AFFINE_LRELU: a = lrelu(A[m,k]*p_scale[k] + p_shift[k], p_slope)   (BatchNorm apply + LeakyReLU
   *                 of the previous layer fused into the operand load)
   * A_EDGE:         row e=(i,r): a = lrelu((A[idx[e],k] - A[i,k] + e_bias[k])*p_scale[k] + p_shift[k], p_slope)
   *                 (conv_w.0 of EdgeBlock restructured per point, Generator.py:57-59,78) */
  const float* p_scale; const float* p_shift; float p_slope;
  const int32_t* e_idx; int e_k; const float* e_bias;
  int epi_mode;               /* SPGAN_EPI_* */
  /* EPI_LINEAR: y = act(acc + bias[n] + rowbias[m / rows_per_group, n]); optional column statistics */
  const float* bias; const float* rowbias; int rows_per_group; int ld_rowbias;
  int act; float act_slope;
  float* stats;               /* NULL or partials [ceil(M/128), N, 2]: (sum, centred M2) per 128-row tile (pre-activation) */
  /* EPI_MASK_OUT:   y = acc * (ref[m,n] > 0 ? 1 : slope)                 (LeakyReLU backward from its output)
   * EPI_BNBWD:      z = ref*b_scale[n]+b_shift[n]; g = acc*(z>0?1:slope); xhat=(ref-b_mean[n])*b_invstd[n];
   *                 y = g; stats partials [tilesM, N, 2] = (sum g, sum g*xhat)   (plain sums)
   * EPI_EDGE_BNBWD: same with ref[e,n] := (P[idx[e],n] - P[i,n] + e_bias2[n])  (P = ref, ld_ref) */
  const float* ref; int ld_ref;
  const float* b_scale; const float* b_shift; const float* b_mean; const float* b_invstd; float b_slope;
  const float* e_bias2;
} spgan_gemm_nt_args;

int spgan_gemm_nt(const spgan_gemm_nt_args* a, spgan_stream_t s);

typedef struct spgan_gemm_tn_args {
  /* C[Na,Nb] = beta*C + sum_m A[m,Na]^T . prologue(B)[m,Nb]  -- weight gradients (reduction over points/edges).
   * Deterministic split over M: partial tiles go to `ws` and are summed in a fixed order. */
  const float* A; int lda;
  const float* B; int ldb;
  float* C; int ldc;
  int M, Na, Nb;
  int b_mode;                 /* SPGAN_A_* applied to B (per column of B) */
  const float* p_scale; const float* p_shift; float p_slope;
  const int32_t* e_idx; int e_k; const float* e_bias;
  float beta;
  float* ws; size_t ws_bytes; /* >= spgan_gemm_tn_ws_bytes(M,Na,Nb) */
} spgan_gemm_tn_args;

size_t spgan_gemm_tn_ws_bytes(int M, int Na, int Nb);
int spgan_gemm_tn(const spgan_gemm_tn_args* a, spgan_stream_t s);

/* Column reductions over row groups (group = G consecutive rows; M % G == 0).  Partials are
 * [groups * ceil(G/128)][C][2] floats: one (a, b) pair per 128-row tile and column -- the format
 * spgan_gemm_nt's `stats` epilogue writes (there: one group of M rows).
 * finalize mode 0 (Welford/Chan): partials (sum, centred M2) -> out0 = mean, out1 = biased variance.
 * finalize mode 1 (plain):         partials (s0, s1)          -> out0 = sum s0, out1 = sum s1.
 * Everything is combined in a fixed order: results are run-to-run deterministic. */
size_t spgan_colreduce_ws_bytes(int M, int C, int G);
int spgan_colstats_finalize(const float* partials, int groups, int tiles_per_group, int C, int G, int mode,
                            float* out0, float* out1, spgan_stream_t s);
/* mean / biased variance over each group of lrelu(X, slope) (slope = 1: plain).  InstanceNorm1d statistics of
 * AdaptivePointNorm (Generator.py:29,42) with G = N; BatchNorm statistics with G = M.  ws >= spgan_colreduce_ws_bytes. */
int spgan_colstats(const float* X, int ldx, int M, int C, int G, float slope, float* out_mean, float* out_var,
                   float* ws, size_t ws_bytes, spgan_stream_t s);
/* out[g, c] = sum over the G rows of group g of X[m,c]  (bias gradients; per-shape bias gradients).
 * ws >= spgan_colreduce_ws_bytes(M,C,G) + (M/G)*C*4. */
int spgan_colsum(const float* X, int ldx, int M, int C, int G, float* out, float* ws, size_t ws_bytes, spgan_stream_t s);

/* ------------------------------------------------------------------------------------------
 * Train-mode BatchNorm bookkeeping (torch defaults: eps 1e-5, momentum 0.1, biased var for the
 * normalisation, unbiased var into running_var).  Generator.py:58,61,67,121,124; Discriminator.py:57-79.
 * ---------------------------------------------------------------------------------------- */
/* training=1: from batch (mean,var): scale=gamma*invstd, shift=beta-mean*scale, invstd; updates running stats
 *             when running_mean != NULL.   training=0: uses running stats instead of (mean,var). */
int spgan_bn_prepare(const float* mean, const float* var, const float* gamma, const float* beta, int C,
                     int count, float eps, float momentum, int training,
                     float* running_mean, float* running_var,
                     float* scale, float* shift, float* invstd, float* mean_used, spgan_stream_t s);
/* dy[m,c] = gamma[c]*invstd[c]*( g[m,c] - sums[c]/count - xhat[m,c]*sums[C+c]/count ),  xhat=(y-mean)*invstd */
int spgan_bn_bwd_apply(const float* g, const float* y, int ld, int M, int C, const float* mean, const float* invstd,
                       const float* gamma, const float* sums, int count, float* dy, spgan_stream_t s);

/* ------------------------------------------------------------------------------------------
 * Global max over the N points of each shape (Generator.py:183; Discriminator.py:104) fused with
 * the preceding affine + LeakyReLU:  out[b,c] = max_n lrelu(y[b*N+n,c]*scale[c]+shift[c]).
 * ---------------------------------------------------------------------------------------- */
int spgan_maxpool(const float* y, int ld, int B, int N, int C, const float* scale, const float* shift, float slope,
                  float* out, int32_t* argmax, spgan_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* SPGAN_HIP_H */
