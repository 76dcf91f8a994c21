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
#define SPGAN_ECOMM (-70) /* collective layer: RCCL not loadable, or an RCCL call failed (spgan_comm_last_error) */

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

/* The same selection with a caller-provided scratch buffer (csrc/knn_pipe.hip): for 16 < C <= 64, k <= 10, mode 0 a pre-pass writes every
 * 32-row tile once as three bfloat16 planes + squared norms into `ws`, and the scan is software-pipelined over those images; identical
 * indices, ~0.6x the time.  spgan_knn_ws_bytes returns the scratch size that route needs (16-byte aligned buffer), or 0 when the shape
 * is served by spgan_knn's kernels -- spgan_knn_ws then forwards to spgan_knn and ignores ws. */
size_t spgan_knn_ws_bytes(int B, int N, int C, int k, int mode);
int spgan_knn_ws(const float* x_pm, int B, int N, int C, int k, int mode, int32_t* idx, void* ws, size_t ws_bytes, spgan_stream_t s);

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

/* Column tail of the M <= 64 product kernel (the per-shape linears: one workgroup owns its columns entirely, so what would be a
 * follow-up spgan_colstats_finalize* launch is finished in the producing launch):
 *   mode 0: the column records are (sum, centred M2) -> out0 = mean, out1 = biased variance (either may be NULL) and, when
 *           scale != NULL, the train-mode BatchNorm bookkeeping of spgan_bn_prepare (scale, shift, invstd, mean_out; running
 *           statistics updated with momentum when rmean != NULL; count_rep multiplies the row count of the unbiased-variance factor);
 *   mode 1: plain sums -> out0 = sum of x, out1 = sum of y.
 * enabled = 0: off.  Only accepted with M <= 64.  (Round 2 carried a generalisation to many row tiles -- last-arriving workgroup
 * merges -- that measured slower than the finalize launch on every shape of the step; it lives in tools/exp/fanin.hpp now.) */
typedef struct spgan_coltail {
  int enabled;
  int mode;
  float* out0; float* out1;
  const float* gamma; const float* beta; float* rmean; float* rvar;
  float* scale; float* shift; float* invstd; float* mean_out;
  float eps, momentum;
  int count_rep;
} spgan_coltail;

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
   * EPI_BNBWD:      z = ref*b_scale[n]+b_shift[n]; g = (acc + bias[n] + rowbias[m / rows_per_group, n])*(z>0?1:slope)
   *                 (bias / rowbias optional); xhat=(ref-b_mean[n])*b_invstd[n];
   *                 y = g; stats partials [tilesM, N, 2] = (sum g, sum g*xhat)   (plain sums)
   * EPI_EDGE_BNBWD: same with ref[e,n] := (P[idx[e],n] - P[i,n] + e_bias2[n])  (P = ref, ld_ref) */
  const float* ref; int ld_ref;
  const float* b_scale; const float* b_shift; const float* b_mean; const float* b_invstd; float b_slope;
  const float* e_bias2;
  /* Optional sparse addend of the A operand (A_AFFINE_LRELU only): a += sp_val[b,k] where sp_arg[b,k] == m, b = m / sp_rows.
   * With p_slope = 1 this expresses the BatchNorm backward behind a global max-pool without materialising it:
   *   dy[m,k] = alpha[k]*y[m,k] + beta[k] + (argmax[b,k]==m ? coef[k]*gval[b,k] : 0)   (Discriminator.py:77-81,104) */
  const float* sp_val; const int32_t* sp_arg; int sp_rows;
  /* Optional pooling partials (EPI_LINEAR, M > 64): per 128-row tile and column the max and min of the pre-activation output
   * and their rows (first row on ties): pool_val / pool_arg [ceil(M/128), N, 2] = (max, min) / (arg-max row, arg-min row).
   * With them Y may be NULL (the output is not stored): adaptive_max_pool1d behind BatchNorm + LeakyReLU
   * (Discriminator.py:77-81,104) is finished by spgan_pool_finalize once the batch statistics are known. */
  float* pool_val; int32_t* pool_arg;
  /* 1: round the operands to fp16 when staging them (after the prologue) and multiply with the fp16 MFMA, fp32 accumulation
   * (BASELINE configs[4] "fp16 MFMA MLPs"); aligned operands and N > 32 only, otherwise the fp32 path is used.  Default 0.
   * 2: split every fp32 operand value exactly into three bfloat16 terms and evaluate the six leading cross products on the bf16 matrix
   * pipe with fp32 accumulation: fp32-equivalent products (dropped terms <= 3*2^-24 relative) at 6/16 of the fp32-MFMA time. */
  int mfma_f16;
  /* batch > 1 (A_PLAIN + EPI_LINEAR without stats / rowbias / pooling only): `batch` independent products in one launch,
   * product z uses A + z*batch_stride_a, W + z*batch_stride_w, Y + z*batch_stride_y (strides in floats; bias is shared).
   * The per-shape [N,N] contractions of the --attn variant (Generation/modules.py:554-556).  Default 0 / 1: a single product. */
  int batch; long batch_stride_a, batch_stride_w, batch_stride_y;
  /* tail.enabled (needs `stats`, M <= 64): the column records are finished in this launch, see spgan_coltail */
  spgan_coltail tail;
  /* Tile geometry.  0: automatic -- 256 x 256 tiles (csrc/gemm_wide.hip) for large aligned products whose tiles fill the chip, the
   * 128-row kernels otherwise; 1: 128-row kernels only; 2: 256 x 256 tiles whenever the problem is eligible (M % 256 == 0,
   * N % 256 == 0, K % 32 == 0, 16-byte aligned rows, fp32 operands, no per-edge mode / batching): for tests and A/B runs. */
  int tile_hint;
  /* p_group_rows > 0: the rows form M / p_group_rows groups of p_group_rows consecutive rows, and p_scale / p_shift hold one vector per
   * group ([groups, K], row g for the rows of group g) -- several passes of a network with their own train-mode BatchNorm statistics
   * evaluated as ONE product (D(real), D(fake) and D(x_hat) of a D step).  A multiple of 128 (of 256 for the 256 x 256-tile kernel),
   * M a multiple of it, M > 64.  0: one vector for all rows.  Also accepted without a prologue (a_mode PLAIN): then it only tells the
   * automatic tile-size rule (tile_hint 0) to decide from the rows of ONE group, so that a grouped launch runs the kernel its groups
   * would run as separate calls (bit-identical results). */
  int p_group_rows;
  /* A2 != NULL (with a_mode = SPGAN_A_AFFINE_LRELU, no sparse addend): the operand is a = A*p_scale[k] + A2*p_scale2[k] + p_shift[k]
   * -- two tensors of the same shape [M,K] (leading dimensions lda, lda2), no activation (p_slope is ignored): the BatchNorm-backward
   * tensor dy = p*g + q*y + r (spgan_bn_bwd_coeffs) evaluated on the operand load of its consumers instead of by a pass of its own.
   * Epilogues LINEAR / BNBWD / EDGE_BNBWD; M > 64; 128-row kernels only. */
  const float* A2; int lda2; const float* p_scale2;
  /* 16-bit operand / result storage, "f16" operand mode (mfma_f16 == 1) only:
   * a_half = 1: A points at IEEE fp16 values (lda in elements; a_mode PLAIN): staged into LDS as they are (128-row kernels);
   * y_bf16 = 1: Y points at bfloat16 storage (ldy in elements; LINEAR epilogue of the 256 x 256-tile kernel: spgan_gemm_nt_y16_ok()). */
  int a_half, y_bf16;
  /* y_half = 1: Y points at IEEE fp16 storage (ldy in elements; same kernels and conditions as y_bf16; also the edge operand mode: the
   * EdgeBlock's h2pre, a pre-BatchNorm activation whose statistics are still taken from the fp32 accumulators).
   * a_half = 1 together with A2 (epi_mode EDGE_BNBWD): A points at bfloat16 values (g2 of spgan_edge_attend_bwd_b), A2 at fp16 values
   * (h2pre): the EdgeBlock's lazy BatchNorm-backward operand p*A + q*A2 + r with both tensors in 16-bit storage. */
  int y_half;
  /* w_image != NULL (mfma_f16 == 2 only): the split-bf16 image of W written by spgan_split_bf16x3_image (same N, K): the 256-row-tile kernel
   * copies W's three bf16 planes from it instead of splitting the fp32 rows of W again in every workgroup (weights: split once per optimiser
   * step).  Kernels that do not use it ignore it; W must still be given. */
  const void* w_image;
  /* gout_add != NULL (epi_mode BNBWD, a_mode AFFINE_LRELU, mfma_f16 == 2 on a problem its 256-row-tile kernel takes; SPGAN_EINVAL otherwise):
   * the stored tile is gout_add[m, :] + gout_scale[:] * g[m, :] (the statistics stay those of g), as spgan_gemm_dual_args.gout_add: phase B of
   * the double backward hands X = xbarA + gamma*g to the next BatchNorm backward. */
  const float* gout_add; int ld_gout_add; const float* gout_scale;
} spgan_gemm_nt_args;
/* 1 when spgan_gemm_nt will honour y_bf16 for this problem (it runs on the 256 x 256-tile kernel with fp16 operands) */
int spgan_gemm_nt_y16_ok(const spgan_gemm_nt_args* a);
/* 1 when spgan_gemm_nt will run this problem on the M <= 64 kernel, i.e. when `tail.enabled` is acceptable (else the launch
 * returns SPGAN_EINVAL for a tail request) */
int spgan_gemm_nt_owns_columns(const spgan_gemm_nt_args* a);
/* number of column blocks (N-tiles) spgan_gemm_nt uses for this problem */
int spgan_gemm_nt_col_blocks(const spgan_gemm_nt_args* a);

int spgan_gemm_nt(const spgan_gemm_nt_args* a, spgan_stream_t s);
/* Split-bf16 image of a row-major fp32 matrix W [N,K] (N % 128 == 0, K % 16 == 0): every value as three bfloat16 terms hi + mid + lo (exact:
 * round to nearest at each level, each residual representable), laid out as the LDS tiles of the split-bf16 gemm_nt read them:
 * image[k / 16][plane][n][16 bf16], the two 16-byte halves of a row swapped where bit 3 of n is set, rows whose 32-row tile index is odd (bit 5 of
 * n) negated (the kernel's sign checkerboard, csrc/gemm_wide3.hip).  spgan_split_bf16x3_image_bytes: 6*N*K. */
size_t spgan_split_bf16x3_image_bytes(int N, int K);
int spgan_split_bf16x3_image(const float* W, int ldw, int N, int K, void* image, spgan_stream_t s);
/* 1 when spgan_gemm_nt would read a w_image for this problem (worth making one) */
int spgan_gemm_nt_uses_w_image(const spgan_gemm_nt_args* a);
/* pooled[b,c] = max_n lrelu(scale[c]*y[b*rows+n, c] + shift[c], slope) from the tile partials above (rows % 128 == 0, so
 * that no tile straddles two shapes): scale >= 0 takes the tile maxima, scale < 0 the minima.  argmax = global row,
 * yarg = the pre-BatchNorm value there.  Ties go to the lowest row of equal PRE-activation values; scale == 0 (all rows tie):
 * argmax = the first row like torch.max, yarg = the column maximum (y at the first row is not available without Y). */
int spgan_pool_finalize(const float* pool_val, const int32_t* pool_arg, int B, int rows, int C, const float* scale, const float* shift,
                        float slope, float* pooled, int32_t* argmax, float* yarg, spgan_stream_t s);
/* The same for B = groups * shapes_per_group shapes that belong to `groups` passes with their own BatchNorm: scale / shift are
 * [groups, group_stride >= C] (vector g for the shapes of group g).  relative_rows != 0: argmax counts rows from the first row of the
 * shape's own group (what a per-group view of the batched tensors needs). */
int spgan_pool_finalize_groups(const float* pool_val, const int32_t* pool_arg, int B, int rows, int C, const float* scale, const float* shift,
                               int group_stride, int shapes_per_group, float slope, float* pooled, int32_t* argmax, float* yarg,
                               int relative_rows, spgan_stream_t s);

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
  /* Optional prologue on A (per column of A): a = A*a_scale[c] + a_shift[c] + (a_sp_arg[b,c]==m ? a_sp_val[b,c] : 0), b = m / a_sp_rows.
   * a_scale == NULL: A is used as is. */
  const float* a_scale; const float* a_shift; const float* a_sp_val; const int32_t* a_sp_arg; int a_sp_rows;
  /* 1: only write the split partials into ws; the caller finishes C = beta*C + sum(partials) later with
   * spgan_splitk_reduce_multi (one launch for all the weight gradients of a backward pass).  Not with a_sp_val.  Default 0. */
  int defer_reduce;
  /* 1: both operands are rounded to bfloat16 (after the fp32 prologues) when they are staged into LDS and multiplied on the bf16
   * matrix pipe, fp32 accumulation / partials / reduction (BASELINE configs[4]; bf16 rather than fp16: per-point gradients of
   * magnitude 1e-8 must not flush).  Honoured for 16-byte aligned operands with Na, Nb, lda, ldb multiples of 4 outside the
   * skinny (Na or Nb <= 4) path; ignored otherwise.  Default 0: exact fp32 products.
   * 2: every fp32 operand value split exactly into three bfloat16 terms, the six leading cross products on the bf16 matrix pipe, fp32
   * accumulation -- fp32-equivalent products (dropped terms <= 2^-26 relative), as spgan_gemm_nt_args.mfma_f16 == 2.  Honoured where the output
   * tiles as 256 x 256, 256 x 128 or 128 x 256 (Na, Nb multiples of 128, not both odd multiples), M % 32 == 0, fp32-stored operands, b_mode PLAIN /
   * AFFINE_LRELU with slopes in [0, 1]: csrc/gemm_tn_wide3.hip, with its own split plan (spgan_gemm_tn_splits_lp / _ws_bytes_lp below); every
   * other problem runs the exact fp32 kernel on that plan. */
  int mfma_lp;
  /* A2 != NULL (needs a_scale, a_shift; not with a_sp_val): a = A*a_scale[c] + A2*a_scale2[c] + a_shift[c] -- the same two-tensor
   * operand as spgan_gemm_nt_args.A2, on the A side of the weight-gradient product. */
  const float* A2; int lda2; const float* a_scale2;
  /* != NULL: also write the column sums of the (transformed) A operand over the rows of every split: a_colsum_ws [splits, Na]
   * (splits = spgan_gemm_tn_splits(M, Na, Nb)); their fixed-order sum over the splits -- e.g. one more entry (Na = 1, Nb = Na) of
   * spgan_splitk_reduce_multi -- is colsum(A): the bias gradient that belongs to this weight gradient, without a pass of its own.
   * Not with a_sp_val, not on the Na / Nb <= 4 streaming path. */
  float* a_colsum_ws;
  /* a_lrelu = 1 (needs a_scale, a_shift; not with A2 / a_sp_val): the A-side prologue is a = lrelu(A*a_scale[c] + a_shift[c], a_slope)
   * -- a BatchNorm + LeakyReLU'd activation as the A operand without materialising it (the Gram matrix a^T a of the collapsed
   * backward takes the same pre-activation tensor on both sides).  0: the affine map alone. */
  int a_lrelu; float a_slope;
  /* b_half = 1 (with mfma_lp == 1, b_mode PLAIN): B points at IEEE fp16 values (ldb in elements). */
  int b_half;
  /* a_half = 1 (with mfma_lp == 1): A points at bfloat16 values, A2 (when given) at IEEE fp16 values (lda / lda2 in elements). */
  int a_half;
} spgan_gemm_tn_args;

/* Coefficient vectors of the BatchNorm backward as an affine combination of two tensors (Generator.py:58-67 / Discriminator.py:57-79
 * backward): dy = gamma*invstd*(g - S0/count - xhat*S1/count), xhat = (y - mean)*invstd  ==  p*g + q*y + r  with
 *   p = gamma*invstd,  q = -p*invstd*S1/count,  r = -p*S0/count - q*mean        (gamma == NULL: 1)
 * sums = [S0 | S1] (2C); coef [3, C] = [p | q | r].  Consumers: spgan_gemm_nt_args.A2 / spgan_gemm_tn_args.A2. */
/* spgan_colstats_finalize (mode 1: plain sums, one group) that ALSO emits those coefficient vectors from the sums it just merged:
 * the finalize launch behind a BNBWD-epilogue GEMM hands the lazy operand to the next products without a launch of its own. */
int spgan_colstats_finalize_bnbwd(const float* partials, int tiles, int C, int G, int tile_rows, const float* mean, const float* invstd,
                                  const float* gamma, float count, float* s0, float* s1, float* coef, spgan_stream_t s);
/* spgan_colstats_finalize (mode 1, one group) that ALSO runs phase B of the BatchNorm double backward on the sums it just merged
 * (spgan_bn_dbl_phaseb_sums with s0 / s1 := the merged sums): -> s0, s1 [C], sums2C, dgamma */
int spgan_colstats_finalize_phaseb(const float* partials, int tiles, int C, int G, int tile_rows, const float* U0, const float* U1, const float* Ugz,
                                   const float* S0, const float* S1, const float* gamma, const float* invstd, int count, float* s0, float* s1,
                                   float* sums2C, float* dgamma, spgan_stream_t s);
int spgan_bn_bwd_coeffs(const float* sums, const float* mean, const float* invstd, const float* gamma, int C, float count, float* coef, spgan_stream_t s);

size_t spgan_gemm_tn_ws_bytes(int M, int Na, int Nb);
/* The same, and the number of split partials, for a launch with spgan_gemm_tn_args.mfma_lp = mfma_lp: the split-bf16 kernel (mfma_lp == 2) works
 * on larger output tiles and has its own split plan; for mfma_lp 0 / 1 these equal spgan_gemm_tn_ws_bytes / spgan_gemm_tn_splits. */
size_t spgan_gemm_tn_ws_bytes_lp(int M, int Na, int Nb, int mfma_lp);
int spgan_gemm_tn_splits_lp(int M, int Na, int Nb, int mfma_lp);
int spgan_gemm_tn(const spgan_gemm_tn_args* a, spgan_stream_t s);
/* `count` (<= 4) streaming products with a narrow B (Nb <= 4: the weight gradient of D's first conv against the three input coordinates) of ONE
 * shape as one launch, partials only (defer_reduce != 0 required; summed by spgan_splitk_reduce_multi): real / fake / double-backward pass of a
 * grouped D step.  A plain or two-tensor (A2) A operand. */
int spgan_gemm_tn_skinny_multi(const spgan_gemm_tn_args* a, int count, spgan_stream_t s);

/* The backward of one 1x1-conv layer behind a train-mode BatchNorm + LeakyReLU as ONE launch (csrc/gemm_dual.hip): the weight-gradient
 * product and the input-gradient product from ONE staging of the incoming gradient tile -- what autograd does with two convolution
 * backward kernels for conv_w.3 of an EdgeBlock (Generation/Generator.py:56-63,78), mlps.3 / mlps.6 of the Discriminator and the
 * collapsed form of its fc2.0 (Generation/Discriminator.py:55-65,77-81,104); before: spgan_gemm_tn + spgan_gemm_nt with the BNBWD /
 * EDGE_BNBWD epilogue.
 *   dy[m,:]  = A[m,:]                                          a_mode 0 (dense)                      [M, Na]
 *            = A[m,:]*p + A2[m,:]*q + r                        a_mode 1 (the lazy BatchNorm-backward operand of spgan_bn_bwd_coeffs)
 *            = lrelu(A[m,:]*p + r, a_slope)                    a_mode 2 (an activation formed on load: the collapsed layer's a3)
 *   pre[m,:] = B[m,:]                                                                               [M, Nb]   the previous layer's pre-BatchNorm output, or
 *            = B[e_idx[m],:] - B[m / e_k,:] + e_bias   (e_idx != NULL: per-edge operand, B = the point tensor, Nb = 64)
 *   ws[run]  = sum over the rows of run `run` of dy^T . lrelu(pre*b_scale + b_shift, slope)          [runs, Na, Nb] partials of dW
 *   G[m,:]   = (dy[m,:] . W + bias + rowadd[m,:]) * (pre*b_scale + b_shift > 0 ? 1 : slope)          [M, Nb]   (bias [Nb], rowadd [M, Nb]: optional)
 *   stats    = per run (sum G, sum G*xhat), xhat = (pre - b_mean)*b_invstd                           [runs, Nb, 2] plain sums
 *   colsum_ws (optional) = per run the column sums of dy                                             [runs, Na]
 * runs = spgan_gemm_dual_wgs(M, Na, Nb, e_k) (0: shape not supported -- (Na, Nb) = (128, 64), (256, 128) or (256, 256), M % 32 == 0,
 * M >= 8192, e_k 10 only with (128, 64); the caller then uses the two separate launches).  dW = the fixed-order sum of the partials
 * (spgan_splitk_reduce_multi with splits = runs); the statistics are finished by spgan_colstats_finalize(_bnbwd) with tiles = runs,
 * tile_rows = spgan_gemm_dual_rows_per_wg(M, Na, Nb).  fp32 operands, 16-byte aligned rows. */
typedef struct spgan_gemm_dual_args {
  const float* A; int lda;
  const float* A2; int lda2; const float* p; const float* q; const float* r;
  const float* W; int ldw;                       /* [Na, Nb]: the layer's weight as stored (row = output channel) */
  const float* B; int ldb;
  const int32_t* e_idx; int e_k; const float* e_bias;
  const float* b_scale; const float* b_shift; const float* b_mean; const float* b_invstd; float slope;
  float* G; int ldg;
  float* stats; float* ws;
  int M, Na, Nb;
  int a_mode; float a_slope;
  const float* bias; const float* rowadd; int ld_rowadd;
  float* colsum_ws;
  /* optional: the stored tile is gout_add[m, :] + gout_scale[:] * G[m, :] (statistics: of G).  The double backward's phase B: the adjoint
   * X = xbarA + gamma*g that the next BatchNorm backward consumes leaves this launch, not a pass of its own. */
  const float* gout_add; int ld_gout_add; const float* gout_scale;
} spgan_gemm_dual_args;
int spgan_gemm_dual_wgs(int M, int Na, int Nb, int e_k);
int spgan_gemm_dual_rows_per_wg(int M, int Na, int Nb);   /* rows per run = the `tile_rows` of the statistics partials (tiles = runs = ceil(M / rows)) */
int spgan_gemm_dual(const spgan_gemm_dual_args* a, spgan_stream_t s);
/* Grouped launch: `count` (<= SPGAN_GROUP_MAX) independent spgan_gemm_dual problems of ONE geometry (equal M, Na, Nb; plain pre tensors) as a
 * single grid, problem g on the workgroups [g*grid1, (g+1)*grid1): every problem's results are bit-identical to its stand-alone launch.
 * The D step issues each Discriminator layer's backward for the real pass, the fake pass and phase B of the penalty's double backward
 * (Generation/Discriminator.py:97-115 called three times per D step, Common/gradient_penalty.py:19-37); count == 1 is spgan_gemm_dual. */
#define SPGAN_GROUP_MAX 4
int spgan_gemm_dual_multi(const spgan_gemm_dual_args* a, int count, spgan_stream_t s);
/* The launches around a grouped spgan_gemm_dual_multi, grouped the same way (every problem runs the body of its stand-alone kernel:
 * bit-identical results).
 * spgan_colstats_finalize_multi: `count` mode-1 finalize launches (plain sums s0/s1 of [tiles, C, 2] records) with per-problem tail --
 *   kind 0: none;  kind 1: spgan_colstats_finalize_bnbwd's coefficients (mean, invstd, gamma | NULL, count -> coef [3,C]);
 *   kind 2: spgan_colstats_finalize_phaseb (U0, U1, Ugz, S0, S1, gamma, invstd, count -> sums [2C], dgamma [C]).  tiles < 2048. */
typedef struct spgan_colfinalize_args {
  const float* partials; int tiles, C, G, tile_rows;
  float* s0; float* s1;
  int kind;
  const float* mean; const float* invstd; const float* gamma; float count;
  float* coef;
  const float* U0; const float* U1; const float* Ugz; const float* S0; const float* S1;
  float* sums; float* dgamma;
  float* pb_coef;          /* kind 2, optional (needs mean): coef [3,C] of the lazy operand p*X + q*y + r = the BatchNorm backward (gamma = 1) with `sums` */
} spgan_colfinalize_args;
int spgan_colstats_finalize_multi(const spgan_colfinalize_args* a, int count, spgan_stream_t s);
/* spgan_pool_bwd_stats_prep for `count` passes (the real and the fake pass behind one grouped forward) as one launch. */
typedef struct spgan_pool_bwd_args {
  const float* gpool; const float* pooled; const int32_t* argmax; const float* y; int ld;
  const float* mean; const float* invstd; float slope; int B, C;
  const float* gamma; int count;
  float* gval; float* sums; float* alpha; float* beta; float* cg;
} spgan_pool_bwd_args;
int spgan_pool_bwd_stats_prep_multi(const spgan_pool_bwd_args* a, int count, spgan_stream_t s);

/* Row-sparse products with the max-pool gradient pattern S (Discriminator.py:104 backward): one (value, row) pair per
 * shape b and channel c, val/arg [B, Cs], arg = global row (b*rows + local).  They let the backward of the layer in front
 * of the pool collapse algebraically (DESIGN.md "collapsed L4 backward"): with dz = alpha*y + beta + S and y = a.W^T + b,
 *   dz.W      = a.(W^T diag(alpha) W) + (alpha*b + beta).W + S.W          -- a [M,Cin]x[Cin,Cin] GEMM instead of [M,Cout]x[Cout,Cin]
 *   dz^T.a    = diag(alpha).W.(a^T a) + (alpha*b + beta) (x) colsum(a) + S^T.a
 * spgan_sparse_rows_nt: E[m, n]  = sum_{c: arg[b,c]==m} val[b,c] * W[c, n]      E [B*rows, N] is fully written (zero rows too)
 * spgan_sparse_rows_tn: C[c, n] += sum_b val[b,c] * pro(Bm)[arg[b,c], n]         pro = lrelu(x*p_scale[n]+p_shift[n], p_slope) or none
 * Both sum in ascending c / b order (deterministic). */
/* The two weight-only operands of that collapsed backward in one launch (csrc/collapse.hip), W [C, K]:
 *   G[i,j] = sum_c W[c,i]*alpha[c]*W[c,j]   (W^T diag(alpha) W, [K,K]);   cvec[j] = sum_c (alpha[c]*bias[c] + beta[c])*W[c,j]   (cvec NULL: skipped)
 * C % 256 == 0, K % 32 == 0; deterministic (fixed-order sums). */
int spgan_wt_diag_w(const float* W, int ldw, int C, int K, const float* alpha, const float* beta, const float* bias, float* G, int ldg,
                    float* cvec, spgan_stream_t s);
/* Both of the above's launches for one collapsed backward pass -- spgan_wt_diag_w (nprob = 1 .. 4 problems on the same W: the double backward needs
 * W^T diag(c1) W and W^T diag(c2) W; the grouped D step adds the real and the fake pass) and nsparse x spgan_sparse_rows_nt (E [B*rows, K] = S.W, Cs = C) -- as ONE launch: the weight-only part is latency-bound on
 * a fraction of the chip and finishes under the part that streams E out.  Results bit-identical to the separate launches.  cvec[p] / beta[p] /
 * bias[p] NULL: no cvec for problem p. */
typedef struct spgan_collapse_prep_args {
  const float* W; int ldw, C, K;
  int nprob;                                            /* 1 .. SPGAN_GROUP_MAX weight problems on the same W */
  const float* alpha[SPGAN_GROUP_MAX]; const float* beta[SPGAN_GROUP_MAX]; const float* bias[SPGAN_GROUP_MAX];
  float* G[SPGAN_GROUP_MAX]; int ldg; float* cvec[SPGAN_GROUP_MAX];
  int nsparse;                                          /* 1 .. SPGAN_GROUP_MAX sparse-row products E[q] = S[q].W, equal B and rows */
  const float* sp_val[SPGAN_GROUP_MAX]; const int32_t* sp_arg[SPGAN_GROUP_MAX]; int B, rows;
  float* E[SPGAN_GROUP_MAX]; int lde;
} spgan_collapse_prep_args;
int spgan_collapse_prep(const spgan_collapse_prep_args* a, spgan_stream_t s);
/* The weight gradient of the collapsed layer, all of its terms in one launch (csrc/collapse.hip):
 *   out[a,n] (+)= a1[a] * sum_k W[a,k]*X1[n,k]  +  (a1[a]*b1[a] + d1[a]) * v1[n]  +  a2[a] * sum_k W[a,k]*X2[n,k]
 *                 +  sum_b sp_val[b,a] * pro(Bm)[sp_arg[b,a], n]          (pro = lrelu(x*p_scale[n] + p_shift[n], p_slope), or identity)
 * W [C,K], X1 [N,K] (a Gram matrix / q^T a of the double backward), K <= 256, C, N, K multiples of 32, 16-byte aligned rows.  Optional: the rank-1 term
 * (v1, b1, d1), the second product (X2 [N,K], or [K,N] with x2_t = 1: used transposed; a2), the sparse term (sp_val / sp_arg [B,C], global rows of
 * Bm [B*rows, N]; B <= 64; shapes in ascending order), T [C,N] = the raw first product, accumulate != 0: out += .  Replaces spgan_gemm_nt +
 * spgan_rowscale_outer + spgan_sparse_rows_tn (and, in the double backward, a transpose, a second product and two axpby). */
typedef struct spgan_wgrad_collapse_args {
  const float* W; int ldw, C, K;
  const float* X1; int ldx1;
  const float* a1; const float* b1; const float* d1; const float* v1;
  const float* X2; int ldx2, x2_t; const float* a2;
  const float* sp_val; const int32_t* sp_arg; int B, rows;
  const float* Bm; int ldb; const float* p_scale; const float* p_shift; float p_slope;
  float* T; int ldt;
  float* out; int ldo, N, accumulate;
} spgan_wgrad_collapse_args;
int spgan_wgrad_collapse(const spgan_wgrad_collapse_args* a, spgan_stream_t s);
/* `count` (<= SPGAN_GROUP_MAX) problems with equal C, N, K as one launch (count == 1: spgan_wgrad_collapse) */
int spgan_wgrad_collapse_multi(const spgan_wgrad_collapse_args* a, int count, spgan_stream_t s);
int spgan_sparse_rows_nt(const float* val, const int32_t* arg, int B, int rows, int Cs, const float* W, int ldw, int N, float* E, int lde,
                         spgan_stream_t s);
int spgan_sparse_rows_tn(const float* val, const int32_t* arg, int B, int rows, int Cs, const float* Bm, int ldb, int Nb,
                         const float* p_scale, const float* p_shift, float p_slope, float* C, int ldc, spgan_stream_t s);
/* out[m,c] = lrelu(X[m,c]*scale[c] + shift[c], slope)   (train-mode BatchNorm + LeakyReLU output, Discriminator.py:57-64) */
int spgan_affine_act(const float* X, int ldx, size_t M, int C, const float* scale, const float* shift, float slope, float* out, spgan_stream_t s);
/* out[r,c] = a[r]*X[r,c] + (a[r]*b[r] + d[r])*v[c]   (v == NULL: first term only); weight-shaped [R,C] tensors */
/* accumulate != 0: out += that (the terms of a weight gradient summed in place instead of by axpby launches) */
int spgan_rowscale_outer(const float* X, int ldx, int R, int C, const float* a, const float* b, const float* d, const float* v, float* out,
                         int ldo, int accumulate, spgan_stream_t s);

/* Column reductions over row groups (group = G consecutive rows; M % G == 0).  Partials are
 * [groups * ceil(G/128)][C][2] floats: one (a, b) pair per 128-row tile and column -- the format
 * spgan_gemm_nt's `stats` epilogue writes (there: one group of M rows).
 * finalize mode 0 (Welford/Chan): partials (sum, centred M2) -> out0 = mean, out1 = biased variance.
 * finalize mode 1 (plain):         partials (s0, s1)          -> out0 = sum s0, out1 = sum s1.
 * Everything is combined in a fixed order: results are run-to-run deterministic. */
size_t spgan_colreduce_ws_bytes(int M, int C, int G);
int spgan_colstats_finalize(const float* partials, int groups, int tiles_per_group, int C, int G, int mode,
                            int tile_rows /* 0 -> 128 */, float* out0, float* out1, spgan_stream_t s);
/* One group of G rows: finalize (sum, M2) partials AND do the train-mode BatchNorm bookkeeping of spgan_bn_prepare in the
 * same launch (scale, shift, invstd, mean; running stats updated when given). */
int spgan_colstats_finalize_bn(const float* partials, int tiles, int C, int G, int tile_rows, const float* gamma, const float* beta,
                               float eps, float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                               float* invstd, float* mean_out, spgan_stream_t s);
/* spgan_colstats_finalize_bn for `groups` consecutive row groups of G rows each (records [groups * tiles_per_group, C, 2]) that go
 * through the SAME BatchNorm layer one after the other: out [4, groups, C] = scale | shift | invstd | mean, each a contiguous [groups, C]
 * block (what spgan_gemm_nt_args.p_group_rows and spgan_pool_finalize_groups read); the running statistics are updated group after
 * group (group 0 first) exactly as `groups` separate calls would. */
int spgan_colstats_finalize_bn_groups(const float* partials, int groups, int tiles_per_group, int C, int G, int tile_rows, const float* gamma,
                                      const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* out,
                                      spgan_stream_t s);
/* The same for TWO BatchNorm layers whose channels lie side by side in one record set (columns [0,split) -> layer A, [split,C) ->
 * layer B, each with its own gamma/beta/running buffers): the two per-edge BatchNorms of an EdgeBlock (Generator.py:57-58,66-67) from
 * spgan_edge_stats' records in one launch.  out4 [4,C] = scale | shift | invstd | mean.  count_rep: the rows stand for count_rep
 * identical copies (only the unbiased-variance count of the running statistics changes). */
int spgan_colstats_finalize_bn2(const float* partials, int tiles, int C, int G, int tile_rows, int split, const float* gammaA,
                                const float* betaA, float* rmeanA, float* rvarA, const float* gammaB, const float* betaB, float* rmeanB,
                                float* rvarB, float eps, float momentum, int count_rep, float* out4, spgan_stream_t s);
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
/* The same with g := g + g2scale[c]*g2 formed on the fly (contiguous [M,C], C % 4 == 0, 16-byte aligned): the double backward's
 * xbarA + gamma*g (BatchNorm backward of the gradient-penalty graph) without a separate elementwise pass. */
int spgan_bn_bwd_apply2(const float* g, const float* g2, const float* g2scale, const float* y, int M, int C, const float* mean,
                        const float* invstd, const float* gamma, const float* sums, int count, float* dy, spgan_stream_t s);

/* ------------------------------------------------------------------------------------------
 * Global max over the N points of each shape (Generator.py:183; Discriminator.py:104) fused with
 * the preceding affine + LeakyReLU:  out[b,c] = max_n lrelu(y[b*N+n,c]*scale[c]+shift[c]).
 * ---------------------------------------------------------------------------------------- */
int spgan_maxpool(const float* y, int ld, int B, int N, int C, const float* scale, const float* shift, float slope,
                  float* out, int32_t* argmax, spgan_stream_t s);

/* ------------------------------------------------------------------------------------------
 * EdgeBlock (Generation/Generator.py:47-88), restructured per point:
 *   conv_w.0(x_j - x_i) = (P_j - P_i) + b1,  conv_x.0([x_i, x_j-x_i]) = (R_i + Q_j) + bx,
 *   PQR[M, H+2F] = x . Wcat^T,  Wcat = [W1; Wd; Wc-Wd],  Wx = [Wc | Wd],  H = F/2.
 * ---------------------------------------------------------------------------------------- */
/* WcatT (nullable) [C, H+2F] = Wcat^T from the same launch: the operand of the block's input-gradient GEMM (dx = dPQR . Wcat). */
int spgan_edge_wcat(const float* Ww0 /*[H,C]*/, const float* Wx /*[F,2C]*/, int H, int F, int C, float* Wcat, float* WcatT, spgan_stream_t s);
/* conv_out.weight [F,F,1,k] (Generator.py:70: Conv2d(F, F, [1,k])) in the two layouts the point-major EdgeBlock multiplies with:
 * Wo [F, k*F] with K index r*F + c (the order of the per-point feature row T[m, r*F + c]) for the forward product, and WoT (nullable)
 * [k*F, F] = Wo^T for the input gradient dT = dout . Wo -- one launch instead of a permuted copy now and a transpose in the backward. */
int spgan_conv_out_weight_pm(const float* w, int F, int k, float* Wo, float* WoT, spgan_stream_t s);
int spgan_edge_wcat_bwd(const float* dWcat, int H, int F, int C, float* dWw0, float* dWx, spgan_stream_t s);
/* BatchNorm2d statistics over the M*k edges of both per-edge pre-activations (Generator.py:58,67):
 * partials [ceil(M/32)][H+F][2] in the finalize-mode-0 format with tile_rows = spgan_edge_stats_tile_rows(k). */
int spgan_edge_stats_tile_rows(int k);
int spgan_edge_stats(const float* PQR, int ld, const int32_t* idx, int M, int k, int H, int F, const float* b1, const float* bx,
                     float* partials, spgan_stream_t s);
/* T[i, r*F+f] = softmax_r(lrelu(h2pre[i,r,f]*sc2+sh2)) * lrelu(((R_i+Q_j)+bx)*scx+shx)    (Generator.py:79,81-82) */
int spgan_edge_attend_fwd(const float* h2pre, const float* sc2, const float* sh2, const float* PQR, int ld, int H, int F,
                          const int32_t* idx, int M, int k, const float* bx, const float* scx, const float* shx, float slope,
                          float* T, spgan_stream_t s);
/* Backward of edge_attend: g2/gy = gradients w.r.t. the two BatchNorm outputs, and plain-sum partials
 * [ceil(M/spgan_edge_attend_bwd_tile_points())][2F][2]: col f -> (sum g2, sum g2*xhat2), col F+f -> (sum gy, sum gy*xhaty). */
/* 16-bit storage variants for the "f16" operand mode (BASELINE configs[4]).  Forward: T is written as IEEE fp16 (consumed by conv_out's
 * products: spgan_gemm_nt_args.a_half / spgan_gemm_tn_args.b_half).  Backward: dT is read as bfloat16 (written by spgan_gemm_nt_args.y_bf16: a
 * gradient keeps fp32's exponent range); g2 -- a GEMM operand only (a_half of spgan_gemm_tn_args / spgan_gemm_nt_args) -- and gy -- consumed
 * by spgan_edge_scatter_b -- are written as bfloat16; the partials stay float.  h2_half = 1: h2pre lies in memory as fp16 (written by the
 * edge GEMM with spgan_gemm_nt_args.y_half).  k = 10 and F % 4 == 0 only; everything else as the fp32 entry points. */
int spgan_edge_attend_fwd_h(const void* h2pre, int h2_half, const float* sc2, const float* sh2, const float* PQR, int ld, int H, int F,
                            const int32_t* idx, int M, int k, const float* bx, const float* scx, const float* shx, float slope, uint16_t* T_f16,
                            spgan_stream_t s);
int spgan_edge_attend_bwd_b(const uint16_t* dT_bf16, const void* h2pre, int h2_half, const float* sc2, const float* sh2, const float* mean2,
                            const float* inv2, const float* PQR, int ld, int H, int F, const int32_t* idx, int M, int k, const float* bx,
                            const float* scx, const float* shx, const float* meanx, const float* invx, float slope, uint16_t* g2_bf16,
                            uint16_t* gy_bf16, float* partials, spgan_stream_t s);
/* spgan_edge_scatter with the gy operand as written by spgan_edge_attend_bwd_b (bfloat16; k = 10) */
int spgan_edge_scatter_b(const float* g1, const uint16_t* gy_bf16, const float* PQR, int ld, int H, int F, const int32_t* idx,
                         const int32_t* rowptr, const int32_t* src, int M, int k, const float* b1, const float* mean1, const float* inv1,
                         const float* gam1, const float* sums1, const float* bx, const float* meanx, const float* invx, const float* gamx,
                         const float* sumsx, float* dPQR, spgan_stream_t s);
int spgan_edge_attend_bwd_tile_points(void);
int spgan_edge_attend_bwd(const float* dT, const float* h2pre, const float* sc2, const float* sh2, const float* mean2,
                          const float* inv2, const float* PQR, int ld, int H, int F, const int32_t* idx, int M, int k,
                          const float* bx, const float* scx, const float* shx, const float* meanx, const float* invx,
                          float slope, float* g2, float* gy, float* partials, spgan_stream_t s);
/* BatchNorm backward of both per-edge pre-activations fused with the reduction onto points (gather over the CSR
 * in-edge lists; the backward obligation of modules.py:708-720): dPQR[M, H+2F]. sums* = [sum g | sum g*xhat]. */
int spgan_edge_scatter(const float* g1, const float* gy, const float* PQR, int ld, int H, int F, const int32_t* idx,
                       const int32_t* rowptr, const int32_t* src, int M, int k, const float* b1, const float* mean1,
                       const float* inv1, const float* gam1, const float* sums1, const float* bx, const float* meanx,
                       const float* invx, const float* gamx, const float* sumsx, float* dPQR, spgan_stream_t s);

/* ------------------------------------------------------------------------------------------
 * AdaptivePointNorm (Generator.py:24-45): out = gamma*xhat + beta with xhat = InstanceNorm1d(lrelu(x, slope))
 * (eps 1e-5, biased variance over the N points of each shape; slope = 1 for the bare module, 0.2 when the
 * Generator's lrelu1/lrelu2 is fused in, Generator.py:175-176,179-180) and [gamma|beta] = gb[M,2C].
 * ---------------------------------------------------------------------------------------- */
int spgan_adain_fwd(const float* x, int M, int C, int N, float slope, const float* imean, const float* ivar, float eps,
                    const float* gb, float* out, spgan_stream_t s);
/* dgb = [dout*xhat | dout]; partials [(M/N)*ceil(N/128)][C][2] = (sum dxh, sum dxh*xhat), dxh = dout*gamma (finalize mode 1) */
int spgan_adain_bwd1(const float* dout, const float* x, int M, int C, int N, float slope, const float* imean, const float* ivar,
                     float eps, const float* gb, float* dgb, float* partials, spgan_stream_t s);
int spgan_adain_bwd2(const float* dout, const float* x, int M, int C, int N, float slope, const float* imean, const float* ivar,
                     float eps, const float* gb, const float* S0, const float* S1, float* dx, spgan_stream_t s);

/* ------------------------------------------------------------------------------------------
 * Backward through [BatchNorm1d -> LeakyReLU -> global max over N] (Discriminator.py:77-81,104): the incoming
 * gradient is non-zero only at the arg-max rows, the BatchNorm backward makes it dense again.
 * ---------------------------------------------------------------------------------------- */
int spgan_pool_bwd_stats(const float* gpool, const float* pooled, const int32_t* argmax, const float* y, int ld, const float* mean,
                         const float* invstd, float slope, int B, int C, float* gval, float* sums /*[2C]*/, spgan_stream_t s);
/* spgan_pool_bwd_stats followed by spgan_sparse_bn_prep (the coefficients alpha, beta [C] and cg [B,C] of the lazily evaluated BatchNorm
 * backward behind the max-pool) in ONE launch: both are per-channel work on the same [B,C] values. */
int spgan_pool_bwd_stats_prep(const float* gpool, const float* pooled, const int32_t* argmax, const float* y, int ld, const float* mean,
                              const float* invstd, float slope, int B, int C, const float* gamma, int count, float* gval, float* sums /*[2C]*/,
                              float* alpha, float* beta, float* cg, spgan_stream_t s);
int spgan_bn_bwd_apply_sparse(const float* gval, const int32_t* argmax, const float* y, int ld, int M, int C, int N,
                              const float* mean, const float* invstd, const float* gamma, const float* sums, int count, float* dy,
                              spgan_stream_t s);
/* dst[argmax[b,c], c] += dpool[b,c]   (torch.max backward, Generator.py:183) */
int spgan_maxpool_bwd_add(const float* dpool, const int32_t* argmax, int B, int C, float* dst, int ld, spgan_stream_t s);
int spgan_tanh_bwd(const float* dy, const float* y, size_t n, float* out, spgan_stream_t s);
/* out = dy * act'(y) from the activation OUTPUT y (act = SPGAN_ACT_*; in-place LeakyReLU semantics, Generator.py:110-133) */
int spgan_act_bwd(const float* dy, const float* y, size_t n, int act, float slope, float* out, spgan_stream_t s);
/* out[M,C] = 0, out[argmax[b,c], c] = val[b,c];   out[b,c] = src[argmax[b,c], c] */
int spgan_scatter_rows(const float* val, const int32_t* argmax, int B, int C, int M, float* out, spgan_stream_t s);
int spgan_gather_rows(const float* src, int ld, const int32_t* argmax, int B, int C, float* out, spgan_stream_t s);

/* ------------------------------------------------------------------------------------------
 * WGAN-GP double backward through train-mode BatchNorm (Common/gradient_penalty.py:31-35 + .backward();
 * derivation in DESIGN.md).  u = adjoint of the BN-backward output, gz = first-order gradient w.r.t. the BN output.
 *   stats partials [ceil(M/128)][2C][2] (finalize mode 1): col c -> (sum u, sum u*xhat), col C+c -> (sum u*gz, 0)
 *   apply: q = gamma*invstd*(u - U0/M - xhat*U1/M)*lrelu'(y*scale+shift);  xbar = -(gamma*invstd/M)*(u*S1 + gz*U1)
 * ---------------------------------------------------------------------------------------- */
int spgan_bn_dbl_stats(const float* u, const float* y, const float* gz, int M, int C, const float* mean, const float* invstd,
                       float* partials, spgan_stream_t s);
int spgan_bn_dbl_apply(const float* u, const float* y, const float* gz, int M, int C, const float* mean, const float* invstd,
                       const float* scale, const float* shift, float slope, const float* gamma, const float* S1, const float* U0,
                       const float* U1, float* q, float* xbar, spgan_stream_t s);
/* ------------------------------------------------------------------------------------------
 * Losses on the [B,1] logits (Common/loss_utils.py:727-802 gen_loss, 854-972 dis_loss) with both logit
 * gradients in the same launch.  mode: 0 ls (default, config.py:72), 1 wgan, 2 hinge, 3 gan (BCE-with-logits);
 * which: 0 discriminator loss, 1 generator loss.  Labels [B] or NULL (ls only; the reference's F.mse_loss
 * broadcast of [B,1] logits against [B] labels to [B,B] is kept).
 * out5 = (loss, fake term, real term, real_acc, fake_acc).
 * ---------------------------------------------------------------------------------------- */
int spgan_gan_loss(int mode, int which, const float* d_real, const float* d_fake, const float* real_label, const float* fake_label,
                   int B, float* out5, float* g_real, float* g_fake, spgan_stream_t s);
/* WGAN-GP (Common/gradient_penalty.py:19-37): x_hat = real + alpha[b]*(fake-real);
 * penalty = lambda*mean_b(((||g_b||-gamma)/gamma)^2) and its gradient w.r.t. g (times upstream[0] if given). */
int spgan_lerp_rows(const float* real, const float* fake, const float* alpha, int B, size_t L, float* out, spgan_stream_t s);
int spgan_gp_penalty_fwd(const float* g, int B, size_t L, float gamma, float lambda, float* norms, float* loss, spgan_stream_t s);
int spgan_gp_penalty_bwd(const float* g, const float* norms, int B, size_t L, float gamma, float lambda, const float* upstream,
                         float* v, spgan_stream_t s);
/* spgan_gp_penalty_fwd and spgan_gp_penalty_bwd (upstream = 1) as two launches instead of three: row norms, then v and the penalty together;
 * optional loss_total[0] = loss_add[0] + penalty (the D step's reported loss).  Values bit-identical to the separate calls. */
int spgan_gp_penalty_fwd_bwd(const float* g, int B, size_t L, float gamma, float lambda, float* norms, float* loss, const float* loss_add,
                             float* loss_total, float* v, spgan_stream_t s);
/* ------------------------------------------------------------------------------------------
 * Ball-query / grouping family (Common/pointnet_util.py, Common/pointconv_util.py; orphans in the reference,
 * named by the north star).  xyz/new_xyz/points are [B,N,C] row-major like the reference; indices are int64.
 * ---------------------------------------------------------------------------------------- */
/* out[b,n,m] = -2<src_n,dst_m> + |src_n|^2 + |dst_m|^2            pointnet_util.py:19-40 */
int spgan_square_distance(const float* src, const float* dst, int B, int N, int M, int C, float* out, spgan_stream_t s);
/* out[b,s,:] = points[b, idx[b,s], :]  (idx [B,S] or [B,S,K] flattened to S*K)   pointnet_util.py:43-60 */
int spgan_index_points(const float* points, const int64_t* idx, int B, int N, int C, int S, float* out, spgan_stream_t s);
/* iterative farthest point sampling from start[b] (NULL: 0); dist_ws: B*N floats   pointnet_util.py:63-84, pointconv_util.py:60-83 */
int spgan_farthest_point_sample(const float* xyz, int B, int N, int npoint, const int64_t* start, int64_t* out, float* dist_ws,
                                spgan_stream_t s);
/* first nsample indices (ascending) with d^2 <= r^2, padded with the first hit     pointnet_util.py:87-107 */
int spgan_query_ball_point(float radius, int nsample, const float* xyz, const float* new_xyz, int B, int N, int S, int C,
                           int64_t* out, spgan_stream_t s);
/* nsample nearest (self included), ascending (distance, index)                      pointconv_util.py:107-118 */
int spgan_knn_point(int nsample, const float* xyz, const float* new_xyz, int B, int N, int S, int C, int64_t* out, spgan_stream_t s);
/* out[b,s,j,:] = [xyz[b,idx] - center[b,s] | feat[b,idx]]   pointnet_util.py:127-139, pointconv_util.py:186-195 */
int spgan_group_concat(const float* xyz, const float* center, const float* feat, const int64_t* idx, int B, int N, int S, int K,
                       int C, int D, float* out, spgan_stream_t s);
/* Adjoints of the gathers above (the reference gets them from torch indexing: index_put / the side-car's atomicAdd scatter,
 * metrics/pointops/src/grouping/grouping_cuda_kernel.cu:28-45, metrics/pointnet2/src/group_points_gpu.cu:8-30).  Deterministic:
 * spgan_gather_csr lists, per point, the gather slots that read it (ascending), the sums run over those lists in order.
 *   spgan_gather_csr:           idx int64 [B,S] local indices -> rowptr int32 [B*N,2] (begin,end) into src int32 [B*S]; *bad |= 1 on an index outside [0,N)
 *   spgan_scatter_slots:        dpoints[n,c] = sum_{e in slots(n)} dout[e*ld + col0 + c]      (index_points / grouping backward)
 *   spgan_group_center_bwd:     dcenter[q,c] = -sum_j dout[(q*K+j)*ld + c]                    (pointnet_util.py:128, pointconv_util.py:189)
 *   spgan_edge_features_cm_bwd: dx [B,C,N] from dE [B,2C,N,k] (central + "-central" + in-edge terms, Generation/modules.py:708-720) */
int spgan_gather_csr(const int64_t* idx, int B, int S, int N, int32_t* rowptr, int32_t* src, int32_t* bad, spgan_stream_t s);
int spgan_scatter_slots(const float* dout, int ld, int col0, int C, const int32_t* rowptr, const int32_t* src, int BN, float* dpoints,
                        spgan_stream_t s);
int spgan_group_center_bwd(const float* dout, int ld, int Q, int K, int C, float* dcenter, spgan_stream_t s);
int spgan_edge_features_cm_bwd(const float* dE, const int32_t* rowptr, const int32_t* src, int B, int C, int N, int k, float* dx,
                               spgan_stream_t s);
/* Per-channel scalar algebra of the double backward, one launch each:
 *   coeffs out4C = [dgammaA | sbarA | xsum0 | xsum1];  phaseb: sums2C = [xsum0+gamma*s0 | xsum1+gamma*s1+invstd*sbarA], dgamma = dgammaA+s1 */
int spgan_bn_dbl_coeffs(const float* U0, const float* U1, const float* Ugz, const float* S0, const float* S1, const float* gamma,
                        const float* invstd, int C, int count, float* out4C, spgan_stream_t s);
int spgan_bn_dbl_phaseb(const float* coeffs4C, const float* gamma, const float* invstd, const float* s0, const float* s1, int C,
                        float* sums2C, float* dgamma, spgan_stream_t s);
/* spgan_bn_dbl_coeffs followed by spgan_bn_dbl_phaseb in one launch (same arithmetic, the [4,C] coefficient block is not stored) */
int spgan_bn_dbl_phaseb_sums(const float* U0, const float* U1, const float* Ugz, const float* S0, const float* S1, const float* gamma,
                             const float* invstd, const float* s0, const float* s1, int C, int count, float* sums2C, float* dgamma,
                             spgan_stream_t s);
/* Collapsed double backward of the layer in front of the max-pool (Discriminator.py:74-81,104): the dense [M,C]
 * tensors of the generic path are only needed as per-channel sums and at the B*C arg-max positions.
 * spgan_gather_rowdot: out[b,c] = Q[arg[b,c], :] . W[c, :]          (u = q.W^T at the arg-max rows)
 * spgan_rowdot:        out[r]   = X[r,:] . Y[r,:]
 * spgan_bn_dbl_pool:   per channel: U1, Ugz, the phase-A coefficients; t [B,C] = adjoint of the pooled gradient;
 *                      out4C = [dgamma | c1 | c2 | c3], spB [B,C]: phase B's ybar = c1*u + c2*y + c3 + scatter(spB). */
int spgan_gather_rowdot(const float* Q, int ldq, const int32_t* arg, const float* W, int ldw, int B, int C, int K, float* out, spgan_stream_t s);
/* spgan_gather_rowdot (uarg [B,C] = Q[arg[b,c],:] . W[c,:]), spgan_rowdot (quad [C] = W[c,:] . T[c,:]) and U0 [C] = W[c,:] . cq in ONE launch:
 * the three independent per-channel dot products of the collapsed double backward's phase A (Discriminator.py:77-81 behind the max-pool). */
int spgan_dbl_top_dots(const float* Q, int ldq, const int32_t* arg, const float* W, int ldw, const float* T, int ldt, const float* cq, int B, int C,
                       int K, float* uarg, float* quad, float* U0, spgan_stream_t s);
int spgan_rowdot(const float* X, int ldx, const float* Y, int ldy, int R, int K, float* out, spgan_stream_t s);
int spgan_bn_dbl_pool(const float* uarg, const float* gval, const float* yarg, const float* pooled, const float* U0, const float* quad,
                      const float* bias, const float* mean, const float* invstd, const float* gamma, const float* S0, const float* S1,
                      int B, int C, int count, float slope, float* t, float* spB, float* out4C, spgan_stream_t s);
/* Lazy-operand form of the BatchNorm backward behind the max-pool: dy = alpha[c]*y + beta[c] + (argmax hit ? cg[b,c] : 0) */
int spgan_sparse_bn_prep(const float* gval, const float* mean, const float* invstd, const float* gamma, const float* sums, int B,
                         int C, int count, float* alpha, float* beta, float* cg, spgan_stream_t s);
/* out[m,c] = a[m,c] + gamma[c]*b[m,c] */
int spgan_col_scale_add(const float* a, const float* b, const float* gamma, int M, int C, float* out, spgan_stream_t s);
/* dst[t][i] += src[t][i], t < count <= SPGAN_MULTI_MAX, in ONE launch (gradient accumulation into the flat buffer) */
#define SPGAN_MULTI_MAX 64
typedef struct spgan_multi_add_args {
  int count;
  float* dst[SPGAN_MULTI_MAX];
  const float* src[SPGAN_MULTI_MAX];
  int n[SPGAN_MULTI_MAX];
} spgan_multi_add_args;
int spgan_multi_add(const spgan_multi_add_args* a, spgan_stream_t s);
/* dst[t] = ((dst[t] + src[0][t]) + src[1][t]) + src[2][t] with nsrc[t] in 1..3 sources: the per-pass parameter gradients of a grouped backward
 * accumulated in one launch, element for element the sums of nsrc successive spgan_multi_add calls. */
#define SPGAN_MULTI_ADDN_MAX 32
typedef struct spgan_multi_addn_args {
  int count;
  float* dst[SPGAN_MULTI_ADDN_MAX];
  const float* src[3][SPGAN_MULTI_ADDN_MAX];
  int nsrc[SPGAN_MULTI_ADDN_MAX];
  int n[SPGAN_MULTI_ADDN_MAX];
} spgan_multi_addn_args;
int spgan_multi_addn(const spgan_multi_addn_args* a, spgan_stream_t s);
/* The same for pairs that are 3-D strided views of one shape [n0, n1, n2] (n = n0*n1*n2; strides in elements):
 * dst[i0*ds[0] + i1*ds[1] + i2*ds[2]] += src[i0*ss[0] + i1*ss[1] + i2*ss[2]].  A permuted source (the conv_out weight gradient is computed
 * as [F,k,F] and accumulated into [F,F,1,k]) or a column block of the destination; n1 = n2 = 1 is the plain contiguous pair. */
typedef struct spgan_multi_add3_args {
  int count;
  float* dst[SPGAN_MULTI_MAX];
  const float* src[SPGAN_MULTI_MAX];
  int n[SPGAN_MULTI_MAX], n1[SPGAN_MULTI_MAX], n2[SPGAN_MULTI_MAX];
  long ds[3][SPGAN_MULTI_MAX], ss[3][SPGAN_MULTI_MAX];
} spgan_multi_add3_args;
int spgan_multi_add3(const spgan_multi_add3_args* a, spgan_stream_t s);
/* The local step of a ONE-HOP all-reduce of a flat gradient buffer over W ranks (on a fully connected xGMI node an
 * all-to-all is one hop per pair, so reduce-scatter = all-to-all + this sum, all-gather = one more hop; 2 hops instead of a ring's
 * 2(W-1) steps for the 2.3 / 3.9 MB latency-bound messages).  recv [parts, n] holds this rank's chunk as received from every rank
 * (row j from rank j); out[i] = sum_j recv[j, i] in ascending j -- every chunk is summed exactly once, on one rank, in a fixed order, so
 * all ranks end up with bit-identical sums.  recv and out 16-byte aligned.  The exchanges themselves are torch.distributed / RCCL
 * (spgan.parallel.DataParallel(collective="one_hop")). */
int spgan_reduce_chunks(const float* recv, int parts, size_t n, float* out, spgan_stream_t s);
/* ---- allreduce_flat: the data-parallel exchange of a train step -- what replaces nn.DataParallel's per-call
 * scatter / parameter broadcast / gather (Generation/model.py:79-84): ONE all-reduce (sum, in place) of a network's flat gradient
 * buffer over RCCL, enqueued on the caller's stream (capturable into a hipGraph together with the kernels around it).  RCCL is
 * dlopen'ed at first use (no link-time dependency).  Rendezvous: rank 0 calls spgan_comm_unique_id and hands the 128 bytes to
 * every rank through the host program's own channel; then every rank calls spgan_comm_init with the HIP device it trains on
 * current.  spgan_comm_available() == 0 (or status SPGAN_ECOMM) when librccl cannot be loaded; spgan_comm_last_error() is the
 * ncclResult_t of the last failing call.  The caller divides by the world size (spgan_adam_step's grad_scale). */
int spgan_comm_available(void);
int spgan_comm_last_error(void* comm);   /* RCCL code of the last failed call ON THIS communicator; comm == NULL: of the calling thread's last failed spgan_comm_unique_id / spgan_comm_init */
int spgan_comm_unique_id(void* id128);
int spgan_comm_init(const void* id128, int rank, int world, void** comm);
int spgan_comm_world(void* comm);
int spgan_allreduce_flat(void* comm, float* buf, size_t n, spgan_stream_t s);
int spgan_comm_destroy(void* comm);
/* dst[t][i] = src[t][i] for the same argument block: up to SPGAN_MULTI_MAX device-to-device copies in ONE launch. */
int spgan_multi_copy(const spgan_multi_add_args* a, spgan_stream_t s);
/* Finish up to SPGAN_MULTI_MAX deferred spgan_gemm_tn products in one launch: C[e] = beta[e]*C[e] + fixed-order sum of the
 * splits[e] = spgan_gemm_tn_splits(M,Na,Nb) partials [splits, Na, Nb] in ws[e].  block_start[e] = sum_{f<e}
 * spgan_splitk_reduce_blocks(splits[f], Na[f], Nb[f]) (64 outputs per workgroup, or 4 -- one wave each -- for many partials of few outputs). */
typedef struct spgan_splitk_multi_args {
  int count;
  const float* ws[SPGAN_MULTI_MAX];
  float* C[SPGAN_MULTI_MAX];
  int splits[SPGAN_MULTI_MAX], Na[SPGAN_MULTI_MAX], Nb[SPGAN_MULTI_MAX], ldc[SPGAN_MULTI_MAX];
  float beta[SPGAN_MULTI_MAX];
  int block_start[SPGAN_MULTI_MAX + 1];
} spgan_splitk_multi_args;
/* dst[e] (cols x rows, contiguous) = src[e]^T (rows x cols, row stride ld) for count <= SPGAN_MULTI_MAX matrices in ONE launch: the
 * transposed weights read by the input-gradient GEMMs, refreshed once per optimiser step.  tile_start[e] = sum_{f<e} of
 * ceil(rows/32)*ceil(cols/32). */
typedef struct spgan_multi_transpose_args {
  int count;
  const float* src[SPGAN_MULTI_MAX];
  float* dst[SPGAN_MULTI_MAX];
  int rows[SPGAN_MULTI_MAX], cols[SPGAN_MULTI_MAX], ld[SPGAN_MULTI_MAX];
  int tile_start[SPGAN_MULTI_MAX + 1];
} spgan_multi_transpose_args;
int spgan_multi_transpose(const spgan_multi_transpose_args* a, spgan_stream_t s);
int spgan_gemm_tn_splits(int M, int Na, int Nb);
int spgan_splitk_reduce_blocks(int splits, int Na, int Nb);
int spgan_splitk_reduce_multi(const spgan_splitk_multi_args* a, spgan_stream_t s);
/* y = a*x + b*y */
int spgan_axpby(float a, const float* x, float b, float* y, size_t n, spgan_stream_t s);
/* torch.optim.Adam step on a flat buffer (Generation/model.py:94-97: lr 1e-4, betas (0.5,0.99)); g is scaled by grad_scale first */
int spgan_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, int step,
                    float grad_scale, spgan_stream_t s);
/* The same update with its step-dependent state in device memory: state[0] = int step count (advanced by this call), state[1..2] = its
 * bias corrections, state[3] = a multiplier on `lr` (1 = lr as passed; the StepLR schedule of Generation/model.py:99-110,309-312 writes
 * it): no host value changes from one step to the next, so a captured hipGraph of the train step replays it.  state: 4 floats.
 * zero_grad != 0: g is zeroed after it was read -- the optimizer.zero_grad() of the next iteration (model.py:243,268) without a launch. */
int spgan_adam_step_dev(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                        float* state, float grad_scale, int zero_grad, spgan_stream_t s);

/* Measurement plumbing (bench.py's `roofline` entry): device timestamps that survive a hipGraph capture, where HIP events
 * cannot be queried.  spgan_stamp_begin stores the constant-rate wall clock in *slot; spgan_stamp_end adds (now - *slot) to acc2[0] and 1
 * to acc2[1]; launched on the kernel's stream directly in front of / behind it.  spgan_wall_clock_khz: ticks per millisecond (0: unknown). */
int spgan_stamp_begin(uint64_t* slot, spgan_stream_t s);
int spgan_stamp_end(const uint64_t* slot, uint64_t* acc2, spgan_stream_t s);
int spgan_wall_clock_khz(void);

/* ------------------------------------------------------------------------------------------
 * Evaluation metrics: Chamfer distance.
 * ---------------------------------------------------------------------------------------- */
/* dist[b,i] = min_j |xyz1[b,i] - xyz2[b,j]|^2, idx[b,i] = the first j attaining it (index into xyz2[b]): one direction of
 * chamfer.forward (metrics/CD_EMD/cd/chamferdist/chamfer.cu:12-113, ChamferDistance.py:13-31); xyz1 [B,N,3], xyz2 [B,M,3]. */
int spgan_nn_distance(const float* xyz1, const float* xyz2, int B, int N, int M, float* dist, int32_t* idx, spgan_stream_t s);
/* grad_a[b,i] = 2*ga[b,i]*(xa_i - xb[idxa[b,i]]) + sum_{j: idxb[b,j]==i} 2*gb[b,j]*(xa_i - xb_j): d/d(xa) of
 * sum(ga*dist_a) + sum(gb*dist_b) (chamfer.cu:155-195, gather form: deterministic, no float atomics).  Call once per cloud. */
int spgan_chamfer_bwd(const float* xa, const float* xb, int B, int Na, int Nb, const float* ga, const int32_t* idxa, const float* gb,
                      const int32_t* idxb, float* grad_a, spgan_stream_t s);
/* out[s,r] = mean_i min_j |A[s,i]-Bc[r,j]|^2 + mean_j min_i |A[s,i]-Bc[r,j]|^2 for every pair of clouds A[s] ([S,N,3]) and
 * Bc[r] ([R,M,3]): the all-pairs Chamfer matrix behind MMD-CD / COV-CD / 1-NNA-CD (metrics/evaluation_metrics.py:89-126). */
int spgan_chamfer_pairs(const float* A, const float* Bc, int S, int R, int N, int M, float* out, spgan_stream_t s);
/* Occupancy-grid statistics of the JSD metric (metrics/evaluation_metrics.py:247-283, entropy_of_occupancy_grid): cell [S,N] holds
 * the nearest grid cell of every point (spgan_nn_distance against the grid); counters[g] += points in cell g over all clouds,
 * bernoulli[g] += clouds with at least one point in g.  Accumulates: the caller zeroes both int32 [G] arrays.  G <= 524288. */
int spgan_occupancy_counts(const int32_t* cell, int S, int N, int G, int32_t* counters, int32_t* bernoulli, spgan_stream_t s);
/* Minimum matching distance / coverage over a distance matrix dist [S,R] between S sample clouds and R reference clouds
 * (metrics/evaluation_metrics.py:161-173, lgan_mmd_cov): out3 = [mean over references of their closest sample's distance,
 * fraction of references that are the closest reference of some sample (first one on ties), mean over samples of their closest
 * reference's distance].  ws: (S + 2R) floats of scratch.  Fixed-order sums. */
int spgan_mmd_cov(const float* dist, int S, int R, float* out3, float* ws, spgan_stream_t s);
/* Leave-one-out k-nearest-neighbour two-sample test (metrics/evaluation_metrics.py:129-158, knn): Mxx [n0,n0], Mxy [n0,n1],
 * Myy [n1,n1] are the blocks of the joint distance matrix (never concatenated); every cloud is classified by the majority label of
 * its k nearest OTHER clouds (votes >= k/2 -> first set; lower index on distance ties); take_sqrt compares sqrt(|d|).
 * out9 = [tp, fp, fn, tn, precision, recall, acc_t, acc_f, acc]; pred: int32 [n0+n1] scratch (the per-cloud predictions). */
int spgan_two_sample_knn(const float* Mxx, const float* Mxy, const float* Myy, int n0, int n1, int k, int take_sqrt, float* out9,
                         int32_t* pred, spgan_stream_t s);

/* Approximate earth mover's distance by a synchronous auction: the algorithm of the reference's emd module
 * (metrics/emd/emd_cuda.cu:93-236 behind metrics/CD_EMD/emd_/emd_module.py:33-75: `emd.forward(xyz1, xyz2, dist, assignment, price,
 * assignment_inv, bid, bid_increments, max_increments, ..., eps, iters)`), with the scratch arrays folded into one workspace and a
 * timing-independent winner rule (highest increment, lowest bidder index).  xyz1/xyz2 [B,n,3] (coordinates normalised to [0,1] as
 * there); dist [B,n] = squared distance of every point of xyz1 to its assigned point, assignment [B,n] = that point's index in
 * xyz2[b] (not guaranteed to be a bijection when `iters` ends before the auction does).  Any n (the reference: n % 1024 == 0). */
size_t spgan_emd_ws_bytes(int B, int n);
int spgan_emd_forward(const float* xyz1, const float* xyz2, int B, int n, float eps, int iters, float* dist, int32_t* assignment,
                      void* ws, size_t ws_bytes, spgan_stream_t s);
/* grad_xyz1[b,i] = 2*grad_dist[b,i]*(xyz1[b,i] - xyz2[b,assignment[b,i]]) (emd_cuda.cu:279-313; xyz2 receives no gradient there) */
int spgan_emd_backward(const float* xyz1, const float* xyz2, int B, int n, const float* grad_dist, const int32_t* assignment,
                       float* grad_xyz1, spgan_stream_t s);

/* ------------------------------------------------------------------------------------------
 * --attn variant: `Attention(640)` between the concat and the tail (Generation/modules.py:534-558,
 * Generator.py:116-117,191-192).  The projections and the per-shape [N,N] contractions are spgan_gemm_nt / spgan_gemm_tn
 * calls; these are the HBM-bound pieces between them.
 * ---------------------------------------------------------------------------------------- */
/* S[r,:] = softmax(S[r,:]) in place, rows x cols contiguous (F.softmax(.,-1), modules.py:554) */
int spgan_softmax_rows(float* S, long rows, int cols, spgan_stream_t s);
/* dP[r,:] = P[r,:] * (dP[r,:] - sum_j dP[r,j]*P[r,j]) in place: the softmax Jacobian applied to the upstream gradient */
int spgan_softmax_rows_bwd(const float* P, float* dP, long rows, int cols, spgan_stream_t s);
/* y = gamma[0]*o + x (gamma: device scalar, the learnable gate of modules.py:546,558); n % 4 == 0 */
int spgan_scale_residual(const float* o, const float* x, const float* gamma, float* y, size_t n, spgan_stream_t s);
/* d_o = gamma[0]*dy, dgamma[0] = sum(dy*o) (two-stage, fixed order).  ws: spgan_scale_residual_bwd_ws_bytes(n) bytes. */
size_t spgan_scale_residual_bwd_ws_bytes(size_t n);
int spgan_scale_residual_bwd(const float* dy, const float* o, const float* gamma, float* d_o, float* dgamma, void* ws, size_t ws_bytes,
                             size_t n, spgan_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* SPGAN_HIP_H */
