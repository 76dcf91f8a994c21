// 256 x 256-tile gemm_nt for large aligned products (gemm_wide.hip); launch_nt (gemm.hip) routes eligible problems here.
#pragma once
#include "common.hpp"

constexpr int SPGAN_WIDE_A_SPARSE = 3;  // internal operand mode: A_AFFINE_LRELU with the sparse addend (sp_val != NULL)

bool spgan_nt_wide_eligible(const spgan_gemm_nt_args& a);  // the kernel can run this problem
bool spgan_nt_wide_pays(const spgan_gemm_nt_args& a);      // ... and is expected to be faster than the 128-row kernels (tile_hint 0)
// what launch_nt does: tile_hint 1 -> never, 2 -> when eligible, 0 -> when eligible and expected to pay; the environment variable
// SPGAN_NT_WIDE=0 (read once) turns the kernel off altogether (A/B measurements of whole programs)
bool spgan_nt_wide_selected(const spgan_gemm_nt_args& a);
int spgan_launch_nt_wide(const spgan_gemm_nt_args& a, hipStream_t s);

// Split-bf16 operands (mfma_f16 == 2) on 256-row tiles (gemm_wide3.hip): 256 x 256 or 256 x 128 tiles.
bool spgan_nt_wide3_eligible(const spgan_gemm_nt_args& a);
int spgan_nt_wide3_config(const spgan_gemm_nt_args& a);     // waves (rows x columns, of 128 x 64 each) as WGM*10 + WGN: 24, 22 or 14
int spgan_nt_wide3_tile_n(const spgan_gemm_nt_args& a);     // 256 or 128: the tile width an eligible problem runs with
// tile_hint 1 -> never, 2 -> when eligible, 0 -> when eligible and large enough; SPGAN_NT_WIDE3=0 (read once) turns the kernel off
bool spgan_nt_wide3_selected(const spgan_gemm_nt_args& a);
int spgan_launch_nt_wide3(const spgan_gemm_nt_args& a, hipStream_t s);

// fp16 operands (mfma_f16 == 1) on 256-row tiles in the row-pipelined form (gemm_wide16.hip): 256 x 256 or 256 x 128 tiles; preferred over
// gemm_wide.hip's fp16 instantiation where eligible (K % 64 == 0, N % 128 == 0); SPGAN_NT_WIDE16=0 (read once) turns it off.
bool spgan_nt_wide16_eligible(const spgan_gemm_nt_args& a);
int spgan_nt_wide16_tile_n(const spgan_gemm_nt_args& a);
bool spgan_nt_wide16_selected(const spgan_gemm_nt_args& a);
int spgan_launch_nt_wide16(const spgan_gemm_nt_args& a, hipStream_t s);
