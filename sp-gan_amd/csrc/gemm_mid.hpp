// 32 x 32-tile, one-shot-K gemm_nt for small weight-by-weight products (gemm_mid.hip); launch_nt (gemm.hip) routes eligible problems here.
#pragma once
#include "common.hpp"

bool spgan_nt_mid_selected(const spgan_gemm_nt_args& a);
int spgan_launch_nt_mid(const spgan_gemm_nt_args& a, hipStream_t s);
