// Split-bf16 weight gradients on large output tiles (gemm_tn_wide3.hip); launch_tn (gemm.hip) routes eligible mfma_lp == 2 problems here.
#pragma once
#include "common.hpp"

int spgan_tn_wide3_config(int M, int Na, int Nb);                          // output tile as WGM*10 + WGN (128*WGM x 64*WGN), 0: not this kernel's shape
void spgan_tn_wide3_plan(int M, int Na, int Nb, int* splits, int* rows);   // its split plan (config != 0)
bool spgan_tn_wide3_eligible(const spgan_gemm_tn_args& a);
int spgan_launch_tn_wide3(const spgan_gemm_tn_args& a, int splits, int rows, hipStream_t s);
