"""ctypes binding of libspgan_hip.so (the C ABI declared in include/spgan_hip.h).

There is deliberately no fallback: if the shared library is missing or does not export a
symbol, importing/using the ops raises.  A CPU path for these ops does not exist in the product.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libspgan_hip.so")

c_f32p = C.c_void_p     # device pointers travel as integers
c_i32p = C.c_void_p
c_i64p = C.c_void_p
stream_t = C.c_void_p


class ColTail(C.Structure):
    _fields_ = [("enabled", C.c_int), ("mode", C.c_int),
                ("out0", c_f32p), ("out1", c_f32p),
                ("gamma", c_f32p), ("beta", c_f32p), ("rmean", c_f32p), ("rvar", c_f32p),
                ("scale", c_f32p), ("shift", c_f32p), ("invstd", c_f32p), ("mean_out", c_f32p),
                ("eps", C.c_float), ("momentum", C.c_float), ("count_rep", C.c_int)]


class GemmNTArgs(C.Structure):
    _fields_ = [
        ("A", c_f32p), ("lda", C.c_int),
        ("W", c_f32p), ("ldw", C.c_int),
        ("Y", c_f32p), ("ldy", C.c_int),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("a_mode", C.c_int),
        ("p_scale", c_f32p), ("p_shift", c_f32p), ("p_slope", C.c_float),
        ("e_idx", c_i32p), ("e_k", C.c_int), ("e_bias", c_f32p),
        ("epi_mode", C.c_int),
        ("bias", c_f32p), ("rowbias", c_f32p), ("rows_per_group", C.c_int), ("ld_rowbias", C.c_int),
        ("act", C.c_int), ("act_slope", C.c_float),
        ("stats", c_f32p),
        ("ref", c_f32p), ("ld_ref", C.c_int),
        ("b_scale", c_f32p), ("b_shift", c_f32p), ("b_mean", c_f32p), ("b_invstd", c_f32p), ("b_slope", C.c_float),
        ("e_bias2", c_f32p),
        ("sp_val", c_f32p), ("sp_arg", c_i32p), ("sp_rows", C.c_int),
        ("pool_val", c_f32p), ("pool_arg", c_i32p),
        ("mfma_f16", C.c_int),
        ("batch", C.c_int), ("batch_stride_a", C.c_long), ("batch_stride_w", C.c_long), ("batch_stride_y", C.c_long),
        ("tail", ColTail),
        ("tile_hint", C.c_int),
        ("p_group_rows", C.c_int),
        ("A2", c_f32p), ("lda2", C.c_int), ("p_scale2", c_f32p),
        ("a_half", C.c_int), ("y_bf16", C.c_int), ("y_half", C.c_int),
        ("w_image", C.c_void_p),
        ("gout_add", c_f32p), ("ld_gout_add", C.c_int), ("gout_scale", c_f32p),
    ]


class GemmTNArgs(C.Structure):
    _fields_ = [
        ("A", c_f32p), ("lda", C.c_int),
        ("B", c_f32p), ("ldb", C.c_int),
        ("C", c_f32p), ("ldc", C.c_int),
        ("M", C.c_int), ("Na", C.c_int), ("Nb", C.c_int),
        ("b_mode", C.c_int),
        ("p_scale", c_f32p), ("p_shift", c_f32p), ("p_slope", C.c_float),
        ("e_idx", c_i32p), ("e_k", C.c_int), ("e_bias", c_f32p),
        ("beta", C.c_float),
        ("ws", c_f32p), ("ws_bytes", C.c_size_t),
        ("a_scale", c_f32p), ("a_shift", c_f32p), ("a_sp_val", c_f32p), ("a_sp_arg", c_i32p), ("a_sp_rows", C.c_int),
        ("defer_reduce", C.c_int),
        ("mfma_lp", C.c_int),
        ("A2", c_f32p), ("lda2", C.c_int), ("a_scale2", c_f32p),
        ("a_colsum_ws", c_f32p),
        ("a_lrelu", C.c_int), ("a_slope", C.c_float),
        ("b_half", C.c_int), ("a_half", C.c_int),
    ]


class GemmDualArgs(C.Structure):
    _fields_ = [
        ("A", c_f32p), ("lda", C.c_int),
        ("A2", c_f32p), ("lda2", C.c_int), ("p", c_f32p), ("q", c_f32p), ("r", c_f32p),
        ("W", c_f32p), ("ldw", C.c_int),
        ("B", c_f32p), ("ldb", C.c_int),
        ("e_idx", c_i32p), ("e_k", C.c_int), ("e_bias", c_f32p),
        ("b_scale", c_f32p), ("b_shift", c_f32p), ("b_mean", c_f32p), ("b_invstd", c_f32p), ("slope", C.c_float),
        ("G", c_f32p), ("ldg", C.c_int),
        ("stats", c_f32p), ("ws", c_f32p),
        ("M", C.c_int), ("Na", C.c_int), ("Nb", C.c_int),
        ("a_mode", C.c_int), ("a_slope", C.c_float),
        ("bias", c_f32p), ("rowadd", c_f32p), ("ld_rowadd", C.c_int),
        ("colsum_ws", c_f32p),
        ("gout_add", c_f32p), ("ld_gout_add", C.c_int), ("gout_scale", c_f32p),
    ]


MULTI_MAX = 64


class MultiAddArgs(C.Structure):
    _fields_ = [("count", C.c_int), ("dst", C.c_void_p * MULTI_MAX), ("src", C.c_void_p * MULTI_MAX), ("n", C.c_int * MULTI_MAX)]


class MultiAdd3Args(C.Structure):
    _fields_ = [("count", C.c_int), ("dst", C.c_void_p * MULTI_MAX), ("src", C.c_void_p * MULTI_MAX), ("n", C.c_int * MULTI_MAX),
                ("n1", C.c_int * MULTI_MAX), ("n2", C.c_int * MULTI_MAX), ("ds", (C.c_long * MULTI_MAX) * 3), ("ss", (C.c_long * MULTI_MAX) * 3)]


GROUP_MAX = 4           # SPGAN_GROUP_MAX: problems per grouped launch


class CollapsePrepArgs(C.Structure):
    _fields_ = [("W", C.c_void_p), ("ldw", C.c_int), ("C", C.c_int), ("K", C.c_int), ("nprob", C.c_int),
                ("alpha", C.c_void_p * GROUP_MAX), ("beta", C.c_void_p * GROUP_MAX), ("bias", C.c_void_p * GROUP_MAX), ("G", C.c_void_p * GROUP_MAX),
                ("ldg", C.c_int), ("cvec", C.c_void_p * GROUP_MAX), ("nsparse", C.c_int), ("sp_val", C.c_void_p * GROUP_MAX),
                ("sp_arg", C.c_void_p * GROUP_MAX), ("B", C.c_int), ("rows", C.c_int), ("E", C.c_void_p * GROUP_MAX), ("lde", C.c_int)]


class ColFinalizeArgs(C.Structure):
    _fields_ = [("partials", C.c_void_p), ("tiles", C.c_int), ("C", C.c_int), ("G", C.c_int), ("tile_rows", C.c_int),
                ("s0", C.c_void_p), ("s1", C.c_void_p), ("kind", C.c_int),
                ("mean", C.c_void_p), ("invstd", C.c_void_p), ("gamma", C.c_void_p), ("count", C.c_float), ("coef", C.c_void_p),
                ("U0", C.c_void_p), ("U1", C.c_void_p), ("Ugz", C.c_void_p), ("S0", C.c_void_p), ("S1", C.c_void_p),
                ("sums", C.c_void_p), ("dgamma", C.c_void_p), ("pb_coef", C.c_void_p)]


class PoolBwdArgs(C.Structure):
    _fields_ = [("gpool", C.c_void_p), ("pooled", C.c_void_p), ("argmax", C.c_void_p), ("y", C.c_void_p), ("ld", C.c_int),
                ("mean", C.c_void_p), ("invstd", C.c_void_p), ("slope", C.c_float), ("B", C.c_int), ("C", C.c_int),
                ("gamma", C.c_void_p), ("count", C.c_int),
                ("gval", C.c_void_p), ("sums", C.c_void_p), ("alpha", C.c_void_p), ("beta", C.c_void_p), ("cg", C.c_void_p)]


MULTI_ADDN_MAX = 32


class MultiAddNArgs(C.Structure):
    _fields_ = [("count", C.c_int), ("dst", C.c_void_p * MULTI_ADDN_MAX), ("src", (C.c_void_p * MULTI_ADDN_MAX) * 3),
                ("nsrc", C.c_int * MULTI_ADDN_MAX), ("n", C.c_int * MULTI_ADDN_MAX)]


class WgradCollapseArgs(C.Structure):
    _fields_ = [("W", C.c_void_p), ("ldw", C.c_int), ("C", C.c_int), ("K", C.c_int), ("X1", C.c_void_p), ("ldx1", C.c_int),
                ("a1", C.c_void_p), ("b1", C.c_void_p), ("d1", C.c_void_p), ("v1", C.c_void_p),
                ("X2", C.c_void_p), ("ldx2", C.c_int), ("x2_t", C.c_int), ("a2", C.c_void_p),
                ("sp_val", C.c_void_p), ("sp_arg", C.c_void_p), ("B", C.c_int), ("rows", C.c_int),
                ("Bm", C.c_void_p), ("ldb", C.c_int), ("p_scale", C.c_void_p), ("p_shift", C.c_void_p), ("p_slope", C.c_float),
                ("T", C.c_void_p), ("ldt", C.c_int), ("out", C.c_void_p), ("ldo", C.c_int), ("N", C.c_int), ("accumulate", C.c_int)]


class MultiTransposeArgs(C.Structure):
    _fields_ = [("count", C.c_int), ("src", C.c_void_p * MULTI_MAX), ("dst", C.c_void_p * MULTI_MAX),
                ("rows", C.c_int * MULTI_MAX), ("cols", C.c_int * MULTI_MAX), ("ld", C.c_int * MULTI_MAX),
                ("tile_start", C.c_int * (MULTI_MAX + 1))]


class SplitKMultiArgs(C.Structure):
    _fields_ = [("count", C.c_int), ("ws", C.c_void_p * MULTI_MAX), ("C", C.c_void_p * MULTI_MAX),
                ("splits", C.c_int * MULTI_MAX), ("Na", C.c_int * MULTI_MAX), ("Nb", C.c_int * MULTI_MAX), ("ldc", C.c_int * MULTI_MAX),
                ("beta", C.c_float * MULTI_MAX), ("block_start", C.c_int * (MULTI_MAX + 1))]


I, F, P, SZ = C.c_int, C.c_float, C.c_void_p, C.c_size_t

# name -> (restype, argtypes).  Must list every symbol include/spgan_hip.h declares
# (tests/test_abi.py cross-checks this table against the header and the .so).
SIGNATURES = {
    "spgan_version": (I, []),
    "spgan_arch": (C.c_char_p, []),
    "spgan_knn": (I, [P, I, I, I, I, I, P, P]),
    "spgan_knn_ws_bytes": (SZ, [I, I, I, I, I]),
    "spgan_knn_ws": (I, [P, I, I, I, I, I, P, P, SZ, P]),
    "spgan_csr_build": (I, [P, I, I, I, P, P, P]),
    "spgan_edge_features_cm": (I, [P, P, I, I, I, I, P, P]),
    "spgan_idx_to_local64": (I, [P, I, I, I, P, P]),
    "spgan_idx_from_local64": (I, [P, I, I, I, P, P]),
    "spgan_cm_to_pm": (I, [P, I, I, I, P, P]),
    "spgan_pm_to_cm": (I, [P, I, I, I, P, P]),
    "spgan_concat2": (I, [P, I, P, I, I, P, P]),
    "spgan_gemm_nt": (I, [C.POINTER(GemmNTArgs), P]),
    "spgan_gemm_nt_col_blocks": (I, [C.POINTER(GemmNTArgs)]),
    "spgan_split_bf16x3_image_bytes": (C.c_size_t, [I, I]),
    "spgan_split_bf16x3_image": (I, [P, I, I, I, P, P]),
    "spgan_gemm_nt_uses_w_image": (I, [C.POINTER(GemmNTArgs)]),
    "spgan_gemm_nt_owns_columns": (I, [C.POINTER(GemmNTArgs)]),
    "spgan_pool_finalize": (I, [P, P, I, I, I, P, P, F, P, P, P, P]),
    "spgan_pool_finalize_groups": (I, [P, P, I, I, I, P, P, I, I, F, P, P, P, I, P]),
    "spgan_gemm_tn_ws_bytes": (SZ, [I, I, I]),
    "spgan_gemm_tn_ws_bytes_lp": (SZ, [I, I, I, I]),
    "spgan_gemm_tn_splits_lp": (I, [I, I, I, I]),
    "spgan_gemm_tn": (I, [C.POINTER(GemmTNArgs), P]),
    "spgan_gemm_tn_skinny_multi": (I, [C.POINTER(GemmTNArgs), I, P]),
    "spgan_sparse_rows_nt": (I, [P, P, I, I, I, P, I, I, P, I, P]),
    "spgan_sparse_rows_tn": (I, [P, P, I, I, I, P, I, I, P, P, F, P, I, P]),
    "spgan_affine_act": (I, [P, I, SZ, I, P, P, F, P, P]),
    "spgan_rowscale_outer": (I, [P, I, I, I, P, P, P, P, P, I, I, P]),
    "spgan_colreduce_ws_bytes": (SZ, [I, I, I]),
    "spgan_colstats_finalize": (I, [P, I, I, I, I, I, I, P, P, P]),
    "spgan_colstats_finalize_bn": (I, [P, I, I, I, I, P, P, F, F, P, P, P, P, P, P, P]),
    "spgan_colstats_finalize_bn_groups": (I, [P, I, I, I, I, I, P, P, F, F, P, P, P, P]),
    "spgan_colstats_finalize_bn2": (I, [P, I, I, I, I, I, P, P, P, P, P, P, P, P, F, F, I, P, P]),
    "spgan_colstats": (I, [P, I, I, I, I, F, P, P, P, SZ, P]),
    "spgan_colsum": (I, [P, I, I, I, I, P, P, SZ, P]),
    "spgan_bn_prepare": (I, [P, P, P, P, I, I, F, F, I, P, P, P, P, P, P, P]),
    "spgan_bn_bwd_apply": (I, [P, P, I, I, I, P, P, P, P, I, P, P]),
    "spgan_bn_bwd_apply2": (I, [P, P, P, P, I, I, P, P, P, P, I, P, P]),
    "spgan_maxpool": (I, [P, I, I, I, I, P, P, F, P, P, P]),
    "spgan_edge_wcat": (I, [P, P, I, I, I, P, P, P]),
    "spgan_conv_out_weight_pm": (I, [P, I, I, P, P, P]),
    "spgan_edge_wcat_bwd": (I, [P, I, I, I, P, P, P]),
    "spgan_edge_stats_tile_rows": (I, [I]),
    "spgan_edge_stats": (I, [P, I, P, I, I, I, I, P, P, P, P]),
    "spgan_edge_attend_fwd": (I, [P, P, P, P, I, I, I, P, I, I, P, P, P, F, P, P]),
    "spgan_edge_attend_fwd_h": (I, [P, I, P, P, P, I, I, I, P, I, I, P, P, P, F, P, P]),
    "spgan_edge_attend_bwd_b": (I, [P, P, I, P, P, P, P, P, I, I, I, P, I, I, P, P, P, P, P, F, P, P, P, P]),
    "spgan_gemm_nt_y16_ok": (I, [P]),
    "spgan_edge_attend_bwd_tile_points": (I, []),
    "spgan_edge_attend_bwd": (I, [P, P, P, P, P, P, P, I, I, I, P, I, I, P, P, P, P, P, F, P, P, P, P]),
    "spgan_edge_scatter_b": (I, [P, P, P, I, I, I, P, P, P, I, I, P, P, P, P, P, P, P, P, P, P, P, P]),
    "spgan_edge_scatter": (I, [P, P, P, I, I, I, P, P, P, I, I, P, P, P, P, P, P, P, P, P, P, P, P]),
    "spgan_adain_fwd": (I, [P, I, I, I, F, P, P, F, P, P, P]),
    "spgan_adain_bwd1": (I, [P, P, I, I, I, F, P, P, F, P, P, P, P]),
    "spgan_adain_bwd2": (I, [P, P, I, I, I, F, P, P, F, P, P, P, P, P]),
    "spgan_pool_bwd_stats": (I, [P, P, P, P, I, P, P, F, I, I, P, P, P]),
    "spgan_pool_bwd_stats_prep": (I, [P, P, P, P, I, P, P, F, I, I, P, I, P, P, P, P, P, P]),
    "spgan_bn_bwd_apply_sparse": (I, [P, P, P, I, I, I, I, P, P, P, P, I, P, P]),
    "spgan_maxpool_bwd_add": (I, [P, P, I, I, P, I, P]),
    "spgan_tanh_bwd": (I, [P, P, SZ, P, P]),
    "spgan_act_bwd": (I, [P, P, SZ, I, F, P, P]),
    "spgan_scatter_rows": (I, [P, P, I, I, I, P, P]),
    "spgan_gather_rows": (I, [P, I, P, I, I, P, P]),
    "spgan_bn_dbl_stats": (I, [P, P, P, I, I, P, P, P, P]),
    "spgan_bn_dbl_apply": (I, [P, P, P, I, I, P, P, P, P, F, P, P, P, P, P, P, P]),
    "spgan_bn_dbl_coeffs": (I, [P, P, P, P, P, P, P, I, I, P, P]),
    "spgan_bn_dbl_phaseb": (I, [P, P, P, P, P, I, P, P, P]),
    "spgan_bn_dbl_phaseb_sums": (I, [P, P, P, P, P, P, P, P, P, I, I, P, P, P]),
    "spgan_gather_rowdot": (I, [P, I, P, P, I, I, I, I, P, P]),
    "spgan_rowdot": (I, [P, I, P, I, I, I, P, P]),
    "spgan_dbl_top_dots": (I, [P, I, P, P, I, P, I, P, I, I, I, P, P, P, P]),
    "spgan_bn_dbl_pool": (I, [P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, F, P, P, P, P]),
    "spgan_sparse_bn_prep": (I, [P, P, P, P, P, I, I, I, P, P, P, P]),
    "spgan_col_scale_add": (I, [P, P, P, I, I, P, P]),
    "spgan_gan_loss": (I, [I, I, P, P, P, P, I, P, P, P, P]),
    "spgan_lerp_rows": (I, [P, P, P, I, SZ, P, P]),
    "spgan_gp_penalty_fwd": (I, [P, I, SZ, F, F, P, P, P]),
    "spgan_gp_penalty_bwd": (I, [P, P, I, SZ, F, F, P, P, P]),
    "spgan_gp_penalty_fwd_bwd": (I, [P, I, SZ, F, F, P, P, P, P, P, P]),
    "spgan_square_distance": (I, [P, P, I, I, I, I, P, P]),
    "spgan_index_points": (I, [P, P, I, I, I, I, P, P]),
    "spgan_farthest_point_sample": (I, [P, I, I, I, P, P, P, P]),
    "spgan_query_ball_point": (I, [F, I, P, P, I, I, I, I, P, P]),
    "spgan_knn_point": (I, [I, P, P, I, I, I, I, P, P]),
    "spgan_group_concat": (I, [P, P, P, P, I, I, I, I, I, I, P, P]),
    "spgan_wt_diag_w": (I, [P, I, I, I, P, P, P, P, I, P, P]),
    "spgan_collapse_prep": (I, [C.POINTER(CollapsePrepArgs), P]),
    "spgan_wgrad_collapse": (I, [C.POINTER(WgradCollapseArgs), P]),
    "spgan_wgrad_collapse_multi": (I, [C.POINTER(WgradCollapseArgs), I, P]),
    "spgan_gemm_dual_wgs": (I, [I, I, I, I]),
    "spgan_gemm_dual_rows_per_wg": (I, [I, I, I]),
    "spgan_gemm_dual": (I, [C.POINTER(GemmDualArgs), P]),
    "spgan_gemm_dual_multi": (I, [C.POINTER(GemmDualArgs), I, P]),
    "spgan_colstats_finalize_multi": (I, [C.POINTER(ColFinalizeArgs), I, P]),
    "spgan_pool_bwd_stats_prep_multi": (I, [C.POINTER(PoolBwdArgs), I, P]),
    "spgan_gather_csr": (I, [P, I, I, I, P, P, P, P]),
    "spgan_scatter_slots": (I, [P, I, I, I, P, P, I, P, P]),
    "spgan_group_center_bwd": (I, [P, I, I, I, I, P, P]),
    "spgan_edge_features_cm_bwd": (I, [P, P, P, I, I, I, I, P, P]),
    "spgan_nn_distance": (I, [P, P, I, I, I, P, P, P]),
    "spgan_chamfer_bwd": (I, [P, P, I, I, I, P, P, P, P, P, P]),
    "spgan_chamfer_pairs": (I, [P, P, I, I, I, I, P, P]),
    "spgan_occupancy_counts": (I, [P, I, I, I, P, P, P]),
    "spgan_mmd_cov": (I, [P, I, I, P, P, P]),
    "spgan_two_sample_knn": (I, [P, P, P, I, I, I, I, P, P, P]),
    "spgan_stamp_begin": (I, [P, P]),
    "spgan_stamp_end": (I, [P, P, P]),
    "spgan_wall_clock_khz": (I, []),
    "spgan_emd_ws_bytes": (SZ, [I, I]),
    "spgan_emd_forward": (I, [P, P, I, I, F, I, P, P, P, SZ, P]),
    "spgan_emd_backward": (I, [P, P, I, I, P, P, P, P]),
    "spgan_softmax_rows": (I, [P, C.c_long, I, P]),
    "spgan_softmax_rows_bwd": (I, [P, P, C.c_long, I, P]),
    "spgan_scale_residual": (I, [P, P, P, P, SZ, P]),
    "spgan_scale_residual_bwd_ws_bytes": (SZ, [SZ]),
    "spgan_scale_residual_bwd": (I, [P, P, P, P, P, P, SZ, SZ, P]),
    "spgan_multi_add": (I, [C.POINTER(MultiAddArgs), P]),
    "spgan_multi_addn": (I, [C.POINTER(MultiAddNArgs), P]),
    "spgan_multi_add3": (I, [C.POINTER(MultiAdd3Args), P]),
    "spgan_reduce_chunks": (I, [P, I, C.c_size_t, P, P]),
    "spgan_bn_bwd_coeffs": (I, [P, P, P, P, I, F, P, P]),
    "spgan_colstats_finalize_bnbwd": (I, [P, I, I, I, I, P, P, P, F, P, P, P, P]),
    "spgan_colstats_finalize_phaseb": (I, [P, I, I, I, I, P, P, P, P, P, P, P, I, P, P, P, P, P]),
    "spgan_comm_available": (I, []),
    "spgan_comm_last_error": (I, [C.c_void_p]),
    "spgan_comm_unique_id": (I, [P]),
    "spgan_comm_init": (I, [P, I, I, C.POINTER(C.c_void_p)]),
    "spgan_comm_world": (I, [P]),
    "spgan_allreduce_flat": (I, [P, P, C.c_size_t, P]),
    "spgan_comm_destroy": (I, [P]),
    "spgan_multi_copy": (I, [C.POINTER(MultiAddArgs), P]),
    "spgan_multi_transpose": (I, [C.POINTER(MultiTransposeArgs), P]),
    "spgan_gemm_tn_splits": (I, [I, I, I]),
    "spgan_splitk_reduce_blocks": (I, [I, I, I]),
    "spgan_splitk_reduce_multi": (I, [C.POINTER(SplitKMultiArgs), P]),
    "spgan_axpby": (I, [F, P, F, P, SZ, P]),
    "spgan_adam_step": (I, [P, P, P, P, SZ, F, F, F, F, I, F, P]),
    "spgan_adam_step_dev": (I, [P, P, P, P, SZ, F, F, F, F, P, F, I, P]),
}

_lib = None


class SpganLibraryError(RuntimeError):
    pass


def load():
    """dlopen libspgan_hip.so and type every entry point.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SpganLibraryError(
            "libspgan_hip.so not found at %s -- build it with `make -C sp-gan_amd` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise SpganLibraryError("libspgan_hip.so does not export %s" % name) from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, what: str, **shapes):
    """Status -> exception (SURVEY 8(b) error convention: no exit(-1), no silent failure)."""
    if status != 0:
        detail = ", ".join("%s=%s" % kv for kv in shapes.items())
        raise RuntimeError("spgan HIP op %s failed with status %d (%s)" % (what, status, detail))
