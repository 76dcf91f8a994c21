"""Tensor-level wrappers of the C ABI (include/spgan_hip.h).

Each function validates its arguments (device, dtype, layout -- the reference's CHECK_INPUT,
metrics/pointops/src/knnquery/knnquery_cuda.cpp:10-18), allocates outputs/workspace through
PyTorch's caching allocator, and launches on the *current* torch HIP stream.  Native code keeps
no pointers after return.  PyTorch is only memory + stream plumbing here.

Shapes: "pm" = point-major [M, C]; idx = int32 [M, k] global row ids.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import GemmDualArgs, GemmNTArgs, GemmTNArgs, check

A_PLAIN, A_AFFINE_LRELU, A_EDGE = 0, 1, 2
EPI_LINEAR, EPI_MASK_OUT, EPI_BNBWD, EPI_EDGE_BNBWD = 0, 1, 2, 3
ACT_NONE, ACT_LRELU, ACT_TANH = 0, 1, 2
BN_EPS, BN_MOMENTUM = 1e-5, 0.1
# momentum the train-mode BatchNorm bookkeeping launches use for the running statistics (nets.edgeblock_forward(bn_repeats=r) sets
# 1 - (1 - BN_MOMENTUM)**r: r updates with the same batch statistics in one)
_BN_MOM = [BN_MOMENTUM]


class bn_momentum:
    def __init__(self, m: float):
        self.m = float(m)

    def __enter__(self):
        self.prev = _BN_MOM[0]
        _BN_MOM[0] = self.m

    def __exit__(self, *exc):
        _BN_MOM[0] = self.prev
ROW_TILE = 128

Tensor = torch.Tensor


def _s():
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _f32(t: Tensor, name: str, dims: Optional[int] = None) -> Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU (spgan has no CPU path); got device %s" % (name, t.device))
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))
    if dims is not None and t.dim() != dims:
        raise ValueError("%s must be %d-D, got shape %s" % (name, dims, tuple(t.shape)))
    return t


def _rowmajor2d(t: Tensor, name: str) -> Tensor:
    """2-D, unit column stride (row stride may exceed the width: column slices are fine)."""
    _f32(t, name, 2)
    if t.shape[1] > 1 and t.stride(1) != 1:
        raise ValueError("%s must have unit column stride, got strides %s" % (name, t.stride()))
    return t


def _rowmajor2d_as(t: Tensor, name: str, dtype) -> Tensor:
    """_rowmajor2d for a 16-bit storage tensor (the "f16" operand mode's T / dT): 2-D, unit column stride, on the GPU, of `dtype`."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("%s must be a tensor on the GPU (spgan has no CPU path)" % name)
    if t.dtype != dtype or t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError("%s must be 2-D %s with unit column stride, got %s %s strides %s" % (name, dtype, t.dtype, tuple(t.shape), t.stride()))
    return t


def _vec(t: Optional[Tensor], n: int, name: str) -> Optional[Tensor]:
    if t is None:
        return None
    _f32(t, name)
    if not t.is_contiguous() or t.numel() != n:
        raise ValueError("%s must be contiguous with %d elements, got %s" % (name, n, tuple(t.shape)))
    return t


def _i32(t: Tensor, name: str) -> Tensor:
    if not t.is_cuda or t.dtype != torch.int32 or not t.is_contiguous():
        raise TypeError("%s must be a contiguous int32 GPU tensor" % name)
    return t


def _ld(t: Tensor) -> int:
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))


# Measurement hook (bench.py's roofline / MFMA-accounting legs): launch_timer(kind, args) may return a callable that is invoked
# right after the launch was enqueued -- the bench brackets launches with HIP events on the launch stream.  kind is "gemm_nt"
# (args: GemmNTArgs -- every gemm_nt flavour), "gemm_tn" (GemmTNArgs) or "knn" (_KnnInfo).  None in normal use.
launch_timer = None

# Operand precision of the matrix-core contractions behind gemm_nt / gemm_nt_maskout / gemm_nt_bnbwd / gemm_bn_pool:
# "f32" (default: exact fp32 products) or "f16" (operands rounded to fp16 at the LDS staging, fp32 accumulation: BASELINE
# configs[4] "fp16 MFMA MLPs"; the weight gradients that reduce over the points / edges then use bfloat16 operands, gemm_tn).  kNN
# distances, statistics, every prologue / epilogue and all accumulation stay fp32.
_MFMA_F16 = [0]


_MFMA_KINDS = {"f32": 0, "f16": 1, "bf16x3": 2}


def set_mfma_operands(kind: str) -> None:
    """"f32": fp32 MFMA (exact fp32 products, the default);
    "f16": operands rounded to fp16 at the LDS staging, fp32 accumulation (BASELINE configs[4] "fp16 MFMA MLPs");
    "bf16x3": each fp32 operand split exactly into three bf16 terms, six cross products on the bf16 matrix pipe, fp32
              accumulation -- fp32-equivalent products (|error| <= 3*2^-24 relative per product) at 6/16 of the fp32-MFMA time."""
    if kind not in _MFMA_KINDS:
        raise ValueError("mfma operands must be one of %s" % sorted(_MFMA_KINDS))
    _MFMA_F16[0] = _MFMA_KINDS[kind]


def get_mfma_operands() -> str:
    return {v: k for k, v in _MFMA_KINDS.items()}[_MFMA_F16[0]]


# Split-bf16 weight images ("bf16x3" operand mode): the 256-row-tile gemm_nt kernel copies W's three bf16 planes from a pre-split image of W
# (split_image below: made once per weight and optimiser step) instead of splitting W's fp32 rows again in every workgroup.  ops does not
# know which operands are weights and when they change -- the provider does (nets.w_image: parameters and their cached transposes, refreshed
# when the weights' epoch / version moves, all stale images of a network in one launch); None, or a provider returning None: W is split
# on the fly.  Signature: provider(W) -> uint8 image tensor | None.
w_image_provider = None


def split_image(W: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """The split-bf16 image of a row-major fp32 matrix W [N,K] (N % 128 == 0, K % 16 == 0; spgan_split_bf16x3_image): 6*N*K bytes."""
    _rowmajor2d(W, "W")
    N, K = W.shape
    lib = _lib.load()
    if out is None:
        out = torch.empty((int(lib.spgan_split_bf16x3_image_bytes(N, K)),), dtype=torch.uint8, device=W.device)
    check(lib.spgan_split_bf16x3_image(_p(W), _ld(W), N, K, _p(out), _s()), "split_bf16x3_image", N=N, K=K)
    return out


def _attach_w_image(a, W: Tensor):
    """Called when the argument block is complete: hands the kernel W's pre-split image when the launch would read one.  Returns the image
    (the caller keeps it alive across the launch)."""
    if a.mfma_f16 != 2 or w_image_provider is None or not _lib.load().spgan_gemm_nt_uses_w_image(C.byref(a)):
        return None
    img = w_image_provider(W)
    if img is not None:
        a.w_image = _p(img)
    return img


# gemm_nt tile geometry (spgan_gemm_nt_args.tile_hint): 0 automatic, 1 the 128-row kernels only, 2 the 256 x 256-tile kernel
# (csrc/gemm_wide.hip) whenever the problem is eligible.  1 / 2 are for tests and A/B measurements.
_NT_TILE_HINT = [0]


class nt_tile_hint:
    """with ops.nt_tile_hint(2): ...   -- every gemm_nt launch inside uses the given tile_hint."""

    def __init__(self, hint: int):
        if hint not in (0, 1, 2):
            raise ValueError("tile_hint must be 0, 1 or 2")
        self.hint = hint

    def __enter__(self):
        self.prev = _NT_TILE_HINT[0]
        _NT_TILE_HINT[0] = self.hint

    def __exit__(self, *exc):
        _NT_TILE_HINT[0] = self.prev


# Bumped by every optimiser step that rewrites parameters through a HIP kernel (invisible to torch's version counters);
# host-side caches of weight-derived tensors (nets._t) key on it.
WEIGHTS_EPOCH = [0]
# The same per flat parameter buffer (keyed by its storage address): a cache of something derived from the GENERATOR's weights
# need not be refreshed because the discriminator's optimiser stepped in between (and vice versa).
WEIGHTS_EPOCH_OF = {}


def bump_weights_epoch(flat: Tensor) -> None:
    WEIGHTS_EPOCH[0] += 1
    key = flat.untyped_storage().data_ptr()
    WEIGHTS_EPOCH_OF[key] = WEIGHTS_EPOCH_OF.get(key, 0) + 1


def weights_epoch_of(p: Tensor) -> int:
    return WEIGHTS_EPOCH_OF.get(p.untyped_storage().data_ptr(), 0)


def capturing() -> bool:
    """Is the current stream being captured into a hipGraph?  (False on a CPU-only build: the host-composition tests.)"""
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


class _KnnInfo:
    def __init__(self, B, N, C, k, mode):
        self.B, self.N, self.C, self.k, self.mode = B, N, C, k, mode


# ----------------------------------------------------------------------------- graph
KNN_PIPELINED = [True]  # test / A-B hook: False keeps every shape on spgan_knn's single-launch kernels


def knn(x_pm: Tensor, B: int, N: int, k: int, mode: int = 0) -> Tensor:
    """x_pm [B*N, C] -> idx int32 [B*N, k] (global rows), sorted ascending, rank 0 dropped.
    mode 1 = fp64 direct differences (coordinate inputs, C<=4)."""
    _f32(x_pm, "x_pm", 2)
    if not x_pm.is_contiguous() or x_pm.shape[0] != B * N:
        raise ValueError("x_pm must be contiguous [B*N, C]")
    idx = torch.empty((B * N, k), dtype=torch.int32, device=x_pm.device)
    done = launch_timer("knn", _KnnInfo(B, N, x_pm.shape[1], k, mode)) if launch_timer is not None else None
    lib, C = _lib.load(), x_pm.shape[1]
    ws_bytes = lib.spgan_knn_ws_bytes(B, N, C, k, mode) if KNN_PIPELINED[0] else 0
    if ws_bytes:  # the tile images of csrc/knn_pipe.hip (freed with the call; inside a capture they live in the graph's pool)
        ws = torch.empty((ws_bytes // 4,), dtype=torch.float32, device=x_pm.device)
        check(lib.spgan_knn_ws(_p(x_pm), B, N, C, k, mode, _p(idx), _p(ws), ws_bytes, _s()), "knn", B=B, N=N, C=C, k=k)
    else:
        check(lib.spgan_knn(_p(x_pm), B, N, C, k, mode, _p(idx), _s()), "knn", B=B, N=N, C=C, k=k)
    if done is not None:
        done()
    return idx


def csr_build(idx: Tensor, B: int, N: int) -> Tuple[Tensor, Tensor]:
    _i32(idx, "idx")
    k = idx.shape[1]
    rowptr = torch.empty((B * N + 1,), dtype=torch.int32, device=idx.device)
    src = torch.empty((B * N * k,), dtype=torch.int32, device=idx.device)
    check(_lib.load().spgan_csr_build(_p(idx), B, N, k, _p(rowptr), _p(src), _s()), "csr_build", B=B, N=N, k=k)
    return rowptr, src


def edge_features_cm(x_cm: Tensor, idx_local: Tensor, k: int) -> Tensor:
    _f32(x_cm, "x", 3)
    B, Cc, N = x_cm.shape
    if not x_cm.is_contiguous():
        raise ValueError("x must be contiguous [B,C,N]")
    if idx_local.dtype != torch.int64 or not idx_local.is_contiguous() or idx_local.numel() != B * N * k:
        raise ValueError("idx must be contiguous int64 with B*N*k elements")
    ee = torch.empty((B, 2 * Cc, N, k), dtype=torch.float32, device=x_cm.device)
    check(_lib.load().spgan_edge_features_cm(_p(x_cm), _p(idx_local), B, Cc, N, k, _p(ee), _s()), "edge_features", B=B, C=Cc, N=N, k=k)
    return ee


def edge_features_cm_bwd(dE: Tensor, idx_local: Tensor, k: int) -> Tensor:
    """Adjoint of edge_features_cm w.r.t. x: dE [B,2C,N,k], idx int64 [B,N*k] local -> dx [B,C,N] (deterministic: per-point slot lists)."""
    _f32(dE, "dE", 4)
    B, C2, N, kk = dE.shape
    if kk != k or C2 % 2 or idx_local.dtype != torch.int64 or idx_local.numel() != B * N * k:
        raise ValueError("edge_features_cm_bwd: dE [B,2C,N,k] and idx int64 [B,N*k] expected")
    lib = _lib.load()
    idx_local = idx_local.contiguous()
    rowptr = torch.empty((B * N, 2), dtype=torch.int32, device=dE.device)
    src = torch.empty((B * N * k,), dtype=torch.int32, device=dE.device)
    bad = index_check_flag(dE.device)
    check(lib.spgan_gather_csr(_p(idx_local), B, N * k, N, _p(rowptr), _p(src), None if bad is None else _p(bad), _s()), "gather_csr", B=B, S=N * k, N=N)
    index_check_raise(bad, "edge_features backward: a neighbour index lies outside [0, %d)" % N)
    dx = torch.empty((B, C2 // 2, N), dtype=torch.float32, device=dE.device)
    check(lib.spgan_edge_features_cm_bwd(_p(dE), _p(rowptr), _p(src), B, C2 // 2, N, k, _p(dx), _s()), "edge_features_cm_bwd", B=B, C=C2 // 2, N=N, k=k)
    return dx


def index_check_flag(device) -> Optional[Tensor]:
    """The gather adjoints drop out-of-range indices silently (the forward gather would already have read out of bounds).  With
    SPGAN_CHECK_INDICES=1 the CSR builder reports them through a device flag that `index_check_raise` reads back -- one host
    synchronisation per backward, so it is a debugging switch, not the default."""
    if os.environ.get("SPGAN_CHECK_INDICES", "0") != "1" or capturing():
        return None
    return torch.zeros(1, dtype=torch.int32, device=device)


def index_check_raise(bad: Optional[Tensor], msg: str) -> None:
    if bad is not None and int(bad.item()) != 0:
        raise IndexError(msg)


def idx_to_local64(idx: Tensor, B: int, N: int) -> Tensor:
    _i32(idx, "idx")
    k = idx.shape[1]
    out = torch.empty((B, N * k), dtype=torch.int64, device=idx.device)
    check(_lib.load().spgan_idx_to_local64(_p(idx), B, N, k, _p(out), _s()), "idx_to_local64")
    return out


def idx_from_local64(idx_local: Tensor, B: int, N: int, k: int) -> Tensor:
    if idx_local.dtype != torch.int64 or not idx_local.is_cuda:
        raise TypeError("idx must be an int64 GPU tensor")
    idx_local = idx_local.contiguous()
    out = torch.empty((B * N, k), dtype=torch.int32, device=idx_local.device)
    check(_lib.load().spgan_idx_from_local64(_p(idx_local), B, N, k, _p(out), _s()), "idx_from_local64")
    return out


# ----------------------------------------------------------------------------- layout
def cm_to_pm(x_cm: Tensor) -> Tensor:
    _f32(x_cm, "x", 3)
    x_cm = x_cm.contiguous()
    B, Cc, N = x_cm.shape
    y = torch.empty((B * N, Cc), dtype=torch.float32, device=x_cm.device)
    check(_lib.load().spgan_cm_to_pm(_p(x_cm), B, Cc, N, _p(y), _s()), "cm_to_pm")
    return y


def pm_to_cm(x_pm: Tensor, B: int, N: int) -> Tensor:
    _f32(x_pm, "x", 2)
    x_pm = x_pm.contiguous()
    Cc = x_pm.shape[1]
    y = torch.empty((B, Cc, N), dtype=torch.float32, device=x_pm.device)
    check(_lib.load().spgan_pm_to_cm(_p(x_pm), B, Cc, N, _p(y), _s()), "pm_to_cm")
    return y


def concat2(a: Tensor, b: Tensor) -> Tensor:
    _f32(a, "a", 2); _f32(b, "b", 2)
    a = a.contiguous(); b = b.contiguous()
    M = a.shape[0]
    out = torch.empty((M, a.shape[1] + b.shape[1]), dtype=torch.float32, device=a.device)
    check(_lib.load().spgan_concat2(_p(a), a.shape[1], _p(b), b.shape[1], M, _p(out), _s()), "concat2")
    return out


# ----------------------------------------------------------------------------- column tail of the M <= 64 kernel (spgan_coltail)
# The per-shape linears (M = batch <= 64) run on a kernel whose workgroups own their columns entirely: their column statistics are
# finished inside the producing launch instead of a finalize launch.  Larger products always use the separate finalize launches
# (round 2's last-arriving-workgroup merge measured slower on every shape of the step: tools/exp/fanin.hpp, profiles/r02_fanin_ab.txt).
def _owns_columns(lib, a) -> bool:
    return bool(lib.spgan_gemm_nt_owns_columns(C.byref(a)))


def _tail_bn(tail, gamma, beta, rm, rv, out4, count_rep: int = 1):
    tail.enabled = 1
    tail.mode = 0
    tail.gamma, tail.beta, tail.rmean, tail.rvar = _p(gamma), _p(beta), _p(rm), _p(rv)
    tail.scale, tail.shift, tail.invstd, tail.mean_out = _p(out4[0]), _p(out4[1]), _p(out4[2]), _p(out4[3])
    tail.eps, tail.momentum, tail.count_rep = BN_EPS, _BN_MOM[0], int(count_rep)


# ----------------------------------------------------------------------------- contractions
def _finalize(partials: Tensor, groups: int, tpg: int, Cn: int, G: int, mode: int, tile_rows: int = 0) -> Tuple[Tensor, Tensor]:
    out = torch.empty((2, groups, Cn), dtype=torch.float32, device=partials.device)
    check(_lib.load().spgan_colstats_finalize(_p(partials), groups, tpg, Cn, G, mode, tile_rows, _p(out[0]), _p(out[1]), _s()), "colstats_finalize")
    return out[0], out[1]


def gemm_nt(A: Tensor, W: Tensor, bias: Optional[Tensor] = None, *, pro=None, edge=None, rowbias: Optional[Tensor] = None,
            rows_per_group: int = 0, act: int = ACT_NONE, slope: float = 0.0, stats: bool = False, M: Optional[int] = None, bn=None,
            out: Optional[Tensor] = None, exact: bool = False, count_rep: int = 1, out_bf16: bool = False, out_half: bool = False):
    """Y[M,N] = act( pro(A) @ W^T + bias + rowbias[m // rows_per_group] ).
    16-bit storage ("f16" operand mode only; see storage16()): A may be a float16 tensor (plain operand: the EdgeBlock's T, written
    by edge_attend_fwd(half=True)); out_bf16=True returns Y as bfloat16 (plain linear product: the EdgeBlock's dT); out_half=True
    returns Y as float16 (no activation; statistics / bn still from the fp32 accumulators: the EdgeBlock's h2pre).
    out  = destination [M,N] (unit column stride; may be a column slice of a wider buffer) instead of a fresh tensor.
    bn = (gamma, beta, running_mean | None, running_var | None): train-mode BatchNorm of Y fused behind the GEMM: returns
         (Y, (scale, shift, invstd, mean)) and updates the running statistics -- column statistics in the epilogue, merged by a
         finalize launch (in the producing launch itself for M <= 64: spgan_coltail).  count_rep: the rows stand for count_rep
         identical copies (only the unbiased-variance count of the running statistics changes).
    pro  = (scale[K], shift[K], slope): operand a = lrelu(A*scale+shift)            (A_AFFINE_LRELU)
    edge = (idx[M,k], ebias[K]) with pro: rows are edges, a = lrelu((A[j]-A[i]+ebias)*scale+shift)  (A_EDGE)
    stats=True additionally returns (mean[N], biased var[N]) of the pre-activation output over all M rows.
    exact=True keeps fp32 operands even in fp16-operand mode: for products whose operands are sums over all points (Gram matrices,
    column sums: they grow with B*N and leave fp16's range at full size) rather than per-point activations.
    A may be an Affine2 operand (a lazy BatchNorm-backward tensor; plain products only: no pro / edge)."""
    a2 = A if isinstance(A, Affine2) else None
    if a2 is not None:
        if pro is not None or edge is not None or a2.shape[0] <= 64 or M is not None:
            a2, A = None, A.dense()
        else:
            A = a2.g
    a_half = isinstance(A, torch.Tensor) and A.dtype == torch.float16
    if a_half:
        if pro is not None or edge is not None or a2 is not None or exact or _MFMA_F16[0] != 1:
            raise ValueError("a float16 A is a plain operand of the fp16-operand mode (set_mfma_operands('f16'))")
        _rowmajor2d_as(A, "A", torch.float16)
    else:
        _rowmajor2d(A, "A")
    _rowmajor2d(W, "W")
    N, K = W.shape
    a = GemmNTArgs(); a.mfma_f16 = 0 if exact else _MFMA_F16[0]; a.tile_hint = _NT_TILE_HINT[0]
    a.a_half = 1 if a_half else 0
    if edge is not None:
        idx, ebias = edge
        _i32(idx, "idx")
        if pro is None:
            raise ValueError("edge operand needs pro=(scale, shift, slope)")
        M_ = idx.shape[0] * idx.shape[1]
        a.a_mode = A_EDGE; a.e_idx = _p(idx); a.e_k = idx.shape[1]; a.e_bias = _p(_vec(ebias, K, "ebias"))
    else:
        M_ = A.shape[0] if M is None else M
        a.a_mode = A_PLAIN if pro is None else A_AFFINE_LRELU
    if A.shape[1] != K:
        raise ValueError("A has %d columns but W has K=%d" % (A.shape[1], K))
    if pro is not None:
        sc, sh, ps = pro
        a.p_scale = _p(_vec(sc, K, "pro.scale")); a.p_shift = _p(_vec(sh, K, "pro.shift")); a.p_slope = float(ps)
    if a2 is not None:
        a.a_mode = A_AFFINE_LRELU
        a.p_scale = _p(_vec(a2.p, K, "p")); a.p_shift = _p(_vec(a2.r, K, "r")); a.p_slope = 1.0
        a.A2 = _p(a2.y); a.lda2 = _ld(a2.y); a.p_scale2 = _p(_vec(a2.q, K, "q"))
    if out_bf16:
        if out is not None or stats or bn is not None or act != ACT_NONE:
            raise ValueError("out_bf16 goes with a plain linear product into a fresh tensor")
        Y = torch.empty((M_, N), dtype=torch.bfloat16, device=A.device)
        a.y_bf16 = 1
    elif out_half:
        if out is not None or act != ACT_NONE:
            raise ValueError("out_half goes with a linear product (no activation) into a fresh tensor")
        Y = torch.empty((M_, N), dtype=torch.float16, device=A.device)
        a.y_half = 1
    elif out is None:
        Y = torch.empty((M_, N), dtype=torch.float32, device=A.device)
    else:
        Y = _rowmajor2d(out, "out")
        if tuple(Y.shape) != (M_, N):
            raise ValueError("out must be [%d,%d], got %s" % (M_, N, tuple(Y.shape)))
    a.A = _p(A); a.lda = _ld(A); a.W = _p(W); a.ldw = _ld(W); a.Y = _p(Y); a.ldy = _ld(Y)
    a.M, a.N, a.K = M_, N, K
    a.epi_mode = EPI_LINEAR
    a.bias = _p(_vec(bias, N, "bias"))
    if rowbias is not None:
        _rowmajor2d(rowbias, "rowbias")
        if rows_per_group <= 0 or rowbias.shape[1] != N or rowbias.shape[0] * rows_per_group < M_:
            raise ValueError("rowbias [%s] does not cover M=%d rows in groups of %d" % (tuple(rowbias.shape), M_, rows_per_group))
        a.rowbias = _p(rowbias); a.rows_per_group = rows_per_group; a.ld_rowbias = _ld(rowbias)
    a.act = act; a.act_slope = float(slope)
    part = None
    if stats or bn is not None:
        tiles = (M_ + ROW_TILE - 1) // ROW_TILE
        part = torch.empty((tiles, N, 2), dtype=torch.float32, device=A.device)
        a.stats = _p(part)
    lib = _lib.load()
    fused = part is not None and _owns_columns(lib, a)
    res = None
    if fused:
        if bn is not None:
            res = torch.empty((4, N), dtype=torch.float32, device=A.device)
            _tail_bn(a.tail, bn[0], bn[1], bn[2], bn[3], res, count_rep)
        else:
            res = torch.empty((2, N), dtype=torch.float32, device=A.device)
            a.tail.enabled = 1; a.tail.mode = 0; a.tail.out0 = _p(res[0]); a.tail.out1 = _p(res[1])
    _wimg = _attach_w_image(a, W)      # noqa: F841 (kept alive across the launch)
    done = launch_timer("gemm_nt", a) if launch_timer is not None else None
    check(lib.spgan_gemm_nt(C.byref(a), _s()), "gemm_nt", M=M_, N=N, K=K, a_mode=a.a_mode)
    if done is not None:
        done()
    if bn is not None:
        if fused:
            return Y, (res[0], res[1], res[2], res[3])
        gamma, beta, rm, rv = bn
        if count_rep != 1:
            mean, var = _finalize(part, 1, part.shape[0], N, M_, 0)
            return Y, bn_prepare(mean[0].contiguous(), var[0].contiguous(), gamma, beta, M_ * count_rep, True, rm, rv)
        out = torch.empty((4, N), dtype=torch.float32, device=A.device)
        check(lib.spgan_colstats_finalize_bn(_p(part), part.shape[0], N, M_, 0, _p(gamma), _p(beta), BN_EPS, _BN_MOM[0], _p(rm), _p(rv),
                                             _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]), _s()), "colstats_finalize_bn", N=N, M=M_)
        return Y, (out[0], out[1], out[2], out[3])
    if stats:
        if fused:
            return Y, res[0], res[1]
        mean, var = _finalize(part, 1, part.shape[0], N, M_, 0)
        return Y, mean[0], var[0]
    return Y


def _batched3d(t: Tensor, name: str) -> Tensor:
    _f32(t, name, 3)
    if (t.shape[2] > 1 and t.stride(2) != 1) or t.stride(1) < t.shape[2]:
        raise ValueError("%s must be [batch, rows, cols] with unit column stride, got strides %s" % (name, t.stride()))
    return t


def gemm_nt_batched(A: Tensor, W: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """Y[z] = A[z] @ W[z]^T for z < batch in ONE launch: A [Z,M,K], W [Z,N,K] -> [Z,M,N].  Row and batch strides are free
    (column slices of wider buffers, or stride 0 to share an operand); `out` may be such a view too."""
    _batched3d(A, "A"); _batched3d(W, "W")
    Z, M_, K = A.shape
    if W.shape[0] != Z or W.shape[2] != K:
        raise ValueError("A %s and W %s do not form a batched A @ W^T" % (tuple(A.shape), tuple(W.shape)))
    N = W.shape[1]
    if out is None:
        out = torch.empty((Z, M_, N), dtype=torch.float32, device=A.device)
    elif tuple(_batched3d(out, "out").shape) != (Z, M_, N):
        raise ValueError("out must be [%d,%d,%d], got %s" % (Z, M_, N, tuple(out.shape)))
    a = GemmNTArgs(); a.mfma_f16 = _MFMA_F16[0]; a.tile_hint = _NT_TILE_HINT[0]
    a.A = _p(A); a.lda = A.stride(1); a.W = _p(W); a.ldw = W.stride(1); a.Y = _p(out); a.ldy = out.stride(1)
    a.M, a.N, a.K = M_, N, K
    a.a_mode = A_PLAIN; a.epi_mode = EPI_LINEAR; a.act = ACT_NONE
    a.batch = Z; a.batch_stride_a = A.stride(0); a.batch_stride_w = W.stride(0); a.batch_stride_y = out.stride(0)
    done = launch_timer("gemm_nt", a) if launch_timer is not None else None
    check(_lib.load().spgan_gemm_nt(C.byref(a), _s()), "gemm_nt_batched", Z=Z, M=M_, N=N, K=K)
    if done is not None:
        done()
    return out


def gemm_nt_maskout(A: Tensor, W: Tensor, ref: Tensor, slope: float, with_colsum: bool = False):
    """Y = (A @ W^T) * (ref > 0 ? 1 : slope): input-gradient through an (in-place) LeakyReLU whose output `ref` was saved.
    with_colsum=True: returns (Y, column sums of Y [N]) -- the bias gradient of the layer below; for the per-shape linears (M <= 64,
    aligned operands) they come out of the same launch, otherwise from a column-sum launch."""
    _rowmajor2d(A, "A"); _rowmajor2d(W, "W"); _rowmajor2d(ref, "ref")
    N, K = W.shape
    M_ = A.shape[0]
    Y = torch.empty((M_, N), dtype=torch.float32, device=A.device)
    a = GemmNTArgs(); a.mfma_f16 = _MFMA_F16[0]; a.tile_hint = _NT_TILE_HINT[0]
    a.A = _p(A); a.lda = _ld(A); a.W = _p(W); a.ldw = _ld(W); a.Y = _p(Y); a.ldy = N
    a.M, a.N, a.K = M_, N, K
    a.a_mode = A_PLAIN; a.epi_mode = EPI_MASK_OUT
    a.ref = _p(ref); a.ld_ref = _ld(ref); a.b_slope = float(slope)
    lib = _lib.load()
    res = part = None
    # the M <= 64 kernel is the only one with this epilogue's column sums
    if with_colsum and _owns_columns(lib, a):
        part = torch.empty((1, N, 2), dtype=torch.float32, device=A.device)
        res = torch.empty((2, N), dtype=torch.float32, device=A.device)
        a.stats = _p(part)
        a.tail.enabled = 1; a.tail.mode = 1; a.tail.out0 = _p(res[0]); a.tail.out1 = _p(res[1])
    _wimg = _attach_w_image(a, W)      # noqa: F841 (kept alive across the launch)
    done = launch_timer("gemm_nt", a) if launch_timer is not None else None
    check(lib.spgan_gemm_nt(C.byref(a), _s()), "gemm_nt_maskout", M=M_, N=N, K=K)
    if done is not None:
        done()
    if not with_colsum:
        return Y
    return Y, (res[0] if res is not None else colsum(Y)[0])


class SparseAffine:
    """A lazily evaluated [M,C] operand  y*alpha[c] + beta[c] + (sp_arg[b,c] == m ? sp_val[b,c] : 0),  b = m // rows.
    It stands for the BatchNorm backward behind D's global max-pool (dense, but a function of y and B*C numbers):
    the GEMMs that consume it evaluate it on their operand load instead of reading a materialised [M,C] tensor."""

    def __init__(self, y: Optional[Tensor], alpha: Tensor, beta: Tensor, sp_val: Tensor, sp_arg: Tensor, rows: int):
        # y may be None when the consumer is the collapsed backward (nets.d_backward), which never evaluates the operand
        self.y, self.alpha, self.beta, self.sp_val, self.sp_arg, self.rows = y, alpha.contiguous(), beta.contiguous(), sp_val.contiguous(), sp_arg, rows
        self.shape = y.shape if y is not None else (sp_val.shape[0] * rows, sp_val.shape[1])
        self.device = sp_val.device


class Affine2:
    """A lazily evaluated [M,C] operand  g*p[c] + y*q[c] + r[c]: the BatchNorm backward dy = gamma*invstd*(g - S0/count - xhat*S1/count)
    written as an affine combination of the incoming gradient g and the layer's pre-BatchNorm output y (spgan_bn_bwd_coeffs).  The
    weight-gradient and input-gradient GEMMs that consume dy evaluate it on their operand loads (spgan_gemm_tn_args.A2 /
    spgan_gemm_nt_args.A2) -- the [M,C] tensor is never written or read back."""

    def __init__(self, g: Tensor, y: Tensor, coef: Tensor):
        if g.shape != y.shape or coef.shape != (3, g.shape[1]):
            raise ValueError("Affine2: g and y must have equal shapes, coef [3,C]")
        self.half = g.dtype == torch.bfloat16       # 16-bit storage ("f16" operand mode): g bfloat16 + y float16, both or neither
        if self.half != (y.dtype == torch.float16):
            raise ValueError("Affine2: a bfloat16 g goes with a float16 y (16-bit storage), float32 with float32")
        self.g, self.y, self.coef = g, y, coef
        self.p, self.q, self.r = coef[0], coef[1], coef[2]
        self.shape, self.device = g.shape, g.device

    def dense(self) -> Tensor:
        """The materialised tensor (for a consumer without the two-tensor operand): p*g + (q*y + r), the kernels' expression."""
        return torch.addcmul(torch.addcmul(self.r, self.y.float(), self.q), self.g.float(), self.p)


def bn_bwd_lazy(g: Tensor, y: Tensor, mean: Tensor, invstd: Tensor, gamma: Optional[Tensor], sums: Tensor, count: int) -> Affine2:
    """bn_bwd_apply as a lazy operand: one C-sized launch for the coefficients instead of a pass over [M,C]."""
    if g.dtype == torch.bfloat16:               # 16-bit storage: the tensors are only handed on, the coefficients come from the sums
        _rowmajor2d_as(g, "g", torch.bfloat16); _rowmajor2d_as(y, "y", torch.float16)
    else:
        _f32(g, "g", 2); _f32(y, "y", 2)
    if not (g.is_contiguous() and y.is_contiguous()) or g.shape != y.shape:
        raise ValueError("g and y must be contiguous with equal shapes")
    Cn = g.shape[1]
    coef = torch.empty((3, Cn), dtype=torch.float32, device=g.device)
    check(_lib.load().spgan_bn_bwd_coeffs(_p(_vec(sums, 2 * Cn, "sums")), _p(_vec(mean, Cn, "mean")), _p(_vec(invstd, Cn, "invstd")),
                                          _p(None if gamma is None else _vec(gamma, Cn, "gamma")), Cn, float(count), _p(coef), _s()), "bn_bwd_coeffs", C=Cn)
    return Affine2(g, y, coef)


def sparse_bn_bwd_operand(gval: Tensor, argmax: Tensor, y: Tensor, N: int, mean, invstd, gamma, sums, count: int) -> SparseAffine:
    """The same quantity bn_bwd_apply_sparse materialises, as a lazy operand (O(C) + O(B*C) preparation only)."""
    B, Cn = gval.shape
    ab = torch.empty((2, Cn), dtype=torch.float32, device=gval.device)
    cg = torch.empty_like(gval)
    check(_lib.load().spgan_sparse_bn_prep(_p(gval.contiguous()), _p(_vec(mean, Cn, "mean")), _p(_vec(invstd, Cn, "invstd")), _p(_vec(gamma, Cn, "gamma")),
                                           _p(_vec(sums, 2 * Cn, "sums")), B, Cn, count, _p(ab[0]), _p(ab[1]), _p(cg), _s()), "sparse_bn_prep")
    return SparseAffine(y, ab[0], ab[1], cg, argmax, N)


def bn_dbl_coeffs(U0, U1, Ugz, S0, S1, gamma, invstd, count: int) -> Tensor:
    """-> [4, C] = [dgammaA | sbarA | xsum0 | xsum1] (phase A of the double backward, per channel)."""
    Cn = U0.numel()
    out = torch.empty((4, Cn), dtype=torch.float32, device=U0.device)
    v = lambda t, n: _p(_vec(t.contiguous(), Cn, n))
    check(_lib.load().spgan_bn_dbl_coeffs(v(U0, "U0"), v(U1, "U1"), v(Ugz, "Ugz"), v(S0, "S0"), v(S1, "S1"), v(gamma, "gamma"), v(invstd, "invstd"),
                                          Cn, count, _p(out), _s()), "bn_dbl_coeffs")
    return out


def bn_dbl_phaseb(coeffs, gamma: Tensor, invstd: Tensor, s0: Optional[Tensor], s1: Optional[Tensor]):
    """-> (sums [2C] for bn_bwd_apply, dgamma [C]) (phase B of the double backward, per channel).
    coeffs: the [4,C] block of bn_dbl_coeffs, or the tuple (U0, U1, Ugz, S0, S1, count) it would be computed from -- then both steps run as one launch."""
    Cn = gamma.numel()
    sums = torch.empty((2 * Cn,), dtype=torch.float32, device=gamma.device)
    dg = torch.empty((Cn,), dtype=torch.float32, device=gamma.device)
    if isinstance(coeffs, tuple):
        U0, U1, Ugz, S0, S1, count = coeffs
        v = lambda t, n: _p(_vec(t.contiguous(), Cn, n))
        check(_lib.load().spgan_bn_dbl_phaseb_sums(v(U0, "U0"), v(U1, "U1"), v(Ugz, "Ugz"), v(S0, "S0"), v(S1, "S1"), v(gamma, "gamma"), v(invstd, "invstd"),
                                                   _p(None if s0 is None else s0.contiguous()), _p(None if s1 is None else s1.contiguous()), Cn, int(count),
                                                   _p(sums), _p(dg), _s()), "bn_dbl_phaseb_sums")
        return sums, dg
    check(_lib.load().spgan_bn_dbl_phaseb(_p(coeffs), _p(_vec(gamma.contiguous(), Cn, "gamma")), _p(_vec(invstd.contiguous(), Cn, "invstd")),
                                          _p(None if s0 is None else s0.contiguous()), _p(None if s1 is None else s1.contiguous()), Cn, _p(sums), _p(dg), _s()),
          "bn_dbl_phaseb")
    return sums, dg


def gemm_nt_bnbwd(A, W: Tensor, y_ref: Tensor, scale: Tensor, shift: Tensor, mean: Tensor, invstd: Tensor, slope: float,
                  edge=None, pro=None, bias: Optional[Tensor] = None, rowadd: Optional[Tensor] = None, coef_bn=None, phaseb=None, gout=None):
    """g = (pro(A) @ W^T + bias + rowadd) * lrelu'(z), z = y*scale+shift; returns (g, sum_c g, sum_c g*xhat), xhat = (y-mean)*invstd.
    With edge=(idx, ebias) y is the per-edge difference y[e] = P[idx[e]] - P[i] + ebias of the point tensor P=y_ref.
    A may be a SparseAffine or an Affine2 operand; pro=(scale[K], shift[K], slope) as in gemm_nt; rowadd is a dense [M,N] addend.
    coef_bn = (gamma | None, count): also return the BatchNorm-backward coefficients coef [3,N] of bn_bwd_lazy(g, y_ref, mean, invstd,
    gamma, [sum g | sum g*xhat], count) as a fourth result -- emitted by the launch that finishes the column sums (not with edge).
    phaseb / gout: as gemm_dual's (the double backward's phase B: the finalize launch also runs bn_dbl_phaseb on the sums it merges, the
    stored tensor is gout[0] + gout[1]*g) -> (g, s0, s1, sums, dgamma [, coef]); M > 64, no per-edge operand, not together with coef_bn."""
    if (phaseb is not None or gout is not None) and (edge is not None or coef_bn is not None):
        raise ValueError("gemm_nt_bnbwd: phaseb / gout take neither a per-edge operand nor coef_bn")
    sa = A if isinstance(A, SparseAffine) else None
    a2 = A if isinstance(A, Affine2) else None
    if sa is not None:
        A = sa.y
    if a2 is not None:
        if A.shape[0] <= 64 or pro is not None:
            a2, A = None, A.dense()              # the two-tensor operand is a path of the 128-row kernels
        else:
            A = a2.g
    if a2 is not None and a2.half:
        if edge is None or _MFMA_F16[0] != 1:
            raise ValueError("a 16-bit Affine2 operand is the EdgeBlock's (edge=...) in the 'f16' operand mode")
        _rowmajor2d_as(A, "A", torch.bfloat16)
    else:
        _rowmajor2d(A, "A")
    _rowmajor2d(W, "W"); _rowmajor2d(y_ref, "y_ref")
    N, K = W.shape
    M_ = A.shape[0]
    g = torch.empty((M_, N), dtype=torch.float32, device=A.device)
    tiles = (M_ + ROW_TILE - 1) // ROW_TILE
    part = torch.empty((tiles, N, 2), dtype=torch.float32, device=A.device)
    a = GemmNTArgs(); a.mfma_f16 = _MFMA_F16[0]; a.tile_hint = _NT_TILE_HINT[0]
    a.A = _p(A); a.lda = _ld(A); a.W = _p(W); a.ldw = _ld(W); a.Y = _p(g); a.ldy = N
    a.M, a.N, a.K = M_, N, K
    a.a_mode = A_PLAIN
    if sa is not None:
        a.a_mode = A_AFFINE_LRELU
        a.p_scale = _p(_vec(sa.alpha, K, "alpha")); a.p_shift = _p(_vec(sa.beta, K, "beta")); a.p_slope = 1.0
        a.sp_val = _p(sa.sp_val); a.sp_arg = _p(_i32(sa.sp_arg, "sp_arg")); a.sp_rows = sa.rows
    elif a2 is not None:
        a.a_mode = A_AFFINE_LRELU
        a.p_scale = _p(_vec(a2.p, K, "p")); a.p_shift = _p(_vec(a2.r, K, "r")); a.p_slope = 1.0
        a.A2 = _p(a2.y); a.lda2 = _ld(a2.y); a.p_scale2 = _p(_vec(a2.q, K, "q"))
        a.a_half = 1 if a2.half else 0
    elif pro is not None:
        a.a_mode = A_AFFINE_LRELU
        a.p_scale = _p(_vec(pro[0], K, "pro.scale")); a.p_shift = _p(_vec(pro[1], K, "pro.shift")); a.p_slope = float(pro[2])
    if bias is not None:
        a.bias = _p(_vec(bias, N, "bias"))
    if rowadd is not None:
        _rowmajor2d(rowadd, "rowadd")
        if rowadd.shape != (M_, N):
            raise ValueError("rowadd must be [M,N]")
        a.rowbias = _p(rowadd); a.rows_per_group = 1; a.ld_rowbias = _ld(rowadd)
    a.ref = _p(y_ref); a.ld_ref = _ld(y_ref)
    a.b_scale = _p(_vec(scale, N, "scale")); a.b_shift = _p(_vec(shift, N, "shift"))
    a.b_mean = _p(_vec(mean, N, "mean")); a.b_invstd = _p(_vec(invstd, N, "invstd")); a.b_slope = float(slope)
    a.stats = _p(part)
    if gout is not None:
        gadd, gsc = gout
        _rowmajor2d(gadd, "gout.add")
        if gadd.shape != (M_, N):
            raise ValueError("gout add must be [M,N]")
        a.gout_add = _p(gadd); a.ld_gout_add = _ld(gadd); a.gout_scale = _p(_vec(gsc, N, "gout.scale"))
    if edge is None:
        a.epi_mode = EPI_BNBWD
    else:
        idx, ebias = edge
        _i32(idx, "idx")
        a.epi_mode = EPI_EDGE_BNBWD; a.e_idx = _p(idx); a.e_k = idx.shape[1]; a.e_bias2 = _p(_vec(ebias, N, "ebias"))
    lib = _lib.load()
    res = None
    if phaseb is not None and M_ <= 64:
        raise ValueError("gemm_nt_bnbwd: phaseb needs the tile partials of the M > 64 kernels")
    if _owns_columns(lib, a):
        res = torch.empty((2, N), dtype=torch.float32, device=A.device)          # [sum g | sum g*xhat], contiguous (nets._cat2)
        a.tail.enabled = 1; a.tail.mode = 1; a.tail.out0 = _p(res[0]); a.tail.out1 = _p(res[1])
    _wimg = _attach_w_image(a, W)      # noqa: F841 (kept alive across the launch)
    done = launch_timer("gemm_nt", a) if launch_timer is not None else None
    check(lib.spgan_gemm_nt(C.byref(a), _s()), "gemm_nt_bnbwd", M=M_, N=N, K=K)
    if done is not None:
        done()
    if phaseb is not None:
        # one finalize launch: the plain sums, bn_dbl_phaseb on them and (4-tuple) the lazy-operand coefficients, as gemm_dual_multi issues it
        from ._lib import ColFinalizeArgs
        (U0, U1, Ugz, S0, S1, count), pg, pinv = phaseb[:3]
        fa = (ColFinalizeArgs * 1)()
        f = fa[0]
        fin = torch.empty((2, N), dtype=torch.float32, device=A.device)
        sums = torch.empty((2 * N,), dtype=torch.float32, device=A.device)
        dg = torch.empty((N,), dtype=torch.float32, device=A.device)
        vs = [_vec(t.contiguous(), N, nm) for t, nm in ((U0, "U0"), (U1, "U1"), (Ugz, "Ugz"), (S0, "S0"), (S1, "S1"), (pg, "gamma"), (pinv, "invstd"))]
        f.partials = _p(part); f.tiles = tiles; f.C = N; f.G = M_; f.tile_rows = 0; f.s0 = _p(fin[0]); f.s1 = _p(fin[1])
        f.kind = 2; f.U0, f.U1, f.Ugz, f.S0, f.S1, f.gamma, f.invstd = [_p(t) for t in vs]
        f.count = float(count); f.sums = _p(sums); f.dgamma = _p(dg)
        tail = (sums, dg)
        if len(phaseb) == 4:
            pmean = _vec(phaseb[3].contiguous(), N, "mean")
            coefB = torch.empty((3, N), dtype=torch.float32, device=A.device)
            f.mean = _p(pmean); f.pb_coef = _p(coefB)
            tail = (sums, dg, coefB)
        check(lib.spgan_colstats_finalize_multi(fa, 1, _s()), "colstats_finalize_multi", count=1)
        return (g, fin[0], fin[1]) + tail
    if coef_bn is not None:
        gamma, count = coef_bn
        if res is not None:                     # the M <= 64 kernel finished its sums itself: coefficients by the small kernel
            return g, res[0], res[1], bn_bwd_lazy(g, y_ref, mean, invstd, gamma, torch.cat([res[0], res[1]]), count).coef
        out = torch.empty((2, N), dtype=torch.float32, device=A.device)
        coef = torch.empty((3, N), dtype=torch.float32, device=A.device)
        check(lib.spgan_colstats_finalize_bnbwd(_p(part), tiles, N, M_, 0, _p(_vec(mean, N, "mean")), _p(_vec(invstd, N, "invstd")),
                                                _p(None if gamma is None else _vec(gamma, N, "gamma")), float(count), _p(out[0]), _p(out[1]), _p(coef), _s()),
              "colstats_finalize_bnbwd", N=N, M=M_)
        return g, out[0], out[1], coef
    if res is not None:
        return g, res[0], res[1]
    s0, s1 = _finalize(part, 1, tiles, N, M_, 1)
    return g, s0[0], s1[0]


class ActOperand:
    """A GEMM operand a = lrelu(x*scale + shift, slope) that is never materialised: the BatchNorm + LeakyReLU'd activation of a layer as
    the `dy` side of gemm_dual (the collapsed backward of the layer in front of the max-pool: dy := a3 = lrelu(bn3(y3)))."""

    def __init__(self, x: Tensor, scale: Tensor, shift: Tensor, slope: float):
        self.x, self.scale, self.shift, self.slope = x, scale, shift, float(slope)
        self.shape, self.device = x.shape, x.device

    def dense(self) -> Tensor:
        return affine_act(self.x, self.scale, self.shift, self.slope)


def gemm_dual_ok(dy, W: Tensor, y_ref: Tensor, edge=None) -> bool:
    """True when gemm_dual takes this layer backward (csrc/gemm_dual.hip: fp32 operands, (columns of dy, input channels) = (128, 64),
    (256, 128) or (256, 256), M % 32 == 0, M >= 8192, per-edge operand with k = 10 at (128, 64) only); otherwise the caller issues
    gemm_tn + gemm_nt_bnbwd."""
    # ("bf16x3" keeps this route: the split mode emulates exact fp32 products, so a layer whose two products leave ONE exact-fp32 launch is
    #  inside its contract; the "f16" mode rounds operands and has its own 16-bit routes)
    if _MFMA_F16[0] == 1 or _NT_TILE_HINT[0] != 0 or not GEMM_DUAL[0]:
        return False
    a2 = dy if isinstance(dy, Affine2) else None
    g = a2.g if a2 is not None else (dy.x if isinstance(dy, ActOperand) else dy)
    if not isinstance(g, Tensor) or g.dtype != torch.float32 or (a2 is not None and (a2.half or a2.y.dtype != torch.float32)):
        return False
    if W.dim() != 2 or y_ref.dtype != torch.float32:
        return False
    ek = 0 if edge is None else int(edge[0].shape[1])
    return bool(_lib.load().spgan_gemm_dual_wgs(g.shape[0], W.shape[0], W.shape[1], ek))


# "bf16x3": the collapsed 256 -> 1024 layer's two products (Gram matrix a3^T a3, input gradient a3.G4) as ONE exact-fp32 gemm_dual launch or as a
# pair on the split-bf16 kernels (gemm_tn_wide3 + gemm_wide3): the pair is faster (step 7.75 -> 7.63 ms for the two first-order passes alone,
# profiles/r06_split_pair_ab.txt).  SPGAN_SPLIT_PAIR=0 / SPLIT_PAIR[0] = False: the fused launch (A/B switch).
SPLIT_PAIR = [os.environ.get("SPGAN_SPLIT_PAIR", "1") != "0"]
_PAIR_OK: dict = {}


def collapsed_pair_preferred(M: int, K: int) -> bool:
    """True when the collapsed layer's backward should run as gemm_tn + gemm_nt_bnbwd although gemm_dual would take it: the split-bf16
    mode, with both products on its 256-row-tile kernels."""
    if not (_MFMA_F16[0] == 2 and SPLIT_PAIR[0] and TN_SPLIT_BF16[0] and _NT_TILE_HINT[0] == 0 and M >= 32768):
        return False
    key = (int(M), int(K))
    if key not in _PAIR_OK:
        # the input-gradient launch carries the stored-tile epilogue (gout), which only gemm_wide3.hip has: ask the library whether it runs there
        a = GemmNTArgs(); a.mfma_f16 = 2; a.M, a.N, a.K = key[0], key[1], key[1]; a.lda = a.ldw = a.ldy = a.ld_ref = a.ld_rowbias = key[1]
        a.a_mode = A_AFFINE_LRELU; a.p_slope = 0.2; a.epi_mode = EPI_BNBWD; a.rows_per_group = 1
        _PAIR_OK[key] = bool(_lib.load().spgan_gemm_nt_uses_w_image(C.byref(a)))
    return _PAIR_OK[key]


GEMM_DUAL = [True]      # test hook: False sends every layer backward through the two separate launches (tests compare the two routes)


def gemm_dual(dy, W: Tensor, y_ref: Tensor, scale: Tensor, shift: Tensor, mean: Tensor, invstd: Tensor, slope: float, edge=None, coef_bn=None,
              defer: bool = True, out: Optional[Tensor] = None, beta: float = 0.0, bias: Optional[Tensor] = None, rowadd: Optional[Tensor] = None,
              with_colsum: bool = False, phaseb=None, gout=None):
    """The weight-gradient AND the masked input-gradient product of one conv layer behind BatchNorm + LeakyReLU in one launch:
      dW [Na,Nb] = dy^T . lrelu(pre*scale + shift),   g = (dy . W + bias + rowadd) * lrelu'(pre*scale + shift),   s0 = sum g,  s1 = sum g*xhat
    dy: Affine2 (lazy BatchNorm backward), ActOperand (an activation formed on load) or a dense [M,Na] tensor; W [Na,Nb] the layer's weight
    as stored; pre = y_ref [M,Nb], or with edge=(idx, ebias) the per-edge difference y_ref[idx[e]] - y_ref[e // k] + ebias of the point
    tensor y_ref; bias [Nb] / rowadd [M,Nb]: optional addends of the input gradient in front of the mask.
    Returns (dW, g, s0, s1) [+ coef [3,Nb] with coef_bn=(gamma, count): the lazy-operand coefficients of the NEXT BatchNorm backward]
    [+ colsum(dy) [Na] with with_colsum].  dW's (and the column sums') split sum is deferred like gemm_tn(defer=True): valid after
    flush_tn(); out / beta: dW = beta*out + sum (accumulated in place).
    phaseb = ((U0, U1, Ugz, S0, S1, count), gamma, invstd) of the layer BELOW (the double backward): the finalize launch also runs
    bn_dbl_phaseb on the sums it merges -> the return tuple ends with (sums [2Nb], dgamma [Nb]); with a fourth element `mean` (of that layer)
    also the coefficients coef [3,Nb] of the lazy operand p*X + q*y + r = bn_bwd_apply(X, y, mean, invstd, None, sums, count) -> (..., sums, dgamma, coef).
    gout = (add [M,Nb], scale [Nb]): the returned g tensor holds add + scale*g instead of g (s0, s1 stay the sums of g): phase B's adjoint
    X = xbarA + gamma*g leaves this launch."""
    if phaseb is not None and len(phaseb) == 4:
        if edge is not None or coef_bn is not None:
            raise ValueError("gemm_dual: the phase-B coefficient tail (4-tuple phaseb) takes neither a per-edge operand nor coef_bn")
        return gemm_dual_multi([dict(dy=dy, W=W, y_ref=y_ref, scale=scale, shift=shift, mean=mean, invstd=invstd, slope=slope, out=out, beta=beta, bias=bias,
                                     rowadd=rowadd, with_colsum=with_colsum, phaseb=phaseb, gout=gout)], defer=defer, _force=True)[0]
    b = _gemm_dual_build(dy, W, y_ref, scale, shift, mean, invstd, slope, edge=edge, out=out, beta=beta, bias=bias, rowadd=rowadd, with_colsum=with_colsum,
                         gout=gout)
    a = b["a"]
    done = launch_timer("gemm_dual", a) if launch_timer is not None else None
    check(_lib.load().spgan_gemm_dual(C.byref(a), _s()), "gemm_dual", M=b["M"], Na=b["Na"], Nb=b["Nb"], k=b["ek"])
    if done is not None:
        done()
    _gemm_dual_pending(b)
    if not defer:
        flush_tn()
    lib = _lib.load()
    M_, Nb, runs, rows_wg, part = b["M"], b["Nb"], b["runs"], b["rows_wg"], b["part"]
    dW, g = b["dW"], b["g"]
    extra = () if b["cs_out"] is None else (b["cs_out"],)
    if coef_bn is not None:
        gamma, count = coef_bn
        fin = torch.empty((2, Nb), dtype=torch.float32, device=g.device)
        coef = torch.empty((3, Nb), dtype=torch.float32, device=g.device)
        check(lib.spgan_colstats_finalize_bnbwd(_p(part), runs, Nb, M_, rows_wg, _p(_vec(mean, Nb, "mean")), _p(_vec(invstd, Nb, "invstd")),
                                                _p(None if gamma is None else _vec(gamma, Nb, "gamma")), float(count), _p(fin[0]), _p(fin[1]), _p(coef), _s()),
              "colstats_finalize_bnbwd", N=Nb, M=M_)
        return (dW, g, fin[0], fin[1], coef) + extra
    if phaseb is not None:
        (U0, U1, Ugz, S0, S1, count), pg, pinv = phaseb
        fin = torch.empty((2, Nb), dtype=torch.float32, device=g.device)
        sums = torch.empty((2 * Nb,), dtype=torch.float32, device=g.device)
        dg = torch.empty((Nb,), dtype=torch.float32, device=g.device)
        v = lambda t, n: _p(_vec(t.contiguous(), Nb, n))
        check(lib.spgan_colstats_finalize_phaseb(_p(part), runs, Nb, M_, rows_wg, v(U0, "U0"), v(U1, "U1"), v(Ugz, "Ugz"), v(S0, "S0"), v(S1, "S1"),
                                                 v(pg, "gamma"), v(pinv, "invstd"), int(count), _p(fin[0]), _p(fin[1]), _p(sums), _p(dg), _s()),
              "colstats_finalize_phaseb", N=Nb, M=M_)
        return (dW, g, fin[0], fin[1]) + extra + (sums, dg)
    s0, s1 = _finalize(part, 1, runs, Nb, M_, 1, rows_wg)
    return (dW, g, s0[0], s1[0]) + extra


def _gemm_dual_build(dy, W: Tensor, y_ref: Tensor, scale: Tensor, shift: Tensor, mean: Tensor, invstd: Tensor, slope: float, edge=None,
                     out: Optional[Tensor] = None, beta: float = 0.0, bias: Optional[Tensor] = None, rowadd: Optional[Tensor] = None,
                     with_colsum: bool = False, gout=None) -> dict:
    """The argument block and the output / workspace tensors of one spgan_gemm_dual problem (shared by the stand-alone and the grouped launch)."""
    a2 = dy if isinstance(dy, Affine2) else None
    act = dy if isinstance(dy, ActOperand) else None
    A = a2.g if a2 is not None else (act.x if act is not None else dy)
    _rowmajor2d(A, "dy"); _rowmajor2d(W, "W"); _rowmajor2d(y_ref, "y_ref")
    M_, Na = A.shape
    Nb = W.shape[1]
    if W.shape[0] != Na:
        raise ValueError("W must be [Na, Nb] with Na = columns of dy")
    lib = _lib.load()
    ek = 0 if edge is None else int(edge[0].shape[1])
    runs = lib.spgan_gemm_dual_wgs(M_, Na, Nb, ek)
    if not runs:
        raise ValueError("gemm_dual: unsupported shape M=%d Na=%d Nb=%d k=%d (see gemm_dual_ok)" % (M_, Na, Nb, ek))
    rows_wg = lib.spgan_gemm_dual_rows_per_wg(M_, Na, Nb)
    a = GemmDualArgs()
    a.A = _p(A); a.lda = _ld(A)
    if a2 is not None:
        _rowmajor2d(a2.y, "dy.y")
        a.a_mode = 1
        a.A2 = _p(a2.y); a.lda2 = _ld(a2.y)
        a.p = _p(_vec(a2.p, Na, "p")); a.q = _p(_vec(a2.q, Na, "q")); a.r = _p(_vec(a2.r, Na, "r"))
    elif act is not None:
        a.a_mode = 2; a.a_slope = act.slope
        a.p = _p(_vec(act.scale, Na, "scale")); a.r = _p(_vec(act.shift, Na, "shift"))
    a.W = _p(W); a.ldw = _ld(W)
    a.B = _p(y_ref); a.ldb = _ld(y_ref)
    if edge is not None:
        idx, ebias = edge
        _i32(idx, "idx")
        if idx.shape[0] * idx.shape[1] != M_:
            raise ValueError("edge operand: dy must have one row per edge")
        a.e_idx = _p(idx); a.e_k = ek; a.e_bias = _p(_vec(ebias, Nb, "ebias"))
    elif y_ref.shape[0] != M_:
        raise ValueError("dy and y_ref disagree on the rows: %d vs %d" % (M_, y_ref.shape[0]))
    a.b_scale = _p(_vec(scale, Nb, "scale")); a.b_shift = _p(_vec(shift, Nb, "shift"))
    a.b_mean = _p(_vec(mean, Nb, "mean")); a.b_invstd = _p(_vec(invstd, Nb, "invstd")); a.slope = float(slope)
    if bias is not None:
        a.bias = _p(_vec(bias, Nb, "bias"))
    if rowadd is not None:
        _rowmajor2d(rowadd, "rowadd")
        if tuple(rowadd.shape) != (M_, Nb):
            raise ValueError("rowadd must be [M,Nb]")
        a.rowadd = _p(rowadd); a.ld_rowadd = _ld(rowadd)
    if gout is not None:
        gadd, gscale = gout
        _rowmajor2d(gadd, "gout.add")
        if tuple(gadd.shape) != (M_, Nb):
            raise ValueError("gout: add must be [M,Nb]")
        a.gout_add = _p(gadd); a.ld_gout_add = _ld(gadd); a.gout_scale = _p(_vec(gscale, Nb, "gout.scale"))
    g = torch.empty((M_, Nb), dtype=torch.float32, device=A.device)
    part = torch.empty((runs, Nb, 2), dtype=torch.float32, device=A.device)
    ws = torch.empty((runs, Na, Nb), dtype=torch.float32, device=A.device)
    if out is None:
        dW, beta = torch.empty((Na, Nb), dtype=torch.float32, device=A.device), 0.0
    else:
        dW = _rowmajor2d(out, "out")
        if tuple(out.shape) != (Na, Nb):
            raise ValueError("out must be [Na, Nb]")
    cs_ws = cs_out = None
    if with_colsum:
        cs_ws = torch.empty((runs, Na), dtype=torch.float32, device=A.device)
        cs_out = torch.empty((Na,), dtype=torch.float32, device=A.device)
        a.colsum_ws = _p(cs_ws)
    a.G = _p(g); a.ldg = Nb; a.stats = _p(part); a.ws = _p(ws)
    a.M, a.Na, a.Nb = M_, Na, Nb
    return dict(a=a, M=M_, Na=Na, Nb=Nb, ek=ek, runs=runs, rows_wg=rows_wg, g=g, part=part, ws=ws, dW=dW, beta=float(beta), cs_ws=cs_ws, cs_out=cs_out,
                mean=mean, invstd=invstd)


def _gemm_dual_pending(b: dict) -> None:
    _PENDING_TN.append((b["ws"], b["dW"], b["runs"], b["Na"], b["Nb"], _ld(b["dW"]), b["beta"]))
    if b["cs_ws"] is not None:
        _PENDING_TN.append((b["cs_ws"], b["cs_out"], b["runs"], 1, b["Na"], b["Na"], 0.0))           # [runs][1 x Na] partials: one more entry of the multi-reduce


GROUPED = [os.environ.get("SPGAN_GROUPED", "1") != "0"]      # test / A-B hook: False issues every problem of a grouped call as its stand-alone launch


def gemm_dual_multi(specs, defer: bool = True, _force: bool = False):
    """Several gemm_dual problems of ONE geometry (equal M, Na, Nb, no per-edge operand) as one launch (spgan_gemm_dual_multi), followed by ONE
    finalize launch for all of them (spgan_colstats_finalize_multi): the same layer's backward of the D step's real pass, fake pass and of
    phase B of the penalty's double backward.  specs: one dict of gemm_dual's arguments per problem (dy, W, y_ref, scale, shift, mean, invstd,
    slope [, coef_bn | phaseb, out, beta, bias, rowadd, with_colsum]); -> the list of gemm_dual's return tuples, bit-identical to separate calls.
    defer=False: the weight-gradient sums (and column sums) of all problems are finished by one flush_tn()."""
    if (len(specs) == 1 or not GROUPED[0]) and not _force:
        return [gemm_dual(defer=defer, **sp) for sp in specs]
    from ._lib import ColFinalizeArgs, GROUP_MAX
    if len(specs) > GROUP_MAX:
        raise ValueError("gemm_dual_multi: at most %d problems per launch" % GROUP_MAX)
    tails = [(sp.get("coef_bn"), sp.get("phaseb")) for sp in specs]
    bs = [_gemm_dual_build(**{k: v for k, v in sp.items() if k not in ("coef_bn", "phaseb")}) for sp in specs]
    if any(b["ek"] for b in bs) or len({(b["M"], b["Na"], b["Nb"]) for b in bs}) != 1:
        raise ValueError("gemm_dual_multi: one geometry (M, Na, Nb), plain pre tensors")
    lib = _lib.load()
    n = len(bs)
    arr = (GemmDualArgs * n)(*[b["a"] for b in bs])
    done = None
    if launch_timer is not None:            # measurement hook: one record for the grouped launch, rows = the rows of all its problems
        agg = GemmDualArgs()
        agg.M, agg.Na, agg.Nb = n * bs[0]["M"], bs[0]["Na"], bs[0]["Nb"]
        done = launch_timer("gemm_dual", agg)
    check(lib.spgan_gemm_dual_multi(arr, n, _s()), "gemm_dual_multi", M=bs[0]["M"], Na=bs[0]["Na"], Nb=bs[0]["Nb"], count=n)
    if done is not None:
        done()
    for b in bs:
        _gemm_dual_pending(b)
    if not defer:
        flush_tn()
    fa = (ColFinalizeArgs * n)()
    res, keep = [], []
    for i, (b, (coef_bn, phaseb)) in enumerate(zip(bs, tails)):
        Nb, dev = b["Nb"], b["g"].device
        f = fa[i]
        fin = torch.empty((2, Nb), dtype=torch.float32, device=dev)
        f.partials = _p(b["part"]); f.tiles = b["runs"]; f.C = Nb; f.G = b["M"]; f.tile_rows = b["rows_wg"]; f.s0 = _p(fin[0]); f.s1 = _p(fin[1])
        extra = () if b["cs_out"] is None else (b["cs_out"],)
        if coef_bn is not None:
            gamma, count = coef_bn
            coef = torch.empty((3, Nb), dtype=torch.float32, device=dev)
            f.kind = 1; f.mean = _p(_vec(b["mean"], Nb, "mean")); f.invstd = _p(_vec(b["invstd"], Nb, "invstd"))
            f.gamma = _p(None if gamma is None else _vec(gamma, Nb, "gamma")); f.count = float(count); f.coef = _p(coef)
            res.append((b["dW"], b["g"], fin[0], fin[1], coef) + extra)
        elif phaseb is not None:
            (U0, U1, Ugz, S0, S1, count), pg, pinv = phaseb[:3]
            sums = torch.empty((2 * Nb,), dtype=torch.float32, device=dev)
            dg = torch.empty((Nb,), dtype=torch.float32, device=dev)
            vs = [_vec(t.contiguous(), Nb, nm) for t, nm in ((U0, "U0"), (U1, "U1"), (Ugz, "Ugz"), (S0, "S0"), (S1, "S1"), (pg, "gamma"), (pinv, "invstd"))]
            keep.append(vs)
            f.kind = 2; f.U0, f.U1, f.Ugz, f.S0, f.S1, f.gamma, f.invstd = [_p(t) for t in vs]
            f.count = float(count); f.sums = _p(sums); f.dgamma = _p(dg)
            tail = (sums, dg)
            if len(phaseb) == 4:      # + the lazy-operand coefficients of the BatchNorm backward these sums belong to
                pmean = _vec(phaseb[3].contiguous(), Nb, "mean")
                keep.append(pmean)
                coefB = torch.empty((3, Nb), dtype=torch.float32, device=dev)
                f.mean = _p(pmean); f.pb_coef = _p(coefB)
                tail = (sums, dg, coefB)
            res.append((b["dW"], b["g"], fin[0], fin[1]) + extra + tail)
        else:
            f.kind = 0
            res.append((b["dW"], b["g"], fin[0], fin[1]) + extra)
    check(lib.spgan_colstats_finalize_multi(fa, n, _s()), "colstats_finalize_multi", count=n)
    return res


_PENDING_TN: list = []      # deferred split-K reductions: (ws, out, splits, Na, Nb, ldc, beta); see gemm_tn(defer=True) / flush_tn()


def flush_tn() -> None:
    """Finish every gemm_tn(defer=True) product issued so far: one launch per 64 pending products."""
    if not _PENDING_TN:
        return
    from ._lib import MULTI_MAX, SplitKMultiArgs
    lib = _lib.load()
    pend = list(_PENDING_TN)
    _PENDING_TN.clear()
    for i0 in range(0, len(pend), MULTI_MAX):
        chunk = pend[i0:i0 + MULTI_MAX]
        a = SplitKMultiArgs()
        a.count = len(chunk)
        start = 0
        for e, (ws, out, splits, Na, Nb, ldc, beta) in enumerate(chunk):
            a.ws[e] = ws.data_ptr(); a.C[e] = out.data_ptr(); a.splits[e] = splits; a.Na[e] = Na; a.Nb[e] = Nb; a.ldc[e] = ldc; a.beta[e] = beta
            a.block_start[e] = start
            start += lib.spgan_splitk_reduce_blocks(splits, Na, Nb)
        a.block_start[len(chunk)] = start
        check(lib.spgan_splitk_reduce_multi(C.byref(a), _s()), "splitk_reduce_multi", count=len(chunk))


STORAGE16 = [os.environ.get("SPGAN_F16_STORAGE", "1") != "0"]
STORAGE16_MIN_EDGES = [81920]     # below this the fp32 tensors are small change


def storage16(E: int, F_: int, k: int) -> bool:
    """True when the EdgeBlock keeps its per-edge tensors in 16 bits in HBM: the "f16" operand mode (BASELINE configs[4]) at sizes where
    its 16-bit kernels are the ones that run.  Activations (h2pre, T) as float16, gradients (dT, g2, gy) as bfloat16 (fp32's exponent
    range); BatchNorm statistics and all column sums are taken from the fp32 accumulators / unrounded values."""
    return STORAGE16[0] and _MFMA_F16[0] == 1 and k == 10 and F_ % 8 == 0 and F_ > 64 and E >= max(STORAGE16_MIN_EDGES[0], k * TN_LP_MIN_ROWS)    # F/2 > 32 columns: the fp16 128-row kernels; E/k points: conv_out's bf16 weight-gradient kernel


# "bf16x3" operand mode: the weight gradients whose [Na, Nb] output tiles as 256 x 256, 256 x 128 or 128 x 256 run on the split-bf16 kernel
# (spgan_gemm_tn_args.mfma_lp == 2, csrc/gemm_tn_wide3.hip: every staged value split once per 128..256 columns of the other operand); every other
# shape keeps the exact-fp32 MFMA kernel.  Measured on MI355X (tools/tn3_bench.py, incl. the split-sum reduction): 65536 x 256 x 256: 112 -> 64 us,
# 196608 x 256 x 256: 323 -> 154, 65536 x 128 x 1280: 263 -> 179, 65536 x 256 x 128: 59 -> 49.  SPGAN_TN_SPLIT=0 / TN_SPLIT_BF16[0] = False: A/B switch.
TN_SPLIT_BF16 = [os.environ.get("SPGAN_TN_SPLIT", "1") != "0"]
TN_LP_MIN_ROWS = 8192     # "f16" operand mode: weight gradients reduce over >= this many points/edges on the bf16 matrix pipe


def gemm_tn(A: Tensor, Bm: Tensor, *, pro=None, edge=None, out: Optional[Tensor] = None, beta: float = 0.0, defer: bool = False,
            exact: bool = False, with_colsum: bool = False, a_pro=None):
    """C[Na,Nb] = beta*C + A^T @ pro(Bm): weight gradient, reduction over the M rows (points or edges).
    A may be a SparseAffine operand (evaluated on load).
    In the "f16" operand mode (set_mfma_operands) the products that reduce over the points / edges (M >= TN_LP_MIN_ROWS) round
    both operands to bfloat16 at the LDS staging -- fp32's exponent range: per-point gradients are 1e-5..1e-8 -- and accumulate in
    fp32 (spgan_gemm_tn_args.mfma_lp); the small weight-by-weight products and exact=True calls keep fp32 operands.
    defer=True: the returned tensor is NOT valid until flush_tn() ran -- the split-K partial sums of all the weight gradients of a
    backward pass are then finished by one launch (functions._deliver flushes before it hands gradients on).
    a_pro = (scale [Na], shift [Na], slope): the A operand is lrelu(A*scale + shift, slope), evaluated on load (plain A only).
    with_colsum=True: -> (C, colsum(A) [Na]): the column sums of the (transformed) A operand -- the bias gradient that belongs to this
    weight gradient -- come out of the same launch (spgan_gemm_tn_args.a_colsum_ws) and are finished by the same split reduction
    (deferred like C with defer=True) instead of a colsum pass of their own."""
    sa = A if isinstance(A, SparseAffine) else None
    a2 = A if isinstance(A, Affine2) else None
    if sa is not None:
        A = sa.y
    if a2 is not None:
        A = a2.g
    b_half = Bm.dtype == torch.float16
    a16 = a2 is not None and a2.half
    if a16:
        _rowmajor2d_as(A, "A", torch.bfloat16)
    else:
        _rowmajor2d(A, "A")
    if b_half:
        _rowmajor2d_as(Bm, "B", torch.float16)
        if pro is not None or edge is not None or sa is not None:
            raise ValueError("a float16 B is a plain operand (the EdgeBlock's T)")
    else:
        _rowmajor2d(Bm, "B")
    M_, Na = A.shape
    Nb = Bm.shape[1]
    a = GemmTNArgs()
    a.b_half = 1 if b_half else 0
    if a2 is not None:
        a.a_scale = _p(_vec(a2.p, Na, "p")); a.a_shift = _p(_vec(a2.r, Na, "r"))
        a.A2 = _p(a2.y); a.lda2 = _ld(a2.y); a.a_scale2 = _p(_vec(a2.q, Na, "q"))
    if a_pro is not None:
        if sa is not None or a2 is not None:
            raise ValueError("a_pro goes with a plain A operand")
        a.a_scale = _p(_vec(a_pro[0], Na, "a_pro.scale")); a.a_shift = _p(_vec(a_pro[1], Na, "a_pro.shift"))
        a.a_lrelu = 1; a.a_slope = float(a_pro[2])
    if sa is not None:
        a.a_scale = _p(_vec(sa.alpha, Na, "alpha")); a.a_shift = _p(_vec(sa.beta, Na, "beta"))
        a.a_sp_val = _p(sa.sp_val); a.a_sp_arg = _p(_i32(sa.sp_arg, "sp_arg")); a.a_sp_rows = sa.rows
    if edge is not None:
        idx, ebias = edge
        _i32(idx, "idx")
        if idx.shape[0] * idx.shape[1] != M_ or pro is None:
            raise ValueError("edge operand: A must have one row per edge and pro must be given")
        a.b_mode = A_EDGE; a.e_idx = _p(idx); a.e_k = idx.shape[1]; a.e_bias = _p(_vec(ebias, Nb, "ebias"))
    else:
        if Bm.shape[0] != M_:
            raise ValueError("A and B disagree on the reduction length: %d vs %d" % (M_, Bm.shape[0]))
        a.b_mode = A_PLAIN if pro is None else A_AFFINE_LRELU
    if pro is not None:
        sc, sh, ps = pro
        a.p_scale = _p(_vec(sc, Nb, "pro.scale")); a.p_shift = _p(_vec(sh, Nb, "pro.shift")); a.p_slope = float(ps)
    if out is None:
        out = torch.empty((Na, Nb), dtype=torch.float32, device=A.device)
        beta = 0.0
    else:
        _rowmajor2d(out, "out")
    lib = _lib.load()
    a.mfma_lp = 1 if (_MFMA_F16[0] == 1 and not exact and M_ >= TN_LP_MIN_ROWS) else 0
    if _MFMA_F16[0] == 2 and TN_SPLIT_BF16[0] and not exact and M_ >= TN_LP_MIN_ROWS:
        a.mfma_lp = 2                   # split-bf16 weight gradient (see TN_SPLIT_BF16)
    if (b_half or a16) and a.mfma_lp != 1:
        raise ValueError("16-bit stored operands need the bf16 weight-gradient kernel ('f16' operand mode, M >= %d, exact=False)" % TN_LP_MIN_ROWS)
    wsb = lib.spgan_gemm_tn_ws_bytes_lp(M_, Na, Nb, a.mfma_lp)       # (the split-bf16 kernel has its own split plan)
    ws = torch.empty((wsb // 4,), dtype=torch.float32, device=A.device)
    a.A = _p(A); a.lda = _ld(A); a.B = _p(Bm); a.ldb = _ld(Bm); a.C = _p(out); a.ldc = _ld(out)
    a.M, a.Na, a.Nb = M_, Na, Nb
    a.beta = float(beta); a.ws = _p(ws); a.ws_bytes = wsb
    defer = bool(defer) and sa is None
    a.defer_reduce = 1 if defer else 0
    a.a_half = 1 if a16 else 0
    splits = lib.spgan_gemm_tn_splits_lp(M_, Na, Nb, a.mfma_lp)
    cs_out = cs_ws = None
    streaming = (Na <= 4 or Nb <= 4) and Na <= 2048 and Nb <= 2048 and pro is None and edge is None and sa is None and a_pro is None and (a2 is None or Nb <= 4)   # 3-column layers: the streaming kernel + a colsum pass stay cheaper (mirrors launch_tn: a narrow-A two-tensor operand runs on the MFMA kernel)
    if with_colsum and sa is not None:
        raise NotImplementedError("gemm_tn(with_colsum=True) with a SparseAffine operand")
    if with_colsum and not streaming:
        cs_ws = torch.empty((splits, Na), dtype=torch.float32, device=A.device)
        cs_out = torch.empty((Na,), dtype=torch.float32, device=A.device)
        a.a_colsum_ws = _p(cs_ws)
    done = launch_timer("gemm_tn", a) if launch_timer is not None else None
    check(lib.spgan_gemm_tn(C.byref(a), _s()), "gemm_tn", M=M_, Na=Na, Nb=Nb)
    if done is not None:
        done()
    if defer:
        _PENDING_TN.append((ws, out, splits, Na, Nb, _ld(out), float(beta)))
    if not with_colsum:
        return out
    if cs_ws is None:                       # the streaming kernels have no such by-product: the separate reduction (of the TRANSFORMED operand)
        return out, colsum(a2.dense() if a2 is not None else A)[0]
    _PENDING_TN.append((cs_ws, cs_out, splits, 1, Na, Na, 0.0))           # [splits][1 x Na] partials: one more entry of the multi-reduce
    if not defer:
        flush_tn()
    return out, cs_out


def gemm_tn_narrow_multi(specs):
    """Several deferred gemm_tn(A, Bm, out=, beta=, defer=True) products with a narrow Bm (<= 4 columns: D's first conv against the three input
    coordinates) of one shape as ONE launch (spgan_gemm_tn_skinny_multi); A: Affine2 or a dense tensor.  specs: dicts(A, Bm[, out, beta]);
    -> the list of result tensors (valid after flush_tn(), like gemm_tn(defer=True)), bit-identical to the separate calls."""
    if len(specs) == 1 or not GROUPED[0] or _MFMA_F16[0] == 1:
        return [gemm_tn(sp["A"], sp["Bm"], out=sp.get("out"), beta=sp.get("beta", 0.0), defer=True) for sp in specs]
    lib = _lib.load()
    n = len(specs)
    arr = (GemmTNArgs * n)()
    res, keep, pend = [], [], []
    for i, sp in enumerate(specs):
        A, Bm, out, beta = sp["A"], sp["Bm"], sp.get("out"), float(sp.get("beta", 0.0))
        a2 = A if isinstance(A, Affine2) else None
        if a2 is not None:
            if a2.half:
                raise ValueError("gemm_tn_narrow_multi: fp32 operands")
            A = a2.g
        _rowmajor2d(A, "A"); _rowmajor2d(Bm, "B")
        M_, Na = A.shape
        Nb = Bm.shape[1]
        if Bm.shape[0] != M_ or Nb > 4:
            raise ValueError("gemm_tn_narrow_multi: Bm [M, <= 4]")
        a = arr[i]
        if a2 is not None:
            a.a_scale = _p(_vec(a2.p, Na, "p")); a.a_shift = _p(_vec(a2.r, Na, "r"))
            a.A2 = _p(a2.y); a.lda2 = _ld(a2.y); a.a_scale2 = _p(_vec(a2.q, Na, "q"))
        if out is None:
            out, beta = torch.empty((Na, Nb), dtype=torch.float32, device=A.device), 0.0
        else:
            _rowmajor2d(out, "out")
        wsb = lib.spgan_gemm_tn_ws_bytes(M_, Na, Nb)
        ws = torch.empty((wsb // 4,), dtype=torch.float32, device=A.device)
        a.A = _p(A); a.lda = _ld(A); a.B = _p(Bm); a.ldb = _ld(Bm); a.C = _p(out); a.ldc = _ld(out)
        a.M, a.Na, a.Nb = M_, Na, Nb
        a.beta = beta; a.ws = _p(ws); a.ws_bytes = wsb; a.defer_reduce = 1; a.b_mode = A_PLAIN
        pend.append((ws, out, lib.spgan_gemm_tn_splits(M_, Na, Nb), Na, Nb, _ld(out), beta))
        res.append(out)
    check(lib.spgan_gemm_tn_skinny_multi(arr, n, _s()), "gemm_tn_skinny_multi", count=n)
    _PENDING_TN.extend(pend)
    return res


def multi_addn(dsts, srcs_per_dst) -> None:
    """dsts[t] = ((dsts[t] + s0) + s1) + s2 for the one to three contiguous sources srcs_per_dst[t] of every destination, in ONE launch: the per-pass
    parameter gradients of a grouped backward into the flat .grad buffers -- the sums of as many successive multi_add calls."""
    from ._lib import MULTI_ADDN_MAX, MultiAddNArgs
    lib = _lib.load()
    items = [(d, [x for x in ss]) for d, ss in zip(dsts, srcs_per_dst) if len(ss)]
    for i0 in range(0, len(items), MULTI_ADDN_MAX):
        chunk = items[i0:i0 + MULTI_ADDN_MAX]
        a = MultiAddNArgs()
        a.count = len(chunk)
        keep = []
        for t, (d, ss) in enumerate(chunk):
            if not (d.is_cuda and d.is_contiguous() and d.dtype == torch.float32) or not 1 <= len(ss) <= 3:
                raise ValueError("multi_addn: contiguous float32 GPU destinations, one to three sources each")
            a.dst[t] = d.data_ptr(); a.n[t] = d.numel(); a.nsrc[t] = len(ss)
            for j, x in enumerate(ss):
                x = _f32(x, "src")
                if x.numel() != d.numel():
                    raise ValueError("multi_addn: source and destination sizes differ")
                x = x.contiguous()
                keep.append(x)
                a.src[j][t] = x.data_ptr()
        check(lib.spgan_multi_addn(C.byref(a), _s()), "multi_addn", count=len(chunk))


# ----------------------------------------------------------------------------- reductions / norms
def wt_diag_w(W: Tensor, alpha: Tensor, beta: Optional[Tensor] = None, bias: Optional[Tensor] = None):
    """G = W^T diag(alpha) W [K,K] and (with beta, bias) cvec = (alpha*bias + beta) . W [K] of W [C,K] in one launch: the weight-only operands
    of the collapsed backward of the layer in front of the max-pool.  Returns G, or (G, cvec)."""
    _rowmajor2d(W, "W")
    Cn, K = W.shape
    if Cn % 256 or K % 32:
        raise ValueError("wt_diag_w: W [C,K] with C % 256 == 0 and K % 32 == 0")
    G = torch.empty((K, K), dtype=torch.float32, device=W.device)
    cvec = torch.empty((K,), dtype=torch.float32, device=W.device) if beta is not None else None
    check(_lib.load().spgan_wt_diag_w(_p(W), _ld(W), Cn, K, _p(_vec(alpha, Cn, "alpha")), _p(None if beta is None else _vec(beta, Cn, "beta")),
                                      _p(None if beta is None else _vec(bias, Cn, "bias")), _p(G), K, _p(cvec), _s()), "wt_diag_w", C=Cn, K=K)
    return G if cvec is None else (G, cvec)


def collapse_prep(W: Tensor, problems, val, arg, rows: int):
    """What the collapsed backward of the layer in front of the max-pool needs before its big launch, in ONE launch:
    problems = one to four (alpha, beta | None, bias | None) -> wt_diag_w(W, alpha, beta, bias) each, and E = sparse_rows_nt(val, arg, rows, W).
    -> ([G or (G, cvec), ...], E); bit-identical to the separate launches (the weight-only part, latency-bound on a fraction of the chip,
    finishes under the part that streams E out).  val / arg may be LISTS of equally shaped sets (the passes of a grouped D step share W):
    then E is the list of their products."""
    from ._lib import CollapsePrepArgs, GROUP_MAX
    many = isinstance(val, (list, tuple))
    vals, args_ = (list(val), list(arg)) if many else ([val], [arg])
    _rowmajor2d(W, "W")
    for v in vals:
        _f32(v, "val", 2)
    Cn, K = W.shape
    B, Cs = vals[0].shape
    if (Cn % 256 or K % 32 or Cs != Cn or not 1 <= len(problems) <= GROUP_MAX or not 1 <= len(vals) <= GROUP_MAX or len(vals) != len(args_)
            or any(tuple(v.shape) != (B, Cs) for v in vals)):
        raise ValueError("collapse_prep: W [C,K] with C % 256 == 0, K % 32 == 0, val [B,C] (up to four equally shaped sets), one to four problems")
    if not GROUPED[0] and (len(problems) > 2 or many):       # A/B hook: the launches of the ungrouped route
        outs, Es = [], []
        for q in range(max(len(vals), (len(problems) + 1) // 2)):
            pr = problems[2 * q:2 * q + 2]
            if q < len(vals) and pr:
                o, E = collapse_prep(W, pr, vals[q], args_[q], rows)
                outs += o; Es.append(E)
            elif pr:
                outs += [wt_diag_w(W, *p3) for p3 in pr]
            else:
                Es.append(sparse_rows_nt(vals[q], args_[q], rows, W))
        return outs, (Es if many else Es[0])
    a = CollapsePrepArgs()
    a.W = _p(W); a.ldw = _ld(W); a.C = Cn; a.K = K; a.nprob = len(problems); a.ldg = K
    outs, keep = [], []
    for i, (alpha, beta, bias) in enumerate(problems):
        G = torch.empty((K, K), dtype=torch.float32, device=W.device)
        cvec = torch.empty((K,), dtype=torch.float32, device=W.device) if beta is not None else None
        va = _vec(alpha, Cn, "alpha"); vb = None if beta is None else _vec(beta, Cn, "beta"); vc = None if beta is None else _vec(bias, Cn, "bias")
        keep += [va, vb, vc]
        a.alpha[i] = _p(va); a.beta[i] = _p(vb); a.bias[i] = _p(vc); a.G[i] = _p(G); a.cvec[i] = _p(cvec)
        outs.append(G if cvec is None else (G, cvec))
    Es = []
    a.nsparse = len(vals); a.B = B; a.rows = rows; a.lde = K
    for q, (v, ar) in enumerate(zip(vals, args_)):
        vc_, ac_ = v.contiguous(), _i32(ar, "arg")
        keep += [vc_, ac_]
        E = torch.empty((B * rows, K), dtype=torch.float32, device=W.device)
        a.sp_val[q] = _p(vc_); a.sp_arg[q] = _p(ac_); a.E[q] = _p(E)
        Es.append(E)
    check(_lib.load().spgan_collapse_prep(C.byref(a), _s()), "collapse_prep", C=Cn, K=K, B=B, rows=rows)
    return outs, (Es if many else Es[0])


WGRAD_COLLAPSE = [True]   # test / A-B hook: False keeps the collapsed layer's weight gradient on its separate launches


def wgrad_collapse_ok(W: Tensor, X1: Tensor, B: int = 0) -> bool:
    """Does spgan_wgrad_collapse take this problem (W [C,K], X1 [N,K]; B shapes in the sparse term)?"""
    Cn, K = W.shape
    return (WGRAD_COLLAPSE[0] and W.is_cuda and Cn % 32 == 0 and X1.shape[0] % 32 == 0 and K % 32 == 0 and 32 <= K <= 256 and X1.shape[1] == K and B <= 64
            and W.stride(0) % 4 == 0 and X1.stride(0) % 4 == 0 and W.data_ptr() % 16 == 0 and X1.data_ptr() % 16 == 0)


def wgrad_collapse(W: Tensor, X1: Tensor, a1: Tensor, b1: Optional[Tensor] = None, d1: Optional[Tensor] = None, v1: Optional[Tensor] = None, *,
                   X2: Optional[Tensor] = None, x2_t: bool = False, a2: Optional[Tensor] = None, sparse=None, out: Optional[Tensor] = None,
                   accumulate: bool = False, want_T: bool = False):
    """The collapsed layer's weight gradient, all terms in one launch (spgan_wgrad_collapse):
      out[a,n] (+)= a1[a]*(W . X1^T)[a,n] + (a1*b1 + d1)[a]*v1[n] + a2[a]*(W . X2^T)[a,n] + sum_b val[b,a]*pro(Bm)[arg[b,a], n]
    W [C,K], X1 [N,K]; X2 [N,K] or, with x2_t, [K,N]; sparse = (val [B,C], arg int32 [B,C] global rows, rows, Bm [B*rows,N], pro | None) with
    pro = (scale[N], shift[N], slope).  -> out, or (out, T) with want_T (T = W . X1^T)."""
    a, res, _keep = _wgrad_collapse_build(W, X1, a1, b1, d1, v1, X2=X2, x2_t=x2_t, a2=a2, sparse=sparse, out=out, accumulate=accumulate, want_T=want_T)
    check(_lib.load().spgan_wgrad_collapse(C.byref(a), _s()), "wgrad_collapse", C=a.C, N=a.N, K=a.K)
    return res


def wgrad_collapse_multi(specs):
    """Several wgrad_collapse problems with equal shapes (the passes of a grouped D step) as one launch; specs: dicts of wgrad_collapse's
    arguments; -> the list of its results, bit-identical to separate calls."""
    if len(specs) == 1 or not GROUPED[0]:
        return [wgrad_collapse(**sp) for sp in specs]
    from ._lib import WgradCollapseArgs
    builds = [_wgrad_collapse_build(**sp) for sp in specs]
    n = len(builds)
    arr = (WgradCollapseArgs * n)(*[b[0] for b in builds])
    check(_lib.load().spgan_wgrad_collapse_multi(arr, n, _s()), "wgrad_collapse_multi", count=n)
    return [b[1] for b in builds]


def _wgrad_collapse_build(W: Tensor, X1: Tensor, a1: Tensor, b1: Optional[Tensor] = None, d1: Optional[Tensor] = None, v1: Optional[Tensor] = None, *,
                          X2: Optional[Tensor] = None, x2_t: bool = False, a2: Optional[Tensor] = None, sparse=None, out: Optional[Tensor] = None,
                          accumulate: bool = False, want_T: bool = False):
    from ._lib import WgradCollapseArgs
    _rowmajor2d(W, "W"); _rowmajor2d(X1, "X1")
    Cn, K = W.shape
    N = X1.shape[0]
    if out is None:
        if accumulate:
            raise ValueError("wgrad_collapse(accumulate=True) needs out")
        out = torch.empty((Cn, N), dtype=torch.float32, device=W.device)
    else:
        _rowmajor2d(out, "out")
        if tuple(out.shape) != (Cn, N):
            raise ValueError("wgrad_collapse: out must be [%d,%d]" % (Cn, N))
    a = WgradCollapseArgs()
    a.W = _p(W); a.ldw = _ld(W); a.C = Cn; a.K = K; a.X1 = _p(X1); a.ldx1 = _ld(X1); a.N = N
    keep = [_vec(a1, Cn, "a1")]
    a.a1 = _p(keep[0])
    if v1 is not None:
        keep += [_vec(b1, Cn, "b1"), _vec(d1, Cn, "d1"), _vec(v1, N, "v1")]
        a.b1, a.d1, a.v1 = _p(keep[1]), _p(keep[2]), _p(keep[3])
    if X2 is not None:
        _rowmajor2d(X2, "X2")
        if tuple(X2.shape) != ((K, N) if x2_t else (N, K)):
            raise ValueError("wgrad_collapse: X2 must be [N,K] (or [K,N] with x2_t)")
        keep.append(_vec(a2, Cn, "a2"))
        a.X2 = _p(X2); a.ldx2 = _ld(X2); a.x2_t = int(x2_t); a.a2 = _p(keep[-1])
    if sparse is not None:
        val, arg, rows, Bm, pro = sparse
        _f32(val, "val", 2); _rowmajor2d(Bm, "Bm")
        B = val.shape[0]
        if val.shape[1] != Cn or Bm.shape[0] != B * rows or Bm.shape[1] != N:
            raise ValueError("wgrad_collapse: sparse term shapes")
        keep += [val.contiguous(), _i32(arg, "arg")]
        a.sp_val = _p(keep[-2]); a.sp_arg = _p(keep[-1]); a.B = B; a.rows = rows; a.Bm = _p(Bm); a.ldb = _ld(Bm)
        if pro is not None:
            keep += [_vec(pro[0], N, "pro.scale"), _vec(pro[1], N, "pro.shift")]
            a.p_scale = _p(keep[-2]); a.p_shift = _p(keep[-1]); a.p_slope = float(pro[2])
    T = torch.empty((Cn, N), dtype=torch.float32, device=W.device) if want_T else None
    a.T = _p(T); a.ldt = N
    a.out = _p(out); a.ldo = _ld(out); a.accumulate = int(accumulate)
    return a, ((out, T) if want_T else out), keep


def sparse_rows_nt(val: Tensor, arg: Tensor, rows: int, W: Tensor) -> Tensor:
    """E[m,:] = sum_{c: arg[b,c]==m} val[b,c] * W[c,:]  (b = m // rows): the row-sparse product S @ W, written densely [B*rows, N]."""
    _f32(val, "val", 2); _rowmajor2d(W, "W")
    B, Cs = val.shape
    if W.shape[0] != Cs:
        raise ValueError("W must have one row per channel of val")
    N = W.shape[1]
    E = torch.empty((B * rows, N), dtype=torch.float32, device=val.device)
    check(_lib.load().spgan_sparse_rows_nt(_p(val.contiguous()), _p(_i32(arg, "arg")), B, rows, Cs, _p(W), _ld(W), N, _p(E), N, _s()),
          "sparse_rows_nt", B=B, rows=rows, Cs=Cs, N=N)
    return E


def sparse_rows_tn(val: Tensor, arg: Tensor, rows: int, Bm: Tensor, out: Tensor, pro=None) -> Tensor:
    """out[c,:] += sum_b val[b,c] * pro(Bm)[arg[b,c],:]  -- S^T @ pro(Bm) accumulated into out [Cs, Nb]."""
    _f32(val, "val", 2); _rowmajor2d(Bm, "Bm"); _rowmajor2d(out, "out")
    B, Cs = val.shape
    Nb = Bm.shape[1]
    if out.shape != (Cs, Nb) or Bm.shape[0] != B * rows:
        raise ValueError("shape mismatch in sparse_rows_tn")
    sc = sh = None; slope = 1.0
    if pro is not None:
        sc, sh, slope = _vec(pro[0], Nb, "pro.scale"), _vec(pro[1], Nb, "pro.shift"), float(pro[2])
    check(_lib.load().spgan_sparse_rows_tn(_p(val.contiguous()), _p(_i32(arg, "arg")), B, rows, Cs, _p(Bm), _ld(Bm), Nb, _p(sc), _p(sh), slope,
                                           _p(out), _ld(out), _s()), "sparse_rows_tn", B=B, rows=rows, Cs=Cs, Nb=Nb)
    return out


def affine_act(X: Tensor, scale: Tensor, shift: Tensor, slope: float) -> Tensor:
    """lrelu(X*scale[c] + shift[c], slope) materialised (train-mode BatchNorm + LeakyReLU output)."""
    _rowmajor2d(X, "X")
    M_, Cn = X.shape
    out = torch.empty((M_, Cn), dtype=torch.float32, device=X.device)
    check(_lib.load().spgan_affine_act(_p(X), _ld(X), M_, Cn, _p(_vec(scale, Cn, "scale")), _p(_vec(shift, Cn, "shift")), float(slope), _p(out), _s()),
          "affine_act", M=M_, C=Cn)
    return out


def rowscale_outer(X: Tensor, a: Tensor, b: Optional[Tensor] = None, d: Optional[Tensor] = None, v: Optional[Tensor] = None,
                   out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """out[r,c] (+)= a[r]*X[r,c] + (a[r]*b[r] + d[r])*v[c]   (v None: first term only).  out: destination [R,C] with unit column stride;
    accumulate=True adds into it (a term of a weight gradient summed in place)."""
    _rowmajor2d(X, "X")
    R, Cn = X.shape
    if out is None:
        if accumulate:
            raise ValueError("rowscale_outer(accumulate=True) needs out")
        out = torch.empty((R, Cn), dtype=torch.float32, device=X.device)
    else:
        _rowmajor2d(out, "out")
        if tuple(out.shape) != (R, Cn):
            raise ValueError("rowscale_outer: out must be [%d,%d]" % (R, Cn))
    vb = vd = vv = None
    if v is not None:
        vb, vd, vv = _vec(b, R, "b"), _vec(d, R, "d"), _vec(v, Cn, "v")
    check(_lib.load().spgan_rowscale_outer(_p(X), _ld(X), R, Cn, _p(_vec(a, R, "a")), _p(vb), _p(vd), _p(vv), _p(out), _ld(out), int(accumulate), _s()),
          "rowscale_outer", R=R, C=Cn)
    return out


def gather_rowdot(Q: Tensor, arg: Tensor, W: Tensor) -> Tensor:
    """out[b,c] = Q[arg[b,c], :] . W[c, :]"""
    _rowmajor2d(Q, "Q"); _rowmajor2d(W, "W"); _i32(arg, "arg")
    B, Cn = arg.shape
    K = W.shape[1]
    if W.shape[0] != Cn or Q.shape[1] != K:
        raise ValueError("shape mismatch in gather_rowdot")
    out = torch.empty((B, Cn), dtype=torch.float32, device=Q.device)
    check(_lib.load().spgan_gather_rowdot(_p(Q), _ld(Q), _p(arg), _p(W), _ld(W), B, Cn, K, _p(out), _s()), "gather_rowdot", B=B, C=Cn, K=K)
    return out


def dbl_top_dots(Q: Tensor, arg: Tensor, W: Tensor, T: Tensor, cq: Tensor):
    """-> (uarg [B,C] = Q[arg[b,c],:] . W[c,:],  quad [C] = W[c,:] . T[c,:],  U0 [C] = W[c,:] . cq): gather_rowdot, rowdot and a matrix-vector
    product -- three independent launches of the collapsed double backward's phase A -- as one."""
    _rowmajor2d(Q, "Q"); _rowmajor2d(W, "W"); _rowmajor2d(T, "T"); _i32(arg, "arg")
    B, Cn = arg.shape
    K = W.shape[1]
    if W.shape[0] != Cn or Q.shape[1] != K or tuple(T.shape) != (Cn, K):
        raise ValueError("shape mismatch in dbl_top_dots")
    uarg = torch.empty((B, Cn), dtype=torch.float32, device=Q.device)
    qu = torch.empty((2, Cn), dtype=torch.float32, device=Q.device)
    check(_lib.load().spgan_dbl_top_dots(_p(Q), _ld(Q), _p(arg), _p(W), _ld(W), _p(T), _ld(T), _p(_vec(cq, K, "cq")), B, Cn, K, _p(uarg), _p(qu[0]),
                                         _p(qu[1]), _s()), "dbl_top_dots", B=B, C=Cn, K=K)
    return uarg, qu[0], qu[1]


def rowdot(X: Tensor, Y: Tensor) -> Tensor:
    """out[r] = X[r,:] . Y[r,:]"""
    _rowmajor2d(X, "X"); _rowmajor2d(Y, "Y")
    if X.shape != Y.shape:
        raise ValueError("shape mismatch in rowdot")
    R, K = X.shape
    out = torch.empty((R,), dtype=torch.float32, device=X.device)
    check(_lib.load().spgan_rowdot(_p(X), _ld(X), _p(Y), _ld(Y), R, K, _p(out), _s()), "rowdot", R=R, K=K)
    return out


def bn_dbl_pool(uarg: Tensor, gval: Tensor, yarg: Tensor, pooled: Tensor, U0, quad, bias, mean, invstd, gamma, S0, S1, count: int, slope: float):
    """Per-channel algebra of the collapsed double backward in front of the max-pool -> (t [B,C], spB [B,C], [dgamma | c1 | c2 | c3] [4,C])."""
    B, Cn = uarg.shape
    t = torch.empty((B, Cn), dtype=torch.float32, device=uarg.device)
    spB = torch.empty_like(t)
    out4 = torch.empty((4, Cn), dtype=torch.float32, device=uarg.device)
    m = lambda x, n: _p(_f32(x.contiguous(), n, 2))
    v = lambda x, n: _p(_vec(x.contiguous(), Cn, n))
    check(_lib.load().spgan_bn_dbl_pool(m(uarg, "uarg"), m(gval, "gval"), m(yarg, "yarg"), m(pooled, "pooled"), v(U0, "U0"), v(quad, "quad"),
                                        v(bias, "bias"), v(mean, "mean"), v(invstd, "invstd"), v(gamma, "gamma"), v(S0, "S0"), v(S1, "S1"),
                                        B, Cn, count, float(slope), _p(t), _p(spB), _p(out4), _s()), "bn_dbl_pool", B=B, C=Cn)
    return t, spB, out4


def colstats(X: Tensor, G: int, slope: float = 1.0) -> Tuple[Tensor, Tensor]:
    """mean / biased var of lrelu(X, slope) over each group of G rows -> ([M/G, C], [M/G, C])."""
    _rowmajor2d(X, "X")
    M_, Cn = X.shape
    lib = _lib.load()
    wsb = lib.spgan_colreduce_ws_bytes(M_, Cn, G)
    ws = torch.empty((wsb // 4,), dtype=torch.float32, device=X.device)
    out = torch.empty((2, M_ // G, Cn), dtype=torch.float32, device=X.device)
    check(lib.spgan_colstats(_p(X), _ld(X), M_, Cn, G, float(slope), _p(out[0]), _p(out[1]), _p(ws), wsb, _s()), "colstats", M=M_, C=Cn, G=G)
    return out[0], out[1]


def colsum(X: Tensor, G: Optional[int] = None) -> Tensor:
    """Column sums over each group of G rows (default: all rows) -> [M/G, C]."""
    _rowmajor2d(X, "X")
    M_, Cn = X.shape
    G = M_ if G is None else G
    lib = _lib.load()
    wsb = lib.spgan_colreduce_ws_bytes(M_, Cn, G) + (M_ // G) * Cn * 4
    ws = torch.empty((wsb // 4,), dtype=torch.float32, device=X.device)
    out = torch.empty((M_ // G, Cn), dtype=torch.float32, device=X.device)
    check(lib.spgan_colsum(_p(X), _ld(X), M_, Cn, G, _p(out), _p(ws), wsb, _s()), "colsum", M=M_, C=Cn, G=G)
    return out


def bn_prepare(mean: Optional[Tensor], var: Optional[Tensor], gamma: Optional[Tensor], beta: Optional[Tensor], count: int,
               training: bool = True, running_mean: Optional[Tensor] = None, running_var: Optional[Tensor] = None,
               momentum: Optional[float] = None, eps: float = BN_EPS):
    """-> (scale, shift, invstd, mean_used) each [C]; updates the running statistics in place (train mode)."""
    if momentum is None:
        momentum = _BN_MOM[0]
    ref = mean if mean is not None else running_mean
    Cn = ref.numel()
    out = torch.empty((4, Cn), dtype=torch.float32, device=ref.device)
    check(_lib.load().spgan_bn_prepare(_p(mean), _p(var), _p(gamma), _p(beta), Cn, count, eps, momentum, 1 if training else 0,
                                       _p(running_mean), _p(running_var), _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]), _s()),
          "bn_prepare", C=Cn)
    return out[0], out[1], out[2], out[3]


def bn_bwd_apply(g: Tensor, y: Tensor, mean: Tensor, invstd: Tensor, gamma: Optional[Tensor], sums: Tensor, count: int,
                 add: Optional[Tuple[Tensor, Tensor]] = None) -> Tensor:
    """dy = gamma*invstd*(g - sum_g/count - xhat*sum_gx/count);  sums = [sum_g | sum_gx] (2C).
    add = (g2 [M,C], scale [C]): g is replaced by g + scale[c]*g2, formed on the fly."""
    _f32(g, "g", 2); _f32(y, "y", 2)
    if not (g.is_contiguous() and y.is_contiguous()) or g.shape != y.shape:
        raise ValueError("g and y must be contiguous with equal shapes")
    M_, Cn = g.shape
    dy = torch.empty_like(g)
    if add is not None:
        g2, sc = add
        if Cn % 4 == 0 and g2.is_contiguous() and g2.shape == g.shape:
            check(_lib.load().spgan_bn_bwd_apply2(_p(g), _p(_f32(g2, "g2", 2)), _p(_vec(sc, Cn, "scale")), _p(y), M_, Cn, _p(mean), _p(invstd),
                                                  _p(gamma), _p(_vec(sums, 2 * Cn, "sums")), count, _p(dy), _s()), "bn_bwd_apply2", M=M_, C=Cn)
            return dy
        g = col_scale_add(g, g2, sc)
    check(_lib.load().spgan_bn_bwd_apply(_p(g), _p(y), Cn, M_, Cn, _p(mean), _p(invstd), _p(gamma), _p(_vec(sums, 2 * Cn, "sums")),
                                         count, _p(dy), _s()), "bn_bwd_apply", M=M_, C=Cn)
    return dy


def maxpool(y: Tensor, B: int, N: int, scale: Optional[Tensor] = None, shift: Optional[Tensor] = None, slope: float = 1.0):
    """out[b,c] = max_n lrelu(y[b*N+n,c]*scale[c]+shift[c], slope); argmax = global row index (int32)."""
    _rowmajor2d(y, "y")
    Cn = y.shape[1]
    out = torch.empty((B, Cn), dtype=torch.float32, device=y.device)
    arg = torch.empty((B, Cn), dtype=torch.int32, device=y.device)
    check(_lib.load().spgan_maxpool(_p(y), _ld(y), B, N, Cn, _p(scale), _p(shift), float(slope), _p(out), _p(arg), _s()), "maxpool", B=B, N=N, C=Cn)
    return out, arg


def gemm_bn_pool(A: Tensor, W: Tensor, bias: Optional[Tensor], bn, rows: int, slope: float, pro=None, keep_y: bool = False, before_finalize=None):
    """Y = pro(A) @ W^T + bias, train-mode BatchNorm over all rows, LeakyReLU, max over each group of `rows` rows -- the
    tail of the Discriminator's conv stack (Discriminator.py:74-81,104) in one GEMM launch + two small finalize launches; Y
    itself is only written when keep_y.  bn = (gamma, beta, running_mean | None, running_var | None).
    before_finalize: a callable issued between the GEMM launch and the finalize launches (which update the running statistics) -- for a
    caller that has to advance the same running statistics with ANOTHER pass first but wants this GEMM to follow its producer directly.
    Returns (Y | None, (scale, shift, invstd, mean), pooled [B,C], argmax int32 [B,C] (global rows), yarg [B,C])."""
    _rowmajor2d(A, "A"); _rowmajor2d(W, "W")
    N, K = W.shape
    M_ = A.shape[0]
    if rows % ROW_TILE or M_ % rows:
        raise ValueError("gemm_bn_pool needs rows %% %d == 0 and M %% rows == 0" % ROW_TILE)
    B = M_ // rows
    tiles = M_ // ROW_TILE
    dev = A.device
    a = GemmNTArgs(); a.mfma_f16 = _MFMA_F16[0]; a.tile_hint = _NT_TILE_HINT[0]
    a.A = _p(A); a.lda = _ld(A); a.W = _p(W); a.ldw = _ld(W)
    Y = torch.empty((M_, N), dtype=torch.float32, device=dev) if keep_y else None
    a.Y = _p(Y); a.ldy = N
    a.M, a.N, a.K = M_, N, K
    a.a_mode = A_PLAIN
    if pro is not None:
        a.a_mode = A_AFFINE_LRELU
        a.p_scale = _p(_vec(pro[0], K, "pro.scale")); a.p_shift = _p(_vec(pro[1], K, "pro.shift")); a.p_slope = float(pro[2])
    a.epi_mode = EPI_LINEAR
    a.bias = _p(_vec(bias, N, "bias"))
    part = torch.empty((tiles, N, 2), dtype=torch.float32, device=dev)
    pval = torch.empty((tiles, N, 2), dtype=torch.float32, device=dev)
    parg = torch.empty((tiles, N, 2), dtype=torch.int32, device=dev)
    a.stats = _p(part); a.pool_val = _p(pval); a.pool_arg = _p(parg)
    lib = _lib.load()
    gamma, beta, rm, rv = bn
    st = torch.empty((4, N), dtype=torch.float32, device=dev)
    _wimg = _attach_w_image(a, W)      # noqa: F841 (kept alive across the launch)
    done = launch_timer("gemm_nt", a) if launch_timer is not None else None
    check(lib.spgan_gemm_nt(C.byref(a), _s()), "gemm_bn_pool", M=M_, N=N, K=K)
    if done is not None:
        done()
    if before_finalize is not None:
        before_finalize()
    check(lib.spgan_colstats_finalize_bn(_p(part), tiles, N, M_, 0, _p(gamma), _p(beta), BN_EPS, _BN_MOM[0], _p(rm), _p(rv),
                                         _p(st[0]), _p(st[1]), _p(st[2]), _p(st[3]), _s()), "colstats_finalize_bn", N=N, M=M_)
    pooled = torch.empty((B, N), dtype=torch.float32, device=dev)
    yarg = torch.empty((B, N), dtype=torch.float32, device=dev)
    arg = torch.empty((B, N), dtype=torch.int32, device=dev)
    check(lib.spgan_pool_finalize(_p(pval), _p(parg), B, rows, N, _p(st[0]), _p(st[1]), float(slope), _p(pooled), _p(arg), _p(yarg), _s()),
          "pool_finalize", B=B, rows=rows, C=N)
    return Y, (st[0], st[1], st[2], st[3]), pooled, arg, yarg


def gemm_bn_groups(A: Tensor, W: Tensor, bias: Optional[Tensor], bn, groups: int, pro=None, rows: int = 0, slope: float = 0.0):
    """Several passes through the same layer as ONE product: A [groups*Mg, K] holds the rows of `groups` passes one after the other,
    every pass has its own train-mode BatchNorm statistics -- of its input (pro = (scale [groups,K], shift [groups,K], slope):
    operand a = lrelu(A*scale[g]+shift[g]) for the rows of pass g) and of this layer's output (bn = (gamma, beta, running_mean | None,
    running_var | None): out [4, groups, N] = scale | shift | invstd | mean per pass, the running statistics updated pass after pass
    exactly as `groups` separate calls would).  One GEMM launch + one finalize launch.
    rows = 0: returns (Y [groups*Mg, N], out).
    rows > 0 (the Discriminator's last conv layer, gemm_bn_pool for several passes): Y is not stored; returns
    (out, pooled [B,N], argmax int32 [B,N] -- rows counted from the first row of the shape's OWN pass --, yarg [B,N]) with B = groups*Mg/rows."""
    _rowmajor2d(A, "A"); _rowmajor2d(W, "W")
    N, K = W.shape
    M_ = A.shape[0]
    if groups <= 0 or M_ % groups or (M_ // groups) % ROW_TILE:
        raise ValueError("gemm_bn_groups needs groups | M and whole %d-row tiles per group" % ROW_TILE)
    Mg = M_ // groups
    dev = A.device
    a = GemmNTArgs(); a.mfma_f16 = _MFMA_F16[0]; a.tile_hint = _NT_TILE_HINT[0]
    a.A = _p(A); a.lda = _ld(A); a.W = _p(W); a.ldw = _ld(W)
    a.M, a.N, a.K = M_, N, K
    a.a_mode = A_PLAIN
    if pro is not None:
        sc, sh, ps = pro
        for t, nm in ((sc, "pro.scale"), (sh, "pro.shift")):
            _f32(t, nm, 2)
            if tuple(t.shape) != (groups, K) or not t.is_contiguous():
                raise ValueError("%s must be contiguous [groups, K]" % nm)
        a.a_mode = A_AFFINE_LRELU
        a.p_scale = _p(sc); a.p_shift = _p(sh); a.p_slope = float(ps)
    a.p_group_rows = Mg if groups > 1 else 0      # also without a prologue: the tile-size rule looks at the rows of ONE pass
    a.epi_mode = EPI_LINEAR
    a.bias = _p(_vec(bias, N, "bias"))
    tiles = M_ // ROW_TILE
    part = torch.empty((tiles, N, 2), dtype=torch.float32, device=dev)
    a.stats = _p(part)
    Y = pval = parg = None
    if rows > 0:
        if rows % ROW_TILE or Mg % rows:
            raise ValueError("rows must be a multiple of %d and divide the rows of a group" % ROW_TILE)
        pval = torch.empty((tiles, N, 2), dtype=torch.float32, device=dev)
        parg = torch.empty((tiles, N, 2), dtype=torch.int32, device=dev)
        a.pool_val = _p(pval); a.pool_arg = _p(parg)
        a.Y = None; a.ldy = N
    else:
        Y = torch.empty((M_, N), dtype=torch.float32, device=dev)
        a.Y = _p(Y); a.ldy = N
    lib = _lib.load()
    gamma, beta, rm, rv = bn
    out = torch.empty((4, groups, N), dtype=torch.float32, device=dev)
    _wimg = _attach_w_image(a, W)      # noqa: F841 (kept alive across the launch)
    done = launch_timer("gemm_nt", a) if launch_timer is not None else None
    check(lib.spgan_gemm_nt(C.byref(a), _s()), "gemm_bn_groups", M=M_, N=N, K=K, groups=groups)
    if done is not None:
        done()
    check(lib.spgan_colstats_finalize_bn_groups(_p(part), groups, tiles // groups, N, Mg, 0, _p(gamma), _p(beta), BN_EPS, _BN_MOM[0], _p(rm), _p(rv),
                                                _p(out), _s()), "colstats_finalize_bn_groups", N=N, groups=groups)
    if rows == 0:
        return Y, out
    B = M_ // rows
    pooled = torch.empty((B, N), dtype=torch.float32, device=dev)
    yarg = torch.empty((B, N), dtype=torch.float32, device=dev)
    arg = torch.empty((B, N), dtype=torch.int32, device=dev)
    check(lib.spgan_pool_finalize_groups(_p(pval), _p(parg), B, rows, N, _p(out[0]), _p(out[1]), N, Mg // rows, float(slope),
                                         _p(pooled), _p(arg), _p(yarg), 1, _s()), "pool_finalize_groups", B=B, rows=rows, C=N)
    return out, pooled, arg, yarg


# ----------------------------------------------------------------------------- EdgeBlock gather-side ops
def edge_wcat(Ww0: Tensor, Wx: Tensor, transposed: bool = False):
    """[W1; Wd; Wc-Wd] from conv_w.0.weight [H,C] and conv_x.0.weight [F,2C] = [Wc|Wd] -> [H+2F, C].
    transposed=True: -> (Wcat, Wcat^T [C, H+2F]) from the same launch (the operand of the block's input-gradient GEMM)."""
    _f32(Ww0, "Ww0", 2); _f32(Wx, "Wx", 2)
    H, Cc = Ww0.shape
    F_ = Wx.shape[0]
    if not (Ww0.is_contiguous() and Wx.is_contiguous()) or Wx.shape[1] != 2 * Cc:
        raise ValueError("expected contiguous Ww0 [H,C] and Wx [F,2C]")
    out = torch.empty((H + 2 * F_, Cc), dtype=torch.float32, device=Ww0.device)
    out_t = torch.empty((Cc, H + 2 * F_), dtype=torch.float32, device=Ww0.device) if transposed else None
    check(_lib.load().spgan_edge_wcat(_p(Ww0), _p(Wx), H, F_, Cc, _p(out), _p(out_t), _s()), "edge_wcat")
    return (out, out_t) if transposed else out


def conv_out_weight_pm(w: Tensor) -> Tuple[Tensor, Tensor]:
    """conv_out.weight [F,F,1,k] -> (Wo [F, k*F] with K index r*F + c, Wo^T [k*F, F]) in one launch."""
    _f32(w, "w", 4)
    F_, F2, one, k = w.shape
    if F2 != F_ or one != 1 or not w.is_contiguous():
        raise ValueError("conv_out.weight must be contiguous [F,F,1,k]")
    wo = torch.empty((F_, k * F_), dtype=torch.float32, device=w.device)
    wot = torch.empty((k * F_, F_), dtype=torch.float32, device=w.device)
    check(_lib.load().spgan_conv_out_weight_pm(_p(w), F_, k, _p(wo), _p(wot), _s()), "conv_out_weight_pm", F=F_, k=k)
    return wo, wot


def edge_wcat_bwd(dWcat: Tensor, H: int, F_: int) -> Tuple[Tensor, Tensor]:
    _f32(dWcat, "dWcat", 2)
    Cc = dWcat.shape[1]
    if not dWcat.is_contiguous() or dWcat.shape[0] != H + 2 * F_:
        raise ValueError("dWcat must be contiguous [H+2F, C]")
    dW0 = torch.empty((H, Cc), dtype=torch.float32, device=dWcat.device)
    dWx = torch.empty((F_, 2 * Cc), dtype=torch.float32, device=dWcat.device)
    check(_lib.load().spgan_edge_wcat_bwd(_p(dWcat), H, F_, Cc, _p(dW0), _p(dWx), _s()), "edge_wcat_bwd")
    return dW0, dWx


def _pqr(PQR: Tensor, H: int, F_: int) -> Tensor:
    _f32(PQR, "PQR", 2)
    if not PQR.is_contiguous() or PQR.shape[1] != H + 2 * F_:
        raise ValueError("PQR must be contiguous [M, H+2F]")
    return PQR


def edge_stats(PQR: Tensor, idx: Tensor, b1: Tensor, bx: Tensor) -> Tuple[Tensor, Tensor]:
    """mean/biased var over all M*k edges of [ (P_j-P_i)+b1 | (R_i+Q_j)+bx ] -> ([H+F], [H+F])."""
    H, F_ = b1.numel(), bx.numel()
    _pqr(PQR, H, F_); _i32(idx, "idx")
    M_, k = idx.shape
    lib = _lib.load()
    tr = lib.spgan_edge_stats_tile_rows(k)
    tiles = (M_ * k + tr - 1) // tr
    part = torch.empty((tiles, H + F_, 2), dtype=torch.float32, device=PQR.device)
    check(lib.spgan_edge_stats(_p(PQR), PQR.shape[1], _p(idx), M_, k, H, F_, _p(_vec(b1, H, "b1")), _p(_vec(bx, F_, "bx")), _p(part), _s()),
          "edge_stats", M=M_, k=k, H=H, F=F_)
    mean, var = _finalize(part, 1, tiles, H + F_, M_ * k, 0, tr)
    return mean[0], var[0]


def edge_stats_bn(PQR: Tensor, idx: Tensor, b1: Tensor, bx: Tensor, bn_w, bn_x, count_rep: int = 1):
    """edge_stats followed by the train-mode bookkeeping of BOTH per-edge BatchNorm layers (conv_w.1 over the H channels, conv_x.1
    over the F channels; bn_* = (gamma, beta, running_mean | None, running_var | None)) in ONE finalize launch
    (spgan_colstats_finalize_bn2) instead of a finalize and two bn_prepare launches.
    -> ((scale, shift, invstd, mean) of conv_w.1 [H], the same of conv_x.1 [F])."""
    H, F_ = b1.numel(), bx.numel()
    _pqr(PQR, H, F_); _i32(idx, "idx")
    M_, k = idx.shape
    lib = _lib.load()
    tr = lib.spgan_edge_stats_tile_rows(k)
    tiles = (M_ * k + tr - 1) // tr
    Cn = H + F_
    part = torch.empty((tiles, Cn, 2), dtype=torch.float32, device=PQR.device)
    check(lib.spgan_edge_stats(_p(PQR), PQR.shape[1], _p(idx), M_, k, H, F_, _p(_vec(b1, H, "b1")), _p(_vec(bx, F_, "bx")), _p(part), _s()),
          "edge_stats", M=M_, k=k, H=H, F=F_)
    out = torch.empty((4, Cn), dtype=torch.float32, device=PQR.device)
    gw, bw, rmw, rvw = bn_w
    gx, bx_, rmx, rvx = bn_x
    check(lib.spgan_colstats_finalize_bn2(_p(part), tiles, Cn, M_ * k, tr, H, _p(_vec(gw, H, "gamma_w")), _p(_vec(bw, H, "beta_w")), _p(rmw), _p(rvw),
                                          _p(_vec(gx, F_, "gamma_x")), _p(_vec(bx_, F_, "beta_x")), _p(rmx), _p(rvx), BN_EPS, _BN_MOM[0], int(count_rep),
                                          _p(out), _s()), "colstats_finalize_bn2", C=Cn, G=M_ * k)
    return tuple(out[i, :H] for i in range(4)), tuple(out[i, H:] for i in range(4))


def edge_attend_fwd(h2pre: Tensor, sc2: Tensor, sh2: Tensor, PQR: Tensor, idx: Tensor, bx: Tensor, scx: Tensor, shx: Tensor,
                    slope: float, half: bool = False) -> Tensor:
    """T[M, k*F]: softmax over the k neighbours of lrelu(bn(h2pre)) times lrelu(bn((R_i+Q_j)+bx)).
    half=True: T is stored as float16 (k = 10, F % 4 == 0; consumed by gemm_nt / gemm_tn in the "f16" operand mode: storage16())."""
    F_ = bx.numel()
    H = PQR.shape[1] - 2 * F_
    _pqr(PQR, H, F_); _i32(idx, "idx")
    M_, k = idx.shape
    h2_half = h2pre.dtype == torch.float16      # written by gemm_nt(out_half=True): 16-bit storage, goes with half=True
    if h2_half:
        if not half:
            raise ValueError("a float16 h2pre belongs to the 16-bit storage mode (half=True)")
        _rowmajor2d_as(h2pre, "h2pre", torch.float16)
    else:
        _f32(h2pre, "h2pre", 2)
    if not h2pre.is_contiguous() or h2pre.shape != (M_ * k, F_):
        raise ValueError("h2pre must be contiguous [M*k, F]")
    T = torch.empty((M_, k * F_), dtype=torch.float16 if half else torch.float32, device=PQR.device)
    fn = _lib.load().spgan_edge_attend_fwd_h if half else _lib.load().spgan_edge_attend_fwd
    check(fn(*((_p(h2pre), 1 if h2_half else 0) if half else (_p(h2pre),)), _p(_vec(sc2, F_, "sc2")), _p(_vec(sh2, F_, "sh2")), _p(PQR), PQR.shape[1], H, F_, _p(idx), M_, k,
             _p(_vec(bx, F_, "bx")), _p(_vec(scx, F_, "scx")), _p(_vec(shx, F_, "shx")), float(slope), _p(T), _s()),
          "edge_attend_fwd", M=M_, k=k, F=F_, half=half)
    return T


def edge_attend_bwd(dT: Tensor, h2pre: Tensor, sc2, sh2, mean2, inv2, PQR: Tensor, idx: Tensor, bx, scx, shx, meanx, invx, slope: float):
    """-> (g2 [E,F], gy [E,F], sums2 [2F] = [sum g2 | sum g2*xhat2], sumsy [2F])."""
    F_ = bx.numel()
    H = PQR.shape[1] - 2 * F_
    _pqr(PQR, H, F_); _i32(idx, "idx")
    M_, k = idx.shape
    dT_b = dT.dtype == torch.bfloat16           # bfloat16 storage (gemm_nt(out_bf16=True)): k = 10, F % 4 == 0
    if dT_b:
        _rowmajor2d_as(dT, "dT", torch.bfloat16)
    else:
        _f32(dT, "dT", 2)
    h2_half = h2pre.dtype == torch.float16
    if h2_half:
        if not dT_b:
            raise ValueError("a float16 h2pre belongs to the 16-bit storage mode (bfloat16 dT)")
        _rowmajor2d_as(h2pre, "h2pre", torch.float16)
    else:
        _f32(h2pre, "h2pre", 2)
    if not (dT.is_contiguous() and h2pre.is_contiguous()) or dT.numel() != M_ * k * F_ or h2pre.numel() != M_ * k * F_:
        raise ValueError("dT / h2pre must be contiguous with M*k*F elements")
    lib = _lib.load()
    tp = lib.spgan_edge_attend_bwd_tile_points()
    tiles = (M_ + tp - 1) // tp
    g2 = torch.empty((M_ * k, F_), dtype=torch.bfloat16 if dT_b else torch.float32, device=PQR.device)    # 16-bit storage: a GEMM operand only
    gy = torch.empty((M_ * k, F_), dtype=torch.bfloat16 if dT_b else torch.float32, device=PQR.device)   # 16-bit storage: gy too (edge_scatter reads it)
    part = torch.empty((tiles, 2 * F_, 2), dtype=torch.float32, device=PQR.device)
    v = lambda t, n: _p(_vec(t, F_, n))
    check((lib.spgan_edge_attend_bwd_b if dT_b else lib.spgan_edge_attend_bwd)(_p(dT), *((_p(h2pre), 1 if h2_half else 0) if dT_b else (_p(h2pre),)), v(sc2, "sc2"), v(sh2, "sh2"), v(mean2, "mean2"), v(inv2, "inv2"), _p(PQR), PQR.shape[1],
                                    H, F_, _p(idx), M_, k, v(bx, "bx"), v(scx, "scx"), v(shx, "shx"), v(meanx, "meanx"), v(invx, "invx"),
                                    float(slope), _p(g2), _p(gy), _p(part), _s()), "edge_attend_bwd", M=M_, k=k, F=F_)
    # partial columns are laid out so that the two finalize outputs ARE [sum g2 | sum g2*xhat2] and [sum gy | sum gy*xhaty]
    s0, s1 = _finalize(part, 1, tiles, 2 * F_, tiles * tp, 1, tp)
    return g2, gy, s0[0], s1[0]


def edge_scatter(g1: Tensor, gy: Tensor, PQR: Tensor, idx: Tensor, rowptr: Tensor, src: Tensor, b1, mean1, inv1, gam1, sums1, bx, meanx,
                 invx, gamx, sumsx) -> Tensor:
    """BatchNorm backward of both per-edge pre-activations + reduction onto points -> dPQR [M, H+2F]."""
    H, F_ = b1.numel(), bx.numel()
    _pqr(PQR, H, F_); _i32(idx, "idx"); _i32(rowptr, "rowptr"); _i32(src, "src")
    M_, k = idx.shape
    gy_b = gy.dtype == torch.bfloat16          # written by edge_attend_bwd in the 16-bit storage mode
    _f32(g1, "g1", 2)
    if gy_b:
        _rowmajor2d_as(gy, "gy", torch.bfloat16)
    else:
        _f32(gy, "gy", 2)
    if not (g1.is_contiguous() and gy.is_contiguous()) or g1.shape != (M_ * k, H) or gy.shape != (M_ * k, F_):
        raise ValueError("g1 [E,H] / gy [E,F] shape mismatch")
    out = torch.empty_like(PQR)
    vh = lambda t, n: _p(_vec(t, H, n))
    vf = lambda t, n: _p(_vec(t, F_, n))
    check((_lib.load().spgan_edge_scatter_b if gy_b else _lib.load().spgan_edge_scatter)(_p(g1), _p(gy), _p(PQR), PQR.shape[1], H, F_, _p(idx), _p(rowptr), _p(src), M_, k, vh(b1, "b1"),
                                         vh(mean1, "mean1"), vh(inv1, "inv1"), vh(gam1, "gam1"), _p(_vec(sums1, 2 * H, "sums1")), vf(bx, "bx"),
                                         vf(meanx, "meanx"), vf(invx, "invx"), vf(gamx, "gamx"), _p(_vec(sumsx, 2 * F_, "sumsx")), _p(out), _s()),
          "edge_scatter", M=M_, k=k, H=H, F=F_)
    return out


# ----------------------------------------------------------------------------- AdaIN
def adain_fwd(x: Tensor, N: int, slope: float, imean: Tensor, ivar: Tensor, gb: Tensor) -> Tensor:
    _f32(x, "x", 2); _f32(gb, "gb", 2)
    M_, Cn = x.shape
    if not (x.is_contiguous() and gb.is_contiguous()) or gb.shape != (M_, 2 * Cn):
        raise ValueError("x [M,C] and gb [M,2C] must be contiguous")
    out = torch.empty_like(x)
    check(_lib.load().spgan_adain_fwd(_p(x), M_, Cn, N, float(slope), _p(_vec(imean, (M_ // N) * Cn, "imean")),
                                      _p(_vec(ivar, (M_ // N) * Cn, "ivar")), BN_EPS, _p(gb), _p(out), _s()), "adain_fwd", M=M_, C=Cn, N=N)
    return out


def adain_bwd(dout: Tensor, x: Tensor, N: int, slope: float, imean: Tensor, ivar: Tensor, gb: Tensor) -> Tuple[Tensor, Tensor]:
    """-> (dx [M,C], dgb [M,2C])"""
    _f32(dout, "dout", 2); _f32(x, "x", 2); _f32(gb, "gb", 2)
    M_, Cn = x.shape
    if not (dout.is_contiguous() and x.is_contiguous() and gb.is_contiguous()) or dout.shape != x.shape:
        raise ValueError("dout/x/gb must be contiguous")
    B = M_ // N
    tpg = (N + ROW_TILE - 1) // ROW_TILE
    dgb = torch.empty_like(gb)
    part = torch.empty((B * tpg, Cn, 2), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    check(lib.spgan_adain_bwd1(_p(dout), _p(x), M_, Cn, N, float(slope), _p(imean), _p(ivar), BN_EPS, _p(gb), _p(dgb), _p(part), _s()), "adain_bwd1")
    S0, S1 = _finalize(part, B, tpg, Cn, N, 1)
    dx = torch.empty_like(x)
    check(lib.spgan_adain_bwd2(_p(dout), _p(x), M_, Cn, N, float(slope), _p(imean), _p(ivar), BN_EPS, _p(gb), _p(S0), _p(S1), _p(dx), _s()), "adain_bwd2")
    return dx, dgb


# ----------------------------------------------------------------------------- pooled BN backward, misc
_ROWIDS = {}


def _rowids(B: int, Cn: int, device) -> Tensor:
    key = (B, Cn, str(device))
    if key not in _ROWIDS:
        _ROWIDS[key] = torch.arange(B, dtype=torch.int32, device=device).view(B, 1).expand(B, Cn).contiguous()
    return _ROWIDS[key]


def pool_bwd_stats(gpool: Tensor, pooled: Tensor, argmax: Tensor, y: Tensor, mean: Tensor, invstd: Tensor, slope: float, prep=None):
    """gval = gpool*lrelu'(pooled); sums [2C] = [sum_b gval | sum_b gval*xhat(argmax row)].
    y is either the full [M,C] pre-BN tensor or, when that was never stored, its values at the arg-max rows [B,C].
    prep = (gamma, count, y_full | None, rows): additionally returns the lazily evaluated BatchNorm backward operand that
    sparse_bn_bwd_operand(gval, argmax, y_full, rows, mean, invstd, gamma, sums, count) would build -- from the same launch."""
    _f32(gpool, "gpool", 2); _rowmajor2d(y, "y")
    B, Cn = gpool.shape
    real_arg = argmax
    if y.shape[0] == B:                     # yarg [B,C]: "row b" of it is the arg-max row of shape b
        argmax = _rowids(B, Cn, y.device)
    gpool = gpool.contiguous()
    gval = torch.empty_like(gpool)
    sums = torch.empty((2 * Cn,), dtype=torch.float32, device=y.device)
    if prep is None:
        check(_lib.load().spgan_pool_bwd_stats(_p(gpool), _p(pooled), _p(_i32(argmax, "argmax")), _p(y), _ld(y), _p(mean), _p(invstd), float(slope), B, Cn,
                                               _p(gval), _p(sums), _s()), "pool_bwd_stats")
        return gval, sums
    gamma, count, y_full, rows = prep
    ab = torch.empty((2, Cn), dtype=torch.float32, device=y.device)
    cg = torch.empty_like(gval)
    check(_lib.load().spgan_pool_bwd_stats_prep(_p(gpool), _p(pooled), _p(_i32(argmax, "argmax")), _p(y), _ld(y), _p(_vec(mean, Cn, "mean")),
                                                _p(_vec(invstd, Cn, "invstd")), float(slope), B, Cn, _p(_vec(gamma, Cn, "gamma")), int(count), _p(gval),
                                                _p(sums), _p(ab[0]), _p(ab[1]), _p(cg), _s()), "pool_bwd_stats_prep")
    return gval, sums, SparseAffine(y_full, ab[0], ab[1], cg, real_arg, rows)


def pool_bwd_stats_multi(specs):
    """pool_bwd_stats(..., prep=...) for several passes as one launch; specs: dicts of its arguments (prep required); -> list of its results."""
    if len(specs) == 1 or not GROUPED[0]:
        return [pool_bwd_stats(**sp) for sp in specs]
    from ._lib import PoolBwdArgs
    n = len(specs)
    arr = (PoolBwdArgs * n)()
    res, keep = [], []
    for i, sp in enumerate(specs):
        gpool, pooled, argmax, y, mean, invstd, slope = (sp[k] for k in ("gpool", "pooled", "argmax", "y", "mean", "invstd", "slope"))
        gamma, count, y_full, rows = sp["prep"]
        _f32(gpool, "gpool", 2); _rowmajor2d(y, "y")
        B, Cn = gpool.shape
        real_arg = argmax
        if y.shape[0] == B:
            argmax = _rowids(B, Cn, y.device)
        gpool = gpool.contiguous()
        gval = torch.empty_like(gpool)
        sums = torch.empty((2 * Cn,), dtype=torch.float32, device=y.device)
        ab = torch.empty((2, Cn), dtype=torch.float32, device=y.device)
        cg = torch.empty_like(gval)
        q = arr[i]
        ks = [gpool, pooled.contiguous(), _i32(argmax, "argmax"), _vec(mean, Cn, "mean"), _vec(invstd, Cn, "invstd"), _vec(gamma, Cn, "gamma")]
        keep.append(ks)
        q.gpool = _p(ks[0]); q.pooled = _p(ks[1]); q.argmax = _p(ks[2]); q.y = _p(y); q.ld = _ld(y); q.mean = _p(ks[3]); q.invstd = _p(ks[4])
        q.slope = float(slope); q.B = B; q.C = Cn; q.gamma = _p(ks[5]); q.count = int(count)
        q.gval = _p(gval); q.sums = _p(sums); q.alpha = _p(ab[0]); q.beta = _p(ab[1]); q.cg = _p(cg)
        res.append((gval, sums, SparseAffine(y_full, ab[0], ab[1], cg, real_arg, rows)))
    check(_lib.load().spgan_pool_bwd_stats_prep_multi(arr, n, _s()), "pool_bwd_stats_prep_multi", count=n)
    return res


def bn_bwd_apply_sparse(gval: Tensor, argmax: Tensor, y: Tensor, N: int, mean, invstd, gamma, sums, count: int) -> Tensor:
    _rowmajor2d(y, "y")
    M_, Cn = y.shape
    dy = torch.empty((M_, Cn), dtype=torch.float32, device=y.device)
    check(_lib.load().spgan_bn_bwd_apply_sparse(_p(gval), _p(argmax), _p(y), _ld(y), M_, Cn, N, _p(mean), _p(invstd), _p(gamma),
                                                _p(_vec(sums, 2 * Cn, "sums")), count, _p(dy), _s()), "bn_bwd_apply_sparse")
    return dy


def maxpool_bwd_add(dpool: Tensor, argmax: Tensor, dst: Tensor) -> Tensor:
    """dst[argmax[b,c], c] += dpool[b,c] (in place)."""
    _rowmajor2d(dst, "dst")
    dpool = dpool.contiguous()
    B, Cn = dpool.shape
    check(_lib.load().spgan_maxpool_bwd_add(_p(dpool), _p(_i32(argmax, "argmax")), B, Cn, _p(dst), _ld(dst), _s()), "maxpool_bwd_add")
    return dst


def tanh_bwd(dy: Tensor, y: Tensor) -> Tensor:
    dy = dy.contiguous(); y = y.contiguous()
    out = torch.empty_like(dy)
    check(_lib.load().spgan_tanh_bwd(_p(dy), _p(y), dy.numel(), _p(out), _s()), "tanh_bwd")
    return out


def act_bwd(dy: Tensor, y: Tensor, act: int, slope: float = 0.0) -> Tensor:
    """dy * act'(y) computed from the activation output y."""
    dy = dy.contiguous(); y = y.contiguous()
    if act == ACT_NONE:
        return dy
    _f32(dy, "dy"); _f32(y, "y")
    out = torch.empty_like(dy)
    check(_lib.load().spgan_act_bwd(_p(dy), _p(y), dy.numel(), act, float(slope), _p(out), _s()), "act_bwd")
    return out


def scatter_rows(val: Tensor, argmax: Tensor, M: int) -> Tensor:
    """[B,C] values -> dense [M,C] with out[argmax[b,c], c] = val[b,c]."""
    val = _f32(val, "val", 2).contiguous()
    B, Cn = val.shape
    out = torch.empty((M, Cn), dtype=torch.float32, device=val.device)
    check(_lib.load().spgan_scatter_rows(_p(val), _p(_i32(argmax, "argmax")), B, Cn, M, _p(out), _s()), "scatter_rows")
    return out


def gather_rows(src: Tensor, argmax: Tensor) -> Tensor:
    _rowmajor2d(src, "src")
    B, Cn = argmax.shape
    out = torch.empty((B, Cn), dtype=torch.float32, device=src.device)
    check(_lib.load().spgan_gather_rows(_p(src), _ld(src), _p(_i32(argmax, "argmax")), B, Cn, _p(out), _s()), "gather_rows")
    return out


def bn_dbl_stats(u: Tensor, y: Tensor, gz: Tensor, mean: Tensor, invstd: Tensor):
    """-> (sum u, sum u*xhat, sum u*gz) each [C]."""
    for t, n in ((u, "u"), (y, "y"), (gz, "gz")):
        _f32(t, n, 2)
        if not t.is_contiguous() or t.shape != u.shape:
            raise ValueError("bn_dbl_stats: %s must be contiguous [M,C]" % n)
    M_, Cn = u.shape
    tiles = (M_ + ROW_TILE - 1) // ROW_TILE
    part = torch.empty((tiles, 2 * Cn, 2), dtype=torch.float32, device=u.device)
    check(_lib.load().spgan_bn_dbl_stats(_p(u), _p(y), _p(gz), M_, Cn, _p(_vec(mean, Cn, "mean")), _p(_vec(invstd, Cn, "invstd")), _p(part), _s()),
          "bn_dbl_stats")
    s0, s1 = _finalize(part, 1, tiles, 2 * Cn, M_, 1)
    return s0[0, :Cn].contiguous(), s1[0, :Cn].contiguous(), s0[0, Cn:].contiguous()


def bn_dbl_apply(u, y, gz, mean, invstd, scale, shift, slope: float, gamma, S1, U0, U1, count: int):
    """-> (q, xbar) [M,C]; see spgan_hip.h."""
    M_, Cn = u.shape
    q = torch.empty_like(u); xbar = torch.empty_like(u)
    v = lambda t, n: _p(_vec(t.contiguous(), Cn, n))
    check(_lib.load().spgan_bn_dbl_apply(_p(u), _p(y), _p(gz), M_, Cn, v(mean, "mean"), v(invstd, "invstd"), v(scale, "scale"), v(shift, "shift"),
                                         float(slope), v(gamma, "gamma"), v(S1, "S1"), v(U0, "U0"), v(U1, "U1"), _p(q), _p(xbar), _s()), "bn_dbl_apply")
    return q, xbar


def col_scale_add(a: Tensor, b: Tensor, gamma: Tensor) -> Tensor:
    """a + gamma[c]*b"""
    M_, Cn = a.shape
    out = torch.empty_like(a)
    check(_lib.load().spgan_col_scale_add(_p(a.contiguous()), _p(b.contiguous()), _p(_vec(gamma.contiguous(), Cn, "gamma")), M_, Cn, _p(out), _s()),
          "col_scale_add")
    return out


GAN_MODES = {"ls": 0, "wgan": 1, "hinge": 2, "gan": 3}


def stacked_rows(parts) -> Tensor:
    """torch.cat(parts, dim=0) -- without a launch when the parts are consecutive row blocks of ONE buffer (contiguous, same storage, each starting
    where the previous ends): then the result is a view of that buffer."""
    parts = list(parts)
    first = parts[0]
    ok = all(isinstance(p_, torch.Tensor) and p_.is_contiguous() and p_.dtype == first.dtype and p_.shape[1:] == first.shape[1:] for p_ in parts)
    if ok and len(parts) > 1:
        st = first.untyped_storage().data_ptr()
        end = first.data_ptr() + first.numel() * first.element_size()
        for p_ in parts[1:]:
            if p_.untyped_storage().data_ptr() != st or p_.data_ptr() != end:
                ok = False
                break
            end += p_.numel() * p_.element_size()
    if ok and len(parts) > 1:
        rows = sum(p_.shape[0] for p_ in parts)
        return first.as_strided((rows,) + tuple(first.shape[1:]), first.stride(), first.storage_offset())
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)


def gan_loss(mode: int, which: int, d_real: Optional[Tensor], d_fake: Tensor, real_label: Optional[Tensor] = None,
             fake_label: Optional[Tensor] = None):
    """-> (out5 = [loss, fake term, real term, real_acc, fake_acc], g_real [B,1] | None, g_fake [B,1])."""
    d_fake = _f32(d_fake, "d_fake").contiguous()
    B = d_fake.shape[0]
    if d_real is not None:
        d_real = _f32(d_real, "d_real").contiguous()
    out5 = torch.empty((5,), dtype=torch.float32, device=d_fake.device)
    if d_real is not None and d_real.shape == d_fake.shape:
        # one buffer, real rows first: a consumer that wants the two seeds stacked (the batched head's backward) takes the buffer as it is
        both = torch.empty((2 * B,) + tuple(d_fake.shape[1:]), dtype=torch.float32, device=d_fake.device)
        g_real, g_fake = both[:B], both[B:]
    else:
        g_fake = torch.empty_like(d_fake)
        g_real = torch.empty_like(d_real) if d_real is not None else None
    check(_lib.load().spgan_gan_loss(mode, which, _p(d_real), _p(d_fake), _p(_vec(real_label, B, "real_label")), _p(_vec(fake_label, B, "fake_label")),
                                     B, _p(out5), _p(g_real), _p(g_fake), _s()), "gan_loss", mode=mode, which=which, B=B)
    return out5, g_real, g_fake


def lerp_rows(real: Tensor, fake: Tensor, alpha: Tensor) -> Tensor:
    """real + alpha[b]*(fake-real), alpha has B elements."""
    real = _f32(real, "real").contiguous(); fake = _f32(fake, "fake").contiguous()
    B = real.shape[0]
    out = torch.empty_like(real)
    check(_lib.load().spgan_lerp_rows(_p(real), _p(fake), _p(_vec(alpha.contiguous(), B, "alpha")), B, real.numel() // B, _p(out), _s()), "lerp_rows")
    return out


def gp_penalty_fwd(g: Tensor, gamma: float, lam: float):
    """-> (loss [1], norms [B])"""
    g = _f32(g, "g").contiguous()
    B = g.shape[0]
    norms = torch.empty((B,), dtype=torch.float32, device=g.device)
    loss = torch.empty((1,), dtype=torch.float32, device=g.device)
    check(_lib.load().spgan_gp_penalty_fwd(_p(g), B, g.numel() // B, float(gamma), float(lam), _p(norms), _p(loss), _s()), "gp_penalty_fwd")
    return loss, norms


def gp_penalty_fwd_bwd(g: Tensor, gamma: float, lam: float, loss_add: Optional[Tensor] = None):
    """gp_penalty_fwd and gp_penalty_bwd(upstream = None) with one launch less: -> (loss [1], norms [B], v, total | None); with loss_add [1]:
    total [1] = loss_add + loss from the same launch (the D step's reported loss).  The values of the separate calls, bit for bit."""
    g = _f32(g, "g").contiguous()
    B = g.shape[0]
    norms = torch.empty((B,), dtype=torch.float32, device=g.device)
    loss = torch.empty((1,), dtype=torch.float32, device=g.device)
    v = torch.empty_like(g)
    total = la = None
    if loss_add is not None:
        la = _f32(loss_add, "loss_add").reshape(1)
        total = torch.empty((1,), dtype=torch.float32, device=g.device)
    check(_lib.load().spgan_gp_penalty_fwd_bwd(_p(g), B, g.numel() // B, float(gamma), float(lam), _p(norms), _p(loss), _p(la), _p(total), _p(v), _s()),
          "gp_penalty_fwd_bwd")
    return loss, norms, v, total


def gp_penalty_bwd(g: Tensor, norms: Tensor, gamma: float, lam: float, upstream: Optional[Tensor]) -> Tensor:
    g = g.contiguous()
    B = g.shape[0]
    v = torch.empty_like(g)
    up = None if upstream is None else upstream.reshape(1).contiguous()
    check(_lib.load().spgan_gp_penalty_bwd(_p(g), _p(norms), B, g.numel() // B, float(gamma), float(lam), _p(up), _p(v), _s()), "gp_penalty_bwd")
    return v


def multi_transpose(srcs) -> list:
    """[w^T contiguous for w in srcs] (2-D fp32, unit column stride) in one launch per 64 matrices."""
    from ._lib import MULTI_MAX, MultiTransposeArgs
    lib = _lib.load()
    outs = []
    for i0 in range(0, len(srcs), MULTI_MAX):
        chunk = srcs[i0:i0 + MULTI_MAX]
        a = MultiTransposeArgs()
        a.count = len(chunk)
        start = 0
        for e, w in enumerate(chunk):
            _rowmajor2d(w, "srcs[%d]" % (i0 + e))
            r, c = w.shape
            t = torch.empty((c, r), dtype=torch.float32, device=w.device)
            outs.append(t)
            a.src[e] = w.data_ptr(); a.dst[e] = t.data_ptr(); a.rows[e] = r; a.cols[e] = c; a.ld[e] = _ld(w)
            a.tile_start[e] = start
            start += ((r + 31) // 32) * ((c + 31) // 32)
        a.tile_start[len(chunk)] = start
        check(lib.spgan_multi_transpose(C.byref(a), _s()), "multi_transpose", count=len(chunk))
    return outs


def softmax_rows(S: Tensor) -> Tensor:
    """softmax over the last dimension, in place (contiguous [..., cols])."""
    _f32(S, "S")
    if not S.is_contiguous():
        raise ValueError("softmax_rows needs a contiguous tensor")
    cols = S.shape[-1]
    check(_lib.load().spgan_softmax_rows(_p(S), S.numel() // max(cols, 1), cols, _s()), "softmax_rows", shape=tuple(S.shape))
    return S


def softmax_rows_bwd(P: Tensor, dP: Tensor) -> Tensor:
    """dP <- P * (dP - sum_j dP*P) in place: gradient w.r.t. the softmax input, given its output P."""
    _f32(P, "P"); _f32(dP, "dP")
    if not (P.is_contiguous() and dP.is_contiguous()) or P.shape != dP.shape:
        raise ValueError("softmax_rows_bwd needs contiguous tensors of equal shape")
    cols = P.shape[-1]
    check(_lib.load().spgan_softmax_rows_bwd(_p(P), _p(dP), P.numel() // max(cols, 1), cols, _s()), "softmax_rows_bwd", shape=tuple(P.shape))
    return dP


def scale_residual(o: Tensor, x: Tensor, gamma: Tensor) -> Tensor:
    """gamma*o + x with the scalar gamma read on the device."""
    _f32(o, "o"); _f32(x, "x"); _f32(gamma, "gamma")
    if not (o.is_contiguous() and x.is_contiguous()) or o.shape != x.shape or o.numel() % 4 or gamma.numel() != 1:
        raise ValueError("scale_residual: o and x must be contiguous, equal in shape, with a multiple of 4 elements; gamma a scalar")
    y = torch.empty_like(x)
    check(_lib.load().spgan_scale_residual(_p(o), _p(x), _p(gamma), _p(y), o.numel(), _s()), "scale_residual", n=o.numel())
    return y


def scale_residual_bwd(dy: Tensor, o: Tensor, gamma: Tensor):
    """-> (d_o = gamma*dy, dgamma = sum(dy*o) as a 0-d tensor)."""
    _f32(dy, "dy"); _f32(o, "o"); _f32(gamma, "gamma")
    if not (o.is_contiguous() and dy.is_contiguous()) or o.shape != dy.shape or o.numel() % 4 or gamma.numel() != 1:
        raise ValueError("scale_residual_bwd: dy and o must be contiguous, equal in shape, with a multiple of 4 elements")
    lib = _lib.load()
    d_o = torch.empty_like(o)
    dgamma = torch.empty((), dtype=torch.float32, device=o.device)
    wsb = lib.spgan_scale_residual_bwd_ws_bytes(o.numel())
    ws = torch.empty((max(wsb // 4, 1),), dtype=torch.float32, device=o.device)
    check(lib.spgan_scale_residual_bwd(_p(dy), _p(o), _p(gamma), _p(d_o), _p(dgamma), _p(ws), wsb, o.numel(), _s()), "scale_residual_bwd", n=o.numel())
    return d_o, dgamma


def _strided3(d: Tensor, s: Tensor):
    """(n1, n2, dst strides, src strides) of a pair with one common shape, as 3-D strided views (size-1 dimensions dropped); None when
    both are contiguous (the plain pair).  Raises for pairs that need more than three strided dimensions."""
    if d.is_contiguous() and s.is_contiguous():
        return None
    dims = [(n, ds_, ss_) for n, ds_, ss_ in zip(d.shape, d.stride(), s.stride()) if n != 1]
    merged = []
    for n, ds_, ss_ in dims:      # merge neighbours that are contiguous in BOTH views
        if merged and merged[-1][1] == n * ds_ and merged[-1][2] == n * ss_:
            pn, _, _ = merged[-1]
            merged[-1] = (pn * n, ds_, ss_)
        else:
            merged.append((n, ds_, ss_))
    if len(merged) > 3:
        raise ValueError("multi_add: a pair needs more than three strided dimensions")
    while len(merged) < 3:
        merged.insert(0, (1, 0, 0))
    return merged[1][0], merged[2][0], [m[1] for m in merged], [m[2] for m in merged]


def multi_add(dsts, srcs) -> None:
    """dst[t] += src[t] for a list of fp32 tensor pairs of equal shape in ceil(T/64) launches.  Contiguous pairs of equal size may differ
    in shape; a pair of equal SHAPE may consist of strided views (<= 3 strided dimensions: a permuted source, a column block of the
    destination) -- spgan_multi_add3."""
    from ._lib import MULTI_MAX, MultiAddArgs, MultiAdd3Args
    lib = _lib.load()
    pairs = [(d, s) for d, s in zip(dsts, srcs)]
    for i0 in range(0, len(pairs), MULTI_MAX):
        chunk = pairs[i0:i0 + MULTI_MAX]
        views = []
        for t, (d, s) in enumerate(chunk):
            if d.numel() != s.numel() or d.dtype != torch.float32 or s.dtype != torch.float32:
                raise ValueError("multi_add: pair %d is not an fp32 pair of equal size" % (i0 + t))
            if not (d.is_cuda and s.is_cuda):
                raise RuntimeError("multi_add needs GPU tensors")
            if not (d.is_contiguous() and s.is_contiguous()) and d.shape != s.shape:
                raise ValueError("multi_add: strided pair %d must have one common shape" % (i0 + t))
            views.append(_strided3(d, s))
        if all(v is None for v in views):
            a = MultiAddArgs()
            a.count = len(chunk)
            for t, (d, s) in enumerate(chunk):
                a.dst[t] = d.data_ptr(); a.src[t] = s.data_ptr(); a.n[t] = d.numel()
            check(lib.spgan_multi_add(C.byref(a), _s()), "multi_add", count=len(chunk))
            continue
        a3 = MultiAdd3Args()
        a3.count = len(chunk)
        for t, ((d, s), v) in enumerate(zip(chunk, views)):
            a3.dst[t] = d.data_ptr(); a3.src[t] = s.data_ptr(); a3.n[t] = d.numel()
            a3.n1[t], a3.n2[t] = (1, 1) if v is None else (v[0], v[1])
            for ax in range(3):
                a3.ds[ax][t] = 0 if v is None else v[2][ax]
                a3.ss[ax][t] = 0 if v is None else v[3][ax]
        check(lib.spgan_multi_add3(C.byref(a3), _s()), "multi_add3", count=len(chunk))


def reduce_chunks(recv: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """recv [parts, n] (contiguous) -> out[n] = sum over parts in ascending order (the local step of a one-hop all-reduce)."""
    _f32(recv, "recv", 2)
    if not recv.is_contiguous():
        raise ValueError("recv must be contiguous [parts, n]")
    parts, n = recv.shape
    if out is None:
        out = torch.empty((n,), dtype=torch.float32, device=recv.device)
    elif out.numel() != n or not out.is_contiguous() or out.dtype != torch.float32:
        raise ValueError("out must be a contiguous fp32 tensor of %d elements" % n)
    check(_lib.load().spgan_reduce_chunks(_p(recv), parts, n, _p(out), _s()), "reduce_chunks", parts=parts, n=n)
    return out


def multi_copy(dsts, srcs) -> None:
    """dst[t].copy_(src[t]) for a list of contiguous fp32 GPU tensor pairs of equal size in ceil(T/64) launches."""
    from ._lib import MULTI_MAX, MultiAddArgs
    lib = _lib.load()
    pairs = [(d, s) for d, s in zip(dsts, srcs)]
    for i0 in range(0, len(pairs), MULTI_MAX):
        chunk = pairs[i0:i0 + MULTI_MAX]
        a = MultiAddArgs()
        a.count = len(chunk)
        for t, (d, s) in enumerate(chunk):
            if not (d.is_contiguous() and s.is_contiguous()) or d.numel() != s.numel() or d.dtype != torch.float32 or s.dtype != torch.float32:
                raise ValueError("multi_copy: pair %d is not a contiguous fp32 pair of equal size" % (i0 + t))
            if not (d.is_cuda and s.is_cuda):
                raise RuntimeError("multi_copy needs GPU tensors")
            a.dst[t] = d.data_ptr(); a.src[t] = s.data_ptr(); a.n[t] = d.numel()
        check(lib.spgan_multi_copy(C.byref(a), _s()), "multi_copy", count=len(chunk))


def axpby(a: float, x: Tensor, b: float, y: Tensor) -> Tensor:
    """y = a*x + b*y in place (contiguous fp32)."""
    if not (x.is_contiguous() and y.is_contiguous()) or x.numel() != y.numel():
        raise ValueError("axpby needs contiguous tensors of equal size")
    check(_lib.load().spgan_axpby(float(a), _p(x), float(b), _p(y), x.numel(), _s()), "axpby")
    return y


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float = 1e-4, beta1: float = 0.5, beta2: float = 0.99,
              eps: float = 1e-8, grad_scale: float = 1.0) -> None:
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _f32(t, n)
        if not t.is_contiguous() or t.numel() != p.numel():
            raise ValueError("adam_step: %s must be contiguous with %d elements" % (n, p.numel()))
    check(_lib.load().spgan_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, step, grad_scale, _s()), "adam_step")


def adam_step_dev(p: Tensor, g: Tensor, m: Tensor, v: Tensor, state: Tensor, lr: float = 1e-4, beta1: float = 0.5, beta2: float = 0.99,
                  eps: float = 1e-8, grad_scale: float = 1.0, zero_grad: bool = False) -> None:
    """adam_step with the step count kept in `state` (4 floats on the device: int step bits, 1-beta1^t, 1/sqrt(1-beta2^t), and a
    multiplier on lr that schedules write); every call advances it.  Nothing host-side changes between steps: capturable in a
    hipGraph."""
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _f32(t, n)
        if not t.is_contiguous() or t.numel() != p.numel():
            raise ValueError("adam_step_dev: %s must be contiguous with %d elements" % (n, p.numel()))
    _f32(state, "state")
    if state.numel() != 4 or not state.is_contiguous():
        raise ValueError("adam_step_dev: state must be 4 contiguous floats (step bits, two bias corrections, lr multiplier)")
    check(_lib.load().spgan_adam_step_dev(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, _p(state), grad_scale, 1 if zero_grad else 0, _s()), "adam_step_dev")
