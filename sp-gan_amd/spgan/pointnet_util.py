"""Drop-in for the ball-query / grouping helpers of the reference
(Common/pointnet_util.py:19-143, Common/pointconv_util.py:60-197): same names, argument order and return
shapes/dtypes; HIP kernels underneath (libspgan_hip.so), GPU tensors only.  The index-producing ops carry no gradient (as in the
reference); the gathers -- index_points, group, sample_and_group -- are differentiable in `points` / `xyz` like the reference's torch
indexing (pointnet_util.py:43-60, pointconv_util.py:174-197), with a deterministic adjoint (per-point slot lists, no float atomics).

    square_distance(src, dst)                          [B,N,C],[B,M,C] -> [B,N,M]
    index_points(points, idx)                          [B,N,C],[B,S(,K)] -> [B,S(,K),C]
    farthest_point_sample(xyz, npoint, start=None)     -> int64 [B,npoint]   (start=None: random, like pointnet_util;
                                                          start=0-tensor: pointconv_util's variant)
    query_ball_point(radius, nsample, xyz, new_xyz)    -> int64 [B,S,nsample]
    knn_point(nsample, xyz, new_xyz)                   -> int64 [B,S,nsample]  (ascending; the reference's order is unspecified)
    group(nsample, xyz, points)                        -> (new_points [B,N,K,C+D], grouped_xyz_norm [B,N,K,C])
    sample_and_group(npoint, radius, nsample, xyz, points, returnfps=False, start=None)
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from ._lib import check
from . import ops
from .ops import _f32, _p, _s

Tensor = torch.Tensor


def _xyz(t: Tensor, name: str) -> Tensor:
    _f32(t, name, 3)
    return t.contiguous()


def square_distance(src: Tensor, dst: Tensor) -> Tensor:
    src, dst = _xyz(src, "src"), _xyz(dst, "dst")
    B, N, C = src.shape
    M = dst.shape[1]
    out = torch.empty((B, N, M), dtype=torch.float32, device=src.device)
    check(_lib.load().spgan_square_distance(_p(src), _p(dst), B, N, M, C, _p(out), _s()), "square_distance", B=B, N=N, M=M, C=C)
    return out


def gather_csr(idx: Tensor, N: int):
    """Per point of each shape, the gather slots that read it: idx int64 [B, ...] (local indices) -> (rowptr int32 [B*N,2], src int32 [B*S])."""
    B = idx.shape[0]
    S = idx.numel() // B
    rowptr = torch.empty((B * N, 2), dtype=torch.int32, device=idx.device)
    src = torch.empty((B * S,), dtype=torch.int32, device=idx.device)
    bad = ops.index_check_flag(idx.device)
    check(_lib.load().spgan_gather_csr(_p(idx), B, S, N, _p(rowptr), _p(src), None if bad is None else _p(bad), _s()), "gather_csr", B=B, S=S, N=N)
    ops.index_check_raise(bad, "gather_csr: an index lies outside [0, %d)" % N)
    return rowptr, src


def _scatter_slots(dout2d: Tensor, col0: int, C: int, rowptr: Tensor, src: Tensor, shape) -> Tensor:
    out = torch.empty(shape, dtype=torch.float32, device=dout2d.device)
    check(_lib.load().spgan_scatter_slots(_p(dout2d), dout2d.shape[1], col0, C, _p(rowptr), _p(src), rowptr.shape[0], _p(out), _s()),
          "scatter_slots", ld=dout2d.shape[1], col0=col0, C=C)
    return out


def _index_points_raw(points: Tensor, idx: Tensor) -> Tensor:
    B, N, C = points.shape
    S = idx.numel() // B
    out = torch.empty(tuple(idx.shape) + (C,), dtype=torch.float32, device=points.device)
    check(_lib.load().spgan_index_points(_p(points), _p(idx), B, N, C, S, _p(out), _s()), "index_points", B=B, N=N, C=C, S=S)
    return out


class _IndexPointsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx):
        ctx.save_for_backward(idx)
        ctx.shape = tuple(points.shape)
        return _index_points_raw(points, idx)

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        B, N, C = ctx.shape
        rowptr, src = gather_csr(idx, N)
        return _scatter_slots(dout.contiguous().view(-1, C), 0, C, rowptr, src, ctx.shape), None


def index_points(points: Tensor, idx: Tensor) -> Tensor:
    points = _xyz(points, "points")
    if idx.dtype != torch.int64 or not idx.is_cuda:
        raise TypeError("idx must be an int64 GPU tensor")
    idx = idx.contiguous()
    if points.requires_grad and torch.is_grad_enabled():
        return _IndexPointsFn.apply(points, idx)
    return _index_points_raw(points, idx)


def farthest_point_sample(xyz: Tensor, npoint: int, start: Optional[Tensor] = None) -> Tensor:
    xyz = _xyz(xyz, "xyz")
    B, N, C = xyz.shape
    if C != 3:
        raise ValueError("farthest_point_sample expects xyz [B,N,3]")
    if start is None:
        start = torch.randint(0, N, (B,), dtype=torch.long, device=xyz.device)        # pointnet_util.py:75
    start = start.to(device=xyz.device, dtype=torch.long).contiguous()
    out = torch.empty((B, npoint), dtype=torch.int64, device=xyz.device)
    ws = torch.empty((B, N), dtype=torch.float32, device=xyz.device)
    check(_lib.load().spgan_farthest_point_sample(_p(xyz), B, N, npoint, _p(start), _p(out), _p(ws), _s()), "farthest_point_sample", B=B, N=N)
    return out


def query_ball_point(radius: float, nsample: int, xyz: Tensor, new_xyz: Tensor) -> Tensor:
    xyz, new_xyz = _xyz(xyz, "xyz"), _xyz(new_xyz, "new_xyz")
    B, N, C = xyz.shape
    S = new_xyz.shape[1]
    out = torch.empty((B, S, nsample), dtype=torch.int64, device=xyz.device)
    check(_lib.load().spgan_query_ball_point(float(radius), nsample, _p(xyz), _p(new_xyz), B, N, S, C, _p(out), _s()), "query_ball_point",
          B=B, N=N, S=S, C=C)
    return out


def knn_point(nsample: int, xyz: Tensor, new_xyz: Tensor) -> Tensor:
    xyz, new_xyz = _xyz(xyz, "xyz"), _xyz(new_xyz, "new_xyz")
    B, N, C = xyz.shape
    S = new_xyz.shape[1]
    out = torch.empty((B, S, nsample), dtype=torch.int64, device=xyz.device)
    check(_lib.load().spgan_knn_point(nsample, _p(xyz), _p(new_xyz), B, N, S, C, _p(out), _s()), "knn_point", B=B, N=N, S=S, C=C)
    return out


def _group_concat_raw(xyz: Tensor, center: Tensor, feat: Optional[Tensor], idx: Tensor) -> Tensor:
    B, N, C = xyz.shape
    S, K = idx.shape[1], idx.shape[2]
    D = 0 if feat is None else feat.shape[2]
    out = torch.empty((B, S, K, C + D), dtype=torch.float32, device=xyz.device)
    check(_lib.load().spgan_group_concat(_p(xyz), _p(center.contiguous()), _p(None if feat is None else feat.contiguous()), _p(idx.contiguous()),
                                         B, N, S, K, C, D, _p(out), _s()), "group_concat", B=B, N=N, S=S, K=K)
    return out


class _GroupConcatFn(torch.autograd.Function):
    """out[b,s,j,:] = [xyz[b,idx] - center[b,s] | feat[b,idx]] with gradients for xyz, center and feat."""

    @staticmethod
    def forward(ctx, xyz, center, feat, idx):
        ctx.save_for_backward(idx)
        ctx.shapes = (tuple(xyz.shape), tuple(center.shape), None if feat is None else tuple(feat.shape))
        return _group_concat_raw(xyz, center, feat, idx)

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        sx, sc, sf = ctx.shapes
        B, N, C = sx
        S, K = idx.shape[1], idx.shape[2]
        W = C + (0 if sf is None else sf[2])
        d2 = dout.contiguous().view(-1, W)
        dxyz = dcen = dfeat = None
        if ctx.needs_input_grad[0] or (sf is not None and ctx.needs_input_grad[2]):
            rowptr, src = gather_csr(idx, N)
            if ctx.needs_input_grad[0]:
                dxyz = _scatter_slots(d2, 0, C, rowptr, src, sx)
            if sf is not None and ctx.needs_input_grad[2]:
                dfeat = _scatter_slots(d2, C, sf[2], rowptr, src, sf)
        if ctx.needs_input_grad[1]:
            dcen = torch.empty(sc, dtype=torch.float32, device=dout.device)
            check(_lib.load().spgan_group_center_bwd(_p(d2), W, B * S, K, C, _p(dcen), _s()), "group_center_bwd", Q=B * S, K=K, C=C)
        return dxyz, dcen, dfeat, None


def _group_concat(xyz: Tensor, center: Tensor, feat: Optional[Tensor], idx: Tensor) -> Tensor:
    if torch.is_grad_enabled() and (xyz.requires_grad or center.requires_grad or (feat is not None and feat.requires_grad)):
        return _GroupConcatFn.apply(xyz, center.contiguous(), None if feat is None else feat.contiguous(), idx.contiguous())
    return _group_concat_raw(xyz, center, feat, idx)


def group(nsample: int, xyz: Tensor, points: Optional[Tensor]):
    """kNN-group every point around itself (pointconv_util.py:174-197)."""
    xyz = _xyz(xyz, "xyz")
    idx = knn_point(nsample, xyz.detach(), xyz.detach())
    C = xyz.shape[2]
    new_points = _group_concat(xyz, xyz, None if points is None else _xyz(points, "points"), idx)
    grouped_xyz_norm = new_points[..., :C].contiguous() if points is not None else new_points
    return new_points, grouped_xyz_norm


def sample_and_group(npoint: int, radius: float, nsample: int, xyz: Tensor, points: Optional[Tensor], returnfps: bool = False,
                     start: Optional[Tensor] = None):
    """FPS centres + ball query + centred grouping (pointnet_util.py:110-143)."""
    xyz = _xyz(xyz, "xyz")
    fps_idx = farthest_point_sample(xyz.detach(), npoint, start)
    new_xyz = index_points(xyz, fps_idx)
    idx = query_ball_point(radius, nsample, xyz.detach(), new_xyz.detach())
    new_points = _group_concat(xyz, new_xyz, None if points is None else _xyz(points, "points"), idx)
    if returnfps:
        return new_xyz, new_points, index_points(xyz, idx), fps_idx
    return new_xyz, new_points
