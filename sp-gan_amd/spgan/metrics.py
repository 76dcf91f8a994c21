"""Chamfer-distance evaluation metrics on the GPU, with the reference's names:

  ChamferDistance / chamferFunction     metrics/CD_EMD/cd/chamferdist/ChamferDistance.py:11-57 (over chamfer.cu)
  nn_distance                           what evaluation_metrics.distChamferCUDA calls (evaluation_metrics.py:22-23)
  pairwise_cd, lgan_mmd_cov, knn,       the CD half of metrics/evaluation_metrics.py:89-208
  compute_all_metrics_cd

  unit_cube_grid_point_cloud, entropy_of_occupancy_grid,     the JSD metric, metrics/evaluation_metrics.py:210-322
  jensen_shannon_divergence, jsd_between_point_cloud_sets

  emdFunction / emdModule                the auction EMD of metrics/CD_EMD/emd_/emd_module.py:33-85 (over emd_cuda.cu)
  emd_approx, pairwise_emd, EMD_CD,      the EMD half of metrics/evaluation_metrics.py:26-35,52-126,176-207.  There `emd_approx`
  compute_all_metrics                    calls StructuralLosses.match_cost, an extension that is NOT part of the reference tree;
                                         here it is the tree's own auction module: mean over the points of sqrt(dist).

The distance searches, the all-pairs Chamfer matrix, the occupancy-grid statistics and the auction are HIP kernels
(`csrc/metrics.hip`, `csrc/emd.hip`); what remains here are reductions over the [S,R] matrices and the grid histograms (a few
thousand numbers).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib
from .ops import _f32, _p, _s, check

Tensor = torch.Tensor


def _cloud(t: Tensor, name: str) -> Tensor:
    _f32(t, name, 3)
    if t.shape[2] != 3:
        raise ValueError("%s must be [B, N, 3], got %s" % (name, tuple(t.shape)))
    return t.contiguous()


def _nn(a: Tensor, b: Tensor):
    B, N, _ = a.shape
    M = b.shape[1]
    dist = torch.empty((B, N), dtype=torch.float32, device=a.device)
    idx = torch.empty((B, N), dtype=torch.int32, device=a.device)
    check(_lib.load().spgan_nn_distance(_p(a), _p(b), B, N, M, _p(dist), _p(idx), _s()), "nn_distance", B=B, N=N, M=M)
    return dist, idx


class chamferFunction(Function):
    """forward(xyz1 [B,N,3], xyz2 [B,M,3]) -> dist1 [B,N], dist2 [B,M], idx1, idx2 (int32, indices into the other cloud)."""

    @staticmethod
    def forward(ctx, xyz1, xyz2):
        xyz1, xyz2 = _cloud(xyz1, "xyz1"), _cloud(xyz2, "xyz2")
        if xyz1.shape[0] != xyz2.shape[0]:
            raise ValueError("batch sizes differ")
        dist1, idx1 = _nn(xyz1, xyz2)
        dist2, idx2 = _nn(xyz2, xyz1)
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        return dist1, dist2, idx1, idx2

    @staticmethod
    def backward(ctx, g1, g2, _i1, _i2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        B, N, _ = xyz1.shape
        M = xyz2.shape[1]
        g1 = torch.zeros((B, N), device=xyz1.device) if g1 is None else g1.contiguous()
        g2 = torch.zeros((B, M), device=xyz1.device) if g2 is None else g2.contiguous()
        d1, d2 = torch.empty_like(xyz1), torch.empty_like(xyz2)
        lib = _lib.load()
        check(lib.spgan_chamfer_bwd(_p(xyz1), _p(xyz2), B, N, M, _p(g1), _p(idx1), _p(g2), _p(idx2), _p(d1), _s()), "chamfer_bwd")
        check(lib.spgan_chamfer_bwd(_p(xyz2), _p(xyz1), B, M, N, _p(g2), _p(idx2), _p(g1), _p(idx1), _p(d2), _s()), "chamfer_bwd")
        return d1, d2


class ChamferDistance(nn.Module):
    """ChamferDistance.py:53-57: forward(input1, input2) -> (dist1, dist2, idx1, idx2)."""

    def forward(self, input1, input2):
        return chamferFunction.apply(input1, input2)


def nn_distance(x: Tensor, y: Tensor):
    """-> (dist1 [B,N], dist2 [B,M]): squared distance of every point to its nearest point in the other cloud."""
    d1, d2, _, _ = chamferFunction.apply(x, y)
    return d1, d2


def pairwise_cd(sample_pcs: Tensor, ref_pcs: Tensor) -> Tensor:
    """[S,R] Chamfer matrix, entry = mean_i min_j + mean_j min_i (the CD half of _pairwise_EMD_CD_, evaluation_metrics.py:
    89-126) -- one launch for all S*R pairs instead of S launches over R-sized batches."""
    a, b = _cloud(sample_pcs, "sample_pcs"), _cloud(ref_pcs, "ref_pcs")
    S, N, _ = a.shape
    R, M, _ = b.shape
    out = torch.empty((S, R), dtype=torch.float32, device=a.device)
    check(_lib.load().spgan_chamfer_pairs(_p(a), _p(b), S, R, N, M, _p(out), _s()), "chamfer_pairs", S=S, R=R, N=N, M=M)
    return out


class emdFunction(Function):
    """emd_module.py:33-75: forward(xyz1 [B,n,3], xyz2 [B,n,3], eps=0.005, iters=50) -> (dist [B,n] squared distances to the assigned
    points, assignment [B,n] int32).  Gradient for xyz1 only, as there."""

    @staticmethod
    def forward(ctx, xyz1, xyz2, eps=0.005, iters=50):
        xyz1, xyz2 = _cloud(xyz1, "xyz1"), _cloud(xyz2, "xyz2")
        if xyz1.shape != xyz2.shape:
            raise ValueError("the two clouds must have the same batch size and point count, got %s and %s" % (tuple(xyz1.shape), tuple(xyz2.shape)))
        B, n, _ = xyz1.shape
        lib = _lib.load()
        dist = torch.empty((B, n), dtype=torch.float32, device=xyz1.device)
        assignment = torch.empty((B, n), dtype=torch.int32, device=xyz1.device)
        wsb = lib.spgan_emd_ws_bytes(B, n)
        ws = torch.empty((wsb // 8,), dtype=torch.int64, device=xyz1.device)
        check(lib.spgan_emd_forward(_p(xyz1), _p(xyz2), B, n, float(eps), int(iters), _p(dist), _p(assignment), _p(ws), wsb, _s()),
              "emd_forward", B=B, n=n, iters=iters)
        ctx.save_for_backward(xyz1, xyz2, assignment)
        ctx.mark_non_differentiable(assignment)
        return dist, assignment

    @staticmethod
    def backward(ctx, graddist, _gradidx):
        xyz1, xyz2, assignment = ctx.saved_tensors
        B, n, _ = xyz1.shape
        g1 = torch.empty_like(xyz1)
        check(_lib.load().spgan_emd_backward(_p(xyz1), _p(xyz2), B, n, _p(graddist.contiguous()), _p(assignment), _p(g1), _s()), "emd_backward")
        return g1, torch.zeros_like(xyz2), None, None


class emdModule(nn.Module):
    """emd_module.py:78-85: forward(input1, input2, eps, iters) -> (dist, assignment)."""

    def forward(self, input1, input2, eps=0.005, iters=50):
        return emdFunction.apply(input1, input2, eps, iters)


def emd_approx(sample: Tensor, ref: Tensor, eps: float = 0.005, iters: int = 50) -> Tensor:
    """Per-pair EMD / N (evaluation_metrics.py:26-35), [B]: mean over the points of the distance to the assigned point."""
    dist, _ = emdFunction.apply(sample, ref, eps, iters)
    return dist.sqrt().mean(dim=1)


def pairwise_emd(sample_pcs: Tensor, ref_pcs: Tensor, batch_size: int = 512, eps: float = 0.005, iters: int = 50) -> Tensor:
    """[S,R] matrix of emd_approx over all pairs (the EMD half of _pairwise_EMD_CD_, evaluation_metrics.py:89-126), `batch_size`
    pairs per auction launch sequence."""
    a, b = _cloud(sample_pcs, "sample_pcs"), _cloud(ref_pcs, "ref_pcs")
    S, R = a.shape[0], b.shape[0]
    out = torch.empty((S * R,), dtype=torch.float32, device=a.device)
    si = torch.arange(S, device=a.device).repeat_interleave(R)
    ri = torch.arange(R, device=a.device).repeat(S)
    for lo in range(0, S * R, batch_size):
        out[lo:lo + batch_size] = emd_approx(a[si[lo:lo + batch_size]], b[ri[lo:lo + batch_size]], eps, iters)
    return out.view(S, R)


def EMD_CD(sample_pcs: Tensor, ref_pcs: Tensor, batch_size: int = 512, reduced: bool = True) -> Dict[str, Tensor]:
    """evaluation_metrics.py:52-86: Chamfer and EMD between corresponding clouds."""
    dl, dr = nn_distance(sample_pcs, ref_pcs)
    cd = dl.mean(dim=1) + dr.mean(dim=1)
    emd = torch.cat([emd_approx(sample_pcs[lo:lo + batch_size], ref_pcs[lo:lo + batch_size]) for lo in range(0, sample_pcs.shape[0], batch_size)])
    return {"MMD-CD": cd.mean() if reduced else cd, "MMD-EMD": emd.mean() if reduced else emd}


def _matrix(t: Tensor, name: str) -> Tensor:
    _f32(t, name, 2)
    return t.contiguous()


def lgan_mmd_cov(all_dist: Tensor) -> Dict[str, Tensor]:
    """Minimum matching distance and coverage of a [samples, references] distance matrix -- the quantities (and dictionary keys)
    of evaluation_metrics.py:161-173.  One HIP launch sequence (spgan_mmd_cov: a row pass, a column pass, a fixed-order
    finish); results stay on the device as 0-d tensors."""
    d = _matrix(all_dist, "all_dist")
    S, R = d.shape
    out = torch.empty((3,), dtype=torch.float32, device=d.device)
    ws = torch.empty((S + 2 * R,), dtype=torch.float32, device=d.device)
    check(_lib.load().spgan_mmd_cov(_p(d), S, R, _p(out), _p(ws), _s()), "mmd_cov", S=S, R=R)
    return {"lgan_mmd": out[0], "lgan_cov": out[1], "lgan_mmd_smp": out[2]}


_TWO_SAMPLE_KEYS = ("tp", "fp", "fn", "tn", "precision", "recall", "acc_t", "acc_f", "acc")


def knn(Mxx: Tensor, Mxy: Tensor, Myy: Tensor, k: int, sqrt: bool = False) -> Dict[str, Tensor]:
    """1-NNA-style two-sample test (evaluation_metrics.py:129-158 semantics and result keys): each of the n0 + n1 clouds is
    labelled by the vote of its k nearest other clouds in the joint distance matrix [[Mxx, Mxy], [Mxy^T, Myy]]; the confusion
    counts of those leave-one-out predictions and the ratios derived from them are returned.  The joint matrix is read block-wise
    by spgan_two_sample_knn, not assembled."""
    xx, xy, yy = _matrix(Mxx, "Mxx"), _matrix(Mxy, "Mxy"), _matrix(Myy, "Myy")
    n0, n1 = xx.shape[0], yy.shape[0]
    if xx.shape != (n0, n0) or yy.shape != (n1, n1) or xy.shape != (n0, n1):
        raise ValueError("expected Mxx [n0,n0], Mxy [n0,n1], Myy [n1,n1]; got %s %s %s" % (tuple(xx.shape), tuple(xy.shape), tuple(yy.shape)))
    out = torch.empty((9,), dtype=torch.float32, device=xx.device)
    pred = torch.empty((n0 + n1,), dtype=torch.int32, device=xx.device)
    check(_lib.load().spgan_two_sample_knn(_p(xx), _p(xy), _p(yy), n0, n1, int(k), 1 if sqrt else 0, _p(out), _p(pred), _s()),
          "two_sample_knn", n0=n0, n1=n1, k=k)
    return {name: out[i] for i, name in enumerate(_TWO_SAMPLE_KEYS)}


def unit_cube_grid_point_cloud(resolution: int, clip_sphere: bool = False, device="cuda"):
    """evaluation_metrics.py:210-229: centres of the resolution^3 cells of the unit cube (float32, i*spacing - 0.5 per axis,
    x slowest), optionally only those within the sphere of radius 0.5.  -> (grid, spacing); grid is [R,R,R,3] or, clipped, [G,3]."""
    spacing = 1.0 / float(resolution - 1)
    ax = (torch.arange(resolution, dtype=torch.float64) * spacing - 0.5).to(torch.float32)        # float64 product stored as float32, as there
    grid = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), dim=-1)
    if clip_sphere:
        grid = grid.reshape(-1, 3)
        grid = grid[torch.linalg.norm(grid, dim=1) <= 0.5]
    return grid.to(device), spacing


def _entropy(p: Tensor, base: Optional[float] = None) -> Tensor:
    """scipy.stats.entropy: normalises, sum(-p log p) with 0 log 0 = 0 (float64)."""
    p = p.double()
    p = p / p.sum()
    h = -(torch.where(p > 0, p * torch.log(torch.where(p > 0, p, torch.ones_like(p))), torch.zeros_like(p))).sum()
    return h / math.log(base) if base else h


def entropy_of_occupancy_grid(pclouds: Tensor, grid_resolution: int, in_sphere: bool = False):
    """evaluation_metrics.py:247-290: -> (mean Bernoulli entropy of the cell-activation variables, grid_counters [G]).  Every point
    is assigned to its nearest grid centre by one brute-force search on the GPU (spgan_nn_distance; the reference fits a
    k-d tree per call), the per-cell point and cloud counts come from spgan_occupancy_counts."""
    pc = _cloud(pclouds, "pclouds")
    S, N, _ = pc.shape
    grid, _ = unit_cube_grid_point_cloud(grid_resolution, in_sphere, pc.device)
    grid = grid.reshape(-1, 3).contiguous()
    G = grid.shape[0]
    _, cell = _nn(pc.view(1, S * N, 3), grid.view(1, G, 3))
    counters = torch.zeros(G, dtype=torch.int32, device=pc.device)
    bern = torch.zeros(G, dtype=torch.int32, device=pc.device)
    check(_lib.load().spgan_occupancy_counts(_p(cell), S, N, G, _p(counters), _p(bern), _s()), "occupancy_counts", S=S, N=N, G=G)
    p = bern.double() / float(S)
    q = 1.0 - p
    h = -(torch.where(p > 0, p * torch.log(p.clamp_min(1e-300)), torch.zeros_like(p)) + torch.where(q > 0, q * torch.log(q.clamp_min(1e-300)), torch.zeros_like(q)))
    acc = torch.where(bern > 0, h, torch.zeros_like(h)).sum()
    return acc / G, counters.double()


def jensen_shannon_divergence(P: Tensor, Q: Tensor) -> Tensor:
    """evaluation_metrics.py:293-312 (base-2 entropies)."""
    if bool((P < 0).any()) or bool((Q < 0).any()):
        raise ValueError("Negative values.")
    if P.numel() != Q.numel():
        raise ValueError("Non equal size.")
    P_, Q_ = P.double() / P.double().sum(), Q.double() / Q.double().sum()
    return _entropy((P_ + Q_) / 2.0, 2) - (_entropy(P_, 2) + _entropy(Q_, 2)) / 2.0


def jsd_between_point_cloud_sets(sample_pcs: Tensor, ref_pcs: Tensor, resolution: int = 28) -> Tensor:
    """evaluation_metrics.py:232-244: JSD between the occupancy histograms of two sets of clouds (in the unit sphere of radius 0.5)."""
    sample_grid_var = entropy_of_occupancy_grid(sample_pcs, resolution, True)[1]
    ref_grid_var = entropy_of_occupancy_grid(ref_pcs, resolution, True)[1]
    return jensen_shannon_divergence(sample_grid_var, ref_grid_var)


def compute_all_metrics(sample_pcs: Tensor, ref_pcs: Tensor, batch_size: int = 512) -> Dict[str, Tensor]:
    """evaluation_metrics.py:176-207: MMD / COV / 1-NNA for both Chamfer and EMD."""
    res = compute_all_metrics_cd(sample_pcs, ref_pcs)
    M_rs = pairwise_emd(ref_pcs, sample_pcs, batch_size)
    res.update({"%s-EMD" % k: v for k, v in lgan_mmd_cov(M_rs.t()).items()})
    M_rr, M_ss = pairwise_emd(ref_pcs, ref_pcs, batch_size), pairwise_emd(sample_pcs, sample_pcs, batch_size)
    res.update({"1-NN-EMD-%s" % k: v for k, v in knn(M_rr, M_rs, M_ss, 1, sqrt=False).items() if "acc" in k})
    return res


def compute_all_metrics_cd(sample_pcs: Tensor, ref_pcs: Tensor) -> Dict[str, Tensor]:
    """The Chamfer rows of compute_all_metrics (evaluation_metrics.py:176-207): lgan_mmd-CD, lgan_cov-CD, lgan_mmd_smp-CD and
    the 1-NN-CD accuracies."""
    M_rs = pairwise_cd(ref_pcs, sample_pcs)
    res = {"%s-CD" % k: v for k, v in lgan_mmd_cov(M_rs.t()).items()}
    M_rr, M_ss = pairwise_cd(ref_pcs, ref_pcs), pairwise_cd(sample_pcs, sample_pcs)
    res.update({"1-NN-CD-%s" % k: v for k, v in knn(M_rr, M_rs, M_ss, 1, sqrt=False).items() if "acc" in k})
    return res
