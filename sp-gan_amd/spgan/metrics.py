"""Chamfer-distance evaluation metrics on the GPU, with the reference's names:

  ChamferDistance / chamferFunction     metrics/CD_EMD/cd/chamferdist/ChamferDistance.py:11-57 (over chamfer.cu)
  nn_distance                           what evaluation_metrics.distChamferCUDA calls (evaluation_metrics.py:22-23)
  pairwise_cd, lgan_mmd_cov, knn,       the CD half of metrics/evaluation_metrics.py:89-208
  compute_all_metrics_cd

The distance searches and the all-pairs Chamfer matrix are HIP kernels (`csrc/metrics.hip`); what remains here are reductions
over the [S,R] matrices (a few thousand numbers).  The EMD half needs the auction solver of metrics/emd (not built yet).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

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


def lgan_mmd_cov(all_dist: Tensor) -> Dict[str, Tensor]:
    """evaluation_metrics.py:161-173; all_dist [N_sample, N_ref]."""
    n_ref = all_dist.shape[1]
    min_val_fromsmp, min_idx = torch.min(all_dist, dim=1)
    min_val, _ = torch.min(all_dist, dim=0)
    cov = torch.tensor(float(min_idx.unique().numel()) / float(n_ref)).to(all_dist)
    return {"lgan_mmd": min_val.mean(), "lgan_cov": cov, "lgan_mmd_smp": min_val_fromsmp.mean()}


def knn(Mxx: Tensor, Mxy: Tensor, Myy: Tensor, k: int, sqrt: bool = False) -> Dict[str, Tensor]:
    """Leave-one-out k-NN two-sample test, evaluation_metrics.py:129-158."""
    n0, n1 = Mxx.size(0), Myy.size(0)
    label = torch.cat((torch.ones(n0), torch.zeros(n1))).to(Mxx)
    M = torch.cat((torch.cat((Mxx, Mxy), 1), torch.cat((Mxy.transpose(0, 1), Myy), 1)), 0)
    if sqrt:
        M = M.abs().sqrt()
    _, idx = (M + torch.diag(float("inf") * torch.ones(n0 + n1).to(Mxx))).topk(k, 0, False)
    count = torch.zeros(n0 + n1).to(Mxx)
    for i in range(k):
        count = count + label.index_select(0, idx[i])
    pred = torch.ge(count, (float(k) / 2) * torch.ones(n0 + n1).to(Mxx)).float()
    s = {"tp": (pred * label).sum(), "fp": (pred * (1 - label)).sum(), "fn": ((1 - pred) * label).sum(), "tn": ((1 - pred) * (1 - label)).sum()}
    s.update({"precision": s["tp"] / (s["tp"] + s["fp"] + 1e-10), "recall": s["tp"] / (s["tp"] + s["fn"] + 1e-10),
              "acc_t": s["tp"] / (s["tp"] + s["fn"] + 1e-10), "acc_f": s["tn"] / (s["tn"] + s["fp"] + 1e-10),
              "acc": torch.eq(label, pred).float().mean()})
    return s


def compute_all_metrics_cd(sample_pcs: Tensor, ref_pcs: Tensor) -> Dict[str, Tensor]:
    """The Chamfer rows of compute_all_metrics (evaluation_metrics.py:176-207): lgan_mmd-CD, lgan_cov-CD, lgan_mmd_smp-CD and
    the 1-NN-CD accuracies."""
    M_rs = pairwise_cd(ref_pcs, sample_pcs)
    res = {"%s-CD" % k: v for k, v in lgan_mmd_cov(M_rs.t()).items()}
    M_rr, M_ss = pairwise_cd(ref_pcs, ref_pcs), pairwise_cd(sample_pcs, sample_pcs)
    res.update({"1-NN-CD-%s" % k: v for k, v in knn(M_rr, M_rs, M_ss, 1, sqrt=False).items() if "acc" in k})
    return res
