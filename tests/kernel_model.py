"""Plain-PyTorch models of every op in spgan.ops (TEST INFRASTRUCTURE ONLY).

Two uses:
  * `-m gpu` tests compare each HIP kernel against its model on the same inputs;
  * `-m "not gpu"` tests monkeypatch `spgan.ops` with these models to check the *host
    composition* (forward/backward/double-backward pipelines in spgan.nets) against the
    oracle on the CPU.  The product never imports this file.
Signatures mirror spgan/ops.py one to one.
"""
from typing import Optional

import torch
import torch.nn.functional as F

ACT_NONE, ACT_LRELU, ACT_TANH = 0, 1, 2
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def _lrelu(x, s):
    return torch.where(x > 0, x, x * s)


# ----------------------------------------------------------------------------- graph
def knn(x_pm, B, N, k, mode=0):
    C = x_pm.shape[1]
    x = x_pm.view(B, N, C)
    if mode == 1:
        xd = x.double()
        d = ((xd[:, :, None, :] - xd[:, None, :, :]) ** 2).sum(-1)
    else:
        inner = torch.bmm(x, x.transpose(1, 2))
        sq = (x * x).sum(-1)
        d = (-2 * inner + sq[:, :, None]) + sq[:, None, :]
    order = torch.sort(d, dim=2, stable=True)[1][:, :, 1:k + 1]
    off = (torch.arange(B, device=x_pm.device) * N).view(B, 1, 1)
    return (order + off).reshape(B * N, k).to(torch.int32)


def csr_build(idx, B, N):
    M, k = idx.shape
    flat = idx.reshape(-1).long()
    order = torch.sort(flat, stable=True)[1]          # edge ids grouped by target, ascending within a target
    deg = torch.bincount(flat, minlength=M)
    rowptr = torch.zeros(M + 1, dtype=torch.int64, device=idx.device)
    rowptr[1:] = torch.cumsum(deg, 0)
    return rowptr.to(torch.int32), order.to(torch.int32)


def edge_features_cm(x_cm, idx_local, k):
    B, C, N = x_cm.shape
    idx3 = idx_local.view(B, 1, N * k).expand(B, C, N * k)
    nb = torch.gather(x_cm, 2, idx3).view(B, C, N, k)
    ctr = x_cm.unsqueeze(3).expand(B, C, N, k)
    return torch.cat([ctr, nb - ctr], dim=1).contiguous()


def idx_to_local64(idx, B, N):
    k = idx.shape[1]
    off = (torch.arange(B, device=idx.device) * N).view(B, 1)
    return (idx.view(B, N * k).long() - off).contiguous()


def idx_from_local64(idx_local, B, N, k):
    off = (torch.arange(B, device=idx_local.device) * N).view(B, 1)
    return (idx_local.view(B, N * k) + off).view(B * N, k).to(torch.int32)


# ----------------------------------------------------------------------------- layout
def cm_to_pm(x_cm):
    B, C, N = x_cm.shape
    return x_cm.permute(0, 2, 1).reshape(B * N, C).contiguous()


def pm_to_cm(x_pm, B, N):
    return x_pm.view(B, N, -1).permute(0, 2, 1).contiguous()


def concat2(a, b):
    return torch.cat([a, b], dim=1).contiguous()


# ----------------------------------------------------------------------------- contractions
def _operand(A, pro, edge, K):
    if edge is not None:
        idx, ebias = edge
        k = idx.shape[1]
        i = torch.arange(idx.shape[0], device=A.device).repeat_interleave(k)
        j = idx.reshape(-1).long()
        a = (A[j, :K] - A[i, :K]) + ebias
    else:
        a = A[:, :K]
    if pro is not None:
        sc, sh, ps = pro
        a = _lrelu(a * sc + sh, ps)
    return a


def gemm_nt(A, W, bias=None, *, pro=None, edge=None, rowbias=None, rows_per_group=0, act=ACT_NONE, slope=0.0, stats=False, M=None):
    N, K = W.shape
    a = _operand(A, pro, edge, K)
    if M is not None:
        a = a[:M]
    y = a @ W.t()
    if bias is not None:
        y = y + bias
    if rowbias is not None:
        y = y + rowbias.repeat_interleave(rows_per_group, dim=0)[:y.shape[0]]
    pre = y
    if act == ACT_LRELU:
        y = _lrelu(y, slope)
    elif act == ACT_TANH:
        y = torch.tanh(y)
    if stats:
        return y.contiguous(), pre.mean(0), pre.var(0, unbiased=False)
    return y.contiguous()


def gemm_nt_maskout(A, W, ref, slope):
    return ((A @ W.t()) * torch.where(ref > 0, 1.0, slope)).contiguous()


def gemm_nt_bnbwd(A, W, y_ref, scale, shift, mean, invstd, slope, edge=None):
    N = W.shape[0]
    if edge is not None:
        idx, ebias = edge
        k = idx.shape[1]
        i = torch.arange(idx.shape[0], device=A.device).repeat_interleave(k)
        y = (y_ref[idx.reshape(-1).long(), :N] - y_ref[i, :N]) + ebias
    else:
        y = y_ref[:, :N]
    z = y * scale + shift
    g = (A @ W.t()) * torch.where(z > 0, 1.0, slope)
    xh = (y - mean) * invstd
    return g.contiguous(), g.sum(0), (g * xh).sum(0)


def gemm_tn(A, Bm, *, pro=None, edge=None, out=None, beta=0.0):
    b = _operand(Bm, pro, edge, Bm.shape[1])
    c = A.t() @ b
    if out is None:
        return c.contiguous()
    out.copy_(beta * out + c)
    return out


# ----------------------------------------------------------------------------- reductions / norms
def colstats(X, G, slope=1.0):
    M, C = X.shape
    v = _lrelu(X, slope).reshape(M // G, G, C)
    return v.mean(1), v.var(1, unbiased=False)


def colsum(X, G=None):
    M, C = X.shape
    G = M if G is None else G
    return X.reshape(M // G, G, C).sum(1)


def bn_prepare(mean, var, gamma, beta, count, training=True, running_mean=None, running_var=None, momentum=BN_MOMENTUM, eps=BN_EPS):
    if training:
        m, v = mean, var
        if running_mean is not None:
            unb = v * (count / (count - 1)) if count > 1 else v
            running_mean.mul_(1 - momentum).add_(momentum * m)
            running_var.mul_(1 - momentum).add_(momentum * unb)
    else:
        m, v = running_mean, running_var
    inv = 1.0 / torch.sqrt(v + eps)
    g = gamma if gamma is not None else torch.ones_like(inv)
    b = beta if beta is not None else torch.zeros_like(inv)
    sc = g * inv
    return sc, b - m * sc, inv, m.clone()


def bn_bwd_apply(g, y, mean, invstd, gamma, sums, count):
    C = g.shape[1]
    xh = (y - mean) * invstd
    ga = gamma if gamma is not None else torch.ones_like(mean)
    return (ga * invstd * (g - sums[:C] / count - xh * (sums[C:] / count))).contiguous()


def maxpool(y, B, N, scale=None, shift=None, slope=1.0):
    C = y.shape[1]
    v = y
    if scale is not None:
        v = v * scale + shift
    v = _lrelu(v, slope).reshape(B, N, C)
    out, arg = v.max(1)
    off = (torch.arange(B, device=y.device) * N).view(B, 1)
    return out.contiguous(), (arg + off).to(torch.int32)
