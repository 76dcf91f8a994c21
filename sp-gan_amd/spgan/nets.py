"""Hand-written forward / backward / double-backward pipelines over the HIP ops.

Everything here works on raw tensors in the point-major layout ([M, C], M = B*N) and on
parameter dicts keyed by the reference's state_dict names; `functions.py` wraps these
pipelines into torch.autograd.Functions and `modules.py` into nn.Modules.

All arithmetic goes through `spgan.ops` (HIP kernels).  torch is used for allocation, views,
weight transposes (`.t().contiguous()`: data movement) and a few O(C) per-channel scalar
formulas in the WGAN-GP double backward.
"""
from __future__ import annotations

import os
import weakref
from typing import Dict, List, Optional, Tuple

import torch

from . import ops

Tensor = torch.Tensor
NEG = 0.01     # Generator.py:22, Discriminator.py:19
NEG_2 = 0.2    # Generator.py:23
ZERO_GRAD = 0  # sentinel for a gradient that is exactly zero (e.g. a conv bias in front of a train-mode BatchNorm)


def _w2(w: Tensor) -> Tensor:
    """[Cout, Cin, 1(,1)] conv weight -> [Cout, Cin] view."""
    return w.view(w.shape[0], w.shape[1])


_T_CACHE: Dict[Tuple[int, Tuple[int, ...]], tuple] = {}      # (data_ptr, shape) -> (stamp, w^T, weakref to the owning Parameter)


def owned(p: Tensor) -> Tensor:
    """p.detach() that remembers which Parameter it came from: the autograd Functions hand detached weights to the backward
    pipelines, and the transpose cache below needs the owner to know when a cached copy is stale."""
    d = p.detach()
    if isinstance(p, torch.nn.Parameter):
        d._spgan_owner = weakref.ref(p)
    return d


def _owner(w: Tensor):
    base = w._base if w._base is not None else w
    if isinstance(base, torch.nn.Parameter):
        return base
    ref = getattr(base, "_spgan_owner", None)
    return ref() if ref is not None else None


def _t(w: Tensor) -> Tensor:
    """w^T, contiguous.  For (views of) parameters -- or of their `owned()` detached aliases -- the copy is cached until the
    weights change: the backward passes of a D-step transpose the same matrices five times.  "Changed" = an in-place torch op
    (version counter, shared by detached aliases) or an optimiser step of spgan.optim.Adam, whose HIP kernel updates the flat
    buffer behind torch's back and bumps ops.WEIGHTS_EPOCH instead.  The first stale hit after such a change re-transposes ALL
    stale cached matrices in one launch (ops.multi_transpose) instead of one copy kernel per weight."""
    owner = _owner(w)
    if owner is None:
        return w.t().contiguous()
    key = (w.data_ptr(), tuple(w.shape))
    stamp = (ops.WEIGHTS_EPOCH[0], owner._version)
    hit = _T_CACHE.get(key)
    if hit is not None and hit[2]() is owner:                        # same live Parameter object (not a new one at a recycled address)
        if hit[0] == stamp:
            return hit[1]
        _refresh_transposes(owner)
        hit = _T_CACHE.get(key)
        if hit is not None and hit[0] == stamp:
            return hit[1]
    t = w.t().contiguous()
    _T_CACHE[key] = (stamp, t, weakref.ref(owner), (tuple(w.shape), tuple(w.stride()), w.storage_offset()))
    return t


_COLS_CACHE: Dict[tuple, tuple] = {}


def _cols_from(w: Tensor, c: int) -> Tensor:
    """w[:, c:] as a contiguous (16-byte aligned) matrix.  For (views of) parameters the copy is kept until the weights change (the staleness
    rule of _t): the latent half of head.0's weight, W0[:, 3:], starts 12 bytes into every 524-byte row -- as a strided view it sends a
    [B,128] x [128,128] product to the unaligned path of the 128-row tile kernel (18 us for B = 32), as a copy to the small-row kernel (6 us)."""
    owner = _owner(w)
    if owner is None:
        return w[:, c:].contiguous()
    key = (w.data_ptr(), tuple(w.shape), c)
    stamp = (ops.WEIGHTS_EPOCH[0], owner._version)
    hit = _COLS_CACHE.get(key)
    if hit is not None and hit[2]() is owner and hit[0] == stamp:
        return hit[1]
    t = w[:, c:].contiguous()
    _COLS_CACHE[key] = (stamp, t, weakref.ref(owner))      # (an entry made inside a capture: the replay re-makes the copy, and the stamp changes with every optimiser step anyway)
    return t


def _refresh_transposes(trigger: Tensor) -> None:
    """Re-transpose the stale cached matrices of the network `trigger` belongs to (= the parameters that live in the same flat
    buffer, spgan.optim.flatten_module; an unflattened parameter is its own group) -- never another model's: a captured train
    step must not read the parameters of a model that may be freed while the graph lives."""
    epoch = ops.WEIGHTS_EPOCH[0]
    group = trigger.untyped_storage().data_ptr()
    todo, srcs = [], []
    for key, (stamp, _t_old, oref, view) in list(_T_CACHE.items()):
        o = oref()
        if o is None:
            del _T_CACHE[key]                                        # the model is gone
            continue
        if o.untyped_storage().data_ptr() != group:
            continue
        cur = (epoch, o._version)
        if stamp == cur:
            continue
        shape, stride, offset = view
        try:
            src = torch.as_strided(o.detach(), shape, stride, offset)
        except RuntimeError:
            del _T_CACHE[key]
            continue
        if src.data_ptr() != key[0] or (shape[1] > 1 and stride[1] != 1):
            del _T_CACHE[key]                                        # storage moved / unusual view: rebuilt on its next use
            continue
        todo.append((key, cur, oref, view))
        srcs.append(src)
    if srcs:
        for (key, cur, oref, view), t in zip(todo, ops.multi_transpose(srcs)):
            _T_CACHE[key] = (cur, t, oref, view)


LAZY_BN_BWD = [True]     # test hook: False materialises every BatchNorm-backward tensor (tests/test_kernels2_gpu.py compares the two forms; A/B-measured in round 3)


def _lazy_ok(rows: int, channels: int) -> bool:
    return LAZY_BN_BWD[0] and rows > 64 and channels % 4 == 0


def _bn_bwd(g: Tensor, y: Tensor, mean: Tensor, invstd: Tensor, gamma, sums: Tensor, count: int, lazy: bool = True):
    """The BatchNorm backward dy = gamma*invstd*(g - S0/count - xhat*S1/count) of a layer whose dy is consumed by GEMMs only: as a lazy
    two-tensor operand (ops.Affine2: the weight-gradient and input-gradient products evaluate p*g + q*y + r on their operand loads; one
    C-sized launch instead of a pass that reads g and y and writes dy), or materialised (ops.bn_bwd_apply) for the few consumers without
    that operand mode."""
    if lazy and _lazy_ok(g.shape[0], g.shape[1]):
        return ops.bn_bwd_lazy(g, y, mean, invstd, gamma, sums, count)
    return ops.bn_bwd_apply(g, y, mean, invstd, gamma, sums, count)


def _dense(t):
    return t.dense() if isinstance(t, ops.Affine2) else t


_CONST_VEC: Dict[Tuple[int, float, str], Tensor] = {}


def _const_vec(n: int, value: float, device) -> Tensor:
    """A cached constant vector (created once, outside any later graph capture's allocations)."""
    key = (n, float(value), str(device))
    v = _CONST_VEC.get(key)
    if v is None:
        v = torch.full((n,), float(value), dtype=torch.float32, device=device)
        if not ops.capturing():        # a constant born inside a capture lives in the graph's pool and is only filled on replay: not cached
            _CONST_VEC[key] = v
    return v


def _cat2(s0: Tensor, s1: Tensor) -> Tensor:
    """[s0 | s1] without a copy when the two vectors already sit back to back in one allocation (the finalize kernels
    write (sum0, sum1) into one [2,C] buffer)."""
    n = s0.numel()
    if (s0.is_contiguous() and s1.is_contiguous() and s1.numel() == n and s1.data_ptr() == s0.data_ptr() + 4 * n
            and s0.untyped_storage().data_ptr() == s1.untyped_storage().data_ptr()):
        return torch.as_strided(s0, (2 * n,), (1,), s0.storage_offset())
    return torch.cat([s0, s1])


def _bn_train(mean, var, P, bufs, bn, count, training, update_running):
    g, b = P[bn + ".weight"], P[bn + ".bias"]
    rm = rv = None
    if bufs is not None and (not training or update_running):
        rm, rv = bufs[bn + ".running_mean"], bufs[bn + ".running_var"]
    if training:
        out = ops.bn_prepare(mean, var, g, b, count, True, rm, rv)
        if rm is not None and (bn + ".num_batches_tracked") in bufs:
            pend = bufs.get("__pending_counts__")
            if pend is not None:        # counted on the host, flushed into the int64 buffer by the module's state_dict hook
                pend[bn + ".num_batches_tracked"] = pend.get(bn + ".num_batches_tracked", 0) + 1
            else:
                bufs[bn + ".num_batches_tracked"] += 1
        return out
    return ops.bn_prepare(None, None, g, b, count, False, rm, rv)


def _count_bn_call(bufs, bn):
    if bufs is not None and (bn + ".num_batches_tracked") in bufs:
        pend = bufs.get("__pending_counts__")
        if pend is not None:
            pend[bn + ".num_batches_tracked"] = pend.get(bn + ".num_batches_tracked", 0) + 1
        else:
            bufs[bn + ".num_batches_tracked"] += 1


def _gemm_bn(A, W, b, P, bufs, bn, count, training, update_running, **kw):
    """GEMM followed by a BatchNorm whose statistics come out of the GEMM epilogue: returns (y_pre_bn, (scale, shift, invstd, mean)).
    Train mode: ONE extra launch (finalize + BN bookkeeping) after the GEMM."""
    if training:
        rm = rv = None
        if bufs is not None and update_running:
            rm, rv = bufs[bn + ".running_mean"], bufs[bn + ".running_var"]
        y, st = ops.gemm_nt(A, W, b, bn=(P[bn + ".weight"], P[bn + ".bias"], rm, rv), **kw)
        if rm is not None:
            _count_bn_call(bufs, bn)
        return y, st
    return ops.gemm_nt(A, W, b, **kw), _bn_train(None, None, P, bufs, bn, count, False, False)


# =============================================================================================
# Discriminator (Generation/Discriminator.py:97-115)
# =============================================================================================
D_LAYERS = (("mlps.0", "mlps.1"), ("mlps.3", "mlps.4"), ("mlps.6", "mlps.7"), ("fc2.0", "fc2.1"))
D_MLP = ("mlp.0", "mlp.2", "mlp.4", "mlp.6")


def d_head_forward(P: Dict[str, Tensor], pooled: Tensor):
    """The per-shape MLP head (Discriminator.py:83-95,110-113): pooled [B,C4] -> (logits [B,1], [h0, h1, h2, logits])."""
    h, hs = pooled, []
    for i, name in enumerate(D_MLP):
        last = i == len(D_MLP) - 1
        h = ops.gemm_nt(h, P[name + ".weight"], P[name + ".bias"], act=ops.ACT_NONE if last else ops.ACT_LRELU, slope=NEG)
        hs.append(h)
    return hs[-1], hs


def d_head_backward(P, pooled: Tensor, hs, dout: Tensor, need_dparams: bool):
    """-> (gpool [B,C4], {name: grad} of the head, dhs = gradients w.r.t. the pre-activation output of each head layer)."""
    grads: Dict[str, Tensor] = {}
    acts = [pooled, hs[0], hs[1], hs[2]]          # input of mlp.0/2/4/6
    d = dout
    dsum = None                                   # column sums of d (= the bias gradient), when the producing launch had them
    dhs = [None, None, None, dout]
    gpool = None
    for li in (3, 2, 1, 0):
        name = D_MLP[li]
        if need_dparams:
            grads[name + ".weight"] = ops.gemm_tn(d, acts[li], defer=True)
            grads[name + ".bias"] = dsum if dsum is not None else ops.colsum(d)[0]
        Wt = _t(P[name + ".weight"])
        if li > 0:
            # through the (in-place) LeakyReLU of the layer below; the bias gradient of that layer rides along
            if need_dparams:
                d, dsum = ops.gemm_nt_maskout(d, Wt, acts[li], NEG, with_colsum=True)
            else:
                d = ops.gemm_nt_maskout(d, Wt, acts[li], NEG)
            dhs[li - 1] = d
        else:
            gpool = ops.gemm_nt(d, Wt)
    return gpool, grads, dhs


def d_forward(P: Dict[str, Tensor], bufs: Optional[Dict[str, Tensor]], x_cm: Tensor, training: bool = True,
              update_running: bool = True, head: bool = True):
    """x [B,3,N] -> logits [B,1] plus the saved context for backward.  head=False: stop behind the max-pool and return
    (pooled [B,C4], ctx) -- the head of several passes can then run as ONE batch (it has no BatchNorm: rows are independent)."""
    B, _, N = x_cm.shape
    M = B * N
    x_pm = ops.cm_to_pm(x_cm)
    ys, bns = [], []
    a, pro = x_pm, None
    yarg = None
    for li, (conv, bn) in enumerate(D_LAYERS):
        W, b = _w2(P[conv + ".weight"]), P[conv + ".bias"]
        if li == 3 and training and N % ops.ROW_TILE == 0:
            # fc2.0 + BatchNorm + LeakyReLU + max over N in one GEMM: the [M,1024] output is never stored -- every backward of this
            # layer is collapsed (d_backward / d_double_backward) and needs y4 only at the arg-max rows (yarg)
            rm = rv = None
            if bufs is not None and update_running:
                rm, rv = bufs[bn + ".running_mean"], bufs[bn + ".running_var"]
            y, (sc, sh, inv, mu), pooled, argmax, yarg = ops.gemm_bn_pool(a, W, b, (P[bn + ".weight"], P[bn + ".bias"], rm, rv), N, NEG, pro=pro)
            if rm is not None:
                _count_bn_call(bufs, bn)
            ys.append(y); bns.append((sc, sh, inv, mu))
            break
        y, (sc, sh, inv, mu) = _gemm_bn(a, W, b, P, bufs, bn, M, training, update_running, pro=pro)
        ys.append(y); bns.append((sc, sh, inv, mu))
        a, pro = y, (sc, sh, NEG)
    if yarg is None:
        pooled, argmax = ops.maxpool(ys[3], B, N, bns[3][0], bns[3][1], NEG)    # BN + LeakyReLU + max over N fused
    ctx = dict(B=B, N=N, x_pm=x_pm, ys=ys, bns=bns, pooled=pooled, argmax=argmax, yarg=yarg, hs=None, training=training)
    if not head:
        return pooled, ctx
    logits, ctx["hs"] = d_head_forward(P, pooled)
    return logits, ctx


def d_forward_groups(P: Dict[str, Tensor], bufs: Optional[Dict[str, Tensor]], xs_cm, update_running: bool = True, pm_shape=None):
    """The conv stacks of several train-mode passes D(x_0), D(x_1), ... as ONE batch: every layer is a single GEMM over the rows of all
    passes (ops.gemm_bn_groups), every pass keeps its own BatchNorm batch statistics, and the running statistics (and call counts)
    advance pass after pass in list order -- the activations, statistics and buffers are those of separate d_forward(head=False) calls
    bit for bit (tests/test_kernels2_gpu.py::test_gemm_bn_groups, test_parity_gpu.py::test_discriminator_grouped_forward...).
    Returns [(pooled_g [B,C4], ctx_g)]: the contexts are VIEWS of the batched tensors, laid out exactly like d_forward's, so d_backward /
    d_double_backward run per pass unchanged.  Needs equal shapes and N % 128 == 0."""
    G = len(xs_cm)
    if pm_shape is not None:
        # pm_shape = (B, N): the inputs are POINT-major [B*N, 3] already (TrainStep's internal route: the generator's output before its
        # layout change, the real cloud as the loader delivers it) -- no [B,3,N] round trip; d_backward then returns dx point-major too
        B, N = pm_shape
        M = B * N
        if any(tuple(x.shape) != (M, 3) for x in xs_cm) or N % ops.ROW_TILE:
            raise ValueError("d_forward_groups(pm_shape=(B,N)) needs inputs [B*N,3] with N %% %d == 0" % ops.ROW_TILE)
        x_pm = torch.cat([x.contiguous() for x in xs_cm], dim=0)
    else:
        B, _, N = xs_cm[0].shape
        M = B * N
        if any(tuple(x.shape) != (B, 3, N) for x in xs_cm) or N % ops.ROW_TILE:
            raise ValueError("d_forward_groups needs equally shaped inputs [B,3,N] with N %% %d == 0" % ops.ROW_TILE)
        x_pm = ops.cm_to_pm(torch.cat([x.contiguous() for x in xs_cm], dim=0))            # [G*M, 3]
    ys_all, outs = [], []
    a, pro = x_pm, None
    for li, (conv, bn) in enumerate(D_LAYERS):
        W, b = _w2(P[conv + ".weight"]), P[conv + ".bias"]
        rm = rv = None
        if bufs is not None and update_running:
            rm, rv = bufs[bn + ".running_mean"], bufs[bn + ".running_var"]
        bnp = (P[bn + ".weight"], P[bn + ".bias"], rm, rv)
        if li == 3:
            out, pooled, argmax, yarg = ops.gemm_bn_groups(a, W, b, bnp, G, pro=pro, rows=N, slope=NEG)
            ys_all.append(None)
        else:
            y, out = ops.gemm_bn_groups(a, W, b, bnp, G, pro=pro)
            ys_all.append(y)
            a, pro = y, (out[0], out[1], NEG)
        outs.append(out)
        if rm is not None:
            for _ in range(G):
                _count_bn_call(bufs, bn)
    res = []
    for g in range(G):
        ys = [None if y is None else y[g * M:(g + 1) * M] for y in ys_all]
        bns = [(o[0, g], o[1, g], o[2, g], o[3, g]) for o in outs]
        sl = slice(g * B, (g + 1) * B)
        ctx = dict(B=B, N=N, x_pm=x_pm[g * M:(g + 1) * M], ys=ys, bns=bns, pooled=pooled[sl], argmax=argmax[sl], yarg=yarg[sl], hs=None, training=True,
                   pm_io=pm_shape is not None)
        res.append((pooled[sl], ctx))
    return res


def d_forward_after_stats_pass(P: Dict[str, Tensor], bufs: Dict[str, Tensor], stats_cm: Tensor, x_cm: Tensor, pm_shape=None):
    """d_advance_running_stats(stats_cm) followed by d_forward(x_cm, head=False) -- the G step's D(real) side effect and D(G(z)) -- with
    the first three layers of the two passes evaluated as ONE batch (per-pass BatchNorm, running statistics real first): three GEMM and
    three finalize launches instead of six and six.  The 1024-wide layer stays separate: the statistics pass replaces it by the
    covariance form (d_advance_running_stats), the real pass runs it with the pooling epilogue.  Bit-identical to the two calls.
    Returns (pooled, ctx) of x_cm."""
    if pm_shape is not None:               # point-major inputs [B*N,3] (see d_forward_groups)
        B, N = pm_shape
        M = B * N
        if tuple(stats_cm.shape) != (M, 3) or tuple(x_cm.shape) != (M, 3) or N % ops.ROW_TILE:
            raise ValueError("d_forward_after_stats_pass(pm_shape=(B,N)) needs inputs [B*N,3] with N %% %d == 0" % ops.ROW_TILE)
        x_pm = torch.cat([stats_cm.contiguous(), x_cm.contiguous()], dim=0)
    else:
        B, _, N = x_cm.shape
        M = B * N
        if tuple(stats_cm.shape) != (B, 3, N) or N % ops.ROW_TILE:
            raise ValueError("d_forward_after_stats_pass needs equally shaped inputs [B,3,N] with N %% %d == 0" % ops.ROW_TILE)
        x_pm = ops.cm_to_pm(torch.cat([stats_cm.contiguous(), x_cm.contiguous()], dim=0))
    ys_all, outs = [], []
    a, pro = x_pm, None
    for conv, bn in D_LAYERS[:3]:
        y, out = ops.gemm_bn_groups(a, _w2(P[conv + ".weight"]), P[conv + ".bias"],
                                    (P[bn + ".weight"], P[bn + ".bias"], bufs[bn + ".running_mean"], bufs[bn + ".running_var"]), 2, pro=pro)
        _count_bn_call(bufs, bn); _count_bn_call(bufs, bn)
        ys_all.append(y); outs.append(out)
        a, pro = y, (out[0], out[1], NEG)
    conv, bn = D_LAYERS[3]
    W, b4 = _w2(P[conv + ".weight"]), P[conv + ".bias"]
    # pass 1: the real forward of the 1024-wide layer with BatchNorm + LeakyReLU + max-pool fused -- its GEMM directly behind the layer
    # that produced its operand (still in the last-level cache); pass 0 (statistics only: fc2.1's running statistics from the
    # covariance of a3, see d_advance_running_stats) is issued between that GEMM and its finalize, so the running statistics still
    # see pass 0 first
    y3 = ys_all[2][M:]
    stats_pass = lambda: _advance_top_running_stats(P, bufs, ys_all[2][:M], outs[2][0, 0], outs[2][1, 0], M)
    _, (sc, sh, inv, mu), pooled, argmax, yarg = ops.gemm_bn_pool(y3, W, b4, (P[bn + ".weight"], P[bn + ".bias"], bufs[bn + ".running_mean"], bufs[bn + ".running_var"]),
                                                                 N, NEG, pro=(outs[2][0, 1], outs[2][1, 1], NEG), before_finalize=stats_pass)
    _count_bn_call(bufs, bn)
    ys = [y[M:] for y in ys_all] + [None]
    bns = [(o[0, 1], o[1, 1], o[2, 1], o[3, 1]) for o in outs] + [(sc, sh, inv, mu)]
    ctx = dict(B=B, N=N, x_pm=x_pm[M:], ys=ys, bns=bns, pooled=pooled, argmax=argmax, yarg=yarg, hs=None, training=True, pm_io=pm_shape is not None)
    return pooled, ctx


def _advance_top_running_stats(P, bufs, y3: Tensor, sc3: Tensor, sh3: Tensor, M: int) -> None:
    """fc2.1's running statistics for a pass whose 1024-wide output is never formed: mean4 = mean(a3).W^T + b4, var4[c] = w_c^T Cov(a3) w_c."""
    conv, bn = D_LAYERS[3]
    W, b4 = _w2(P[conv + ".weight"]), P[conv + ".bias"]
    a3 = ops.affine_act(y3, sc3, sh3, NEG)
    mu_a = ops.colstats(a3, M)[0][0]                    # the column means (one group of M rows), no separate scaling launch
    neg_ones, neg_inv_m = _const_vec(mu_a.numel(), -1.0, mu_a.device), _const_vec(mu_a.numel(), -1.0 / M, mu_a.device)
    cov = ops.rowscale_outer(ops.gemm_tn(a3, a3, pro=(neg_ones, mu_a, 1.0)), neg_inv_m)
    mean4 = ops.gemm_nt(mu_a.view(1, -1), W, b4, exact=True)[0]
    var4 = ops.rowdot(W, ops.gemm_nt(W, cov, exact=True))
    _bn_train(mean4.contiguous(), var4, P, bufs, bn, M, True, True)


def d_advance_running_stats(P: Dict[str, Tensor], bufs: Dict[str, Tensor], x_cm: Tensor) -> None:
    """The side effect of a train-mode D(x) whose logits nobody reads: the G-step of the reference loop calls D(real)
    (model.py:272-273) although gen_loss ignores d_real -- only the four BatchNorm layers' running statistics (and call counts)
    move.  Layers 1-3 run as usual; the statistics of the 1024-wide fc2.0 output follow from the 256 x 256 covariance of its
    input, mean4 = mean(a3).W^T + b4, var4[c] = w_c^T Cov(a3) w_c, so that layer's GEMM (a quarter of a D forward), the pool
    and the MLP head are skipped."""
    B, _, N = x_cm.shape
    M = B * N
    a, pro = ops.cm_to_pm(x_cm), None
    for conv, bn in D_LAYERS[:3]:
        y, (sc, sh, inv, mu) = _gemm_bn(a, _w2(P[conv + ".weight"]), P[conv + ".bias"], P, bufs, bn, M, True, True, pro=pro)
        a, pro = y, (sc, sh, NEG)
    # Cov(a3) = a3^T (a3 - 1 mu^T) / M: the second operand is centred on load (gemm_tn's affine prologue with slope 1), so no
    # Gram/M - mu mu^T difference of two large numbers is formed (a3 is a LeakyReLU output: its means are not small); what
    # rounding leaves of a negative variance is clamped by bn_prepare.  Written with constant vectors only: the prologue forms
    # -(a3 - mu) = a3*(-1) + mu and the row scaling multiplies by -1/M.
    _advance_top_running_stats(P, bufs, a, pro[0], pro[1], M)


def d_backward(P, ctx, dout: Tensor, need_dx: bool, need_dparams: bool, keep_for_double: bool = False, gpool: Optional[Tensor] = None):
    """First-order backward.  Returns (dx_cm | None, {name: grad} | None, saved-for-double-backward | None).
    gpool given (the context came from d_forward(head=False)): `dout` is ignored, the backward starts at the max-pool with the
    gradient w.r.t. the pooled features, and no head gradients are produced."""
    B, N = ctx["B"], ctx["N"]
    M = B * N
    ys, bns, hs, pooled, argmax = ctx["ys"], ctx["bns"], ctx["hs"], ctx["pooled"], ctx["argmax"]
    grads: Dict[str, Tensor] = {}
    dhs = None
    if gpool is None:
        dout = dout.contiguous()
        # ---- MLP head (mlp.6 <- mlp.4 <- mlp.2 <- mlp.0)
        gpool, grads, dhs = d_head_backward(P, pooled, hs, dout, need_dparams)
    else:
        gpool = gpool.contiguous()
    # ---- max-pool + BN4 (sparse incoming gradient)
    sc4, sh4, inv4, mu4 = bns[3]
    y4ref = ys[3] if ys[3] is not None else ctx["yarg"]
    if ctx["training"]:
        # dense [M,1024] BatchNorm backward of a sparse gradient: never materialised, evaluated on the GEMM operand loads; its
        # per-channel coefficients come out of the same launch as the statistics
        gval, sums4, dy = ops.pool_bwd_stats(gpool, pooled, argmax, y4ref, mu4, inv4, NEG, prep=(P["fc2.1.weight"], M, ys[3], N))
    else:
        gval, sums4 = ops.pool_bwd_stats(gpool, pooled, argmax, y4ref, mu4, inv4, NEG)
    C4 = gval.shape[1]
    if need_dparams:
        grads["fc2.1.weight"] = sums4[C4:]; grads["fc2.1.bias"] = sums4[:C4]
    if not ctx["training"]:
        sums4 = torch.zeros_like(sums4)
        dy = ops.bn_bwd_apply_sparse(gval, argmax, ys[3], N, mu4, inv4, P["fc2.1.weight"], sums4, M)
    dys = [None, None, None, dy]
    gs = [None, None, None, None]
    sums_all = [None, None, None, sums4]
    # ---- conv stack fc2.0 <- mlps.6 <- mlps.3 <- mlps.0
    for li in (3, 2, 1, 0):
        conv, bn = D_LAYERS[li]
        W = _w2(P[conv + ".weight"])
        if li == 3 and isinstance(dy, ops.SparseAffine):
            # Collapsed backward of the 256->1024 layer: dz4 = alpha*y4 + beta + S is affine in y4 = a3.W^T + b4, so
            #   dz4.W    = a3.(W^T diag(alpha) W) + (alpha*b4 + beta).W + S.W            [M,256]x[256,256] instead of [M,1024]x[1024,256]
            #   dz4^T.a3 = diag(alpha).W.(a3^T a3) + (alpha*b4 + beta) (x) colsum(a3) + S^T.a3
            # -- a quarter of the FLOPs, and y4 is not read at all.
            sc, sh, inv, mu = bns[2]
            alpha, beta, b4 = dy.alpha, dy.beta, P[conv + ".bias"]
            # a3 = lrelu(bn3(y3)) is never materialised: every consumer applies the affine + LeakyReLU to y3 on its operand load (both
            # sides of the Gram product, whose launch also yields colsum(a3); the sparse rows; the input-gradient GEMM)
            pro3 = (sc, sh, NEG)
            if W.shape[0] % 256 == 0 and W.shape[1] % 32 == 0:
                # W^T diag(alpha) W, (alpha*b4 + beta).W and S.W (dense rows) in one launch: the weight-only part finishes under the streaming one
                ((G4, cvec),), E = ops.collapse_prep(W, [(alpha, beta, b4)], dy.sp_val, dy.sp_arg, N)
            else:
                G4 = ops.gemm_tn(W, ops.rowscale_outer(W, alpha))                     # W^T diag(alpha) W
                cvec = ops.gemm_nt(b4.view(1, -1), _t(W), pro=(alpha, beta, 1.0), exact=True)[0]  # (alpha*b4 + beta).W
                E = ops.sparse_rows_nt(dy.sp_val, dy.sp_arg, N, W)                # S.W, dense rows
            lazy = _lazy_ok(M, sc.numel())
            cb = dict(coef_bn=(P[D_LAYERS[2][1] + ".weight"], M)) if lazy else {}     # the finalize launch also emits the lazy operand's coefficients
            a3 = ops.ActOperand(ys[2], sc, sh, NEG)
            if need_dparams and ops.gemm_dual_ok(a3, G4, ys[2]) and not ops.collapsed_pair_preferred(M, G4.shape[0]):
                # the Gram matrix a3^T a3 (+ colsum(a3)) and the input-gradient product a3.G4 from ONE staging of the y3 tile (ops.gemm_dual
                # with dy := a3, W := G4 -- symmetric --, pre := y3): both read the same tensor with the same BatchNorm + LeakyReLU on load
                gram, g, s0, s1, *rest = ops.gemm_dual(a3, G4, ys[2], sc, sh, mu, inv, NEG, bias=cvec, rowadd=E, with_colsum=True, defer=False, **cb)
                coef, cs3 = rest[:-1], rest[-1]
            else:
                if need_dparams:
                    gram, cs3 = ops.gemm_tn(ys[2], ys[2], a_pro=pro3, pro=pro3, with_colsum=True)      # a3^T a3 [256,256], colsum(a3)
                g, s0, s1, *coef = ops.gemm_nt_bnbwd(ys[2], G4, ys[2], sc, sh, mu, inv, NEG, pro=pro3, bias=cvec, rowadd=E, **cb)
            if need_dparams:
                if ops.wgrad_collapse_ok(W, gram, B):
                    # diag(alpha).W.(a3^T a3) + (alpha*b4 + beta) (x) colsum(a3) + S^T.a3 in one launch
                    dW = ops.wgrad_collapse(W, gram, alpha, b4, beta, cs3, sparse=(dy.sp_val, dy.sp_arg, N, ys[2], pro3))
                else:
                    dW = ops.rowscale_outer(ops.gemm_nt(W, gram, exact=True), alpha, b4, beta, cs3)     # gram: a sum over B*N points
                    ops.sparse_rows_tn(dy.sp_val, dy.sp_arg, N, ys[2], dW, pro=pro3)
                grads[conv + ".weight"] = dW.view_as(P[conv + ".weight"])
                grads[conv + ".bias"] = ZERO_GRAD
            if need_dparams:
                grads[D_LAYERS[2][1] + ".weight"] = s1; grads[D_LAYERS[2][1] + ".bias"] = s0
            sums = _cat2(s0, s1)
            dy = ops.Affine2(g, ys[2], coef[0]) if lazy else _bn_bwd(g, ys[2], mu, inv, P[D_LAYERS[2][1] + ".weight"], sums, M, lazy=False)
            dys[2] = dy; gs[2] = g; sums_all[2] = sums
            continue
        # one launch for the weight gradient AND the masked input gradient of this layer (ops.gemm_dual: the dy tile is staged once)
        fused = need_dparams and li > 0 and ctx["training"] and ops.gemm_dual_ok(dy, W, ys[li - 1])
        if need_dparams and not fused:
            if li > 0:
                grads[conv + ".weight"] = ops.gemm_tn(dy, ys[li - 1], pro=(bns[li - 1][0], bns[li - 1][1], NEG), defer=True).view_as(P[conv + ".weight"])
            else:
                grads[conv + ".weight"] = ops.gemm_tn(dy, ctx["x_pm"], defer=True).view_as(P[conv + ".weight"])
        if need_dparams:
            grads[conv + ".bias"] = ZERO_GRAD     # bias before a train-mode BN: exactly zero gradient
            if not ctx["training"]:
                grads[conv + ".bias"] = ops.colsum(_dense(dy))[0]
        if li > 0:
            pconv, pbn = D_LAYERS[li - 1]
            sc, sh, inv, mu = bns[li - 1]
            # lazy for every layer (coefficients from the finalize launch); layer 1's dy (64 channels) is consumed by the 3-column
            # weight gradient (the streaming kernel takes the two-tensor operand on its wide side) and the 3-column input gradient
            lazy = ctx["training"] and _lazy_ok(M, sc.numel())
            cb = dict(coef_bn=(P[pbn + ".weight"], M)) if lazy else {}
            if fused:
                dW, g, s0, s1, *coef = ops.gemm_dual(dy, W, ys[li - 1], sc, sh, mu, inv, NEG, **cb)
                grads[conv + ".weight"] = dW.view_as(P[conv + ".weight"])
            else:
                g, s0, s1, *coef = ops.gemm_nt_bnbwd(dy, _t(W), ys[li - 1], sc, sh, mu, inv, NEG, **cb)
            if need_dparams:
                grads[pbn + ".weight"] = s1; grads[pbn + ".bias"] = s0
            sums = _cat2(s0, s1) if ctx["training"] else torch.zeros(2 * s0.numel(), device=s0.device)
            dy = ops.Affine2(g, ys[li - 1], coef[0]) if lazy else _bn_bwd(g, ys[li - 1], mu, inv, P[pbn + ".weight"], sums, M)
            dys[li - 1] = dy; gs[li - 1] = g; sums_all[li - 1] = sums
    dx_cm = None
    if need_dx:
        dx_pm = ops.gemm_nt(dy, _t(_w2(P["mlps.0.weight"])))
        dx_cm = dx_pm if ctx.get("pm_io") else ops.pm_to_cm(dx_pm, B, N)        # the layout the input came in
    saved = None
    if keep_for_double:
        saved = dict(dout=dout, dhs=dhs, gval=gval, dys=dys, gs=gs, sums=sums_all)
    return dx_cm, (grads if need_dparams else None), saved


def _d_double_top_phase_a(P, ctx, saved, q3: Tensor, grads) -> dict:
    """Phase A of the double backward at fc2.0/fc2.1 (the layer in front of the max-pool) without [M,1024] tensors.
    q3 [M,256] is the adjoint arriving from layer 3.  With W [1024,256], a3 = lrelu(bn3(y3)), y4 = a3.W^T + b4, u = q3.W^T:
      gy4^T q3 = (alpha*y4 + beta + S)^T q3 = diag(alpha).W.(a3^T q3) + (alpha*b4 + beta) (x) colsum(q3) + S^T q3
      U0 = colsum(q3).W^T,  U1[c] = inv*(w_c^T (q3^T a3) w_c + (b4 - mean)*U0),  Ugz[c] = sum_b gval[b,c]*u[arg[b,c], c]
    and the outgoing adjoint is only needed at the arg-max rows (it is gathered there by the max-pool)."""
    B, N = ctx["B"], ctx["N"]
    M = B * N
    ys, bns, pooled, argmax = ctx["ys"], ctx["bns"], ctx["pooled"], ctx["argmax"]
    conv, bn = D_LAYERS[3]
    W, b4, gamma = _w2(P[conv + ".weight"]), P[conv + ".bias"], P[bn + ".weight"]
    sc, sh, inv, mu = bns[3]
    C = W.shape[0]
    dz = saved["dys"][3]
    S0, S1 = saved["sums"][3][:C], saved["sums"][3][C:]
    pro3 = (bns[2][0], bns[2][1], NEG)                                           # a3 = lrelu(bn3(y3)), applied to y3 on the operand loads
    # q3^T a3 [256,256] and colsum(q3) from one launch; their split sums -- and those of phase A's weight gradients below this layer, deferred
    # by _d_double_phase_a -- are finished by ONE reduction launch
    Qqa, cq = ops.gemm_tn(q3, ys[2], pro=pro3, with_colsum=True, defer=True)
    ops.flush_tn()
    if ops.wgrad_collapse_ok(W, Qqa, B):
        # T = W.(a3^T q3) and dW = diag(alpha).T + (alpha*b4 + beta) (x) colsum(q3) + S^T q3 from one launch
        dW, T = ops.wgrad_collapse(W, Qqa, dz.alpha, b4, dz.beta, cq, sparse=(dz.sp_val, dz.sp_arg, N, q3, None), want_T=True)
    else:
        T = ops.gemm_nt(W, Qqa, exact=True)                                      # W.(a3^T q3) [1024,256]; Qqa: a sum over B*N points
        dW = ops.rowscale_outer(T, dz.alpha, b4, dz.beta, cq)
        ops.sparse_rows_tn(dz.sp_val, dz.sp_arg, N, q3, dW)
    grads[conv + ".weight"] = dW
    # u at the arg-max rows [B,1024], w_c^T (q3^T a3) w_c and U0 = colsum(q3).W^T: three independent dot-product launches as one
    uarg, quad, U0 = ops.dbl_top_dots(q3, argmax, W, T, cq)
    yarg = ctx["yarg"] if ctx.get("yarg") is not None else ops.gather_rows(ys[3], argmax)
    t, spB, c4 = ops.bn_dbl_pool(uarg, saved["gval"], yarg, pooled, U0, quad, b4, mu, inv, gamma, S0, S1, M, NEG)
    return dict(t=t, spB=spB, c4=c4, pro3=pro3, q3=q3, Qqa=Qqa)


def _d_double_top_phase_b(P, ctx, top: dict, grads, phaseb=None, gout=None):
    """Phase B at the same layer: ybar = c1*u + c2*y4 + c3 + scatter(spB) (never formed) ->
      ybar^T a3 = diag(c1).W.(q3^T a3) + diag(c2).W.(a3^T a3) + (c2*b4 + c3) (x) colsum(a3) + spB^T a3
      ybar.W    = q3.(W^T diag(c1) W) + a3.(W^T diag(c2) W) + (c2*b4 + c3).W + spB.W      (then layer 3's mask / sums epilogue)"""
    B, N = ctx["B"], ctx["N"]
    ys, bns, argmax = ctx["ys"], ctx["bns"], ctx["argmax"]
    conv, bn = D_LAYERS[3]
    W, b4 = _w2(P[conv + ".weight"]), P[conv + ".bias"]
    c4, pro3, q3, spB = top["c4"], top["pro3"], top["q3"], top["spB"]
    dgamma, c1, c2, c3 = c4[0], c4[1], c4[2], c4[3]
    grads[bn + ".weight"] = dgamma
    grads[bn + ".bias"] = ZERO_GRAD
    psc, psh, pinv, pmu = bns[2]
    if W.shape[0] % 256 == 0 and W.shape[1] % 32 == 0:
        # W^T diag(c1) W;  W^T diag(c2) W, (c2*b4 + c3).W;  spB.W -- one launch (ops.collapse_prep)
        (G1, (G2, cvec)), EB = ops.collapse_prep(W, [(c1, None, None), (c2, c3, b4)], spB, argmax, N)
    else:
        G1, G2 = ops.gemm_tn(W, ops.rowscale_outer(W, c1)), ops.gemm_tn(W, ops.rowscale_outer(W, c2))
        cvec = ops.gemm_nt(b4.view(1, -1), _t(W), pro=(c2, c3, 1.0), exact=True)[0]
        EB = ops.sparse_rows_nt(spB, argmax, N, W)
    part = ops.gemm_nt(q3, G1, rowbias=EB, rows_per_group=1)
    a3 = ops.ActOperand(ys[2], pro3[0], pro3[1], NEG)
    pair = ops.collapsed_pair_preferred(B * N, G2.shape[0])
    if pair:
        # split-bf16 mode: the two products on the mode's own kernels; phase B's sums / the stored X = xbarA + gamma*g as in the fused launch
        gram, cs3 = ops.gemm_tn(ys[2], ys[2], a_pro=pro3, pro=pro3, with_colsum=True)
        abar_g = ops.gemm_nt_bnbwd(ys[2], G2, ys[2], psc, psh, pmu, pinv, NEG, pro=pro3, bias=cvec, rowadd=part, phaseb=phaseb, gout=gout)
    elif ops.gemm_dual_ok(a3, G2, ys[2]):
        # a3^T a3, colsum(a3) and the outgoing adjoint a3.G2 (+ addends, layer 3's mask / sums epilogue) from one staging of the y3 tile
        # (phaseb: layer 2's phase-B sums come out of this launch's finalize)
        gram, g_, s0_, s1_, cs3, *pb = ops.gemm_dual(a3, G2, ys[2], psc, psh, pmu, pinv, NEG, bias=cvec, rowadd=part, with_colsum=True, defer=False,
                                                     phaseb=phaseb, gout=gout)
        abar_g = (g_, s0_, s1_) + tuple(pb)
    else:
        gram, cs3 = ops.gemm_tn(ys[2], ys[2], a_pro=pro3, pro=pro3, with_colsum=True)            # a3^T a3 and colsum(a3) from one launch
        abar_g = ops.gemm_nt_bnbwd(ys[2], G2, ys[2], psc, psh, pmu, pinv, NEG, pro=pro3, bias=cvec, rowadd=part)
    # the four terms are summed into phase A's part of the gradient
    dW = grads[conv + ".weight"]
    if ops.wgrad_collapse_ok(W, gram, B):
        # diag(c2).W.(a3^T a3) + (c2*b4 + c3) (x) colsum(a3) + diag(c1).W.(q3^T a3) + spB^T a3, accumulated in one launch (q3^T a3 is used transposed)
        ops.wgrad_collapse(W, gram, c2, b4, c3, cs3, X2=top["Qqa"], x2_t=True, a2=c1, sparse=(spB, argmax, N, ys[2], pro3), out=dW, accumulate=True)
    else:
        ops.rowscale_outer(ops.gemm_nt(W, gram, exact=True), c2, b4, c3, cs3, out=dW, accumulate=True)
        ops.gemm_nt(ops.rowscale_outer(W, c1), _t(top["Qqa"]), rowbias=dW, rows_per_group=1, out=dW, exact=True)     # + diag(c1).W.(q3^T a3)
        ops.sparse_rows_tn(spB, argmax, N, ys[2], dW, pro=pro3)
    grads[conv + ".weight"] = dW.view_as(P[conv + ".weight"])
    grads[conv + ".bias"] = ZERO_GRAD
    return abar_g


def d_double_backward(P, ctx, saved, v_dx_cm: Tensor, need_dx: bool = False):
    """Gradient of a scalar R(dx) w.r.t. D's parameters, where dx = d_backward(...)[0] (WGAN-GP:
    Common/gradient_penalty.py:31-35 followed by .backward()).  v = dR/d(dx) [B,3,N].
    Phase A walks the first-order backward graph in reverse (layer 1 -> top), phase B is an ordinary
    backward sweep of the adjoints that phase A deposits on the forward activations (derivation in
    DESIGN.md).  Returns ({name: grad}, dR/dx [B,3,N] | None)."""
    grads, top, coeffs, xbarA = _d_double_phase_a(P, ctx, saved, v_dx_cm)
    return _d_double_phase_b(P, ctx, grads, top, coeffs, xbarA, need_dx)


def _d_double_phase_a(P, ctx, saved, v_dx_cm: Tensor):
    """Phase A of d_double_backward (the first-order backward graph in reverse, layer 1 -> top -> MLP head) -> (grads so far, top, coeffs, xbarA)."""
    B, N = ctx["B"], ctx["N"]
    M = B * N
    ys, bns, hs, pooled, argmax = ctx["ys"], ctx["bns"], ctx["hs"], ctx["pooled"], ctx["argmax"]
    dys, gs, sums_all = saved["dys"], saved["gs"], saved["sums"]
    grads: Dict[str, Tensor] = {}
    q = v_dx_cm.contiguous() if ctx.get("pm_io") else ops.cm_to_pm(v_dx_cm.contiguous())     # adjoint of ga_0 [M,3]
    xbarA: List[Optional[Tensor]] = [None] * 4                   # phase-A adjoint on xhat_l
    coeffs: list = [None] * 4                                    # per-channel phase-A sums (see ops.bn_dbl_coeffs / bn_dbl_phaseb)
    # ---------------------------------------------------------------- phase A
    for li in range(4):
        conv, bn = D_LAYERS[li]
        W = _w2(P[conv + ".weight"])
        sc, sh, inv, mu = bns[li]
        gamma = P[bn + ".weight"]
        C = W.shape[0]
        if li == 3 and isinstance(dys[3], ops.SparseAffine):
            # Collapsed top layer: u = q.W^T, y4 = a3.W^T + b4 and gz (the max-pool scatter) enter only through
            # the K x K matrices q^T a3 / a3^T a3, per-channel sums and their values at the B*C arg-max rows -- no [M,1024]
            # tensor is formed in either phase.
            top = _d_double_top_phase_a(P, ctx, saved, q, grads)
            break
        # gy_l^T q_{l-1}; with the collapsed top layer ahead its split sum waits for that layer's reduction launch (_d_double_top_phase_a)
        grads[conv + ".weight"] = ops.gemm_tn(dys[li], q, defer=isinstance(dys[3], ops.SparseAffine))
        u = ops.gemm_nt(q, W)                                                    # adjoint of gy_l
        if li == 3:
            gz = ops.scatter_rows(saved["gval"], argmax, M)
        else:
            gz = gs[li]
        S0, S1 = sums_all[li][:C], sums_all[li][C:]
        U0, U1, Ugz = ops.bn_dbl_stats(u, ys[li], gz, mu, inv)
        coeffs[li] = (U0, U1, Ugz, S0, S1, M)       # -> [dgammaA | sbarA | sum xbarA | sum xbarA*xhat], formed by phase B's own launch (ops.bn_dbl_phaseb)
        q, xbarA[li] = ops.bn_dbl_apply(u, ys[li], gz, mu, inv, sc, sh, NEG, gamma, S1, U0, U1, M)
    else:
        top = None
    # top: ga_4 = scatter(gpool) -> MLP backward graph in reverse
    t = top["t"] if top is not None else ops.gather_rows(q, argmax)             # adjoint of gpool [B,C4]
    dhs, dout = saved["dhs"], saved["dout"]
    acts = [pooled, hs[0], hs[1], hs[2]]
    for li in (0, 1, 2, 3):
        name = D_MLP[li]
        grads[name + ".weight"] = ops.gemm_tn(dhs[li], t, defer=True)                        # (grad wrt pre-act of layer li)^T . adjoint
        grads[name + ".bias"] = ZERO_GRAD
        if li < 3:
            t = ops.gemm_nt_maskout(t, P[name + ".weight"], hs[li], NEG)
    return grads, top, coeffs, xbarA


LAZY_PHASE_B = [True]     # test / A-B hook: False keeps phase B's BatchNorm backward as a pass of its own (ops.bn_bwd_apply) in front of every layer's launch


def _phaseb_below(P, bns, coeffs, xbarA, li: int, M: int):
    """What the launch that produces layer li-1's adjoint carries for that layer: the phase-B sums' inputs (+ the layer's mean: the finalize then
    also emits the lazy BatchNorm-backward coefficients) and the stored-tile addend X = xbarA + gamma*g.  -> (phaseb, gout) or (None, None)."""
    if li == 0 or not isinstance(coeffs[li - 1], tuple):
        return None, None
    gam = P[D_LAYERS[li - 1][1] + ".weight"]
    sc, sh, inv, mu = bns[li - 1]
    if LAZY_PHASE_B[0] and xbarA[li - 1] is not None and _lazy_ok(M, gam.numel()):
        return (coeffs[li - 1], gam, inv, mu), (xbarA[li - 1], gam)
    return (coeffs[li - 1], gam, inv), None


def _d_double_phase_b(P, ctx, grads, top, coeffs, xbarA, need_dx: bool):
    """Phase B of d_double_backward: an ordinary backward sweep of the adjoints phase A deposited on the forward activations."""
    B, N = ctx["B"], ctx["N"]
    M = B * N
    ys, bns = ctx["ys"], ctx["bns"]
    # ---------------------------------------------------------------- phase B
    abar_g = None      # (abar_l * mask_l) and its column sums, produced by the dgrad epilogue of layer l+1
    dx = None
    for li in (3, 2, 1, 0):
        conv, bn = D_LAYERS[li]
        W = _w2(P[conv + ".weight"])
        sc, sh, inv, mu = bns[li]
        gamma = P[bn + ".weight"]
        C = W.shape[0]
        # the finalize of the launch that produces layer l's adjoint also runs layer l's phase-B sums (ops.gemm_dual(phaseb=...)) and the lazy
        # BatchNorm-backward coefficients that go with them, and the launch itself stores X = xbarA + gamma*g (gout): no pass of its own
        pb_below, gout_below = _phaseb_below(P, bns, coeffs, xbarA, li, M)
        if li == 3 and top is not None:
            abar_g = _d_double_top_phase_b(P, ctx, top, grads, phaseb=pb_below, gout=gout_below)
            continue
        add = None
        if abar_g is not None and len(abar_g) == 6:
            X, s0, s1, sums, grads[bn + ".weight"], coefB = abar_g
            grads[bn + ".bias"] = s0
            ybar = ops.Affine2(X, ys[li], coefB)                                 # inv*(X - S0/M - xhat*S1/M), evaluated on the operand loads
        else:
            if abar_g is None:
                sums, grads[bn + ".weight"] = ops.bn_dbl_phaseb(coeffs[li], gamma, inv, None, None)
                grads[bn + ".bias"] = ZERO_GRAD
            else:
                g, s0, s1 = abar_g[:3]
                add = (g, gamma)                                                     # X = xbarA + gamma*g, formed inside bn_bwd_apply
                if len(abar_g) == 5:
                    sums, grads[bn + ".weight"] = abar_g[3], abar_g[4]
                else:
                    sums, grads[bn + ".weight"] = ops.bn_dbl_phaseb(coeffs[li], gamma, inv, s0, s1)
                grads[bn + ".bias"] = s0
            ybar = ops.bn_bwd_apply(xbarA[li], ys[li], mu, inv, None, sums, M, add=add)
        # phase B's weight-gradient term is accumulated onto phase A's by the split-K reduction itself (beta = 1): no separate add
        gA = grads[conv + ".weight"]
        if li > 0:
            psc, psh, pinv, pmu = bns[li - 1]
            if ops.gemm_dual_ok(ybar, W, ys[li - 1]):
                _, *abar = ops.gemm_dual(ybar, W, ys[li - 1], psc, psh, pmu, pinv, NEG, out=gA, beta=1.0, phaseb=pb_below, gout=gout_below)      # both products of this layer in one launch
                abar_g = tuple(abar)
            else:
                ops.gemm_tn(ybar, ys[li - 1], pro=(psc, psh, NEG), out=gA, beta=1.0)
                abar_g = ops.gemm_nt_bnbwd(ybar, _t(W), ys[li - 1], psc, psh, pmu, pinv, NEG)
        else:
            ops.gemm_tn(ybar, ctx["x_pm"], out=gA, beta=1.0)
            if need_dx:
                dx = ops.gemm_nt(_dense(ybar), _t(W))
                dx = dx if ctx.get("pm_io") else ops.pm_to_cm(dx, B, N)
        grads[conv + ".weight"] = gA.view_as(P[conv + ".weight"])
        grads[conv + ".bias"] = ZERO_GRAD
    return grads, dx


def d_joint_ok(P, ctxs) -> bool:
    """Can d_backward_joint run these passes in lock step?  Train-mode contexts of one shape from the grouped forward (the 1024-wide layer
    collapsed: no stored y4), every layer on its fused launch (ops.gemm_dual / wgrad_collapse / the lazy BatchNorm-backward operand)."""
    if not ctxs or not ops.GROUPED[0]:
        return False
    c0 = ctxs[0]
    B, N = c0["B"], c0["N"]
    M = B * N
    for c in ctxs:
        if not c["training"] or (c["B"], c["N"]) != (B, N) or c["ys"][3] is not None or c.get("yarg") is None:
            return False
    W4 = _w2(P[D_LAYERS[3][0] + ".weight"])
    if W4.shape[0] % 256 or W4.shape[1] % 32 or W4.shape[1] > 256 or B > 64:
        return False
    ys = c0["ys"]
    sc3, sh3 = c0["bns"][2][0], c0["bns"][2][1]
    two = _joint_two_launch()
    if not (two or ops.gemm_dual_ok(ops.ActOperand(ys[2], sc3, sh3, NEG), W4[:W4.shape[1]], ys[2])):       # the collapsed layer: dy := a3, W := a [K,K] matrix
        return False
    for li in (2, 1):
        W = _w2(P[D_LAYERS[li][0] + ".weight"])
        if not (_lazy_ok(M, W.shape[0]) and (two or ops.gemm_dual_ok(ys[li], W, ys[li - 1]))):
            return False
    return _lazy_ok(M, ys[0].shape[1]) and ys[2].shape[1] % 32 == 0


JOINT_TWO_LAUNCH = [os.environ.get("SPGAN_JOINT_PAIRS", "1") != "0"]     # A/B hook: False keeps one node per pass in the "f16" operand mode


def _joint_two_launch() -> bool:
    """The "f16" operand mode has no fused layer-backward kernel (ops.gemm_dual_ok is False for every layer): d_backward_joint then issues each
    layer's two launches per pass and keeps what else it groups (the pool / collapse / weight-gradient-collapse launches of all passes as one)."""
    return JOINT_TWO_LAUNCH[0] and ops.get_mfma_operands() == "f16"


def _layer_backward_multi(specs, defer: bool = True):
    """ops.gemm_dual_multi(specs), or -- where the fused kernel does not take the layer (the "f16" operand mode) -- per problem the two launches it
    stands for: gemm_tn (deferred split sum) + gemm_nt_bnbwd, with the finalize tails computed from their sums.  Same result tuples, except that
    a phase-B problem comes back without the stored-tile / lazy-coefficient tail ((dW, g, s0, s1, sums, dgamma): the caller's bn_bwd_apply route)."""
    d0 = specs[0]
    if ops.gemm_dual_ok(d0["dy"], d0["W"], d0["y_ref"]):
        return ops.gemm_dual_multi(specs, defer=defer)
    res = []
    for sp in specs:
        pro = (sp["scale"], sp["shift"], sp["slope"])
        dW = ops.gemm_tn(sp["dy"], sp["y_ref"], pro=pro, out=sp.get("out"), beta=sp.get("beta", 0.0), defer=True)
        kw = dict(coef_bn=sp["coef_bn"]) if sp.get("coef_bn") is not None else {}
        g, s0, s1, *coef = ops.gemm_nt_bnbwd(sp["dy"], _t(sp["W"]), sp["y_ref"], sp["scale"], sp["shift"], sp["mean"], sp["invstd"], sp["slope"], **kw)
        pb = sp.get("phaseb")
        if pb is not None:
            sums, dgam = ops.bn_dbl_phaseb(pb[0], pb[1], pb[2], s0, s1)
            res.append((dW, g, s0, s1, sums, dgam))
        else:
            res.append((dW, g, s0, s1) + tuple(coef))
    if not defer:
        ops.flush_tn()
    return res


def d_backward_joint(P, firsts, dbl=None):
    """The D step's backward work on the conv stack in LOCK STEP (Generation/Discriminator.py:97-115 is called three times per D step,
    Common/gradient_penalty.py:19-37 differentiates the third call twice): `firsts` = [(ctx, gpool), ...] first-order passes that need parameter
    gradients (the real and the fake pass: d_backward(P, ctx, None, False, True, gpool=gpool)), `dbl` = (ctx, saved, v) the penalty's double
    backward (d_double_backward(P, ctx, saved, v)).  Phase A of the double backward runs first; then every layer's launch is issued ONCE for all
    passes (ops.*_multi: the passes' problems on consecutive workgroup ranges of one grid; the first-order passes meet phase B of the double
    backward at the collapsed 1024-wide layer and walk down together).  Every pass computes exactly what its separate call computes -- the
    results are bit-identical (tests/test_parity_gpu.py::test_d_backward_joint_equals_separate_calls).
    -> ([grads of each first-order pass], grads of the double backward | None)."""
    nf = len(firsts)
    ctxs = [c for c, _ in firsts] + ([dbl[0]] if dbl is not None else [])
    c0 = ctxs[0]
    B, N = c0["B"], c0["N"]
    M = B * N
    conv4, bn4 = D_LAYERS[3]
    W4, b4 = _w2(P[conv4 + ".weight"]), P[conv4 + ".bias"]
    C4 = W4.shape[0]
    G: List[Dict[str, Tensor]] = [dict() for _ in ctxs]
    hot = None
    if dbl is not None:
        hctx, hsaved, v = dbl
        hg, top, coeffs, xbarA = _d_double_phase_a(P, hctx, hsaved, v)
        if top is None:
            raise RuntimeError("d_backward_joint: the double backward's context is not the collapsed one (see d_joint_ok)")
        G[nf] = hg
        c4 = top["c4"]
        hg[bn4 + ".weight"] = c4[0]; hg[bn4 + ".bias"] = ZERO_GRAD
        hot = dict(c1=c4[1], c2=c4[2], c3=c4[3], spB=top["spB"], q3=top["q3"], Qqa=top["Qqa"])
    # ---- max-pool + BN4 of the first-order passes (sparse incoming gradient), one launch
    dys = []
    if nf:
        outs = ops.pool_bwd_stats_multi([dict(gpool=gp.contiguous(), pooled=c["pooled"], argmax=c["argmax"], y=c["yarg"], mean=c["bns"][3][3],
                                              invstd=c["bns"][3][2], slope=NEG, prep=(P[bn4 + ".weight"], M, None, N)) for c, gp in firsts])
        for i, (gval, sums4, dy) in enumerate(outs):
            G[i][bn4 + ".weight"] = sums4[C4:]; G[i][bn4 + ".bias"] = sums4[:C4]
            dys.append(dy)
    # ---- the collapsed 256 -> 1024 layer (see d_backward / _d_double_top_phase_b): its weight-only operands and sparse-row products, one launch
    problems = [(dy.alpha, dy.beta, b4) for dy in dys]
    vals, args_ = [dy.sp_val for dy in dys], [dy.sp_arg for dy in dys]
    if hot is not None:
        problems += [(hot["c1"], None, None), (hot["c2"], hot["c3"], b4)]
        vals.append(hot["spB"]); args_.append(hctx["argmax"])
    outs, Es = ops.collapse_prep(W4, problems, vals, args_, N)
    bn3 = D_LAYERS[2][1]
    specs = []
    for i, (c, _) in enumerate(firsts):
        sc, sh, inv, mu = c["bns"][2]
        G4, cvec = outs[i]
        specs.append(dict(dy=ops.ActOperand(c["ys"][2], sc, sh, NEG), W=G4, y_ref=c["ys"][2], scale=sc, shift=sh, mean=mu, invstd=inv, slope=NEG,
                          bias=cvec, rowadd=Es[i], with_colsum=True, coef_bn=(P[bn3 + ".weight"], M)))
    gout_ok = ops.collapsed_pair_preferred(M, W4.shape[1])      # the split-bf16 pair: its input-gradient launch carries the stored tile (gout)
    pair = gout_ok or not ops.gemm_dual_ok(ops.ActOperand(c0["ys"][2], c0["bns"][2][0], c0["bns"][2][1], NEG), W4[:W4.shape[1]], c0["ys"][2])
    pair_res = []
    if pair:
        # split-bf16 mode: the first-order passes' two products on the mode's own kernels, exactly as d_backward issues them
        for sp in specs:
            x3, pro3 = sp["y_ref"], (sp["scale"], sp["shift"], NEG)
            gram, cs3 = ops.gemm_tn(x3, x3, a_pro=pro3, pro=pro3, with_colsum=True, defer=True)      # (one split reduction for all passes, below)
            g, s0, s1, coef = ops.gemm_nt_bnbwd(x3, sp["W"], x3, sp["scale"], sp["shift"], sp["mean"], sp["invstd"], NEG, pro=pro3, bias=sp["bias"],
                                                rowadd=sp["rowadd"], coef_bn=sp["coef_bn"])
            pair_res.append((gram, g, s0, s1, coef, cs3))
        specs = []
    if hot is not None:
        psc, psh, pinv, pmu = hctx["bns"][2]
        G1, (G2, cvec) = outs[nf], outs[nf + 1]
        part = ops.gemm_nt(hot["q3"], G1, rowbias=Es[nf], rows_per_group=1)
        pb, gout = _phaseb_below(P, hctx["bns"], coeffs, xbarA, 3, M)
        if pair:
            x3, pro3 = hctx["ys"][2], (psc, psh, NEG)
            gram, cs3 = ops.gemm_tn(x3, x3, a_pro=pro3, pro=pro3, with_colsum=True, defer=True)
            if gout_ok:
                g_, s0_, s1_, *pbr = ops.gemm_nt_bnbwd(x3, G2, x3, psc, psh, pmu, pinv, NEG, pro=pro3, bias=cvec, rowadd=part, phaseb=pb, gout=gout)
            else:      # no kernel with the stored-tile epilogue: phase B's sums by their own launch, the adjoint goes through bn_bwd_apply below
                g_, s0_, s1_ = ops.gemm_nt_bnbwd(x3, G2, x3, psc, psh, pmu, pinv, NEG, pro=pro3, bias=cvec, rowadd=part)
                pbr = ops.bn_dbl_phaseb(pb[0], pb[1], pb[2], s0_, s1_) if pb is not None else ()
            pair_res.append((gram, g_, s0_, s1_, cs3) + tuple(pbr))
        else:
            specs.append(dict(dy=ops.ActOperand(hctx["ys"][2], psc, psh, NEG), W=G2, y_ref=hctx["ys"][2], scale=psc, shift=psh, mean=pmu, invstd=pinv,
                              slope=NEG, bias=cvec, rowadd=part, with_colsum=True, phaseb=pb, gout=gout))
    if pair:
        ops.flush_tn()
    res = pair_res + (ops.gemm_dual_multi(specs, defer=False) if specs else [])
    wspecs, lazies, abar_g = [], [], None
    for i, (c, _) in enumerate(firsts):
        gram, g, s0, s1, coef, cs3 = res[i]
        sc, sh = c["bns"][2][0], c["bns"][2][1]
        wspecs.append(dict(W=W4, X1=gram, a1=dys[i].alpha, b1=b4, d1=dys[i].beta, v1=cs3, sparse=(dys[i].sp_val, dys[i].sp_arg, N, c["ys"][2], (sc, sh, NEG))))
        G[i][bn3 + ".weight"] = s1; G[i][bn3 + ".bias"] = s0
        lazies.append(ops.Affine2(g, c["ys"][2], coef))
    if hot is not None:
        gram, g_, s0_, s1_, cs3, *pbr = res[nf]
        abar_g = (g_, s0_, s1_) + tuple(pbr)
        wspecs.append(dict(W=W4, X1=gram, a1=hot["c2"], b1=b4, d1=hot["c3"], v1=cs3, X2=hot["Qqa"], x2_t=True, a2=hot["c1"],
                           sparse=(hot["spB"], hctx["argmax"], N, hctx["ys"][2], (psc, psh, NEG)), out=G[nf][conv4 + ".weight"], accumulate=True))
    dWs = ops.wgrad_collapse_multi(wspecs)
    for i in range(len(ctxs)):
        G[i][conv4 + ".weight"] = dWs[i].view_as(P[conv4 + ".weight"]); G[i][conv4 + ".bias"] = ZERO_GRAD
    # ---- mlps.6 <- mlps.3 <- mlps.0
    for li in (2, 1, 0):
        conv, bn = D_LAYERS[li]
        W = _w2(P[conv + ".weight"])
        ybar = None
        if hot is not None:
            # phase B of layer li: the adjoint X = xbarA + gamma*g through the BatchNorm backward (sums from the finalize launch above)
            sc, sh, inv, mu = hctx["bns"][li]
            g, s0, s1, sums, dgam = abar_g[:5]
            G[nf][bn + ".weight"] = dgam; G[nf][bn + ".bias"] = s0
            if len(abar_g) == 6:      # g is X = xbarA + gamma*g already; its BatchNorm backward is a lazy operand
                ybar = ops.Affine2(g, hctx["ys"][li], abar_g[5])
            else:
                ybar = ops.bn_bwd_apply(xbarA[li], hctx["ys"][li], mu, inv, None, sums, M, add=(g, P[bn + ".weight"]))
        if li == 0:
            tsp = [dict(A=lazies[i], Bm=c["x_pm"]) for i, (c, _) in enumerate(firsts)]
            if hot is not None:
                tsp.append(dict(A=ybar, Bm=hctx["x_pm"], out=G[nf][conv + ".weight"], beta=1.0))
            for i, dW in enumerate(ops.gemm_tn_narrow_multi(tsp)):
                G[i][conv + ".weight"] = dW.view_as(P[conv + ".weight"]); G[i][conv + ".bias"] = ZERO_GRAD
            break
        pbn = D_LAYERS[li - 1][1]
        specs = []
        for i, (c, _) in enumerate(firsts):
            sc, sh, inv, mu = c["bns"][li - 1]
            specs.append(dict(dy=lazies[i], W=W, y_ref=c["ys"][li - 1], scale=sc, shift=sh, mean=mu, invstd=inv, slope=NEG, coef_bn=(P[pbn + ".weight"], M)))
        if hot is not None:
            psc, psh, pinv, pmu = hctx["bns"][li - 1]
            pb, gout = _phaseb_below(P, hctx["bns"], coeffs, xbarA, li, M)
            specs.append(dict(dy=ybar, W=W, y_ref=hctx["ys"][li - 1], scale=psc, shift=psh, mean=pmu, invstd=pinv, slope=NEG, out=G[nf][conv + ".weight"],
                              beta=1.0, phaseb=pb, gout=gout))
        res = _layer_backward_multi(specs)
        for i, (c, _) in enumerate(firsts):
            dW, g, s0, s1, coef = res[i]
            G[i][conv + ".weight"] = dW.view_as(P[conv + ".weight"]); G[i][conv + ".bias"] = ZERO_GRAD
            G[i][pbn + ".weight"] = s1; G[i][pbn + ".bias"] = s0
            lazies[i] = ops.Affine2(g, c["ys"][li - 1], coef)
        if hot is not None:
            dW, *abar = res[nf]
            abar_g = tuple(abar)
            G[nf][conv + ".weight"] = dW.view_as(P[conv + ".weight"]); G[nf][conv + ".bias"] = ZERO_GRAD
    return G[:nf], (G[nf] if dbl is not None else None)


def d_double_backward_eval(P, ctx, saved, v_dx_cm: Tensor, need_dx: bool = False):
    """d_double_backward for a Discriminator in eval() mode (Common/gradient_penalty.py:28-33 works in either mode).  With running
    statistics every BatchNorm is a fixed per-channel affine, so the first-order backward is gy_l = sc_l * gz_l, gz_l = ga_l * m_l,
    ga_{l-1} = gy_l.W_l with piecewise-constant masks: R(dx) reaches the parameters only through phase A (the reverse of the backward
    graph) -- nothing is deposited on the forward activations, the biases and BatchNorm shifts get exactly zero, and dR/dx = 0 almost
    everywhere (D is piecewise linear in x).
      u_l = q_{l-1}.W_l^T;  Wbar_l = gy_l^T q_{l-1};  gammabar_l = invstd_l * sum(u_l*gz_l);  q_l = sc_l * u_l * m_l."""
    B, N = ctx["B"], ctx["N"]
    M = B * N
    ys, bns, hs, pooled, argmax = ctx["ys"], ctx["bns"], ctx["hs"], ctx["pooled"], ctx["argmax"]
    dys, gs = saved["dys"], saved["gs"]
    grads: Dict[str, Tensor] = {}
    q = v_dx_cm.contiguous() if ctx.get("pm_io") else ops.cm_to_pm(v_dx_cm.contiguous())
    for li in range(4):
        conv, bn = D_LAYERS[li]
        W = _w2(P[conv + ".weight"])
        sc, sh, inv, mu = bns[li]
        C = W.shape[0]
        grads[conv + ".weight"] = ops.gemm_tn(_dense(dys[li]), q).view_as(P[conv + ".weight"])
        grads[conv + ".bias"] = ZERO_GRAD
        u = ops.gemm_nt(q, W)
        gz = ops.scatter_rows(saved["gval"], argmax, M) if li == 3 else gs[li]
        _, _, Ugz = ops.bn_dbl_stats(u, ys[li], gz, mu, inv)
        grads[bn + ".weight"] = Ugz * inv                     # O(C) per-channel scalar formula
        grads[bn + ".bias"] = ZERO_GRAD
        zero = _const_vec(C, 0.0, u.device)
        q, _ = ops.bn_dbl_apply(u, ys[li], gz, mu, inv, sc, sh, NEG, P[bn + ".weight"], zero, zero, zero, M)
    t = ops.gather_rows(q, argmax)
    dhs = saved["dhs"]
    for li in (0, 1, 2, 3):
        name = D_MLP[li]
        grads[name + ".weight"] = ops.gemm_tn(dhs[li], t, defer=True)
        grads[name + ".bias"] = ZERO_GRAD
        if li < 3:
            t = ops.gemm_nt_maskout(t, P[name + ".weight"], hs[li], NEG)
    dx = torch.zeros((B * N, 3) if ctx.get("pm_io") else (B, 3, N), dtype=torch.float32, device=q.device) if need_dx else None
    return grads, dx


# =============================================================================================
# EdgeBlock (Generation/Generator.py:47-88), point-major, restructured per point
# =============================================================================================
_WO_CACHE: Dict[int, tuple] = {}      # data_ptr -> (stamp, permuted weight, weakref to the owning Parameter)


def conv_out_weight_pm(w: Tensor) -> Tuple[Tensor, Tensor]:
    """conv_out.weight [F,F,1,k] -> (Wo [F, k*F] with K index r*F + c (matches T's layout), Wo^T) from one launch
    (ops.conv_out_weight_pm): the forward product's operand and the one of the input gradient dT = dout . Wo.  For a Parameter the pair
    is kept until the weights change (both generator forwards of a train step see the same weights: nets._t's staleness rule)."""
    owner = _owner(w)
    if owner is None:
        return ops.conv_out_weight_pm(w.contiguous())
    stamp = (ops.weights_epoch_of(owner), owner._version)          # this network's own optimiser steps, not the other's
    hit = _WO_CACHE.get(w.data_ptr())
    if hit is not None and hit[0] == stamp and hit[2]() is owner:
        return hit[1]
    out = ops.conv_out_weight_pm(w.detach().contiguous())
    _WO_CACHE[w.data_ptr()] = (stamp, out, weakref.ref(owner))
    return out


class CatCols:
    """A weight gradient computed as column blocks [rows, c_i] of one [rows, sum c_i] matrix, not yet concatenated:
    functions._deliver adds every block into its column range of the parameter's .grad (one fused launch for all gradients of the
    pass) instead of a torch.cat launch followed by the accumulation; cat() materialises it for autograd's own accumulation."""

    def __init__(self, parts):
        self.parts = [p_ if p_.dim() == 2 else p_.reshape(p_.shape[0], -1) for p_ in parts]
        self.rows = self.parts[0].shape[0]

    def numel(self) -> int:
        return sum(p_.numel() for p_ in self.parts)

    def cat(self) -> Tensor:
        return torch.cat(self.parts, dim=1)


def drop_weight_caches() -> None:
    """Forget every cached weight-derived tensor (transposes, permuted conv_out weights).  TrainStep calls this right before it
    captures a step into a hipGraph: a cache entry filled by an EAGER call after the last optimiser step (a sample dump, an eval
    forward) would be a hit during the capture, the kernel that derives it would not be recorded, and every replay would read a
    tensor frozen at capture time that lives outside the graph's memory pool."""
    for key, (stamp, t, oref, view) in list(_T_CACHE.items()):
        _T_CACHE[key] = ((-1, -1), t, oref, view)          # stale, not forgotten: the capture's first use re-transposes the whole
    _WO_CACHE.clear()                                      # network's matrices in ONE launch (_refresh_transposes), as every step does
    _COLS_CACHE.clear()


def conv_out_weight_grad_from_pm(g: Tensor, F_: int, k: int) -> Tensor:
    """[F, k*F] (the GEMM's layout) -> the parameter's [F,F,1,k] as a permuted VIEW: functions._deliver adds it into .grad through its
    strides (ops.multi_add), autograd's own accumulation takes it as it is -- no copy launch."""
    return g.view(F_, k, F_).permute(0, 2, 1).unsqueeze(2)


def edgeblock_forward(P, bufs, pre: str, x: Tensor, idx: Tensor, B: int, N: int, training: bool = True, update_running: bool = True,
                      count_rep: int = 1, bn_repeats: int = 1, out: Optional[Tensor] = None):
    """out: the [M,F] tensor (e.g. a row block of a larger buffer) the block's result is written into."""
    if bn_repeats > 1 and training:
        # this forward stands for bn_repeats identical ones (same input, same weights: the two generator forwards of a train step see
        # the same sphere prior): one evaluation, the running statistics advanced bn_repeats times with the same batch statistics
        with ops.bn_momentum(1.0 - (1.0 - ops.BN_MOMENTUM) ** bn_repeats):
            out, ctx = edgeblock_forward(P, bufs, pre, x, idx, B, N, training, update_running, count_rep, 1, out=out)
        if bufs is not None and update_running:
            for _ in range(bn_repeats - 1):
                for bn in (".conv_w.1", ".conv_x.1", ".conv_w.4"):
                    _count_bn_call(bufs, pre + bn)
        return out, ctx
    return _edgeblock_forward(P, bufs, pre, x, idx, B, N, training, update_running, count_rep, out=out)


def _edgeblock_forward(P, bufs, pre: str, x: Tensor, idx: Tensor, B: int, N: int, training: bool = True, update_running: bool = True,
                       count_rep: int = 1, out: Optional[Tensor] = None):
    """x [M,C] (point-major), idx int32 [M,k] -> out [M,F] + ctx.
    count_rep > 1: x stands for count_rep identical copies of these M rows (the tiled sphere prior): batch statistics are those
    of one copy, only the unbiased-variance count of the running statistics is count_rep times larger."""
    M, C = x.shape
    k = idx.shape[1]
    Ww0 = P[pre + ".conv_w.0.weight"]; Wx = P[pre + ".conv_x.0.weight"]
    H, F_ = Ww0.shape[0], Wx.shape[0]
    b1, bx = P[pre + ".conv_w.0.bias"], P[pre + ".conv_x.0.bias"]
    Wcat, WcatT = ops.edge_wcat(_w2(Ww0), _w2(Wx), transposed=True)               # Wcat^T: the input gradient's operand, same launch
    PQR = ops.gemm_nt(x, Wcat)                                                   # [M, H+2F]
    E = M * k
    if training:
        run = lambda n: (bufs[n + ".running_mean"], bufs[n + ".running_var"]) if (bufs is not None and update_running) else (None, None)
        bn1, bnx = ops.edge_stats_bn(PQR, idx, b1, bx,
                                     (P[pre + ".conv_w.1.weight"], P[pre + ".conv_w.1.bias"]) + run(pre + ".conv_w.1"),
                                     (P[pre + ".conv_x.1.weight"], P[pre + ".conv_x.1.bias"]) + run(pre + ".conv_x.1"), count_rep)
        if bufs is not None and update_running:
            _count_bn_call(bufs, pre + ".conv_w.1"); _count_bn_call(bufs, pre + ".conv_x.1")
    else:
        bn1 = _bn_train(None, None, P, bufs, pre + ".conv_w.1", E, False, False)
        bnx = _bn_train(None, None, P, bufs, pre + ".conv_x.1", E, False, False)
    W2, b2 = _w2(P[pre + ".conv_w.3.weight"]), P[pre + ".conv_w.3.bias"]
    # "f16" operand mode at full size: the per-edge tensors live in HBM in 16 bits -- h2pre and T as float16 (activations), dT, g2 and gy as
    # bfloat16 (gradients); BatchNorm statistics and every sum stay fp32 (taken from the accumulators / the unrounded values)
    s16 = ops.storage16(E, F_, k) and _lazy_ok(E, F_)
    h2pre, bn2 = _gemm_bn(PQR[:, :H], W2, b2, P, bufs, pre + ".conv_w.4", E, training, update_running, pro=(bn1[0], bn1[1], NEG), edge=(idx, b1),
                          **({"count_rep": count_rep} if training and count_rep > 1 else {}), **({"out_half": True} if s16 else {}))
    T = ops.edge_attend_fwd(h2pre, bn2[0], bn2[1], PQR, idx, bx, bnx[0], bnx[1], NEG, half=s16)
    Wo, WoT = conv_out_weight_pm(P[pre + ".conv_out.weight"])
    out = ops.gemm_nt(T, Wo, P[pre + ".conv_out.bias"], out=out)
    ctx = dict(x=x, idx=idx, B=B, N=N, PQR=PQR, Wcat=Wcat, WcatT=WcatT, WoT=WoT, bn1=bn1, bnx=bnx, bn2=bn2, h2pre=h2pre, T=T, Wo=Wo, H=H, F=F_, k=k, training=training)
    return out, ctx


def edgeblock_backward(P, pre: str, ctx, dout: Tensor, csr: Tuple[Tensor, Tensor], need_dx: bool = True):
    """-> (dx [M,C] | None, {name: grad})"""
    x, idx, PQR, H, F_, k = ctx["x"], ctx["idx"], ctx["PQR"], ctx["H"], ctx["F"], ctx["k"]
    M = x.shape[0]
    E = M * k
    bn1, bnx, bn2 = ctx["bn1"], ctx["bnx"], ctx["bn2"]
    b1, bx = P[pre + ".conv_w.0.bias"], P[pre + ".conv_x.0.bias"]
    g: Dict[str, Tensor] = {}
    dout = dout.contiguous()
    # conv_out
    gwo, g[pre + ".conv_out.bias"] = ops.gemm_tn(dout, ctx["T"], with_colsum=True, defer=True)      # the bias gradient rides along (column sums of dout); both split sums wait for the pass's one reduction launch
    g[pre + ".conv_out.weight"] = conv_out_weight_grad_from_pm(gwo, F_, k)
    dT = ops.gemm_nt(dout, ctx["WoT"], out_bf16=ctx["T"].dtype == torch.float16)     # [M, k*F]
    # softmax * conv_x product, both LeakyReLUs
    g2, gy, sums2, sumsy = ops.edge_attend_bwd(dT, ctx["h2pre"], bn2[0], bn2[1], bn2[3], bn2[2], PQR, idx, bx, bnx[0], bnx[1], bnx[3], bnx[2], NEG)
    g[pre + ".conv_w.4.weight"] = sums2[F_:]; g[pre + ".conv_w.4.bias"] = sums2[:F_]
    g[pre + ".conv_x.1.weight"] = sumsy[F_:]; g[pre + ".conv_x.1.bias"] = sumsy[:F_]
    if not ctx["training"]:
        sums2 = torch.zeros_like(sums2); sumsy = torch.zeros_like(sumsy)
    # conv_w.4 BN backward -> conv_w.3
    dh2 = _bn_bwd(g2, ctx["h2pre"], bn2[3], bn2[2], P[pre + ".conv_w.4.weight"], sums2, E)     # lazy: consumed by the two edge GEMMs below
    W2 = _w2(P[pre + ".conv_w.3.weight"])
    if ctx["training"] and ops.gemm_dual_ok(dh2, W2, PQR[:, :H], edge=(idx, b1)):
        # conv_w.3's weight gradient and its masked input gradient from ONE staging of the dh2 tile (csrc/gemm_dual.hip): the two [E,F]
        # tensors behind the lazy operand are read once instead of twice
        dW3, g1, s10, s11 = ops.gemm_dual(dh2, W2, PQR[:, :H], bn1[0], bn1[1], bn1[3], bn1[2], NEG, edge=(idx, b1))
        g[pre + ".conv_w.3.weight"] = dW3.view_as(P[pre + ".conv_w.3.weight"])
    else:
        g[pre + ".conv_w.3.weight"] = ops.gemm_tn(dh2, PQR[:, :H], pro=(bn1[0], bn1[1], NEG), edge=(idx, b1), defer=True).view_as(P[pre + ".conv_w.3.weight"])
        g1, s10, s11 = ops.gemm_nt_bnbwd(dh2, _t(W2), PQR[:, :H], bn1[0], bn1[1], bn1[3], bn1[2], NEG, edge=(idx, b1))
    g[pre + ".conv_w.3.bias"] = ZERO_GRAD if ctx["training"] else ops.colsum(_dense(dh2))[0]
    g[pre + ".conv_w.1.weight"] = s11; g[pre + ".conv_w.1.bias"] = s10
    sums1 = _cat2(s10, s11) if ctx["training"] else torch.zeros(2 * H, device=x.device)
    # BN backward of conv_w.0 / conv_x.0 outputs fused with the edge -> point reduction
    dPQR = ops.edge_scatter(g1, gy, PQR, idx, csr[0], csr[1], b1, bn1[3], bn1[2], P[pre + ".conv_w.1.weight"], sums1,
                            bx, bnx[3], bnx[2], P[pre + ".conv_x.1.weight"], sumsy)
    dWcat = ops.gemm_tn(dPQR, x)
    dW0, dWx = ops.edge_wcat_bwd(dWcat, H, F_)
    g[pre + ".conv_w.0.weight"] = dW0.view_as(P[pre + ".conv_w.0.weight"])
    g[pre + ".conv_x.0.weight"] = dWx.view_as(P[pre + ".conv_x.0.weight"])
    # biases in front of a train-mode BatchNorm: mathematically zero gradient
    g[pre + ".conv_w.0.bias"] = ZERO_GRAD
    g[pre + ".conv_x.0.bias"] = ZERO_GRAD
    dx = ops.gemm_nt(dPQR, ctx["WcatT"]) if need_dx else None
    return dx, g


# =============================================================================================
# AdaptivePointNorm (Generator.py:24-45) with the preceding LeakyReLU(0.2) optionally fused (slope)
# =============================================================================================
def adain_forward(P, pre: str, x: Tensor, style: Tensor, N: int, slope: float = 1.0):
    Ws, bs = _w2(P[pre + ".style.weight"]), P[pre + ".style.bias"]
    gb = ops.gemm_nt(style, Ws, bs)                                              # [M, 2C] = [gamma | beta]
    imean, ivar = ops.colstats(x, N, slope)
    imean = imean.contiguous(); ivar = ivar.contiguous()
    out = ops.adain_fwd(x, N, slope, imean, ivar, gb)
    return out, dict(x=x, style=style, gb=gb, imean=imean, ivar=ivar, N=N, slope=slope)


def adain_backward(P, pre: str, ctx, dout: Tensor, need_dx: bool = True, need_dstyle: bool = True, need_dparams: bool = True,
                   dstyle_addend: Optional[Tensor] = None):
    dx, dgb = ops.adain_bwd(dout.contiguous(), ctx["x"], ctx["N"], ctx["slope"], ctx["imean"], ctx["ivar"], ctx["gb"])
    g = {}
    if need_dparams:
        gws, g[pre + ".style.bias"] = ops.gemm_tn(dgb, ctx["style"], defer=True, with_colsum=True)
        g[pre + ".style.weight"] = gws.view_as(P[pre + ".style.weight"])
    dstyle = None
    if need_dstyle:      # dstyle_addend: the style gradient of another AdaIN layer on the same style tensor, added in the GEMM's epilogue
        kw = dict(rowbias=dstyle_addend.contiguous(), rows_per_group=1) if dstyle_addend is not None else {}
        dstyle = ops.gemm_nt(dgb, _t(_w2(P[pre + ".style.weight"])), **kw)
    return (dx if need_dx else None), dstyle, g


# =============================================================================================
# point-wise MLP chains (Generator head / tail) and the global feature branch
# =============================================================================================
def mlp_forward(P, names: List[str], acts: List[int], x: Tensor, slope: float = NEG, rowbias: Optional[Tensor] = None, N: int = 0,
                first_weight: Optional[Tensor] = None):
    """Chain of 1x1 convs with fused bias + activation.  `first_weight` overrides the first layer's
    weight view (tail.0 uses only the last 128 input columns; the global part enters as `rowbias`)."""
    hs = []
    h = x
    for i, (n, a) in enumerate(zip(names, acts)):
        W = first_weight if (i == 0 and first_weight is not None) else _w2(P[n + ".weight"])
        if i == 0 and rowbias is not None:
            h = ops.gemm_nt(h, W, None, rowbias=rowbias, rows_per_group=N, act=a, slope=slope)
        else:
            h = ops.gemm_nt(h, W, P[n + ".bias"], act=a, slope=slope)
        hs.append(h)
    return h, dict(x=x, hs=hs, names=names, acts=acts, slope=slope, first_weight=first_weight, rowbias=rowbias is not None, N=N)


def mlp_backward(P, ctx, dout: Tensor, need_dx: bool = True, need_dparams: bool = True):
    """-> (dx | None, {name: grad}, drowbias | None).  For the first layer with `first_weight`, the weight
    gradient returned under key names[0]+'.weight.part' covers only those columns."""
    names, acts, hs, slope = ctx["names"], ctx["acts"], ctx["hs"], ctx["slope"]
    g: Dict[str, Tensor] = {}
    d = ops.act_bwd(dout.contiguous(), hs[-1], acts[-1], slope)                  # gradient w.r.t. the last pre-activation
    drb = None
    for i in range(len(names) - 1, -1, -1):
        inp = hs[i - 1] if i > 0 else ctx["x"]
        first_part = i == 0 and ctx["first_weight"] is not None
        W = ctx["first_weight"] if first_part else _w2(P[names[i] + ".weight"])
        if need_dparams:
            per_shape_bias = i == 0 and ctx["rowbias"]
            if per_shape_bias:
                gw = ops.gemm_tn(d, inp, defer=True)
            else:                                                                # the bias gradient = column sums of d: a by-product of the product
                gw, g[names[i] + ".bias"] = ops.gemm_tn(d, inp, defer=True, with_colsum=True)
            if first_part:
                g[names[i] + ".weight.part"] = gw
            else:
                g[names[i] + ".weight"] = gw.view_as(P[names[i] + ".weight"])
            if per_shape_bias:
                drb = ops.colsum(d, ctx["N"])                                    # per-shape bias gradient [B, C]
        elif i == 0 and ctx["rowbias"]:
            drb = ops.colsum(d, ctx["N"])
        if i > 0:
            if acts[i - 1] == ops.ACT_LRELU:
                d = ops.gemm_nt_maskout(d, _t(W), hs[i - 1], slope)
            else:
                d = ops.act_bwd(ops.gemm_nt(d, _t(W)), hs[i - 1], acts[i - 1], slope)
        elif need_dx:
            d = ops.gemm_nt(d, _t(W))
        else:
            d = None
    return d, g, drb


def global_feat_forward(P, bufs, a2: Tensor, B: int, N: int, training: bool = True, update_running: bool = True):
    """cat[global feature repeated over N, a2] materialised, [M, 512+C] (Generator.py:183-189) -- only the --attn variant needs
    it as a tensor; the default path folds the global half into a per-shape bias of tail.0."""
    gctx = global_forward(P, bufs, a2, B, N, training, update_running)
    gact = ops.affine_act(gctx["y3"], gctx["bn3"][0], gctx["bn3"][1], NEG)                       # [B,512]
    Cg = gact.shape[1]
    feat = torch.empty((B * N, Cg + a2.shape[1]), dtype=torch.float32, device=a2.device)
    feat[:, :Cg].view(B, N, Cg).copy_(gact.view(B, 1, Cg).expand(B, N, Cg))
    feat[:, Cg:].copy_(a2)
    gctx["Cg"] = Cg
    return feat, gctx


def global_feat_backward(P, gctx, dfeat: Tensor):
    """-> (da2, {name: grad}) for global_feat_forward."""
    Cg, N = gctx["Cg"], gctx["N"]
    da2 = dfeat[:, Cg:].contiguous()
    dg = ops.colsum(dfeat[:, :Cg], N)                                                            # [B,512]: sum over the shape's points
    return da2, global_backward(P, gctx, None, dg, da2)


def g_pair_forward(Ph, Pa1, Pe2, bufs_e2, Pa2, Pgt, bufs_g, x_pm2: Tensor, zb2: Tensor, x1_one: Tensor, B: int, N: int, k: int, slope: float,
                   training: bool = True, idx2=None):
    """The two generator forwards of one train step -- G(x, z_d) of the D step (model.py:246-248, no gradient) and G(x, z_g) of the G step
    (model.py:264-271) -- behind their shared EdgeConv1 as ONE pipeline over the rows of both passes [2*B*N, C]: they see the same prior and
    the same weights, and the second depends on nothing the D step produces.  Everything that acts per point or per shape (the style head,
    both AdaIN layers, the tail) runs once on 2*B shapes; everything with BatchNorm batch statistics (EdgeConv2, global_conv) runs per pass
    on its row block, pass 0 first (running statistics advance in the reference's order).  Every row's values are those of the separate
    forwards.  x_pm2 [2M,3] the prior's rows twice, zb2 [2B,nz] = [z_d; z_g] (one latent per shape), x1_one [N,C1] EdgeConv1's output for ONE copy of the
    prior; idx2: optional (graph of pass 0, graph of pass 1) for EdgeConv2.
    -> dict(fake_d [M,3] (pass 0's cloud), and for pass 1 the (output, saved context) pairs the autograd Functions adopt:
       head, x1, adain1, ec2, adain2, tail)."""
    M = B * N
    W0 = _w2(Ph["head.0.weight"])
    c = x_pm2.shape[1]
    rb = ops.gemm_nt(zb2.contiguous(), _cols_from(W0, c), Ph["head.0.bias"])                           # [2B,128]: the latent half of head.0
    style, mh = mlp_forward(Ph, ["head.0", "head.2"], [ops.ACT_LRELU, ops.ACT_LRELU], x_pm2, NEG, rowbias=rb, N=N, first_weight=W0[:, :c])
    x1 = x1_one.repeat(2 * B, 1)                                                                       # the same EdgeConv1 rows for every shape of both passes
    a1, c1 = adain_forward(Pa1, "a", x1, style, N, slope)
    F_ = _w2(Pe2["e.conv_x.0.weight"]).shape[0]
    x2 = torch.empty((2 * M, F_), dtype=torch.float32, device=x_pm2.device)
    ec = []
    for p_ in (0, 1):                                                                                  # BatchNorm batch statistics: per pass
        rows = slice(p_ * M, (p_ + 1) * M)
        xin = a1[rows]
        idx = idx2[p_] if (idx2 is not None and idx2[p_] is not None) else ops.knn(xin, B, N, k, 0)
        ec.append(edgeblock_forward(Pe2, bufs_e2, "e", xin, idx, B, N, training, True, out=x2[rows]))
    a2, c2 = adain_forward(Pa2, "a", x2, style, N, slope)
    Wt0 = Pgt["tail.0.weight"].view(Pgt["tail.0.weight"].shape[0], -1)
    gcs = [global_forward(Pgt, bufs_g, a2[p_ * M:(p_ + 1) * M], B, N, training, True) for p_ in (0, 1)]
    Cg = gcs[0]["y3"].shape[1]
    rb2 = torch.empty((2 * B, Wt0.shape[0]), dtype=torch.float32, device=x_pm2.device)
    for p_ in (0, 1):
        ops.gemm_nt(gcs[p_]["y3"], Wt0[:, :Cg], Pgt["tail.0.bias"], pro=(gcs[p_]["bn3"][0], gcs[p_]["bn3"][1], NEG), out=rb2[p_ * B:(p_ + 1) * B])
    out, mt = mlp_forward(Pgt, ["tail.0", "tail.2", "tail.4"], [ops.ACT_LRELU, ops.ACT_LRELU, ops.ACT_TANH], a2, NEG, rowbias=rb2, N=N,
                          first_weight=Wt0[:, Cg:])

    def half_mlp(m):
        return dict(m, x=m["x"][M:], hs=[h[M:] for h in m["hs"]])

    def half_adain(cx):
        return dict(cx, x=cx["x"][M:], style=cx["style"][M:], gb=cx["gb"][M:], imean=cx["imean"][B:], ivar=cx["ivar"][B:])
    return dict(fake_d=out[:M], head=(style[M:], half_mlp(mh)), x1=x1[M:], adain1=(a1[M:], half_adain(c1)), ec2=ec[1],
                adain2=(a2[M:], half_adain(c2)), tail=(out[M:], gcs[1], half_mlp(mt)), idx=(ec[0][1]["idx"], ec[1][1]["idx"]),
                x1_in=a1[M:], a2=a2[M:])


def attention_forward(P, pre: str, x: Tensor, B: int, N: int):
    """`Attention.forward` (Generation/modules.py:549-558) on point-major rows: x [M,C] -> gamma * W_o(beta g) + x with
    beta_b = softmax_rows(theta_b phi_b^T) per shape.  The [B,N,N] maps are materialised (537 MB at B=32, N=2048: they are kept
    for the backward pass anyway); the per-shape products are ONE batched gemm_nt launch each (blockIdx.y = shape)."""
    Wt, Wp, Wg, Wo = (_w2(P[pre + n + ".weight"]) for n in (".theta", ".phi", ".g", ".o"))
    M = x.shape[0]
    theta, phi, gv = ops.gemm_nt(x, Wt), ops.gemm_nt(x, Wp), ops.gemm_nt(x, Wg)                  # [M,C/8], [M,C/8], [M,C/2]
    c8, c2 = theta.shape[1], gv.shape[1]
    beta = ops.gemm_nt_batched(theta.view(B, N, c8), phi.view(B, N, c8))                         # [B,N,N] scores
    ops.softmax_rows(beta)
    o = ops.gemm_nt_batched(beta, ops.pm_to_cm(gv, B, N)).view(M, c2)                            # beta_b . g_b
    oo = ops.gemm_nt(o, Wo)                                                                      # [M,C]
    y = ops.scale_residual(oo, x, P[pre + ".gamma"])
    return y, dict(x=x, theta=theta, phi=phi, g=gv, beta=beta, o=o, oo=oo, B=B, N=N)


def attention_backward(P, pre: str, ctx, dy: Tensor, need_dx: bool = True):
    """-> (dx | None, {name: grad}).  dS = softmax'(dP), d theta = dS phi, d phi = dS^T theta, d g = beta^T d_o per shape; the two
    transposed products read explicitly transposed maps, so that all five are batched A.W^T launches."""
    Wt, Wp, Wg, Wo = (_w2(P[pre + n + ".weight"]) for n in (".theta", ".phi", ".g", ".o"))
    x, theta, phi, gv, beta, o, B, N = (ctx[k] for k in ("x", "theta", "phi", "g", "beta", "o", "B", "N"))
    dy = dy.contiguous()
    g: Dict[str, Tensor] = {}
    d_oo, g[pre + ".gamma"] = ops.scale_residual_bwd(dy, ctx["oo"], P[pre + ".gamma"])
    g[pre + ".o.weight"] = ops.gemm_tn(d_oo, o, defer=True).view_as(P[pre + ".o.weight"])
    d_o = ops.gemm_nt(d_oo, _t(Wo))                                                              # [M,C/2]
    c8, c2 = theta.shape[1], gv.shape[1]
    dcat = torch.empty((x.shape[0], 2 * c8 + c2), dtype=torch.float32, device=x.device)         # [d theta | d phi | d g]
    dcat3 = dcat.view(B, N, 2 * c8 + c2)
    dS = ops.gemm_nt_batched(d_o.view(B, N, c2), gv.view(B, N, c2))                              # dP [B,N,N]
    ops.softmax_rows_bwd(beta, dS)
    ops.gemm_nt_batched(dS, ops.pm_to_cm(phi, B, N), out=dcat3[:, :, :c8])
    dST = ops.pm_to_cm(dS.view(B * N, N), B, N)
    del dS
    ops.gemm_nt_batched(dST, ops.pm_to_cm(theta, B, N), out=dcat3[:, :, c8:2 * c8])
    del dST
    ops.gemm_nt_batched(ops.pm_to_cm(beta.view(B * N, N), B, N), ops.pm_to_cm(d_o, B, N), out=dcat3[:, :, 2 * c8:])
    dW = ops.gemm_tn(dcat, x, defer=True)                                                                    # [C/8+C/8+C/2, C]
    for n, lo, hi in ((".theta", 0, c8), (".phi", c8, 2 * c8), (".g", 2 * c8, 2 * c8 + c2)):
        g[pre + n + ".weight"] = dW[lo:hi].reshape(P[pre + n + ".weight"].shape)
    dx = None
    if need_dx:
        dx = ops.gemm_nt(dcat, _t(torch.cat([Wt, Wp, Wg], 0)))
        ops.multi_add([dx], [dy])
    return dx, g


def global_forward(P, bufs, a2: Tensor, B: int, N: int, training: bool = True, update_running: bool = True):
    """max over N -> Linear+BN1d+LReLU -> Linear+BN1d+LReLU (Generator.py:119-126,183-186).  Returns the
    second pre-BN tensor and its BN affine (the activation is applied by the consumer's prologue)."""
    gmax, garg = ops.maxpool(a2, B, N)
    W0, b0 = P["global_conv.0.weight"], P["global_conv.0.bias"]
    W3, b3 = P["global_conv.3.weight"], P["global_conv.3.bias"]
    y0, bn0 = _gemm_bn(gmax, W0, b0, P, bufs, "global_conv.1", B, training, update_running)
    y3, bn3 = _gemm_bn(y0, W3, b3, P, bufs, "global_conv.4", B, training, update_running, pro=(bn0[0], bn0[1], NEG))
    return dict(gmax=gmax, garg=garg, y0=y0, bn0=bn0, y3=y3, bn3=bn3, B=B, N=N, training=training)


_EYE: Dict[Tuple[int, str], Tensor] = {}


def _eye(n: int, device) -> Tensor:
    key = (n, str(device))
    if key not in _EYE:
        _EYE[key] = torch.eye(n, dtype=torch.float32, device=device)
    return _EYE[key]


def global_backward(P, gctx, W_g: Optional[Tensor], drb: Tensor, da2: Tensor):
    """drb [B,256] = gradient w.r.t. the per-shape bias of tail.0 (= W_g . lrelu(bn(y3)) + b).
    W_g None: drb [B,512] is the gradient w.r.t. the global feature lrelu(bn(y3)) itself (--attn: the concat is materialised).
    Adds the max-pool gradient into da2 in place.  -> {name: grad} incl. 'tail.0.weight.global' [256,512]."""
    B = gctx["B"]
    g: Dict[str, Tensor] = {}
    bn0, bn3, y0, y3 = gctx["bn0"], gctx["bn3"], gctx["y0"], gctx["y3"]
    tr = gctx["training"]
    if W_g is None:
        Wg_t = _eye(y3.shape[1], y3.device)
    else:
        g["tail.0.weight.global"] = ops.gemm_tn(drb, y3, pro=(bn3[0], bn3[1], NEG), defer=True)
        g["tail.0.bias"] = ops.colsum(drb)[0]
        Wg_t = _t(W_g)
    g3, s0, s1 = ops.gemm_nt_bnbwd(drb, Wg_t, y3, bn3[0], bn3[1], bn3[3], bn3[2], NEG)
    g["global_conv.4.weight"] = s1; g["global_conv.4.bias"] = s0
    sums = _cat2(s0, s1) if tr else torch.zeros(2 * s0.numel(), device=s0.device)
    dy3 = ops.bn_bwd_apply(g3, y3, bn3[3], bn3[2], P["global_conv.4.weight"], sums, B)
    g["global_conv.3.weight"] = ops.gemm_tn(dy3, y0, pro=(bn0[0], bn0[1], NEG), defer=True)
    g["global_conv.3.bias"] = ZERO_GRAD if tr else ops.colsum(dy3)[0]
    g0, s0, s1 = ops.gemm_nt_bnbwd(dy3, _t(P["global_conv.3.weight"]), y0, bn0[0], bn0[1], bn0[3], bn0[2], NEG)
    g["global_conv.1.weight"] = s1; g["global_conv.1.bias"] = s0
    sums = _cat2(s0, s1) if tr else torch.zeros(2 * s0.numel(), device=s0.device)
    dy0 = ops.bn_bwd_apply(g0, y0, bn0[3], bn0[2], P["global_conv.1.weight"], sums, B)
    g["global_conv.0.weight"] = ops.gemm_tn(dy0, gctx["gmax"], defer=True)
    g["global_conv.0.bias"] = ZERO_GRAD if tr else ops.colsum(dy0)[0]
    dgmax = ops.gemm_nt(dy0, _t(P["global_conv.0.weight"]))
    ops.maxpool_bwd_add(dgmax, gctx["garg"], da2)
    return g
