"""torch.autograd.Function wrappers around the hand-written pipelines of `nets.py`.

autograd is used as plumbing only: it sequences our forward/backward pipelines, sums the two
style gradients, and hands parameter gradients to the optimiser.  Every Function computes its own
backward with HIP kernels; the Discriminator additionally supports double backward (WGAN-GP,
Common/gradient_penalty.py:31-35) by exposing its backward as a second Function.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import threading

import torch
from torch.autograd import Function

from . import nets, ops

Tensor = torch.Tensor


_HAS_ENGINE_QUERY = hasattr(torch._C, "_will_engine_execute_node")     # private API (present in torch 1.13 .. 2.10); guarded, see below
# No process-global mutable flags: the two caller-selected backward modes below are THREAD-LOCAL on the thread that runs the
# forward pass, and every Function records them in its own ctx when its forward runs.  The backward of that node (which autograd executes
# on its device thread, not on the caller's) reads the record of ITS graph -- two threads driving two model pairs, or one thread
# interleaving a fused and a plain backward, cannot see each other's modes.
_TLS = threading.local()


def _mode(name: str) -> bool:
    return getattr(_TLS, name, 0) > 0


class _Mode:
    name = ""

    def __enter__(self):
        setattr(_TLS, self.name, getattr(_TLS, self.name, 0) + 1)

    def __exit__(self, *exc):
        setattr(_TLS, self.name, getattr(_TLS, self.name, 0) - 1)


class input_grad_only(_Mode):
    """Context: Discriminator forwards evaluated inside it build nodes whose backward delivers INPUT gradients only (no parameter
    gradients) -- spgan.losses.GradientPenalty wraps `D(x_hat)` and its `autograd.grad(..., create_graph=True)` in it, which makes the
    WGAN-GP route independent of the private engine query below."""
    name = "input_only"


def _engine_needs(ctx, pos: int, tpos: int) -> bool:
    """True if the running autograd task will actually consume the gradient of input `pos`
    (autograd.grad(inputs=[x]) does not need parameter gradients although they require grad).
    `tpos` is the input's index among the *tensor* arguments (next_functions skips non-tensors).
    Uses torch._C._will_engine_execute_node where this torch has it (so that the REFERENCE's GradientPenalty works unchanged on
    our Discriminator); without it every gradient autograd marks as needed is computed, and only `input_grad_only()` callers get
    the differentiable input-gradient route."""
    if not ctx.needs_input_grad[pos]:
        return False
    if not _HAS_ENGINE_QUERY:
        return True
    try:
        fn = ctx.next_functions[tpos][0]
        if fn is None:
            return False
        return bool(torch._C._will_engine_execute_node(fn))
    except RuntimeError:
        return True


# Parameter gradients are handed back to autograd (AccumulateGrad, hooks, torch.autograd.grad all behave as usual) unless the
# caller opted into the fused route with `fused_grad_accumulation()` -- spgan.TrainStep does, around each of its segments (the mode is
# recorded per graph when a Function's FORWARD runs: the context must enclose the forward pass, wrapping only backward() selects nothing).
class DeliverySink:
    """Where the nodes of ONE backward pass leave their parameter gradients when the caller accumulates them itself (TrainStep, one sink per
    network): every fused `_deliver` appends its (destination, source) pairs instead of launching its own split-sum reduction and accumulation,
    and `flush()` -- called by the owner when backward() has returned, on the same stream -- finishes ALL deferred split sums with one
    reduction launch and adds every gradient with one launch (two when a parameter received more than one gradient: those go through
    spgan_multi_addn in arrival order).  The same sums as the per-node launches (a parameter's gradients are added in the order the nodes ran)."""

    def __init__(self):
        self._pairs = []
        self._lock = threading.Lock()

    def add(self, pairs) -> None:
        with self._lock:
            self._pairs.extend(pairs)

    def clear(self) -> None:
        """Forget what an aborted backward (an exception, a failed stream capture) left behind: its gradients must not reach the next step."""
        with self._lock:
            self._pairs = []

    def flush(self) -> None:
        with self._lock:
            pairs, self._pairs = self._pairs, []
        ops.flush_tn()
        if not pairs:
            return
        by_dst, order = {}, []
        for d, s_ in pairs:
            key = (d.data_ptr(), tuple(d.shape), tuple(d.stride()))
            if key not in by_dst:
                by_dst[key] = (d, [])
                order.append(key)
            by_dst[key][1].append(s_)
        # Two DIFFERENT destinations that share elements (a CatCols column block of a parameter's .grad from one node and the whole .grad
        # from another) must not meet in one launch -- that would be an unordered read-modify-write.  They leave the grouped launches and
        # are added one stream-ordered launch per gradient, in arrival order (what the per-node launches did).
        clash = _overlapping_keys([by_dst[k][0] for k in order])
        if clash:
            hit = {order[i] for i in clash}
            for d, s_ in pairs:
                if (d.data_ptr(), tuple(d.shape), tuple(d.stride())) in hit:
                    ops.multi_add([d], [s_ if s_.shape == d.shape or s_.is_contiguous() else s_.contiguous()])
            order = [k for k in order if k not in hit]
        single = [(by_dst[k][0], by_dst[k][1][0]) for k in order if len(by_dst[k][1]) == 1]
        multi = [by_dst[k] for k in order if len(by_dst[k][1]) > 1]
        rest = []
        for d, ss in multi:      # more than three gradients for one parameter, or strided ones: further rounds of plain adds
            if len(ss) > 3 or not d.is_contiguous() or any(not x.is_contiguous() for x in ss):
                rest.append((d, ss))
        multi = [m for m in multi if not any(m[0] is r[0] for r in rest)]
        if single:
            ops.multi_add([d for d, _ in single], [s_ for _, s_ in single])
        if multi:
            ops.multi_addn([d for d, _ in multi], [ss for _, ss in multi])
        for d, ss in rest:
            for x in ss:
                ops.multi_add([d], [x])


def _extent(t: Tensor):
    """[lo, hi) byte range a (possibly strided) view touches."""
    lo = t.data_ptr()
    span = 1 + sum((n - 1) * st for n, st in zip(t.shape, t.stride()) if n > 0)
    return lo, lo + span * t.element_size()


def _share_elements(a: Tensor, b: Tensor) -> bool:
    """Do two destination views share an element?  Exact for the shapes the sink sees (contiguous tensors; column blocks [rows, w] of one
    row-major matrix: same row stride, unit column stride); conservative (True) for any other pair whose byte ranges intersect."""
    if a.numel() == 0 or b.numel() == 0:
        return False
    (alo, ahi), (blo, bhi) = _extent(a), _extent(b)
    if ahi <= blo or bhi <= alo:
        return False
    if a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1 and a.stride(0) == b.stride(0) and a.shape[0] == b.shape[0] \
            and a.element_size() == b.element_size():
        ld, item = a.stride(0), a.element_size()
        base = min(alo, blo)
        ca, cb = (alo - base) // item, (blo - base) // item
        if (alo - base) % item == 0 and (blo - base) % item == 0 and ca + a.shape[1] <= ld and cb + b.shape[1] <= ld:
            return not (ca + a.shape[1] <= cb or cb + b.shape[1] <= ca)      # column intervals inside one row frame
    return True


def _overlapping_keys(dsts) -> set:
    """Indices of the destinations that share elements with ANOTHER destination of the list (distinct views; O(n log n) sweep over byte
    ranges, the exact test only for range-intersecting pairs)."""
    ext = sorted((*_extent(d), i) for i, d in enumerate(dsts))
    out, live = set(), []
    for lo, hi, i in ext:
        live = [(h, j) for h, j in live if h > lo]
        for _, j in live:
            if _share_elements(dsts[i], dsts[j]):
                out.add(i)
                out.add(j)
        live.append((hi, i))
    return out


class fused_grad_accumulation(_Mode):
    """Context: the nodes of forward passes evaluated inside it add their parameter gradients straight into the pre-bound flat `.grad`
    buffers (spgan.optim.flatten_module) with one fused launch per Function and return None to autograd.  Only for callers that read
    gradients from `.grad` afterwards and use no parameter hooks (TrainStep wraps its segments in it, CapturedBody the caller's body on
    request); everything else gets normal autograd semantics.  The mode is recorded at FORWARD time (`_record_modes`): enclose the forward
    pass -- a context around backward() alone has no effect on graphs built outside it."""
    name = "fused"

    def __init__(self, sink: Optional[DeliverySink] = None):
        """sink: the graphs built inside leave their parameter gradients there (DeliverySink) and the caller flushes it after backward()."""
        self.sink = sink

    def __enter__(self):
        super().__enter__()
        self._prev = getattr(_TLS, "sink", None)
        _TLS.sink = self.sink

    def __exit__(self, *exc):
        _TLS.sink = self._prev
        super().__exit__(*exc)


def _record_modes(ctx, holder=None) -> None:
    """Called in every Function.forward (on the caller's thread): remember the backward modes selected for this graph."""
    ctx.fused = (getattr(_TLS, "sink", None) or True) if _mode("fused") else False      # True, or the DeliverySink the gradients go to
    ctx.input_only = _mode("input_only")
    if holder is not None:                       # nodes created later INSIDE this node's backward (the double-backward node) inherit them
        holder.fused, holder.input_only = ctx.fused, ctx.input_only


def _deliver(params: Sequence[Tensor], grads: Sequence, needs: Sequence[bool], fused: bool = False):
    """Hand parameter gradients back to autograd -- or, for a graph built inside `fused_grad_accumulation()` (`fused`), for every wanted leaf parameter that
    already owns a contiguous `.grad` (the flat buffer of spgan.optim.flatten_module) while no graph is being recorded, add them
    into `.grad` with ONE fused launch (spgan_multi_add) and return None: the same sums AccumulateGrad would form with one
    elementwise launch per parameter tensor (and on the launch stream, which keeps the step capturable as a hipGraph).  Non-leaf
    "parameters" (the scaled weights of equalised-LR layers) get their gradient returned.  Exact-zero gradients (nets.ZERO_GRAD)
    cost nothing on the fused path."""
    sink = fused if isinstance(fused, DeliverySink) else None
    fused = bool(fused) and not torch.is_grad_enabled()
    def to_sink(p, g) -> bool:         # can this gradient wait for the sink's flush as it is (no copy, nothing handed back to autograd)?
        if isinstance(g, int):
            return True
        if not (p.is_leaf and p.grad is not None and p.grad.is_contiguous() and p.grad.numel() == g.numel()):
            return False
        return isinstance(g, nets.CatCols) or g.is_contiguous() or g.shape == p.grad.shape
    if sink is not None and fused and not all(to_sink(p, g) for p, g, need in zip(params, grads, needs) if need and g is not None):
        ops.flush_tn()                 # a gradient that is copied or handed back to autograd below must be complete now
    if sink is None or not fused:
        ops.flush_tn()                 # weight gradients whose split-K sums were deferred (ops.gemm_tn(defer=True)) become valid here
        sink = None                    # (with a sink the owner's flush() finishes them, together with those of every other node)
    out: List[Optional[Tensor]] = [None] * len(params)
    pairs = []
    for i, (p, g, need) in enumerate(zip(params, grads, needs)):
        if not need or g is None:
            continue
        zero = isinstance(g, int)
        if fused and p.is_leaf and p.grad is not None and p.grad.is_contiguous() and (zero or p.grad.numel() == g.numel()):
            if zero:
                continue
            if isinstance(g, nets.CatCols):      # column blocks of one weight: each goes straight into its slice of .grad
                flat, c0 = p.grad.view(g.rows, -1), 0
                for part in g.parts:
                    pairs.append((flat[:, c0:c0 + part.shape[1]], part))
                    c0 += part.shape[1]
            elif g.is_contiguous() or g.shape == p.grad.shape:
                pairs.append((p.grad, g))        # a strided view of the parameter's shape is added through its strides
            else:
                pairs.append((p.grad, g.contiguous()))
        elif zero:
            out[i] = torch.zeros_like(p)
        else:
            if isinstance(g, nets.CatCols):
                g = g.cat()
            out[i] = g.view_as(p) if g.shape != p.shape else g
    if pairs:
        if sink is not None:
            sink.add(pairs)
        else:
            ops.multi_add([d for d, _ in pairs], [s_ for _, s_ in pairs])
    return tuple(out)


class _Holder:
    """Non-tensor bag passed through Function.apply (module buffers, flags)."""
    def __init__(self, **kw):
        self.__dict__.update(kw)


# ---------------------------------------------------------------------------------------------
# layout
# ---------------------------------------------------------------------------------------------
class CmToPm(Function):
    @staticmethod
    def forward(ctx, x_cm):
        ctx.B, _, ctx.N = x_cm.shape
        return ops.cm_to_pm(x_cm)

    @staticmethod
    def backward(ctx, g):
        return ops.pm_to_cm(g.contiguous(), ctx.B, ctx.N)


class PmToCm(Function):
    @staticmethod
    def forward(ctx, x_pm, B, N):
        return ops.pm_to_cm(x_pm, B, N)

    @staticmethod
    def backward(ctx, g):
        return ops.cm_to_pm(g.contiguous()), None, None


# ---------------------------------------------------------------------------------------------
# Discriminator (with double backward)
# ---------------------------------------------------------------------------------------------
class DiscriminatorFn(Function):
    """logits = D(x);  inputs: holder, x [B,3,N], *params (in `holder.names` order)."""

    @staticmethod
    def forward(ctx, holder, x, *params):
        _record_modes(ctx, holder)
        P = dict(zip(holder.names, params))
        pre = getattr(holder, "pre", None)
        if pre is not None:
            # the conv stack of this pass was already evaluated (batched with other passes, nets.d_forward_groups): only the head is left
            pooled, dctx = pre
            out, dctx["hs"] = nets.d_head_forward(P, pooled)
        else:
            out, dctx = nets.d_forward(P, holder.buffers, x, holder.training, True)
        ctx.holder, ctx.dctx = holder, dctx
        ctx.save_for_backward(x, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, *params = ctx.saved_tensors
        names = ctx.holder.names
        if ctx.input_only:
            need_dx, need_dp = bool(ctx.needs_input_grad[1]), False
        else:
            need_dx = _engine_needs(ctx, 1, 0)
            need_dp = any(_engine_needs(ctx, 2 + i, 1 + i) for i in range(len(params)))
        if torch.is_grad_enabled() and need_dx and not need_dp:
            # create_graph=True and only the input gradient is wanted: differentiable backward
            dx = DiscriminatorBackwardFn.apply(ctx.holder, ctx.dctx, dout, x, *params)
            return (None, dx) + (None,) * len(params)
        P = dict(zip(names, [nets.owned(p) for p in params]))
        dx, grads, _ = nets.d_backward(P, ctx.dctx, dout.detach(), need_dx, need_dp, False)
        if grads is None:
            return (None, dx) + (None,) * len(params)
        return (None, dx) + _deliver(params, [grads[n] for n in names], ctx.needs_input_grad[2:], ctx.fused)


class DStackFn(Function):
    """pooled [B,C4] = max_N lrelu(bn(conv stack(x))): the Discriminator without its per-shape MLP head (first-order only).
    inputs: holder(names of the conv-stack parameters, buffers, training), x [B,3,N], *conv-stack params."""

    @staticmethod
    def forward(ctx, holder, x, *params):
        _record_modes(ctx, holder)
        P = dict(zip(holder.names, params))
        pre = getattr(holder, "pre", None)
        if pre is not None:
            pooled, dctx = pre          # evaluated as one of several batched passes (nets.d_forward_groups)
        else:
            pooled, dctx = nets.d_forward(P, holder.buffers, x, holder.training, True, head=False)
        ctx.holder, ctx.dctx = holder, dctx
        ctx.save_for_backward(*params)
        return pooled

    @staticmethod
    def backward(ctx, gpool):
        params = ctx.saved_tensors
        names = ctx.holder.names
        need_dp = any(ctx.needs_input_grad[2:])
        P = dict(zip(names, [nets.owned(p) for p in params]))
        dx, grads, _ = nets.d_backward(P, ctx.dctx, None, ctx.needs_input_grad[1], need_dp, False, gpool=gpool.detach())
        if grads is None:
            return (None, dx) + (None,) * len(params)
        return (None, dx) + _deliver(params, [grads.get(n) for n in names], ctx.needs_input_grad[2:], ctx.fused)


class DStacksJointFn(Function):
    """The conv stacks of the D step's passes behind ONE node, so that their backward work runs in lock step (nets.d_backward_joint):
    outputs = (pooled of every first-order pass ..., gx) with gx = d sum(D(x_hat)) / d x_hat of the `hat` pass (WGAN-GP,
    Common/gradient_penalty.py:28-33; its head and first-order backward run inside this forward).  The backward receives the pooled
    gradients of the first-order passes (from DHeadFn) and the penalty's seed on gx -- the double backward -- together: every layer's
    launch is issued once for all of them.  inputs: holder(names, firsts=[(pooled, dctx)], hat=(pooled, dctx) | None), *all D params."""

    @staticmethod
    def forward(ctx, holder, *params):
        _record_modes(ctx, holder)
        outs = [pooled for pooled, _ in holder.firsts]
        ctx.saved_hat = None
        if holder.hat is not None:
            P = dict(zip(holder.names, params))
            pooled_h, dctx_h = holder.hat
            logits, dctx_h["hs"] = nets.d_head_forward(P, pooled_h)
            ones = nets._const_vec(logits.numel(), 1.0, logits.device).view_as(logits)          # grad_outputs = ones (gradient_penalty.py:29), a cached constant
            dx, _, ctx.saved_hat = nets.d_backward(P, dctx_h, ones, True, False, keep_for_double=True)
            outs.append(dx)
        ctx.holder = holder
        ctx.save_for_backward(*params)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        params = ctx.saved_tensors
        h = ctx.holder
        names = h.names
        P = dict(zip(names, [nets.owned(p) for p in params]))
        nf = len(h.firsts)
        firsts = [(dctx, g.detach()) for (_, dctx), g in zip(h.firsts, gouts[:nf]) if g is not None]
        dbl = None
        if h.hat is not None and gouts[nf] is not None:
            dbl = (h.hat[1], ctx.saved_hat, gouts[nf].detach())
        fg, hg = nets.d_backward_joint(P, firsts, dbl)
        chains = list(fg) + ([hg] if hg is not None else [])
        needs = ctx.needs_input_grad[1:]
        fused = bool(ctx.fused) and not torch.is_grad_enabled()
        sink = ctx.fused if (fused and isinstance(ctx.fused, DeliverySink)) else None
        def to_grad(p_, gs) -> bool:      # added into the pre-bound .grad (by the sink's owner or by the launch below) instead of handed back
            return fused and p_.is_leaf and p_.grad is not None and p_.grad.is_contiguous() and all(g.numel() == p_.grad.numel() for g in gs)
        wanted = []
        for i, (n, p_, need) in enumerate(zip(names, params, needs)):
            gs = [c[n] for c in chains if n in c and not isinstance(c[n], int)] if need else []
            if gs:
                wanted.append((i, p_, gs))
        # The per-pass weight gradients are deferred split-K sums (gemm_dual_multi(defer=True), gemm_tn_narrow_multi): only the sink's owner may
        # leave them unreduced.  Anything summed here with torch adds and handed back to autograd has to be complete first.
        if sink is None or any(not to_grad(p_, gs) for _, p_, gs in wanted):
            ops.flush_tn()
        out: List[Optional[Tensor]] = [None] * len(params)
        dsts, srcs = [], []
        for i, p_, gs in wanted:
            if to_grad(p_, gs):
                dsts.append(p_.grad); srcs.append(gs)
            else:
                tot = gs[0]
                for g in gs[1:]:
                    tot = tot + g
                out[i] = tot.view_as(p_)
        if dsts and sink is not None:
            sink.add([(d, g) for d, gs in zip(dsts, srcs) for g in gs])      # the owner adds them (arrival order) with everything else of this backward
        elif dsts:
            ops.multi_addn(dsts, srcs)      # one launch: ((grad + real) + fake) + double backward, per parameter
        return (None,) + tuple(out)


class JoinRowsFn(Function):
    """cat(parts, dim=0) for the pooled features of passes that one grouped launch wrote as consecutive row blocks of ONE buffer: then the result
    is a view of that buffer (ops.stacked_rows: no copy launch); backward hands every part its row block of the gradient (views)."""

    @staticmethod
    def forward(ctx, *parts):
        ctx.sizes = [p_.shape[0] for p_ in parts]
        return ops.stacked_rows([p_.detach() for p_ in parts])

    @staticmethod
    def backward(ctx, g):
        return tuple(g.split(ctx.sizes, dim=0))


class DHeadFn(Function):
    """logits [B',1] = mlp(pooled [B',C4]) -- the head of one or several stacked passes as one batch.  inputs: holder(names), pooled, *params."""

    @staticmethod
    def forward(ctx, holder, pooled, *params):
        _record_modes(ctx, holder)
        P = dict(zip(holder.names, params))
        pooled = pooled.contiguous()
        logits, hs = nets.d_head_forward(P, pooled)
        ctx.holder, ctx.hs = holder, hs
        ctx.save_for_backward(pooled, *params)
        sizes = getattr(holder, "sizes", None)
        if sizes is None:
            return logits
        # one output per pass (row blocks of the one logits buffer): the caller's per-pass losses then hand their gradients straight to
        # this node -- slicing a single output instead costs a zero-fill, a copy and an add per slice in autograd's slice backward
        return tuple(logits.split(list(sizes), dim=0))

    @staticmethod
    def backward(ctx, *douts):
        pooled, *params = ctx.saved_tensors
        names = ctx.holder.names
        need_dp = any(ctx.needs_input_grad[2:])
        P = dict(zip(names, [nets.owned(p) for p in params]))
        if len(douts) == 1:
            dout = douts[0]
        else:
            sizes = ctx.holder.sizes
            dout = ops.stacked_rows([d if d is not None else pooled.new_zeros((n, 1)) for d, n in zip(douts, sizes)])   # the loss kernel's seeds: one buffer
        gpool, grads, _ = nets.d_head_backward(P, pooled, ctx.hs, dout.detach().contiguous(), need_dp)
        gp = gpool if ctx.needs_input_grad[1] else None
        if not need_dp:
            return (None, gp) + (None,) * len(params)
        return (None, gp) + _deliver(params, [grads.get(n) for n in names], ctx.needs_input_grad[2:], ctx.fused)


class DiscriminatorBackwardFn(Function):
    """dx = dD(x)/dx contracted with dout, as a differentiable node (its backward is the double backward)."""

    @staticmethod
    def forward(ctx, holder, dctx, dout, x, *params):
        ctx.fused, ctx.input_only = getattr(holder, "fused", False), False      # created inside DiscriminatorFn.backward (engine thread): modes inherited from that graph
        P = dict(zip(holder.names, params))
        dx, _, saved = nets.d_backward(P, dctx, dout, True, False, keep_for_double=True)
        ctx.holder, ctx.dctx, ctx.saved = holder, dctx, saved
        ctx.save_for_backward(*params)
        return dx

    @staticmethod
    def backward(ctx, v):
        params = ctx.saved_tensors
        names = ctx.holder.names
        P = dict(zip(names, [nets.owned(p) for p in params]))
        need_x = ctx.needs_input_grad[3]
        dbl = nets.d_double_backward if ctx.dctx["training"] else nets.d_double_backward_eval      # eval(): BatchNorm is a fixed affine
        grads, dx2 = dbl(P, ctx.dctx, ctx.saved, v.detach(), need_dx=need_x)
        # (holder, dctx, dout, x, *params); the gradient w.r.t. dout is not provided (constant ones in WGAN-GP)
        return (None, None, None, dx2) + _deliver(params, [grads[n] for n in names], ctx.needs_input_grad[4:], ctx.fused)


# ---------------------------------------------------------------------------------------------
# Generator pieces (first-order only)
# ---------------------------------------------------------------------------------------------
class EdgeBlockFn(Function):
    """out[M,F] = EdgeBlock(x[M,C]); inputs: holder(prefix, names, buffers, B, N, k, training, knn_mode, idx), x, *params."""

    @staticmethod
    def forward(ctx, holder, x, *params):
        _record_modes(ctx, holder)
        P = dict(zip(holder.names, params))
        x = x.contiguous()
        reuse = getattr(holder, "reuse", None)
        if reuse is not None:
            out, ectx = reuse                       # evaluated by an earlier identical forward (bn_repeats accounted for it there)
            out = out.view_as(out)                  # a fresh tensor object for autograd to hang this node on
            idx = ectx["idx"]
        else:
            idx = holder.idx if holder.idx is not None else ops.knn(x, holder.B, holder.N, holder.k, holder.knn_mode)
            out, ectx = nets.edgeblock_forward(P, holder.buffers, holder.prefix, x, idx, holder.B, holder.N, holder.training,
                                               getattr(holder, "update_running", True),
                                               getattr(holder, "count_rep", 1), getattr(holder, "bn_repeats", 1))
            keep = getattr(holder, "keep", None)
            if keep is not None:
                keep["out"], keep["ctx"] = out, ectx
        ctx.holder, ctx.ectx = holder, ectx
        holder.last_idx = idx
        ctx.save_for_backward(*params)
        return out

    @staticmethod
    def backward(ctx, dout):
        params = ctx.saved_tensors
        h = ctx.holder
        P = dict(zip(h.names, [nets.owned(p) for p in params]))
        cache = getattr(h, "graph_cache", None)                      # static-sphere graph: CSR built once, reused every step
        if cache is not None and cache.get("csr") is not None:
            csr = cache["csr"]
        else:
            csr = ops.csr_build(ctx.ectx["idx"], h.B, h.N)
            if cache is not None:
                cache["csr"] = csr
        dx, g = nets.edgeblock_backward(P, h.prefix, ctx.ectx, dout, csr, need_dx=ctx.needs_input_grad[1])
        return (None, dx) + _deliver(params, [g[n] for n in h.names], ctx.needs_input_grad[2:], ctx.fused)


class EdgeFeaturesFn(Function):
    """ee [B,2C,N,k] = cat[x_i, x_j - x_i] (Generation/modules.py:708-720) with its adjoint w.r.t. x [B,C,N]."""

    @staticmethod
    def forward(ctx, x, idx, k):
        ctx.save_for_backward(idx)
        ctx.k = k
        return ops.edge_features_cm(x, idx, k)

    @staticmethod
    def backward(ctx, dE):
        (idx,) = ctx.saved_tensors
        return ops.edge_features_cm_bwd(dE.contiguous(), idx, ctx.k), None, None


class RepeatRowsFn(Function):
    """[N,C] -> [B*N,C]: B copies of the rows (the EdgeConv1 features of the tiled sphere prior are the same for every shape of the
    batch); backward sums the B row blocks (deterministic column reduction)."""

    @staticmethod
    def forward(ctx, x, B, reuse=None):
        ctx.B = B
        if reuse is not None:                   # the rows were written by an earlier evaluation (nets.g_pair_forward)
            return reuse.view_as(reuse)
        return x.repeat(B, 1)

    @staticmethod
    def backward(ctx, g):
        B = ctx.B
        M, Cn = g.shape
        return ops.colsum(g.contiguous().view(B, (M // B) * Cn))[0].view(M // B, Cn), None, None


class AdaINFn(Function):
    """inputs: holder(prefix, N, slope), x [M,C], style [M,S], weight, bias"""

    @staticmethod
    def forward(ctx, holder, x, style, w, b):
        _record_modes(ctx, holder)
        P = {holder.prefix + ".style.weight": w, holder.prefix + ".style.bias": b}
        reuse = getattr(holder, "reuse", None)
        if reuse is not None:                   # evaluated by nets.g_pair_forward: adopt the rows and the saved context of this pass
            out, actx = reuse
            out = out.view_as(out)
        else:
            out, actx = nets.adain_forward(P, holder.prefix, x.contiguous(), style.contiguous(), holder.N, holder.slope)
        ctx.holder, ctx.actx = holder, actx
        ctx.save_for_backward(w, b)
        return out

    @staticmethod
    def backward(ctx, dout):
        w, b = ctx.saved_tensors
        pre = ctx.holder.prefix
        P = {pre + ".style.weight": nets.owned(w), pre + ".style.bias": nets.owned(b)}
        need_p = ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
        # Two AdaIN layers on one style tensor (holder.style_chain): the layer whose backward runs FIRST (the later layer: adain2) keeps its
        # style gradient in the shared dict and hands autograd nothing; the other one (adain1 -- its backward depends on adain2's through
        # EdgeConv2, so it always runs afterwards) adds that tensor inside its own style-gradient GEMM and returns the sum.
        chain = getattr(ctx.holder, "style_chain", None)
        addend = None
        if chain is not None and ctx.needs_input_grad[2] and chain[1] == "last":
            addend = chain[0].pop("ds", None)
        dx, ds, g = nets.adain_backward(P, pre, ctx.actx, dout, ctx.needs_input_grad[1], ctx.needs_input_grad[2], need_p, dstyle_addend=addend)
        if chain is not None and ctx.needs_input_grad[2] and chain[1] == "first":
            chain[0]["ds"] = ds
            ds = None
        return (None, dx, ds) + _deliver((w, b), [g.get(pre + ".style.weight"), g.get(pre + ".style.bias")], ctx.needs_input_grad[3:], ctx.fused)


class MLPFn(Function):
    """Chain of 1x1 convs + activations.  inputs: holder(names, acts, slope), x [M,C], *params (w0,b0,w1,b1,...)"""

    @staticmethod
    def forward(ctx, holder, x, *params):
        _record_modes(ctx, holder)
        P = {}
        for i, n in enumerate(holder.names):
            P[n + ".weight"], P[n + ".bias"] = params[2 * i], params[2 * i + 1]
        out, mctx = nets.mlp_forward(P, holder.names, holder.acts, x.contiguous(), holder.slope)
        ctx.holder, ctx.mctx = holder, mctx
        ctx.save_for_backward(*params)
        return out

    @staticmethod
    def backward(ctx, dout):
        params = ctx.saved_tensors
        h = ctx.holder
        P = {}
        for i, n in enumerate(h.names):
            P[n + ".weight"], P[n + ".bias"] = nets.owned(params[2 * i]), nets.owned(params[2 * i + 1])
        need_p = any(ctx.needs_input_grad[2:])
        dx, g, _ = nets.mlp_backward(P, ctx.mctx, dout, ctx.needs_input_grad[1], need_p)
        out = []
        for n in h.names:
            out.append(g.get(n + ".weight"))
            out.append(g.get(n + ".bias"))
        return (None, dx) + _deliver(params, out, ctx.needs_input_grad[2:], ctx.fused)


class ScaleFn(Function):
    """c * w: the runtime weight scaling of equalised-LR layers (`EqualLR.compute_weight`, Generation/modules.py:264-268)."""

    @staticmethod
    def forward(ctx, w, c):
        _record_modes(ctx)
        ctx.c = c
        ctx.save_for_backward(w)
        return ops.axpby(c, w.contiguous(), 0.0, torch.empty_like(w, memory_format=torch.contiguous_format))

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        gw = ops.axpby(ctx.c, g.contiguous(), 0.0, torch.empty_like(g, memory_format=torch.contiguous_format))
        return _deliver((w,), [gw], ctx.needs_input_grad[:1], ctx.fused) + (None,)


class HeadFn(Function):
    """The style branch for a latent that is constant over the points of a shape (the default `noise_generator`, model.py:128-131:
    one vector per shape tiled over N): head.0([x, z]) = Wx.x + (Wz.z + b) -- the 128-wide latent half is a per-shape bias
    ([B,128], one small GEMM) and only the three coordinate channels are contracted per point, instead of a 131-wide GEMM over all
    B*N rows of a materialised cat[x, z] (Generator.py:166-169).  inputs: holder(N), x_pm [M,3], zb [B,nz], head.0 w/b, head.2 w/b."""

    @staticmethod
    def forward(ctx, holder, x_pm, zb, w0, b0, w2, b2):
        _record_modes(ctx, holder)
        W0 = nets._w2(w0)
        c = x_pm.shape[1]
        reuse = getattr(holder, "reuse", None)
        if reuse is not None:                   # evaluated by nets.g_pair_forward
            out, mctx = reuse
            out = out.view_as(out)
        else:
            rb = ops.gemm_nt(zb.contiguous(), nets._cols_from(W0, c), b0)                  # [B,128]
            P = {"head.0.weight": w0, "head.0.bias": b0, "head.2.weight": w2, "head.2.bias": b2}
            out, mctx = nets.mlp_forward(P, ["head.0", "head.2"], [ops.ACT_LRELU, ops.ACT_LRELU], x_pm.contiguous(), nets.NEG,
                                         rowbias=rb, N=holder.N, first_weight=W0[:, :c])
        ctx.mctx, ctx.c = mctx, c
        ctx.save_for_backward(zb, w0, b0, w2, b2)
        return out

    @staticmethod
    def backward(ctx, dout):
        zb, w0, b0, w2, b2 = ctx.saved_tensors
        params = (w0, b0, w2, b2)
        P = {"head.0.weight": nets.owned(w0), "head.0.bias": nets.owned(b0), "head.2.weight": nets.owned(w2), "head.2.bias": nets.owned(b2)}
        W0 = nets._w2(P["head.0.weight"])
        ctx.mctx["first_weight"] = W0[:, :ctx.c]
        need_p = any(ctx.needs_input_grad[3:])
        dx, g, drb = nets.mlp_backward(P, ctx.mctx, dout, ctx.needs_input_grad[1], need_p)
        dz = ops.gemm_nt(drb, nets._t(W0[:, ctx.c:])) if ctx.needs_input_grad[2] else None
        grads = [None] * 4
        if need_p:
            gz = ops.gemm_tn(drb, zb.contiguous(), defer=True)                           # [128, nz]; finished with the other deferred sums (_deliver)
            grads = [nets.CatCols([g["head.0.weight.part"], gz]), ops.colsum(drb)[0], g["head.2.weight"], g["head.2.bias"]]
        return (None, dx, dz) + _deliver(params, grads, ctx.needs_input_grad[3:], ctx.fused)


GF_NAMES = ("global_conv.0.weight", "global_conv.0.bias", "global_conv.1.weight", "global_conv.1.bias",
            "global_conv.3.weight", "global_conv.3.bias", "global_conv.4.weight", "global_conv.4.bias")


class GlobalFeatFn(Function):
    """feat[M,640] = cat[global_conv(max_N a2) repeated over N, a2] as a tensor (Generator.py:183-189) -- the --attn variant
    feeds it to `Attention`; the default path (GlobalTailFn) never builds it."""

    @staticmethod
    def forward(ctx, holder, a2, *params):
        _record_modes(ctx, holder)
        P = dict(zip(GF_NAMES, params))
        feat, gctx = nets.global_feat_forward(P, holder.buffers, a2.contiguous(), holder.B, holder.N, holder.training, True)
        ctx.gctx = gctx
        ctx.save_for_backward(*params)
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        params = ctx.saved_tensors
        P = dict(zip(GF_NAMES, [nets.owned(p) for p in params]))
        da2, g = nets.global_feat_backward(P, ctx.gctx, dfeat.contiguous())
        return (None, da2 if ctx.needs_input_grad[1] else None) + _deliver(params, [g[n] for n in GF_NAMES], ctx.needs_input_grad[2:], ctx.fused)


ATTN_NAMES = ("attn.theta.weight", "attn.phi.weight", "attn.g.weight", "attn.o.weight", "attn.gamma")


class AttentionFn(Function):
    """`Attention(ch)` of Generation/modules.py:534-558 on point-major rows [M,ch]; inputs: holder(B, N), x, theta/phi/g/o weights,
    gamma (module parameter order)."""

    @staticmethod
    def forward(ctx, holder, x, *params):
        _record_modes(ctx, holder)
        P = dict(zip(ATTN_NAMES, params))
        y, actx = nets.attention_forward(P, "attn", x.contiguous(), holder.B, holder.N)
        ctx.actx = actx
        ctx.save_for_backward(*params)
        return y

    @staticmethod
    def backward(ctx, dy):
        params = ctx.saved_tensors
        P = dict(zip(ATTN_NAMES, [nets.owned(p) for p in params]))
        dx, g = nets.attention_backward(P, "attn", ctx.actx, dy, ctx.needs_input_grad[1])
        return (None, dx) + _deliver(params, [g[n] for n in ATTN_NAMES], ctx.needs_input_grad[2:], ctx.fused)


GT_NAMES = ("global_conv.0.weight", "global_conv.0.bias", "global_conv.1.weight", "global_conv.1.bias",
            "global_conv.3.weight", "global_conv.3.bias", "global_conv.4.weight", "global_conv.4.bias",
            "tail.0.weight", "tail.0.bias", "tail.2.weight", "tail.2.bias", "tail.4.weight", "tail.4.bias")


class GlobalTailFn(Function):
    """out[M,3] = tanh(tail(cat[global_conv(max_N a2) repeated, a2]))  (Generator.py:183-194).
    The concat never exists: tail.0 = per-shape bias (512 global channels) + per-point GEMM (128 channels)."""

    @staticmethod
    def forward(ctx, holder, a2, *params):
        _record_modes(ctx, holder)
        P = dict(zip(GT_NAMES, params))
        a2 = a2.contiguous()
        B, N = holder.B, holder.N
        reuse = getattr(holder, "reuse", None)
        if reuse is not None:                   # evaluated by nets.g_pair_forward
            out, gctx, mctx = reuse
            out = out.view_as(out)
            Cg = gctx["y3"].shape[1]
        else:
            gctx = nets.global_forward(P, holder.buffers, a2, B, N, holder.training, True)
            Wt0 = P["tail.0.weight"].view(P["tail.0.weight"].shape[0], -1)
            Cg = gctx["y3"].shape[1]
            W_g, W_x = Wt0[:, :Cg], Wt0[:, Cg:]
            rb = ops.gemm_nt(gctx["y3"], W_g, P["tail.0.bias"], pro=(gctx["bn3"][0], gctx["bn3"][1], nets.NEG))     # [B,256]
            out, mctx = nets.mlp_forward(P, ["tail.0", "tail.2", "tail.4"], [ops.ACT_LRELU, ops.ACT_LRELU, ops.ACT_TANH], a2, nets.NEG,
                                         rowbias=rb, N=N, first_weight=W_x)
        ctx.holder, ctx.gctx, ctx.mctx, ctx.Cg = holder, gctx, mctx, Cg
        ctx.save_for_backward(*params)
        return out

    @staticmethod
    def backward(ctx, dout):
        params = ctx.saved_tensors
        P = dict(zip(GT_NAMES, [nets.owned(p) for p in params]))
        Wt0 = P["tail.0.weight"].view(P["tail.0.weight"].shape[0], -1)
        ctx.mctx["first_weight"] = Wt0[:, ctx.Cg:]
        da2, g, drb = nets.mlp_backward(P, ctx.mctx, dout, True, True)
        gg = nets.global_backward(P, ctx.gctx, Wt0[:, :ctx.Cg], drb, da2)
        g.update({k: v for k, v in gg.items() if k != "tail.0.weight.global"})
        g["tail.0.weight"] = nets.CatCols([gg["tail.0.weight.global"], g.pop("tail.0.weight.part")])
        return (None, da2 if ctx.needs_input_grad[1] else None) + _deliver(params, [g[n] for n in GT_NAMES], ctx.needs_input_grad[2:], ctx.fused)
