"""nn.Module surface of the hot path: drop-in for the reference's `Generator`, `Discriminator`,
`EdgeBlock`, `AdaptivePointNorm` (Generation/Generator.py, Generation/Discriminator.py).

The modules own torch.nn layers purely as *parameter containers*: constructed in the reference's
order, so `state_dict()` keys/shapes and default initial values match and reference checkpoints
load both ways.  Their `forward` never calls those layers: it runs the HIP pipelines of
`functions.py` / `nets.py`.  There is no CPU path: inputs must be on the GPU.
"""
from __future__ import annotations

import math
import weakref
from typing import List, Optional

import torch
import torch.nn as nn

from . import functions as Fn
from . import nets, ops
from .functions import _Holder

NEG = nets.NEG


class _EqualLR(nn.Module):
    """Equalised learning rate (`EqualLR` / `EqualConv1d` / `EqualLinear`, Generation/modules.py:202-239,259-288): the wrapped
    layer keeps `weight_orig` ~ N(0,1) and a zero bias as parameters (state_dict keys `<name>.conv.weight_orig`,
    `<name>.conv.bias` / `<name>.linear.*`), and every forward uses weight_orig * sqrt(2 / fan_in)."""
    _inner = "conv"

    def _wrap(self, layer: nn.Module):
        with torch.no_grad():
            layer.weight.normal_()
            layer.bias.zero_()
        w = layer.weight
        del layer._parameters["weight"]
        layer.register_parameter("weight_orig", nn.Parameter(w.data))
        fan_in = w.shape[1] * (w[0][0].numel() if w.dim() > 2 else 1)
        self.scale = math.sqrt(2.0 / fan_in)
        setattr(self, self._inner, layer)

    @property
    def weight(self):
        return Fn.ScaleFn.apply(getattr(self, self._inner).weight_orig, self.scale)

    @property
    def bias(self):
        return getattr(self, self._inner).bias


class EqualConv1d(_EqualLR):
    def __init__(self, *args, **kwargs):
        super().__init__()
        self._wrap(nn.Conv1d(*args, **kwargs))


class EqualLinear(_EqualLR):
    _inner = "linear"

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self._wrap(nn.Linear(in_dim, out_dim))


class Attention(nn.Module):
    """Generation/modules.py:534-558: non-local block over the points of a shape.  forward(x [B,ch,N]) -> [B,ch,N]."""

    def __init__(self, ch: int, name: str = "attention"):
        super().__init__()
        self.ch = ch
        self.theta = nn.Conv1d(ch, ch // 8, 1, bias=False)
        self.phi = nn.Conv1d(ch, ch // 8, 1, bias=False)
        self.g = nn.Conv1d(ch, ch // 2, 1, bias=False)
        self.o = nn.Conv1d(ch // 2, ch, 1, bias=False)
        self.gamma = nn.Parameter(torch.tensor(0.), requires_grad=True)

    def forward_pm(self, x_pm, B: int, N: int):
        return Fn.AttentionFn.apply(_Holder(B=B, N=N), x_pm, self.theta.weight, self.phi.weight, self.g.weight, self.o.weight, self.gamma)

    def forward(self, x, y=None):
        _require_gpu(x, "Attention")
        B, C, N = x.shape
        return Fn.PmToCm.apply(self.forward_pm(Fn.CmToPm.apply(x), B, N), B, N)


def _stack(spec, eql: bool = False):
    """spec: list of ('conv1d'|'conv2d'|'linear', cin, cout[, ksize]) | ('bn1d'|'bn2d', c) | ('lrelu',) | ('tanh',).
    eql: Conv1d / Linear become their equalised-LR wrappers."""
    layers = []
    for s in spec:
        kind = s[0]
        if kind == "conv1d" and eql:
            layers.append(EqualConv1d(s[1], s[2], 1))
        elif kind == "linear" and eql:
            layers.append(EqualLinear(s[1], s[2]))
        elif kind == "conv1d":
            layers.append(nn.Conv1d(s[1], s[2], 1))
        elif kind == "conv2d":
            layers.append(nn.Conv2d(s[1], s[2], s[3] if len(s) > 3 else 1))
        elif kind == "linear":
            layers.append(nn.Linear(s[1], s[2]))
        elif kind == "bn1d":
            layers.append(nn.BatchNorm1d(s[1]))
        elif kind == "bn2d":
            layers.append(nn.BatchNorm2d(s[1]))
        elif kind == "lrelu":
            layers.append(nn.LeakyReLU(NEG, inplace=True))
        elif kind == "tanh":
            layers.append(nn.Tanh())
        else:
            raise ValueError(kind)
    return nn.Sequential(*layers)


def _require_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError("%s: spgan modules run only on the GPU through libspgan_hip.so (no CPU fallback); got a %s tensor" % (what, t.device))


def _named(module: nn.Module, prefix: str = ""):
    names = [n for n, _ in module.named_parameters()]
    params = [p for _, p in module.named_parameters()]
    return [prefix + n for n in names], params


def _buffers(module: nn.Module, prefix: str = ""):
    d = {prefix + n: b for n, b in module.named_buffers()}
    if isinstance(module, _BNCounts):
        d["__pending_counts__"] = module._pending(prefix)
    return d


class _BNCounts:
    """BatchNorm's `num_batches_tracked` is bookkeeping only (momentum is fixed): the increments are counted on the host
    and written to the int64 buffers when somebody looks (state_dict(), flush_bn_counts()), instead of one tiny
    kernel launch per BatchNorm per forward (36 per train step)."""

    def _pending(self, prefix: str) -> dict:
        store = self.__dict__.setdefault("_bn_pending", {})
        return store.setdefault(prefix, {})

    def _flush_own(self):
        bufs = dict(self.named_buffers())
        for prefix, pend in self.__dict__.get("_bn_pending", {}).items():
            for key, n in pend.items():
                if n:
                    bufs[key[len(prefix):]] += n
            pend.clear()

    def flush_bn_counts(self):
        """Write the host-side BatchNorm call counts of this module and its sub-modules into `num_batches_tracked`."""
        for m in self.modules():
            if isinstance(m, _BNCounts):
                m._flush_own()

    def _discard_own(self, *args):
        """load_state_dict replaces `num_batches_tracked`: counts of forwards that ran before the load must not be added later."""
        for pend in self.__dict__.get("_bn_pending", {}).values():
            pend.clear()

    def _install_count_hook(self):
        self.register_state_dict_pre_hook(lambda module, prefix, keep_vars: module._flush_own())
        self._register_load_state_dict_pre_hook(self._discard_own)


class AdaptivePointNorm(nn.Module):
    """Generator.py:24-45: out = gamma*InstanceNorm(x) + beta, [gamma|beta] = Conv1d(style_dim, 2C, 1)(style) per point.
    forward(input [B,C,N], style [B,S,N]) -> [B,C,N]"""

    def __init__(self, in_channel: int, style_dim: int, use_eql: bool = False):
        super().__init__()
        if use_eql:
            raise NotImplementedError("equalised-LR AdaIN (--eql) is not part of the accelerated path yet")
        self.norm = nn.InstanceNorm1d(in_channel)
        self.style = nn.Conv1d(style_dim, in_channel * 2, 1)
        with torch.no_grad():
            self.style.weight.normal_()
            self.style.bias.zero_()
            self.style.bias[:in_channel] = 1

    def forward_pm(self, x_pm, style_pm, N: int, slope: float = 1.0, style_chain=None, reuse=None):
        """style_chain = (dict, role): two AdaIN layers fed by the SAME style tensor (the generator's adain1 / adain2) hand the style
        gradient along instead of leaving its sum to autograd -- see Fn.AdaINFn.backward.  reuse = (out, ctx): already evaluated
        (nets.g_pair_forward), only the autograd node is built."""
        return Fn.AdaINFn.apply(_Holder(prefix="a", N=N, slope=slope, style_chain=style_chain, reuse=reuse), x_pm, style_pm, self.style.weight,
                                self.style.bias)

    def forward(self, input, style):
        _require_gpu(input, "AdaptivePointNorm")
        B, C, N = input.shape
        out = self.forward_pm(Fn.CmToPm.apply(input), Fn.CmToPm.apply(style), N)
        return Fn.PmToCm.apply(out, B, N)


class EdgeBlock(nn.Module, _BNCounts):
    """Generator.py:47-88.  forward(x [B,Fin,N]) -> [B,Fout,N]."""

    def __init__(self, Fin: int, Fout: int, k: int, attn: bool = True):
        super().__init__()
        self.k, self.Fin, self.Fout = k, Fin, Fout
        self.conv_w = _stack([("conv2d", Fin, Fout // 2), ("bn2d", Fout // 2), ("lrelu",), ("conv2d", Fout // 2, Fout), ("bn2d", Fout), ("lrelu",)])
        self.conv_x = _stack([("conv2d", 2 * Fin, Fout, [1, 1]), ("bn2d", Fout), ("lrelu",)])
        self.conv_out = nn.Conv2d(Fout, Fout, [1, k], [1, 1])
        self.last_idx: Optional[torch.Tensor] = None      # int32 [B*N, k] global rows of the most recent forward
        self._install_count_hook()

    def forward_pm(self, x_pm, B: int, N: int, idx: Optional[torch.Tensor] = None, knn_mode: Optional[int] = None,
                   graph_cache: Optional[dict] = None, count_rep: int = 1, bn_repeats: int = 1, reuse=None, keep: Optional[dict] = None,
                   update_running: bool = True):
        """bn_repeats / reuse / keep: one evaluation standing for several identical forwards (Generator: the two forwards of a train
        step see the same sphere prior and the same weights).  keep: a dict that receives (out, ctx) of this evaluation;
        reuse = (out, ctx): skip the evaluation, return `out` with `ctx` behind it for the backward pass."""
        names, params = _named(self, "e.")
        if knn_mode is None:
            knn_mode = 1 if self.Fin <= 4 else 0          # coordinates: exact fp64 order; features: fp32 expanded form
        if graph_cache is not None and idx is None:
            idx = graph_cache.get("idx")
        h = _Holder(prefix="e", names=names, buffers=_buffers(self, "e."), B=B, N=N, k=self.k, training=self.training,
                    knn_mode=knn_mode, idx=idx, last_idx=None, graph_cache=graph_cache, count_rep=count_rep, bn_repeats=bn_repeats,
                    reuse=reuse, keep=keep, update_running=update_running)
        out = Fn.EdgeBlockFn.apply(h, x_pm, *params)
        self.last_idx = h.last_idx
        if graph_cache is not None and graph_cache.get("idx") is None:
            graph_cache["idx"] = h.last_idx
        return out

    def forward(self, x, idx: Optional[torch.Tensor] = None):
        _require_gpu(x, "EdgeBlock")
        B, C, N = x.shape
        if idx is not None and idx.dtype == torch.int64:
            idx = ops.idx_from_local64(idx, B, N, self.k)
        out = self.forward_pm(Fn.CmToPm.apply(x), B, N, idx)
        return Fn.PmToCm.apply(out, B, N)


class Generator(nn.Module, _BNCounts):
    """Generation/Generator.py:91-198.  forward(x [B,N,3], z [B,N,nz]) -> [B,3,N].
    opts fields read: np, nk, nz, softmax, off, attn, use_head, eql, z_norm."""

    def __init__(self, opts):
        super().__init__()
        self.opts = opts
        self.np = opts.np
        self.nk = opts.nk // 2
        self.nz = opts.nz
        self.off = opts.off
        self.use_attn = opts.attn
        self.use_head = opts.use_head
        eql = bool(getattr(opts, "eql", False))      # Generator.py:103-104: head, global_conv's Linear and pc_head only
        dim = 128
        self.head = _stack([("conv1d", 3 + self.nz, dim), ("lrelu",), ("conv1d", dim, dim), ("lrelu",)], eql)
        if self.use_attn:
            self.attn = Attention(dim + 512)
        self.global_conv = _stack([("linear", dim, dim), ("bn1d", dim), ("lrelu",), ("linear", dim, 512), ("bn1d", 512), ("lrelu",)], eql)
        self.tail = _stack([("conv1d", 512 + dim, 256), ("lrelu",), ("conv1d", 256, 64), ("lrelu",), ("conv1d", 64, 3), ("tanh",)])
        if self.use_head:
            self.pc_head = _stack([("conv1d", 3, dim // 2), ("lrelu",), ("conv1d", dim // 2, dim), ("lrelu",)], eql)
            self.EdgeConv1 = EdgeBlock(dim, dim, self.nk)
            self.adain1 = AdaptivePointNorm(dim, dim)
            self.EdgeConv2 = EdgeBlock(dim, dim, self.nk)
            self.adain2 = AdaptivePointNorm(dim, dim)
        else:
            self.EdgeConv1 = EdgeBlock(3, 64, self.nk)
            self.adain1 = AdaptivePointNorm(64, dim)
            self.EdgeConv2 = EdgeBlock(64, dim, self.nk)
            self.adain2 = AdaptivePointNorm(dim, dim)
        self.lrelu1 = nn.LeakyReLU(nets.NEG_2)
        self.lrelu2 = nn.LeakyReLU(nets.NEG_2)
        self._install_count_hook()

    def _params_of(self, names):
        """The tensors behind reference-style names ('global_conv.0.weight'): parameters, or for equalised-LR layers the
        scaled weight computed on the fly."""
        out = []
        for n in names:
            mod, attr = n.rsplit(".", 1)
            out.append(getattr(self.get_submodule(mod), attr))
        return out

    def _mlp2(self, seq: nn.Sequential, x_pm):
        h = _Holder(names=["l0", "l2"], acts=[ops.ACT_LRELU, ops.ACT_LRELU], slope=NEG)
        return Fn.MLPFn.apply(h, x_pm, seq[0].weight, seq[0].bias, seq[2].weight, seq[2].bias)

    def _style(self, x, z):
        """head(cat[x, z]) (Generator.py:163-169).  z [B,N,nz] as in the reference, or [B,1,nz]: one latent per shape, i.e. what the
        default noise_generator tiles over the points -- then the latent half of head.0 is evaluated once per shape (HeadFn)."""
        B, N, _ = x.shape
        if self.opts.z_norm:
            z = z / (z.norm(p=2, dim=-1, keepdim=True) + 1e-8)
        if z.dim() == 3 and z.shape[1] == 1 and N > 1:
            h0, h2 = self.head[0], self.head[2]
            return Fn.HeadFn.apply(_Holder(N=N), x.reshape(B * N, 3), z.reshape(B, -1), h0.weight, h0.bias, h2.weight, h2.bias)
        hz = ops.concat2(x.reshape(B * N, 3), z.reshape(B * N, -1))
        return self._mlp2(self.head, hz)

    def _sphere_entry(self, x):
        """The cache entry of the sphere prior x [B,N,3]: its kNN graph, the CSR of its in-edges and whether every shape carries the same prior."""
        B = x.shape[0]
        # a few entries (most recent first): the training prior stays cached while an occasional call with another tensor -- a
        # sample dump with its own batch size, an eval forward -- comes and goes (building an entry costs a host sync, which a
        # stream capture that finds its entry evicted could not afford)
        sgs = self.__dict__.setdefault("_sphere_graphs", [])
        key = (x._version, tuple(x.shape), self.nk)
        sg = next((e for e in sgs if e["ref"]() is x and e["key"] == key), None)        # same tensor OBJECT (not just address), unmodified
        if sg is None:
            # Does every shape of the batch carry the SAME prior (sphere_generator(static=True) tiles one template,
            # model.py:169-171)?  Checked once per (tensor, version) -- one host sync when the cache entry is built.
            shared = B > 1 and getattr(self, "dedup_sphere", True) and bool(torch.equal(x, x[:1].expand_as(x)))
            sg = {"ref": weakref.ref(x), "key": key, "idx": None, "csr": None, "shared": shared, "idx_full": None}
            sgs[:] = [e for e in sgs if e["ref"]() is not None and e["ref"]() is not x][:3]
        else:
            sgs[:] = [e for e in sgs if e is not sg]
        sgs.insert(0, sg)
        self.__dict__["_sphere_graph"] = sg
        return sg

    def pair_ok(self, x, z_d, z_g) -> bool:
        """Can forward_pair evaluate G(x, z_d) and G(x, z_g) as one pipeline?  The default generator in train mode, one latent per shape
        ([B,1,nz]), every shape carrying the same prior (its cache entry exists or can be built: not inside a stream capture)."""
        if (self.use_head or self.use_attn or self.off or not self.training or bool(getattr(self.opts, "eql", False))
                or x.dim() != 3 or z_d.shape != z_g.shape or z_d.dim() != 3 or z_d.shape[1] != 1 or x.shape[1] <= 1):
            return False
        return bool(self._sphere_entry(x)["shared"])

    def forward_pair(self, x, z_d, z_g):
        """(G(x, z_d).detach(), G(x, z_g)), both point-major [B*N,3]: the two generator forwards of one train step (model.py:246-248 and
        264-271) as ONE pipeline over the rows of both passes (nets.g_pair_forward) -- same prior, same weights, and the second depends on
        nothing the D step in between produces.  EdgeConv1 is evaluated once for one copy of the prior (as in the twin protocol of _body),
        the per-point / per-shape stages run once on 2B shapes, the BatchNorm stages per pass in the reference's order; the autograd graph of
        the second forward is built from the saved contexts of its rows only.  Values, running statistics and call counts are those of the
        two separate calls."""
        B, N, _ = x.shape
        M = B * N
        cache = self._sphere_entry(x)
        pc = x.reshape(M, 3).contiguous()
        slope = nets.NEG_2
        if self.opts.z_norm:
            z_d = z_d / (z_d.norm(p=2, dim=-1, keepdim=True) + 1e-8)
            z_g = z_g / (z_g.norm(p=2, dim=-1, keepdim=True) + 1e-8)
        self.__dict__["_ec1_twin"] = None
        x1_one = self.EdgeConv1.forward_pm(pc[:N], 1, N, knn_mode=1, graph_cache=cache, count_rep=B, bn_repeats=2)
        if cache["idx_full"] is None:
            off = (torch.arange(B, device=x.device, dtype=torch.int32) * N).view(B, 1, 1)
            cache["idx_full"] = (cache["idx"].view(1, N, -1) + off).reshape(B * N, -1).contiguous()
        self.EdgeConv1.last_idx = cache["idx_full"]
        h0, h2 = self.head[0], self.head[2]
        Ph = {"head.0.weight": h0.weight, "head.0.bias": h0.bias, "head.2.weight": h2.weight, "head.2.bias": h2.bias}
        Pa1 = {"a.style.weight": self.adain1.style.weight, "a.style.bias": self.adain1.style.bias}
        Pa2 = {"a.style.weight": self.adain2.style.weight, "a.style.bias": self.adain2.style.bias}
        en, ep = _named(self.EdgeConv2, "e.")
        gt_params = self._params_of(Fn.GT_NAMES)
        q = self.__dict__.get("_graph2_queue")
        idx2 = [q.pop(0) if q else None, q.pop(0) if q else None]
        idx2 = [None if g is None else (g if g.dtype == torch.int32 else ops.idx_from_local64(g.to(torch.int64).reshape(B, -1), B, N, self.nk)) for g in idx2]
        zg = z_g.reshape(B, -1)
        with torch.no_grad():
            pc2 = cache.get("pc2")                  # the prior's rows for both passes: constant, kept with the prior's cache entry
            if pc2 is None:
                pc2 = pc.repeat(2, 1)
                if not ops.capturing():
                    cache["pc2"] = pc2
            zb2 = torch.cat([z_d.reshape(B, -1), zg], dim=0)
            pre = nets.g_pair_forward({k_: nets.owned(v) for k_, v in Ph.items()}, {k_: nets.owned(v) for k_, v in Pa1.items()},
                                      dict(zip(en, [nets.owned(p) for p in ep])), _buffers(self.EdgeConv2, "e."),
                                      {k_: nets.owned(v) for k_, v in Pa2.items()}, dict(zip(Fn.GT_NAMES, [nets.owned(p) for p in gt_params])),
                                      _buffers(self), pc2, zb2, x1_one.detach(), B, N, self.nk, slope, self.training, idx2=idx2)
        # the autograd graph of the second forward: the usual nodes, adopting their rows and saved contexts
        style = Fn.HeadFn.apply(_Holder(N=N, reuse=pre["head"]), pc, zg, h0.weight, h0.bias, h2.weight, h2.bias)
        x1 = Fn.RepeatRowsFn.apply(x1_one, B, pre["x1"])
        chain = {} if torch.is_grad_enabled() else None
        a1 = self.adain1.forward_pm(x1, style, N, slope, style_chain=None if chain is None else (chain, "last"), reuse=pre["adain1"])
        self.last_x1 = pre["x1_in"].detach()
        x2 = self.EdgeConv2.forward_pm(a1, B, N, knn_mode=0, reuse=pre["ec2"])
        a2 = self.adain2.forward_pm(x2, style, N, slope, style_chain=None if chain is None else (chain, "first"), reuse=pre["adain2"])
        self.last_x2 = pre["a2"].detach()
        h = _Holder(buffers=_buffers(self), B=B, N=N, training=self.training, reuse=pre["tail"])
        out = Fn.GlobalTailFn.apply(h, a2, *gt_params)
        self.__dict__["_pair_idx"] = pre["idx"]
        return pre["fake_d"], out

    def _body(self, x, style, pm_out: bool = False):
        B, N, _ = x.shape
        pc = x.reshape(B * N, 3).contiguous()
        feat = self._mlp2(self.pc_head, pc) if self.use_head else pc
        slope = nets.NEG_2
        # The sphere prior is the same tensor every step (Generation/model.py:231): its kNN graph (and the CSR of
        # its in-edges) is built once per (buffer, version, shape) and reused.  Any in-place write
        # to x bumps _version and invalidates the cache.
        cache = None if self.use_head else self._sphere_entry(x)
        if cache is not None and cache["shared"]:
            # EdgeConv1 sees the same N points in every shape: evaluate it for ONE copy (B times less work in forward and
            # backward; exact -- batch statistics of identical copies are those of one copy, and the backward is linear in the
            # upstream gradient, which RepeatRowsFn sums over the copies) and repeat the rows for the per-shape AdaIN.
            # TrainStep announces that this forward and the next one see the same prior and the same weights (D step, then G step):
            # EdgeConv1 is evaluated once, its running statistics advanced twice (`twin_forward`: "first" / "second")
            twin = getattr(self, "twin_forward", None)
            saved = self.__dict__.get("_ec1_twin")
            stamp = (id(cache), tuple(p._version for p in self.EdgeConv1.parameters()), ops.weights_epoch_of(next(self.EdgeConv1.parameters())), B, N)
            if twin == "second" and saved is not None and saved["stamp"] == stamp and self.training:
                x1_one = self.EdgeConv1.forward_pm(feat[:N], 1, N, knn_mode=1, graph_cache=cache, count_rep=B, reuse=(saved["out"], saved["ctx"]))
                self.__dict__["_ec1_twin"] = None
            elif twin == "first" and self.training:
                keep = {"stamp": stamp}
                x1_one = self.EdgeConv1.forward_pm(feat[:N], 1, N, knn_mode=1, graph_cache=cache, count_rep=B, bn_repeats=2, keep=keep)
                self.__dict__["_ec1_twin"] = keep
            elif twin == "second" and saved is not None and self.training:
                # announced as the twin of an earlier forward that already advanced EdgeConv1's running statistics for BOTH, but its
                # output cannot be reused (the weights, the prior or the batch changed in between): evaluate, do not advance them a
                # third time
                self.__dict__["_ec1_twin"] = None
                x1_one = self.EdgeConv1.forward_pm(feat[:N], 1, N, knn_mode=1, graph_cache=cache, count_rep=B, update_running=False)
            else:
                if twin is not None:                 # a forward outside the twin protocol (eval call, sample dump) leaves a pending twin alone
                    self.__dict__["_ec1_twin"] = None
                x1_one = self.EdgeConv1.forward_pm(feat[:N], 1, N, knn_mode=1, graph_cache=cache, count_rep=B)
            if cache["idx_full"] is None:
                off = (torch.arange(B, device=x.device, dtype=torch.int32) * N).view(B, 1, 1)
                cache["idx_full"] = (cache["idx"].view(1, N, -1) + off).reshape(B * N, -1).contiguous()
            self.EdgeConv1.last_idx = cache["idx_full"]
            x1 = Fn.RepeatRowsFn.apply(x1_one, B)
        else:
            x1 = self.EdgeConv1.forward_pm(feat, B, N, knn_mode=0 if self.use_head else 1, graph_cache=cache)
        # both AdaIN layers read the same style tensor: adain2's backward (which runs first) stashes its style gradient, adain1's adds it
        # in the epilogue of its own style-gradient GEMM -- instead of an autograd add over a [B*N, 128] tensor
        chain = {} if (style.requires_grad and torch.is_grad_enabled()) else None
        x1 = self.adain1.forward_pm(x1, style, N, slope, style_chain=None if chain is None else (chain, "last"))    # lrelu1 fused into the instance norm (Generator.py:175-176)
        self.last_x1 = x1.detach()                                 # [M,64] input of EdgeConv2's graph (diagnostics / parity tests)
        # Tie-aware parity protocol: a caller may hand over EdgeConv2's kNN graph(s) for the next forward(s) --
        # `inject_graph2([idx, ...])`, int32 [B*N,k] global rows or the reference's int64 [B,N*k] local indices -- e.g. the graph
        # the reference itself built on this stage, so that everything behind the discrete choice is compared tightly.
        q = self.__dict__.get("_graph2_queue")
        idx2 = q.pop(0) if q else None
        if idx2 is not None and idx2.dtype != torch.int32:
            idx2 = ops.idx_from_local64(idx2.to(torch.int64).reshape(B, -1), B, N, self.nk)
        x2 = self.EdgeConv2.forward_pm(x1, B, N, idx=idx2, knn_mode=0)
        x2 = self.adain2.forward_pm(x2, style, N, slope, style_chain=None if chain is None else (chain, "first"))
        self.last_x2 = x2.detach()                                 # [M,128] adain2's output (parity tests)
        h = _Holder(buffers=_buffers(self), B=B, N=N, training=self.training)
        if self.use_attn:
            feat = Fn.GlobalFeatFn.apply(h, x2, *self._params_of(Fn.GF_NAMES))                    # [M,640], Generator.py:183-189
            feat = self.attn.forward_pm(feat, B, N)                                              # Generator.py:191-192
            th = _Holder(names=["tail.0", "tail.2", "tail.4"], acts=[ops.ACT_LRELU, ops.ACT_LRELU, ops.ACT_TANH], slope=NEG)
            out = Fn.MLPFn.apply(th, feat, *self._params_of(Fn.GT_NAMES[len(Fn.GF_NAMES):]))
        else:
            out = Fn.GlobalTailFn.apply(h, x2, *self._params_of(Fn.GT_NAMES))
        if pm_out and not self.off:
            return out                              # [B*N,3] point-major: TrainStep's internal route hands it to the Discriminator as it is
        out = Fn.PmToCm.apply(out, B, N)
        return x.transpose(2, 1) + out if self.off else out

    def forward(self, x, z, pm_out: bool = False):
        """pm_out=True (TrainStep's internal route; not with --off): return the cloud point-major [B*N,3], i.e. without the final layout
        change to the reference's [B,3,N] -- Discriminator.forward_stacks_grouped / forward(..., pm_shape=(B,N)) take it as it is."""
        _require_gpu(x, "Generator")
        return self._body(x, self._style(x, z), pm_out=pm_out)

    def inject_graph2(self, graphs) -> None:
        """EdgeConv2's neighbour graph for the next len(graphs) forwards, consumed in order (see _body).  None = build it."""
        self.__dict__["_graph2_queue"] = [None if g is None else g.to(next(self.parameters()).device) for g in graphs]

    def interpolate(self, x, z1, z2, selection, alpha, use_latent: bool = False):
        """Generator.py:200-261: blend two latents (or two styles) on the selected points."""
        sel = selection == 1
        if not use_latent:
            z = z1
            z[:, sel] = z1[:, sel] * (1 - alpha) + z2[:, sel] * alpha
            style = self._style(x, z)
        else:
            s1, s2 = self._style(x, z1), self._style(x, z2)
            B, N, _ = x.shape
            s1 = s1.view(B, N, -1); s2 = s2.view(B, N, -1)
            s1[:, sel] = s1[:, sel] * (1 - alpha) + s2[:, sel] * alpha
            style = s1.reshape(B * N, -1)
        return self._body(x, style)


class Discriminator(nn.Module, _BNCounts):
    """Generation/Discriminator.py:48-115.  forward(x [B,3,N]) -> [B,1].  Supports autograd.grad(create_graph=True)
    w.r.t. its input followed by backward() (WGAN-GP)."""

    def __init__(self, opts, num_point: int = 2048):
        super().__init__()
        self.num_point = num_point
        self.small_d = opts.small_d
        self.mlps = _stack([("conv1d", 3, 64), ("bn1d", 64), ("lrelu",), ("conv1d", 64, 128), ("bn1d", 128), ("lrelu",),
                            ("conv1d", 128, 256), ("bn1d", 256), ("lrelu",)])
        self.mode = "max"
        dim = 512 if self.small_d else 1024
        self.fc2 = _stack([("conv1d", 256, dim), ("bn1d", dim), ("lrelu",)])
        self.mlp = _stack([("linear", dim, 512), ("lrelu",), ("linear", 512, 256), ("lrelu",), ("linear", 256, 64), ("lrelu",), ("linear", 64, 1)])
        self._install_count_hook()

    def forward(self, x, pre=None):
        """pre: this input's entry of `forward_stacks_grouped(...)` -- its conv stack was evaluated there, only the head runs here."""
        _require_gpu(x, "Discriminator")
        names, params = _named(self)
        h = _Holder(names=names, buffers=_buffers(self), training=self.training, pre=pre)
        return Fn.DiscriminatorFn.apply(h, x.contiguous(), *params)

    def forward_stack_after_stats_pass(self, stats_x, x, pm_shape=None):
        """The side effect of a train-mode D(stats_x) whose result nobody reads (`advance_running_stats`) followed by the conv stack of
        D(x), the shared first three layers of the two passes as one batch (nets.d_forward_after_stats_pass): the G step's
        D(real); D(G(z)).  Returns the entry to hand to `forward(x, pre=...)`."""
        _require_gpu(x, "Discriminator"); _require_gpu(stats_x, "Discriminator")
        if not self.training:
            raise RuntimeError("forward_stack_after_stats_pass is a train-mode path (batch statistics)")
        from . import nets
        names, params = _named(self)
        with torch.no_grad():
            P = dict(zip(names, [nets.owned(p) for p in params]))
            return nets.d_forward_after_stats_pass(P, _buffers(self), stats_x.detach(), x.detach(), pm_shape=pm_shape)

    def forward_stacks_grouped(self, xs, pm_shape=None):
        """The conv stacks of several train-mode passes as ONE batch (nets.d_forward_groups): one GEMM and one finalize launch per layer
        for all of them, per-pass BatchNorm statistics, running statistics advanced in list order -- bit-identical to separate calls.
        Returns one opaque entry per input, to be handed to `forward(x, pre=...)` or `forward_stack(x, pre=...)` of the same input."""
        for x in xs:
            _require_gpu(x, "Discriminator")
        if not self.training:
            raise RuntimeError("forward_stacks_grouped is a train-mode path (batch statistics)")
        from . import nets
        names, params = _named(self)
        with torch.no_grad():
            P = dict(zip(names, [nets.owned(p) for p in params]))
            return nets.d_forward_groups(P, _buffers(self), [x.detach() for x in xs], pm_shape=pm_shape)


def _discriminator_forward_many(self, *xs):
    """[D(x) for x in xs] with the per-shape MLP head of all passes evaluated as ONE batch: the conv stacks run one after the other
    (each with its own train-mode BatchNorm statistics and running-statistics update, in call order -- exactly what separate calls
    do), the head has no BatchNorm, so its rows are independent and cat[pooled...] goes through it once (4 launches forward and 16
    backward once instead of once per pass).  First-order only: the WGAN-GP route uses forward()."""
    return self.forward_heads([self.forward_stack(x) for x in xs])


def _discriminator_forward_stack(self, x, pre=None):
    """The conv stack up to the max-pool: x [B,3,N] -> pooled [B,C4] (train-mode BatchNorm statistics and running-statistics update
    of this pass).  pre: this input's entry of forward_stacks_grouped(...) (already evaluated there)."""
    _require_gpu(x, "Discriminator")
    sn, sp = _named(self.mlps, "mlps.")
    fn, fp = _named(self.fc2, "fc2.")
    h = _Holder(names=sn + fn, buffers=_buffers(self), training=self.training, pre=pre)
    return Fn.DStackFn.apply(h, x.contiguous(), *(sp + fp))


def _discriminator_forward_heads(self, pooled):
    """The per-shape MLP head of several forward_stack() results as one batch -> one logits tensor per pass."""
    hn, hp = _named(self.mlp, "mlp.")
    pooled = list(pooled)
    joined = pooled[0] if len(pooled) == 1 else Fn.JoinRowsFn.apply(*pooled)       # a view when one grouped launch produced the passes
    outs = Fn.DHeadFn.apply(_Holder(names=hn, sizes=[p.shape[0] for p in pooled]), joined, *hp)
    return list(outs)


def _discriminator_stacks_joint(self, firsts, hat=None):
    """The D step's passes behind one autograd node (Fn.DStacksJointFn): firsts / hat are entries of forward_stacks_grouped(...).
    -> [pooled of every first-order pass ..., gx] with gx = d sum(D(x_hat)) / d x_hat of the `hat` entry (differentiable once more: its
    backward is the penalty's double backward).  The backward work of all these passes then runs in lock step, every layer's launch issued
    once (nets.d_backward_joint) -- results as from forward_stack(pre=...) per pass and the WGAN-GP route of forward(pre=...)."""
    names, params = _named(self)
    h = _Holder(names=names, firsts=list(firsts), hat=hat)
    return list(Fn.DStacksJointFn.apply(h, *params))


def _discriminator_joint_ok(self, entries) -> bool:
    from . import nets
    names, params = _named(self)
    return nets.d_joint_ok(dict(zip(names, params)), [c for _, c in entries])


Discriminator.stacks_joint = _discriminator_stacks_joint
Discriminator.joint_ok = _discriminator_joint_ok
Discriminator.forward_many = _discriminator_forward_many
Discriminator.forward_stack = _discriminator_forward_stack
Discriminator.forward_heads = _discriminator_forward_heads


def _discriminator_advance_running_stats(self, x):
    """Train-mode D(x) reduced to its only lasting effect when the logits are not used: the BatchNorm running statistics and
    call counts (TrainStep uses it for the reference's D(real) call inside the G step, Generation/model.py:272-273)."""
    _require_gpu(x, "Discriminator")
    if not self.training:
        return
    with torch.no_grad():
        nets.d_advance_running_stats(dict(self.named_parameters()), _buffers(self), x.contiguous())


Discriminator.advance_running_stats = _discriminator_advance_running_stats


def get_edge_features(x, k, num=-1, idx=None, return_idx=False):
    """Generation/modules.py:683-725: x [B,C,N] -> ee [B,2C,N,k] (= cat[central, neighbour-central]); optional injected /
    returned idx is int64 [B, N*k] with per-shape local indices, like the reference.  Differentiable in x like the reference's
    gather/concat (the indices carry no gradient): the backward sums, per point, its own k central / difference terms and the
    difference terms of the edges that gathered it, in edge order (Fn.EdgeFeaturesFn; no float atomics).  The Generator path itself
    never materialises ee -- EdgeBlock carries its own fused backward."""
    _require_gpu(x, "get_edge_features")
    B, C, N = x.shape
    x = x.contiguous()
    if idx is None:
        with torch.no_grad():
            gidx = ops.knn(ops.cm_to_pm(x.detach()), B, N, k, 1 if C <= 4 else 0)
            idx = ops.idx_to_local64(gidx, B, N)
    idx = idx.contiguous().view(B, N * k)
    if x.requires_grad and torch.is_grad_enabled():
        ee = Fn.EdgeFeaturesFn.apply(x, idx, k)
    else:
        ee = ops.edge_features_cm(x, idx, k)
    return (ee, idx) if return_idx else ee
