"""GAN losses and the WGAN-GP penalty with the reference's call signatures.

  dis_loss / gen_loss : Common/loss_utils.py:854-972 / 727-802  -> (loss, info-dict)
  GradientPenalty     : Common/gradient_penalty.py:4-37         -> callable(netD, real, fake)

Value and logit-gradients come from one HIP launch (spgan_gan_loss); the penalty's norm,
value and gradient are HIP kernels as well.  Unlike the reference, nothing here forces a
device->host sync: the info dict holds 0-dim device tensors (the reference calls .item()).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch.autograd import Function

from . import ops
from .functions import input_grad_only


class _GanLossFn(Function):
    @staticmethod
    def forward(ctx, mode, which, d_real, d_fake, real_label, fake_label):
        out5, g_real, g_fake = ops.gan_loss(mode, which, d_real if which == 0 else None, d_fake, real_label, fake_label)
        ctx.save_for_backward(g_real, g_fake)
        ctx.which = which
        ctx.mark_non_differentiable(out5)
        return out5[0].clone(), out5

    @staticmethod
    def backward(ctx, gl, _):
        g_real, g_fake = ctx.saved_tensors
        gr = g_real * gl if (ctx.which == 0 and ctx.needs_input_grad[2]) else None
        gf = g_fake * gl if ctx.needs_input_grad[3] else None
        return None, None, gr, gf, None, None


def _smooth_labels(B, device, ran=(0.9, 1.0)):
    """loss_utils.py:698-700: labels drawn uniformly from `ran` -- on the device (torch's generator; capturable in a hipGraph,
    every replay draws fresh values), where the reference draws with host numpy."""
    return (ran[1] - ran[0]) * torch.rand(B, device=device) + ran[0]


def _noisy_labels(y, p_flip=0.05):
    """loss_utils.py:718-725: int(p_flip*B) positions drawn with replacement, y[ix] = 1 - y[ix] (a position drawn twice flips once)."""
    n = int(p_flip * y.shape[0])
    if n == 0:
        return y
    ix = torch.randint(0, y.shape[0], (n,), device=y.device)
    return y.index_put((ix,), 1 - y[ix])


_ONES = {}


def _ones_like(t: torch.Tensor) -> torch.Tensor:
    """A cached all-ones seed of t's shape (constant: created once per shape and device)."""
    key = (tuple(t.shape), str(t.device))
    o = _ONES.get(key)
    if o is None:
        o = torch.ones_like(t)
        if not ops.capturing():        # see nets._const_vec
            _ONES[key] = o
    return o


def _mode(gan: str) -> int:
    g = gan.lower()
    if g not in ops.GAN_MODES:
        raise NotImplementedError("Not implement: %s" % gan)          # same message as loss_utils.py:972
    return ops.GAN_MODES[g]


def dis_loss(d_real, d_fake, gan="wgan", weight=1., d_real_p=None, d_fake_p=None, noise_label=False,
             real_label: Optional[torch.Tensor] = None, fake_label: Optional[torch.Tensor] = None):
    """Discriminator loss; `noise_label` draws smoothed/flipped real labels (loss_utils.py:897-901) on the device, so a captured
    train step draws new ones on every replay.  Explicit label tensors [B] may be passed instead (tests); the labels used are
    returned in the info dict."""
    if d_real_p is not None or d_fake_p is not None:
        raise NotImplementedError("patch logits (d_real_p/d_fake_p) are not produced by this Discriminator")
    mode = _mode(gan)
    B = d_fake.shape[0]
    if mode == 0 and noise_label and real_label is None:
        real_label = _noisy_labels(_smooth_labels(B, d_fake.device))
    loss, out5 = _GanLossFn.apply(mode, 0, d_real, d_fake, real_label, fake_label)
    if weight != 1.:
        loss = loss * weight
    info = {"loss": loss.detach(), "loss_fake": out5[1], "loss_real": out5[2], "real_acc": out5[3], "fake_acc": out5[4]}
    if real_label is not None:
        info["real_label"] = real_label
    return loss, info


def gen_loss(d_real, d_fake, gan="wgan", weight=1., d_real_p=None, d_fake_p=None, noise_label=False,
             fake_label: Optional[torch.Tensor] = None):
    """Generator loss (d_real is accepted and unused, as in the reference for ls/wgan/hinge/gan)."""
    if d_real_p is not None or d_fake_p is not None:
        raise NotImplementedError("patch logits (d_real_p/d_fake_p) are not produced by this Discriminator")
    mode = _mode(gan)
    B = d_fake.shape[0]
    if mode == 0 and noise_label and fake_label is None:
        fake_label = _noisy_labels(torch.ones(B, device=d_fake.device))                                  # loss_utils.py:753-755
    loss, out5 = _GanLossFn.apply(mode, 1, d_real, d_fake, None, fake_label)
    if weight != 1.:
        loss = loss * weight
    info = {"loss": loss.detach(), "g_loss": out5[1]}
    if fake_label is not None:
        info["fake_label"] = fake_label
    return loss, info


def dis_loss_with_grads(d_real, d_fake, gan="wgan", noise_label=False):
    """dis_loss evaluated once with its gradients w.r.t. the two logit tensors: -> (out5 [loss, fake term, real term, real_acc,
    fake_acc], g_real, g_fake).  For callers that seed the backward pass themselves (spgan.TrainStep: torch.autograd.backward(
    [d_real, d_fake], [g_real, g_fake]) -- no loss clone, no multiplication of the gradients by the implicit seed 1)."""
    mode = _mode(gan)
    real_label = None
    if mode == 0 and noise_label:
        real_label = _noisy_labels(_smooth_labels(d_fake.shape[0], d_fake.device))
    return ops.gan_loss(mode, 0, d_real.detach(), d_fake.detach(), real_label, None)


def gen_loss_with_grads(d_fake, gan="wgan", noise_label=False):
    """gen_loss the same way: -> (out5 [loss, g_loss, ...], g_fake)."""
    mode = _mode(gan)
    fake_label = None
    if mode == 0 and noise_label:
        fake_label = _noisy_labels(torch.ones(d_fake.shape[0], device=d_fake.device))
    out5, _, g_fake = ops.gan_loss(mode, 1, None, d_fake.detach(), None, fake_label)
    return out5, g_fake


class _GPPenaltyFn(Function):
    @staticmethod
    def forward(ctx, g, gamma, lam):
        loss, norms = ops.gp_penalty_fwd(g, gamma, lam)
        ctx.save_for_backward(g, norms)
        ctx.gamma, ctx.lam = gamma, lam
        return loss[0].clone()

    @staticmethod
    def backward(ctx, up):
        g, norms = ctx.saved_tensors
        return ops.gp_penalty_bwd(g, norms, ctx.gamma, ctx.lam, up), None, None


class GradientPenalty:
    """WGAN-GP penalty, Common/gradient_penalty.py:4-37:
        alpha ~ U[0,1] per sample; x_hat = real + alpha*(fake-real); g = d netD(x_hat)/d x_hat (create_graph);
        penalty = lambdaGP * mean(((||g_b||_2 - gamma)/gamma)^2).
    `netD` must be an spgan.Discriminator (its input-gradient node is differentiable once more).

    The second GradientPenalty of the reference (Common/loss_utils.py:1087-1131) differs in two ways, both available here:
    `mix="loss_utils"` interpolates alpha*real + (1-alpha)*fake (:1108), and `mapping=True` (:1110-1118) first pairs every fake point
    with a real point by the auction EMD (`emdModule()(fake, real, 0.005, 300)`, spgan.metrics) and interpolates along those pairs:
    alpha*fake + (1-alpha)*real[assignment]."""

    def __init__(self, lambdaGP, gamma=1, vertex_num=2500, device=None, mix: str = "common"):
        if mix not in ("common", "loss_utils"):
            raise ValueError("mix must be 'common' (gradient_penalty.py) or 'loss_utils'")
        self.lambdaGP = lambdaGP
        self.gamma = gamma
        self.vertex_num = vertex_num
        self.device = device
        self.mix = mix

    def _mapped(self, real_data, fake_data, alpha):
        from . import metrics
        B, C, N = real_data.shape
        fake_pm, real_pm = ops.cm_to_pm(fake_data), ops.cm_to_pm(real_data)                   # [B*N,3]
        _, ass = metrics.emdFunction.apply(fake_pm.view(B, N, C), real_pm.view(B, N, C), 0.005, 300)
        rows = (ass + (torch.arange(B, device=ass.device, dtype=torch.int32) * N).view(B, 1)).reshape(B * N, 1)
        paired = ops.gather_rows(real_pm, rows.expand(B * N, C).contiguous())                   # real[b][assignment[b]] (same row for x, y, z)
        mixed = ops.lerp_rows(paired.view(B, N * C), fake_pm.view(B, N * C), alpha.reshape(B))  # paired + alpha*(fake - paired)
        return ops.pm_to_cm(mixed.view(B * N, C), B, N)

    def __call__(self, netD, real_data, fake_data, alpha: Optional[torch.Tensor] = None, mapping: bool = False):
        grads = self.input_gradient(netD, real_data, fake_data, alpha, mapping)
        return _GPPenaltyFn.apply(grads.contiguous().view(grads.shape[0], -1), float(self.gamma), float(self.lambdaGP))

    def with_grads(self, netD, real_data, fake_data, alpha: Optional[torch.Tensor] = None, mapping: bool = False, interpolates=None, pre=None):
        """-> (penalty [1], grads, v): the penalty's value, the differentiable input gradient it is a function of, and
        d penalty / d grads -- for a caller that seeds torch.autograd.backward([grads], [v]) itself.
        interpolates / pre: the x_hat of `interpolate(...)` and its entry of netD.forward_stacks_grouped([..., x_hat]) when the caller
        evaluated D's conv stack on x_hat together with other passes."""
        grads = self.input_gradient(netD, real_data, fake_data, alpha, mapping, interpolates=interpolates, pre=pre)
        g = grads.detach().contiguous().view(grads.shape[0], -1)
        loss, norms = ops.gp_penalty_fwd(g, float(self.gamma), float(self.lambdaGP))
        v = ops.gp_penalty_bwd(g, norms, float(self.gamma), float(self.lambdaGP), None)
        return loss, grads, v.view_as(grads)

    def with_grads_pm(self, netD, x_hat_pm, pre, B: int):
        """with_grads for TrainStep's point-major route: x_hat_pm [B*N,3] = interpolates (already evaluated by
        netD.forward_stacks_grouped(..., pm_shape=(B,N)) as `pre`); the input gradient comes back point-major, the penalty's per-shape norm
        runs over the same 3N values of a shape in either layout."""
        x_hat_pm = x_hat_pm.requires_grad_(True)
        with input_grad_only():
            disc = netD(x_hat_pm, pre=pre)
            grads = torch.autograd.grad(outputs=disc, inputs=x_hat_pm, grad_outputs=_ones_like(disc), create_graph=True, retain_graph=True,
                                        only_inputs=True)[0]
        g = grads.detach().contiguous().view(B, -1)
        loss, norms = ops.gp_penalty_fwd(g, float(self.gamma), float(self.lambdaGP))
        v = ops.gp_penalty_bwd(g, norms, float(self.gamma), float(self.lambdaGP), None)
        return loss, grads, v.view_as(grads)

    def from_input_gradient(self, grads: torch.Tensor, B: int, loss_add: Optional[torch.Tensor] = None):
        """(penalty [1], seed, total) for an input gradient that is already there (TrainStep's joint D-step node, Discriminator.stacks_joint): the
        penalty of gradient_penalty.py:31-35, v = d penalty / d grads (the seed of the double backward) and, with loss_add [1], total = loss_add +
        penalty from the same launch (else None)."""
        g = grads.detach().contiguous().view(B, -1)
        loss, _norms, v, total = ops.gp_penalty_fwd_bwd(g, float(self.gamma), float(self.lambdaGP), loss_add)
        return loss, v.view_as(grads), total

    def interpolate(self, real_data, fake_data, alpha: Optional[torch.Tensor] = None, mapping: bool = False):
        """x_hat [B,3,N] (detached): the points the penalty is evaluated at."""
        B = real_data.size(0)
        fake_data = fake_data[:B]
        if alpha is None:
            alpha = torch.rand(B, 1, 1, device=real_data.device)
        real_d, fake_d = real_data.detach().contiguous(), fake_data.detach().contiguous()
        if mapping:
            return self._mapped(real_d, fake_d, alpha)
        if self.mix == "loss_utils":
            return ops.lerp_rows(fake_d, real_d, alpha.reshape(B))                               # fake + alpha*(real - fake)
        return ops.lerp_rows(real_d, fake_d, alpha.reshape(B))

    def input_gradient(self, netD, real_data, fake_data, alpha: Optional[torch.Tensor] = None, mapping: bool = False, interpolates=None, pre=None):
        """d netD(x_hat) / d x_hat on the interpolates, as a differentiable tensor (create_graph)."""
        if interpolates is None:
            interpolates = self.interpolate(real_data, fake_data, alpha, mapping)
        interpolates = interpolates.requires_grad_(True)
        with input_grad_only():          # explicit: the node this forward builds delivers d disc / d x_hat only, as a differentiable node
            disc = netD(interpolates) if pre is None else netD(interpolates, pre=pre)
            grads = torch.autograd.grad(outputs=disc, inputs=interpolates, grad_outputs=_ones_like(disc),
                                        create_graph=True, retain_graph=True, only_inputs=True)[0]
        return grads
