"""One iteration of the reference's train loop body (Generation/model.py:239-279) over the HIP modules:
D-step (G frozen, no graph through G), then G-step (D frozen: input gradients only), with the
reference's call order -- including the unused D(real) forward in the G-step that advances D's
BatchNorm running statistics -- and Adam(lr, betas=(0.5, 0.99)) (model.py:94-97).

With `use_gp` the discriminator loss is dis_loss(gan) + GradientPenalty(lambda_gp, gamma=1): the
"WGAN-GP" composition of reference pieces that BASELINE config 2 names.
No host<->device synchronisation happens inside step(); losses are returned as device tensors.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from . import ops
from .losses import GradientPenalty, dis_loss, gen_loss
from .optim import Adam
from .parallel import DataParallel


def requires_grad(model: nn.Module, flag: bool = True):
    """Common/network_utils.py:92-94"""
    for p in model.parameters():
        p.requires_grad = flag


class TrainStep:
    def __init__(self, G: nn.Module, D: nn.Module, gan: str = "ls", use_gp: bool = False, lambda_gp: float = 10.0,
                 lr_g: float = 1e-4, lr_d: float = 1e-4, betas=(0.5, 0.99), flip_d: bool = False, flip_g: bool = False,
                 distributed: bool = False, process_group=None):
        self.G, self.D = G, D
        self.gan, self.use_gp = gan, use_gp
        self.flip_d, self.flip_g = flip_d, flip_g
        self.gp = GradientPenalty(lambda_gp, gamma=1)
        self.dpG = DataParallel(G, process_group) if distributed else None
        self.dpD = DataParallel(D, process_group) if distributed else None
        if distributed:
            self.dpG.sync_params(); self.dpD.sync_params()
        self.optG = Adam(G, lr_g, betas)
        self.optD = Adam(D, lr_d, betas)
        G.train(); D.train()

    def step(self, x: torch.Tensor, real: torch.Tensor, z_d: torch.Tensor, z_g: torch.Tensor, alpha: Optional[torch.Tensor] = None,
             keep_grads: bool = False) -> Dict[str, torch.Tensor]:
        """x: sphere [B,N,3]; real [B,N,3]; z_d, z_g [B,N,nz] (noise for the D- and the G-step)."""
        G, D = self.G, self.D
        B, N, _ = real.shape
        info: Dict[str, torch.Tensor] = {}
        # ------------------------------------------------------------ D step (model.py:240-260)
        requires_grad(G, False); requires_grad(D, True)
        self.optD.zero_grad()
        fake = G(x, z_d).detach()
        real_t = ops.pm_to_cm(real.reshape(B * N, 3), B, N)                      # real_points.transpose(2,1)
        d_real = D(real_t)
        d_fake = D(fake)
        loss_d, dinfo = dis_loss(d_real, d_fake, gan=self.gan, noise_label=self.flip_d)
        if self.use_gp:
            loss_d = loss_d + self.gp(D, real_t, fake, alpha=alpha)
        loss_d.backward()
        scale = self.dpD.allreduce_grads() if self.dpD is not None else 1.0
        if keep_grads:
            info["d_grads"] = {n: p.grad.detach().clone() * scale for n, p in D.named_parameters()}
            info["fake_d"] = fake
        self.optD.step(scale)
        # ------------------------------------------------------------ G step (model.py:264-279)
        requires_grad(G, True); requires_grad(D, False)
        self.optG.zero_grad()
        g_fake = G(x, z_g)
        g_real_logit = D(real_t)
        g_fake_logit = D(g_fake)
        loss_g, _ = gen_loss(g_real_logit, g_fake_logit, gan=self.gan, noise_label=self.flip_g)
        loss_g.backward()
        scale = self.dpG.allreduce_grads() if self.dpG is not None else 1.0
        if keep_grads:
            info["g_grads"] = {n: p.grad.detach().clone() * scale for n, p in G.named_parameters()}
            info["fake_g"] = g_fake.detach()
        self.optG.step(scale)
        info.update(loss_d=loss_d.detach(), loss_g=loss_g.detach(), real_acc=dinfo["real_acc"], fake_acc=dinfo["fake_acc"])
        return info
