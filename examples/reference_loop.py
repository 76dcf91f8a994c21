"""The reference's train-loop body (Generation/model.py:239-279), statement for statement, over whatever `Generator`,
`Discriminator`, loss functions and optimisers the caller hands in -- i.e. what `Model.train` executes per iteration once its
imports point at this package.  Nothing here knows about spgan's own harness (`TrainStep`): it is the CALLER's
code, kept under examples/ (not in the shipped package) so that the parity test (tests/test_literal_loop_gpu.py), `bench.py`'s
`literal_loop` leg and `examples/train.py` run the very same statements.

    state = LoopState(G, D, optimizerG, optimizerD, gan="ls")
    lossD, lossG, info = reference_loop_body(state, x, data, z_d, z_g)

Differences from the file, all forced by making it a function: `self.` is `s.`, the two `noise_generator` draws arrive as
arguments (the caller's sampler), `data` is already on the device, and the `.item()` meter updates (model.py:281-285) stay with
the caller (they are host synchronisations and do not belong to a capturable body).  `gp` (None by default, as in the reference
loop) adds the WGAN-GP composition of SURVEY 8(a)7: `lossD + GradientPenalty(lambdaGP)(D, real, fake)`.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
from torch.autograd import Variable

from spgan.train import requires_grad


class LoopState:
    def __init__(self, G, D, optimizerG, optimizerD, gan: str = "ls", flip_d: bool = False, flip_g: bool = False,
                 dis_loss: Optional[Callable] = None, gen_loss: Optional[Callable] = None, gp: Optional[Callable] = None,
                 gp_kwargs: Optional[dict] = None):
        from spgan import losses
        self.G, self.D, self.optimizerG, self.optimizerD = G, D, optimizerG, optimizerD
        self.gan, self.flip_d, self.flip_g = gan, flip_d, flip_g
        self.dis_loss = dis_loss or losses.dis_loss
        self.gen_loss = gen_loss or losses.gen_loss
        self.gp, self.gp_kwargs = gp, (gp_kwargs or {})
        self.keep = None           # a dict: the body leaves its intermediate tensors there (tests)


def _to_device(t, like):
    """`.cuda()` of model.py:249 (a no-op for a tensor that already lives on the device of the prior)."""
    return t if t.device == like.device else t.to(like.device)


def reference_loop_body(s: LoopState, x, data, z_d, z_g):
    requires_grad(s.G, False)                                                   # model.py:240
    requires_grad(s.D, True)                                                    # :241

    s.optimizerD.zero_grad()                                                    # :243

    real_points = Variable(data, requires_grad=True)                            # :245
    z = z_d                                                                     # :246  self.noise_generator(bs=self.opts.bs)

    d_fake_preds = s.G(x, z)                                                    # :248
    real_points = _to_device(real_points.transpose(2, 1), x)                          # :249
    d_fake_preds = d_fake_preds.detach()                                        # :250

    d_real_logit = s.D(real_points)                                             # :253
    d_fake_logit = s.D(d_fake_preds)                                            # :254

    lossD, info = s.dis_loss(d_real_logit, d_fake_logit, gan=s.gan, noise_label=s.flip_d)     # :257
    if s.gp is not None:                                                        # WGAN-GP composition (not wired in the file)
        lossD = lossD + s.gp(s.D, real_points, d_fake_preds, **s.gp_kwargs)

    lossD.backward()                                                            # :259
    if s.keep is not None:
        s.keep["fake_d"] = d_fake_preds
        s.keep["d_grads"] = {n: p.grad.detach().clone() for n, p in s.D.named_parameters()}
    s.optimizerD.step()                                                         # :260

    # -----------------------------------train G-----------------------------------

    requires_grad(s.G, True)                                                    # :264
    requires_grad(s.D, False)                                                   # :265

    s.optimizerG.zero_grad()                                                    # :268

    z = z_g                                                                     # :270
    g_fake_preds = s.G(x, z)                                                    # :271

    g_real_logit = s.D(real_points)                                             # :274
    g_fake_logit = s.D(g_fake_preds)                                            # :275
    lossG, _ = s.gen_loss(g_real_logit, g_fake_logit, gan=s.gan, noise_label=s.flip_g)        # :276

    lossG.backward()                                                            # :278
    if s.keep is not None:
        s.keep["fake_g"] = g_fake_preds.detach()
        s.keep["g_grads"] = {n: p.grad.detach().clone() for n, p in s.G.named_parameters()}
    s.optimizerG.step()                                                         # :279
    return lossD.detach(), lossG.detach(), info
