"""Flat-buffer Adam (torch.optim.Adam semantics, Generation/model.py:94-97) and the flat
parameter/gradient layout shared with the data-parallel reducer.

`flatten_module(m)` re-points every parameter of `m` at a slice of ONE contiguous fp32 buffer and
pre-binds `.grad` to the matching slice of ONE gradient buffer: the optimiser update is a single
HIP launch and the data-parallel all-reduce a single RCCL collective per network.
state_dict() is unaffected (parameters keep their names/shapes).
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.nn as nn

from . import ops


class FlatParams:
    def __init__(self, module: nn.Module):
        params = [p for p in module.parameters()]
        if not params:
            raise ValueError("module has no parameters")
        dev = params[0].device
        # every slice starts on a 16-byte boundary (float4 operand loads in the GEMM fast path); the padding
        # elements stay zero in both buffers, so Adam and the all-reduce may run over the whole buffer.
        self.offsets = []
        n = 0
        for p in params:
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.params = params
        with torch.no_grad():
            for p, off in zip(params, self.offsets):
                k = p.numel()
                self.flat[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat[off:off + k].view_as(p)
                p.grad = self.grad[off:off + k].view_as(p)
        self.numel = n

    def zero_grad(self):
        """Zero in place and re-bind (autograd accumulates into the flat slices)."""
        self.grad.zero_()
        for p, off in zip(self.params, self.offsets):
            k = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + k].view_as(p)


def flatten_module(module: nn.Module) -> FlatParams:
    fp = getattr(module, "_spgan_flat", None)
    if fp is None or any(p.data_ptr() < fp.flat.data_ptr() or p.data_ptr() >= fp.flat.data_ptr() + 4 * fp.numel for p in module.parameters()):
        fp = FlatParams(module)
        module.__dict__["_spgan_flat"] = fp
    return fp


class Adam:
    """Adam over a module's flat buffer.  `step()` == torch.optim.Adam(lr, betas, eps=1e-8).step();
    `zero_grad()` zeroes the flat gradient buffer in place."""

    def __init__(self, module: nn.Module, lr: float = 1e-4, betas=(0.5, 0.99), eps: float = 1e-8, capturable: bool = False,
                 zero_grad_in_step: bool = False):
        self.fp = flatten_module(module)
        self.lr, self.betas, self.eps = lr, betas, eps
        self.m = torch.zeros_like(self.fp.flat)
        self.v = torch.zeros_like(self.fp.flat)
        self.t = 0
        # capturable: the step count (and its bias corrections) live on the device, so that a hipGraph of the train step can
        # be replayed; `t` stays the host-side mirror (state_dict)
        self.capturable = capturable
        self.dev_state = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=torch.float32, device=self.fp.flat.device) if capturable else None
        self.base_lr = lr
        # zero_grad_in_step (capturable mode; TrainStep): step() leaves the flat gradient buffer zeroed -- the kernel has every gradient in a
        # register anyway -- and the zero_grad() that follows it is a no-op instead of a fill launch.  Only for an owner that writes the
        # gradients exclusively between zero_grad() and step().
        self.zero_grad_in_step = bool(zero_grad_in_step and capturable)
        self._grad_clean = False

    def invalidate_grads(self) -> None:
        """Restore "the flat gradient buffer is zero" after anything OUTSIDE the owner's zero_grad() .. step() window wrote into the
        parameters' `.grad` (a diagnostic backward between two steps, a regulariser of the caller's): zero_grad_in_step skips the
        fill on the host flag alone, and a captured step records no fill at all, so such a write would be added to the next step's
        gradients.  Zeroes the buffer now (one fill launch) and forgets the flag.  Contract of zero_grad_in_step otherwise: `p.grad`
        reads as zero after step()."""
        self._grad_clean = False
        self.fp.zero_grad()
        self._grad_clean = self.zero_grad_in_step

    def zero_grad(self, set_to_none: bool = False):
        if self._grad_clean:
            self._grad_clean = False         # the previous step() zeroed the buffer; the parameters' .grad views are bound to it
            return
        self.fp.zero_grad()

    def step(self, grad_scale: float = 1.0):
        self.t += 1
        ops.bump_weights_epoch(self.fp.flat)
        if self.capturable:
            ops.adam_step_dev(self.fp.flat, self.fp.grad, self.m, self.v, self.dev_state, self.lr, self.betas[0], self.betas[1], self.eps, grad_scale,
                              zero_grad=self.zero_grad_in_step)
            self._grad_clean = self.zero_grad_in_step
        else:
            ops.adam_step(self.fp.flat, self.fp.grad, self.m, self.v, self.t, self.lr, self.betas[0], self.betas[1], self.eps, grad_scale)

    def set_lr(self, lr: float) -> None:
        """Change the learning rate (lr schedules).  In capturable mode the captured kernels keep the lr they were recorded with;
        the new value enters through the multiplier in the device state, so replayed graphs follow it."""
        if self.capturable:
            self.dev_state[3:4].fill_(float(lr) / self.base_lr)
            self._lr_now = float(lr)
        else:
            self.lr = float(lr)

    def get_lr(self) -> float:
        return getattr(self, "_lr_now", self.lr) if self.capturable else self.lr

    def state_dict(self):
        return {"m": self.m, "v": self.v, "t": self.t, "lr": self.get_lr(), "betas": self.betas, "eps": self.eps}

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"]); self.v.copy_(sd["v"]); self.t = int(sd["t"])
        self.betas, self.eps = tuple(sd["betas"]), sd["eps"]
        self.set_lr(sd["lr"])
        if self.capturable:
            self.dev_state[:1].view(torch.int32).fill_(self.t)


class StepLR:
    """torch.optim.lr_scheduler.StepLR for spgan.optim.Adam (Generation/model.py:99-110: step_size = lr_decay_feq, gamma =
    lr_decay_rate, stepped once per epoch at model.py:309-312): lr = base_lr * gamma ** (epoch // step_size)."""

    def __init__(self, optimizer: Adam, step_size: int, gamma: float = 0.1):
        self.opt, self.step_size, self.gamma = optimizer, int(step_size), float(gamma)
        self.base_lr = optimizer.get_lr()
        self.last_epoch = 0

    def step(self) -> None:
        self.last_epoch += 1
        self.opt.set_lr(self.base_lr * self.gamma ** (self.last_epoch // self.step_size))

    def get_last_lr(self):
        return [self.opt.get_lr()]
