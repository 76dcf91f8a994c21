"""Data parallelism for the train step: one process per GPU, per-rank BatchNorm statistics (what the
reference's nn.DataParallel does per replica, Generation/model.py:79-84; its vendored sync_bn is
unused), and ONE flat all-reduce per network per step over RCCL/xGMI instead of DataParallel's
per-call scatter / parameter broadcast / gather.

G = 585,155 and D = 980,353 fp32 gradients -> 2.3 MB / 3.9 MB messages: latency-bound, so they are
sent as a single buffer each.  Averaging is folded into the Adam kernel (grad_scale).
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from .optim import flatten_module


def init_process_group_from_env(backend: Optional[str] = None) -> int:
    """torchrun-style rendezvous (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*).  'nccl' is RCCL on ROCm."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    force = os.environ.get("SPGAN_FORCE_DIST", "0") == "1" and "RANK" in os.environ      # exercise the RCCL path with one rank
    if (world <= 1 and not force) or dist.is_initialized():
        return int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend, rank=int(os.environ["RANK"]), world_size=world)
    return dist.get_rank()


class DataParallel(nn.Module):
    """Thin wrapper exposing `.module` (the reference unwraps it when saving, model.py:514,522).
    forward() runs the local replica on the local shard; `allreduce_grads()` sums the flat gradient
    buffer over ranks; `sync_params()` broadcasts rank 0's parameters and buffers once at start."""

    def __init__(self, module: nn.Module, process_group=None):
        super().__init__()
        self.module = module
        self.pg = process_group
        self.flat = flatten_module(module)

    @property
    def world_size(self) -> int:
        return dist.get_world_size(self.pg) if dist.is_initialized() else 1

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def sync_params(self):
        if self.world_size > 1:
            dist.broadcast(self.flat.flat, src=0, group=self.pg)
            for b in self.module.buffers():
                dist.broadcast(b, src=0, group=self.pg)

    def allreduce_grads(self) -> float:
        """Sum gradients over ranks in place; returns the scale (1/world) the optimiser must apply."""
        w = self.world_size
        if w > 1:
            dist.all_reduce(self.flat.grad, op=dist.ReduceOp.SUM, group=self.pg)
        return 1.0 / w


def shard_batch(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Even split on dim 0 (DataParallel's scatter semantics for divisible batches)."""
    b = t.shape[0]
    if b % world:
        raise ValueError("global batch %d is not divisible by world size %d" % (b, world))
    per = b // world
    return t[rank * per:(rank + 1) * per]
