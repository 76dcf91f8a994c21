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


def ops_capturing() -> bool:
    from . import ops
    return ops.capturing()


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

    def __init__(self, module: nn.Module, process_group=None, collective: Optional[str] = None):
        """collective: "all_reduce" (default: one dist.all_reduce of the flat buffer, RCCL picks the algorithm) or "one_hop": reduce-scatter as
        ONE all-to-all (on a fully connected xGMI node every pair is one hop apart) + a local sum of the received chunks in rank order
        (spgan_reduce_chunks) + one all-gather -- two hops for the 2.3 / 3.9 MB latency-bound gradient messages instead of a ring's
        2(W-1) steps.  Both give every rank the same sums (one_hop: bit-identical on all ranks by construction).
        "rccl": the library's own entry point spgan_allreduce_flat (include/spgan_hip.h): one RCCL all-reduce issued by
        libspgan_hip.so itself on the CURRENT stream through a communicator of its own (the 128-byte RCCL id travels from rank 0
        through torch.distributed once) -- what a host program without torch.distributed would call, and capturable into a hipGraph.
        Which is fastest on an 8-GPU MI355X node is NOT measured (no multi-GPU box in this build's loop: tools/collective_probe.py
        is the script for it); the environment variable SPGAN_DP_COLLECTIVE overrides the default for such a measurement."""
        super().__init__()
        self.module = module
        self.pg = process_group
        self.flat = flatten_module(module)
        self.collective = collective or os.environ.get("SPGAN_DP_COLLECTIVE", "all_reduce")
        if self.collective not in ("all_reduce", "one_hop", "rccl"):
            raise ValueError("collective must be 'all_reduce', 'one_hop' or 'rccl'")
        self._hop = None             # (send [W, chunk], recv [W, chunk], mine [chunk], full [W*chunk]) staging buffers of the one-hop path
        self._comm = None            # native RCCL communicator of the "rccl" path
        if self.collective == "rccl" and dist.is_initialized() and self.flat.flat.is_cuda:
            self.native_comm()       # created eagerly: its rendezvous (a broadcast + a host copy) must not first happen inside a stream capture

    @property
    def world_size(self) -> int:
        return dist.get_world_size(self.pg) if dist.is_initialized() else 1

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def sync_params(self):
        if self.world_size > 1:
            src = dist.get_global_rank(self.pg, 0) if self.pg is not None else 0     # the group's first member (a sub-group's need not be global rank 0)
            dist.broadcast(self.flat.flat, src=src, group=self.pg)
            for b in self.module.buffers():
                dist.broadcast(b, src=src, group=self.pg)

    def allreduce_grads(self) -> float:
        """Sum gradients over ranks in place; returns the scale (1/world) the optimiser must apply."""
        self.allreduce_grads_begin()
        return self.allreduce_grads_end()

    def allreduce_grads_begin(self) -> None:
        """Start the reduction of the flat gradient buffer WITHOUT making the current stream wait for it: the collective runs on the
        backend's own stream (it first waits for the work already enqueued on the current stream, i.e. for the backward pass that
        produced the gradients); kernels issued on the current stream afterwards overlap with it.  `allreduce_grads_end()` joins.
        Nothing issued in between may read or write the gradient buffer (TrainStep issues the generator's forward of the G step
        there: it touches neither D's gradients nor D's weights)."""
        self._work = None
        w = self.world_size
        if w > 1:
            if self.collective == "one_hop":
                self._allreduce_one_hop(w)                 # three dependent steps with a local kernel in the middle: issued in order
            elif self.collective == "rccl":
                self._allreduce_native()                   # on the current stream: the following kernels queue behind it
            else:
                self._work = dist.all_reduce(self.flat.grad, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def allreduce_grads_end(self) -> float:
        work, self._work = getattr(self, "_work", None), None
        if work is not None:
            work.wait()                                    # stream-level join for RCCL (the host does not block); a host wait on gloo
        return 1.0 / self.world_size

    def native_comm(self):
        """The library's own RCCL communicator over the ranks of this process group (created on first use): rank 0 draws the 128-byte
        id (spgan_comm_unique_id), torch.distributed carries it to the other ranks once, every rank calls spgan_comm_init."""
        if self._comm is not None:
            return self._comm
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        dev = self.flat.flat.device
        if dev.type != "cuda":
            raise RuntimeError("collective='rccl' needs the parameters on a GPU")
        if not lib.spgan_comm_available():
            raise RuntimeError("collective='rccl': librccl could not be loaded by libspgan_hip.so")
        rank = dist.get_rank(self.pg) if dist.is_initialized() else 0
        w = self.world_size
        if ops_capturing():
            raise RuntimeError("collective='rccl': the communicator must exist before a stream capture starts (DataParallel creates it in "
                               "__init__ when the parameters are on a GPU; call native_comm() once before capturing otherwise)")
        buf = (C.c_ubyte * 128)()
        status = 0
        if rank == 0:
            status = int(lib.spgan_comm_unique_id(buf))       # a failure here must reach every rank: the others wait in the broadcast
        # 128 id bytes + rank 0's status, carried by the process group's first member (a sub-group's rank 0 need not be global rank 0)
        ident = torch.tensor(list(bytes(buf)) + [status & 0xFF], dtype=torch.uint8, device=dev)
        if w > 1:
            src = dist.get_global_rank(self.pg, 0) if self.pg is not None else 0
            dist.broadcast(ident, src=src, group=self.pg)
        host = ident.cpu().tolist()
        if host[128] != 0:
            raise RuntimeError("collective='rccl': spgan_comm_unique_id failed on the group's first rank (status byte %d, RCCL code %d on that "
                               "rank's thread)" % (host[128], lib.spgan_comm_last_error(None) if rank == 0 else -1))
        raw = (C.c_ubyte * 128)(*host[:128])
        comm = C.c_void_p()
        with torch.cuda.device(dev):
            _lib.check(lib.spgan_comm_init(raw, rank, w, C.byref(comm)), "comm_init", rank=rank, world=w, rccl_error=lib.spgan_comm_last_error(None))
        self._comm = comm
        return comm

    def _allreduce_native(self) -> None:
        from . import _lib
        g = self.flat.grad
        comm = self.native_comm()
        lib = _lib.load()
        st = lib.spgan_allreduce_flat(comm, g.data_ptr(), g.numel(), torch.cuda.current_stream().cuda_stream)
        if st != 0:
            _lib.check(st, "allreduce_flat", n=g.numel(), rccl_error=lib.spgan_comm_last_error(comm))

    def __del__(self):
        comm = self.__dict__.get("_comm")          # plain dict access: nn.Module.__setattr__ / __getattr__ are not usable at interpreter shutdown
        if comm is not None:
            self.__dict__["_comm"] = None
            try:
                from . import _lib
                _lib.load().spgan_comm_destroy(comm)
            except Exception:      # noqa: BLE001  (interpreter shutdown)
                pass

    def _allreduce_one_hop(self, w: int) -> None:
        from . import ops
        g = self.flat.grad
        n = g.numel()
        chunk = (n + w - 1) // w
        chunk = (chunk + 3) // 4 * 4                      # 16-byte aligned chunks
        if self._hop is None or self._hop[0].shape != (w, chunk) or self._hop[0].device != g.device:
            mk = lambda *s: torch.zeros(*s, dtype=torch.float32, device=g.device)
            self._hop = (mk(w, chunk), mk(w, chunk), mk(chunk), mk(w * chunk))
        send, recv, mine, full = self._hop
        send.view(-1)[:n].copy_(g)                          # the padding stays zero
        dist.all_to_all_single(recv, send, group=self.pg)   # row j of recv: my chunk as rank j computed it
        ops.reduce_chunks(recv, mine)                       # fixed order j = 0 .. W-1
        dist.all_gather_into_tensor(full, mine, group=self.pg)
        g.copy_(full[:n])


def shard_batch(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Even split on dim 0 (DataParallel's scatter semantics for divisible batches)."""
    b = t.shape[0]
    if b % world:
        raise ValueError("global batch %d is not divisible by world size %d" % (b, world))
    per = b // world
    return t[rank * per:(rank + 1) * per]
