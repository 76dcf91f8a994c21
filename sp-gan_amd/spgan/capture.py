"""hipGraph capture of a CALLER-WRITTEN loop body.

`spgan.TrainStep` replays its own formulation of the train step.  A caller who keeps the reference's loop
(Generation/model.py:239-279: separate `G()`, `D()`, `dis_loss`, `.backward()`, `optimizer.step()` statements, `requires_grad`
toggles) pays the host cost of issuing ~900 launches per iteration -- more than the GPU needs to execute them.  `CapturedBody`
wraps THEIR function:

    body = spgan.CapturedBody(my_loop_body, modules=(G, D), warmup=3)
    for data in loader:
        out = body(x, data, z_d, z_g)            # tensors in, tensors out (static buffers, overwritten by the next call)

The first `warmup` calls run the function eagerly (on a side stream, as torch recommends before a capture), the next one is
captured, all later ones replay.  Contract for the function (the usual stream-capture rules):
  * positional tensor arguments only, static shapes; everything else through the closure;
  * no host synchronisation inside (`.item()`, `.cpu()`, printing a device value): return device tensors instead;
  * optimisers must be capturable: `torch.optim.Adam(..., capturable=True)` or `spgan.Adam(..., capturable=True)`;
  * random draws must come from the device generators;
  * wrap the body from its FIRST iteration: do not run the same modules' backward eagerly on another stream beforehand (autograd
    binds the gradient-accumulation nodes of the parameters to the stream that first used them).
What the wrapper takes care of: input staging (an argument that is the same tensor object at the same version as on the previous
call -- the constant sphere prior -- is adopted without a copy, so the Generator keeps its cached neighbour graph; an argument the
modules derived cached graph structure from -- that prior's kNN graph, its CSR, the "every shape carries the same prior" decision, none
of which is re-derived inside the captured graph -- is compared by CONTENT when its object or version changes: on a difference the
graph is dropped, one call runs eagerly (the caches are rebuilt) and the next one is captured again; a prior that keeps changing,
sphere_generator(static=False), ends in eager issue with a warning after a few re-captures), the host-side
BatchNorm call counters of the spgan modules (replayed per call), the weight-derived host caches (dropped before the capture
so that their kernels are recorded; invalidated after every replay because the captured optimiser kernels changed the weights
behind torch's version counters).  A failed capture falls back to eager issue with a warning.
"""
from __future__ import annotations

import warnings
import weakref
from typing import Callable, Optional, Sequence

import torch
import torch.nn as nn

from . import ops


def _bn_modules(modules: Sequence[nn.Module]):
    from .modules import _BNCounts
    return [m for net in modules for m in net.modules() if isinstance(m, _BNCounts)]


def _bn_snapshot(mods):
    return [{pre: dict(pend) for pre, pend in m.__dict__.get("_bn_pending", {}).items()} for m in mods]


class CapturedBody:
    # a change of the structure-bearing argument (the sphere prior) after this many undisturbed replays no longer counts towards the
    # "keeps changing -> eager" verdict
    RECAPTURE_FORGIVEN_AFTER = 8

    def __init__(self, fn: Callable, modules: Sequence[nn.Module], warmup: int = 3):
        self.fn, self.modules, self.warmup = fn, tuple(modules), warmup
        self._calls = 0
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._static = None
        self._src = None
        self._out = None
        self._bn_delta = None
        self._side = None
        self.eager = False
        self._struct = set()         # positions of arguments the modules hold cached graph structure for (the sphere prior)
        self._recaptures = 0
        self._quiet_replays = 0
        self._captured_once = False

    # ------------------------------------------------------------------ inputs
    def _structure_args(self):
        """Static buffers some module keeps cached graph structure for (Generator._sphere_graphs: kNN graph / CSR / dedup decision of the
        prior, keyed by tensor object and version)."""
        held = []
        for net in self.modules:
            for sub in net.modules():
                for e in sub.__dict__.get("_sphere_graphs", ()):
                    t = e["ref"]()
                    if t is not None:
                        held.append(t)
        return {i for i, st in enumerate(self._static or ()) if any(st is t for t in held)}

    def _bind(self, args) -> bool:
        """Stage the arguments; returns True when an argument that feeds cached graph structure was rewritten (under a captured graph:
        only if its CONTENT changed) -- the modules' cache entries for it are then stale and must be rebuilt by an eager call."""
        if self._static is None:
            self._static = [a.detach().clone() for a in args]
            self._src = [(weakref.ref(a), a._version) for a in args]
            return False
        if len(args) != len(self._static):
            raise ValueError("CapturedBody: %d arguments, captured with %d" % (len(args), len(self._static)))
        dst, src = [], []
        structure_changed = False
        for i, (a, st) in enumerate(zip(args, self._static)):
            if a.shape != st.shape or a.dtype != st.dtype:
                raise ValueError("CapturedBody needs static shapes/dtypes: argument %d is %s %s, captured %s %s"
                                 % (i, tuple(a.shape), a.dtype, tuple(st.shape), st.dtype))
            ref, ver = self._src[i]
            if ref() is a and ver == a._version:
                continue                                   # same object, unmodified: the static copy is current
            self._src[i] = (weakref.ref(a), a._version)
            if i in self._struct:
                # the modules cached structure derived from this buffer and the captured graph does not re-derive it: compare on the
                # device (one host sync PER CALL for a caller that hands over a NEW tensor object every call -- `x = sphere.cuda()` inside
                # the loop; a caller that reuses its tensor object unmodified never reaches this line); a real change goes through torch's copy_, whose version bump
                # invalidates the modules' cache entries
                # -- with or without a captured graph: a caller that re-creates an IDENTICAL prior on every call (x = sphere.cuda() inside
                # the loop) must not be taken for one whose prior changes, or the warm-up would restart forever and no graph would ever
                # be captured (round-4 advisor finding); equal content needs no copy and leaves the modules' cache entries valid
                if not bool(torch.equal(st, a)):
                    structure_changed = True
                    st.copy_(a)
                continue
            if a.is_cuda and a.is_contiguous() and a.dtype == torch.float32:
                dst.append(st); src.append(a)
            else:
                st.copy_(a)
        if dst:
            ops.multi_copy(dst, src)
        return structure_changed

    # ------------------------------------------------------------------ call
    def _invalidate_weight_caches(self):
        seen = set()
        for m in self.modules:
            for p in m.parameters():                       # one entry per storage: a flattened network has one, a plain one has one per tensor
                key = p.untyped_storage().data_ptr()
                if key not in seen:
                    seen.add(key)
                    ops.bump_weights_epoch(p)

    def __call__(self, *args):
        for a in args:
            if not isinstance(a, torch.Tensor) or not a.is_cuda:
                raise TypeError("CapturedBody takes CUDA tensors as positional arguments")
        if self.eager:
            return self.fn(*args)
        if self._bind(args):
            # The prior changed: a captured graph holds the OLD prior's kNN graph, CSR and dedup decision (the modules cache them per
            # tensor and version, so the capture contains no kNN launch), and a capture that found its cache entry stale would have to
            # rebuild it with a host sync.  Drop the graph, run this call eagerly (rebuilds the caches), capture again on the next one.
            # Counts changes in CLOSE succession (also before the first capture: a prior that really changes on every call must reach the
            # eager verdict); a change after RECAPTURE_FORGIVEN_AFTER undisturbed replays starts from zero again -- a new sphere per epoch
            # is re-captured every time, not demoted on the fourth epoch.
            self._recaptures += 1
            self._quiet_replays = 0
            self._graph = None
            if self._recaptures > 3:
                warnings.warn("CapturedBody: an argument the modules derive cached graph structure from (the sphere prior) keeps changing "
                              "between calls; the captured graph depends on it, so the body is issued eagerly from now on")
                self.eager = True
                return self.fn(*args)
            self._calls = self.warmup - 1            # exactly one eager call (also with warmup = 0), then the capture
        if self._graph is None and self._calls < self.warmup:
            if self._side is None:
                self._side = torch.cuda.Stream()
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                out = self.fn(*self._static)
            torch.cuda.current_stream().wait_stream(self._side)
            self._calls += 1
            self._struct = self._structure_args()
            return out
        mods = _bn_modules(self.modules)
        if self._graph is None:
            from . import nets
            nets.drop_weight_caches()
            for m in self.modules:
                m.__dict__["_ec1_twin"] = None
            before = _bn_snapshot(mods)
            g = torch.cuda.CUDAGraph()
            if self._side is None:
                self._side = torch.cuda.Stream()
            try:
                # captured on the stream the warm-up calls ran on: autograd's AccumulateGrad nodes are bound to the stream they were
                # created on, and one that survives from the warm-up must not pull a foreign stream into the capture
                with torch.cuda.graph(g, stream=self._side, capture_error_mode="thread_local"):
                    self._out = self.fn(*self._static)
            except Exception as e:                                   # noqa: BLE001
                warnings.warn("CapturedBody: hipGraph capture failed (%s: %s); issuing the body eagerly from now on" % (type(e).__name__, e))
                after = _bn_snapshot(mods)
                for m, b, a in zip(mods, before, after):              # nothing ran: take the host-side bookkeeping of the attempt back
                    store = m.__dict__.setdefault("_bn_pending", {})
                    for pre, pend in a.items():
                        for k in pend:
                            store[pre][k] = b.get(pre, {}).get(k, 0)
                nets.drop_weight_caches()
                self.eager = True
                torch.cuda.synchronize()
                return self.fn(*args)
            after = _bn_snapshot(mods)
            self._bn_delta = []
            for m, b, a in zip(mods, before, after):
                d = {pre: {k: n - b.get(pre, {}).get(k, 0) for k, n in pend.items()} for pre, pend in a.items()}
                self._bn_delta.append(d)
                store = m.__dict__.setdefault("_bn_pending", {})
                for pre, pend in d.items():
                    for k, n in pend.items():
                        store[pre][k] -= n
            self._graph = g
            self._captured_once = True
            self._struct = self._structure_args()
        self._graph.replay()
        self._quiet_replays += 1
        if self._quiet_replays >= self.RECAPTURE_FORGIVEN_AFTER:
            self._recaptures = 0
        self._invalidate_weight_caches()
        for m, d in zip(mods, self._bn_delta):
            store = m.__dict__.setdefault("_bn_pending", {})
            for pre, pend in d.items():
                tgt = store.setdefault(pre, {})
                for k, n in pend.items():
                    tgt[k] = tgt.get(k, 0) + n
        return self._out
