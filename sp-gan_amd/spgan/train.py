"""One iteration of the reference's train loop body (Generation/model.py:239-279) over the HIP modules:
D-step (G frozen, no graph through G), then G-step (D frozen: input gradients only), with the
reference's call order -- including the unused D(real) forward in the G-step that advances D's
BatchNorm running statistics -- and Adam(lr, betas=(0.5, 0.99)) (model.py:94-97).

With `use_gp` the discriminator loss is dis_loss(gan) + GradientPenalty(lambda_gp, gamma=1): the
"WGAN-GP" composition of reference pieces that BASELINE config 2 names.
No host<->device synchronisation happens inside step(); losses are returned as device tensors.

`graph=True`: after `graph_warmup` eager steps the whole iteration (2 G forwards, 5 D forwards, all backwards, the gradient
penalty's double backward, both Adam updates and, data-parallel, both all-reduces: ~580 kernel launches) is captured once into
a hipGraph and replayed.  The Python/launch cost of a step (~14 ms, as long as the GPU work itself) drops to one graph launch;
inputs are copied into static buffers every step; only the sphere prior x is adopted without a copy while the caller keeps
passing the same unmodified tensor object.  x must be constant in graph mode: the capture depends on its kNN graph, so a changed
x costs an eager step and a re-capture (and after a few changes the harness stays eager).
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import ops
from .functions import DeliverySink, fused_grad_accumulation
from .losses import GradientPenalty, dis_loss_with_grads, gen_loss_with_grads
from .optim import Adam
from .parallel import DataParallel


def requires_grad(model: nn.Module, flag: bool = True):
    """Common/network_utils.py:92-94"""
    for p in model.parameters():
        p.requires_grad = flag


class TrainStep:
    def __init__(self, G: nn.Module, D: nn.Module, gan: str = "ls", use_gp: bool = False, lambda_gp: float = 10.0,
                 lr_g: float = 1e-4, lr_d: float = 1e-4, betas=(0.5, 0.99), flip_d: bool = False, flip_g: bool = False,
                 distributed: bool = False, process_group=None, graph: bool = False, graph_warmup: int = 3,
                 reference_schedule: bool = False):
        self.G, self.D = G, D
        self.gan, self.use_gp = gan, use_gp
        self.flip_d, self.flip_g = flip_d, flip_g
        self.gp = GradientPenalty(lambda_gp, gamma=1)
        self.dpG = DataParallel(G, process_group) if distributed else None
        self.dpD = DataParallel(D, process_group) if distributed else None
        if distributed:
            self.dpG.sync_params(); self.dpD.sync_params()
        self.optG = Adam(G, lr_g, betas, capturable=graph, zero_grad_in_step=True)
        self.optD = Adam(D, lr_d, betas, capturable=graph, zero_grad_in_step=True)
        G.train(); D.train()
        # reference_schedule=True evaluates exactly the calls of model.py:239-279, including the two pieces of work this harness
        # otherwise removes because they are provably redundant: EdgeConv1 on every copy of the tiled sphere prior
        # (Generator.dedup_sphere) and the full D(real) forward of the G step whose logits gen_loss ignores.
        self.reference_schedule = reference_schedule
        if reference_schedule:
            G.dedup_sphere = False
        self.use_graph, self.graph_warmup = graph, graph_warmup
        self._graph = None
        self._static = None          # [x, real, z_d, z_g, alpha]
        self._x_src = None
        self._recaptures = 0
        self._static_info = None
        self._eager_calls = 0
        self._bn_delta = None        # host-side BatchNorm call counts of one step (replayed on the host)
        self._side = None
        # The D step's three passes through D's conv stack -- D(real), D(fake), D(x_hat) of the penalty -- as ONE batch: one GEMM and one
        # finalize launch per layer for all three, per-pass BatchNorm statistics (Discriminator.forward_stacks_grouped; bit-identical to
        # the separate calls, running statistics advanced in the reference's order real, fake, x_hat).  The backward passes run per
        # pass as before, on views of the batched activations.
        self.batch_d_forwards = not reference_schedule and hasattr(D, "forward_stacks_grouped")      # attribute = test hook (the separate-calls side was measured in rounds 2 and 3)
        # The generator's two forwards of a step (D step, G step) see the same sphere prior and the same weights: EdgeConv1, which
        # depends on nothing else, is evaluated once and its BatchNorm running statistics are advanced twice (Generator.twin_forward).
        self.twin_g_forwards = not reference_schedule      # attribute = test hook
        self.pair_g_forwards = os.environ.get("SPGAN_PAIR_G", "1") != "0"      # test / A-B hook: False evaluates the step's two generator forwards separately
        self._pair_gfake = None
        self._sinkD, self._sinkG = DeliverySink(), DeliverySink()      # where the backward nodes of the D / G step leave their parameter gradients
        self.joint_d_backward = os.environ.get("SPGAN_JOINT_D", "1") != "0"      # test / A-B hook: False keeps one autograd node (and one chain of launches) per D pass
        self.point_major = True                            # test hook: False keeps the [B,3,N] layout between the networks (same results up to the penalty norm's summation order)
        # Data parallel: the generator's forward of the G step does not depend on D's update, so it is issued while D's gradient
        # all-reduce is in flight (SPGAN_DP_OVERLAP=0: the strictly sequential schedule, for A/B measurements on a node).
        self.overlap_g_forward = distributed and os.environ.get("SPGAN_DP_OVERLAP", "1") != "0"
        # collective="rccl" (spgan_allreduce_flat) is enqueued on the current stream by the library itself, so the whole data-parallel
        # step -- both collectives included -- can be ONE captured graph.  Opt-in (SPGAN_DP_SINGLE_GRAPH=1): not measured on a
        # multi-GPU node yet (tools/collective_probe.py tries it).
        self.single_graph_dp = (distributed and self.dpD.collective == "rccl" and os.environ.get("SPGAN_DP_SINGLE_GRAPH", "0") == "1")

    # ------------------------------------------------------------------ hipGraph replay
    def _bn_modules(self):
        from .modules import _BNCounts
        return [m for net in (self.G, self.D) for m in net.modules() if isinstance(m, _BNCounts)]

    def _bn_snapshot(self):
        return [{pre: dict(pend) for pre, pend in m.__dict__.get("_bn_pending", {}).items()} for m in self._bn_modules()]

    def _bind(self, tensors):
        """Copy the step's inputs into the static buffers the graph reads.  real, z_d, z_g and alpha are copied on EVERY step
        (16 KB .. 0.8 MB device copies): equal address and version do not prove equal content -- a fresh temporary routinely
        lands on the block the previous step's temporary just freed.  Only the sphere prior x may be adopted without a copy, and
        only when it is the same tensor OBJECT at the same version as on the previous step (the constant template: its cached
        kNN graph stays valid).  Returns True when x changed."""
        import weakref
        if self._static is None:
            self._static = [None if t is None else t.detach().clone() for t in tensors]
            self._x_src = (weakref.ref(tensors[0]), tensors[0]._version)
            return False
        x_changed = False
        pend_d, pend_s = [], []
        for i, t in enumerate(tensors):
            st = self._static[i]
            if t is None or st is None:
                if (t is None) != (st is None):
                    raise ValueError("graph mode: alpha must be given on every step or on none")
                continue
            if t.shape != st.shape:
                raise ValueError("graph mode needs static shapes: got %s, captured %s" % (tuple(t.shape), tuple(st.shape)))
            if i == 0:
                ref, ver = self._x_src
                if ref() is t and ver == t._version:
                    continue
                # a different prior tensor (or the same one modified in place): the content may or may not differ -- compare on
                # the device only when a captured graph depends on it (one host sync, off the steady-state path)
                if self._graph is not None and not bool(torch.equal(st, t)):
                    x_changed = True
                self._x_src = (weakref.ref(t), t._version)
                if self._graph is None or x_changed:
                    st.copy_(t)
                continue
            if t.is_cuda and t.is_contiguous() and t.dtype == torch.float32:
                pend_d.append(st); pend_s.append(t)                              # one launch for all of them (below)
            else:
                st.copy_(t)
        if pend_d:
            ops.multi_copy(pend_d, pend_s)
        return x_changed

    def _graph_step(self, x, real, z_d, z_g, alpha):
        if self._bind((x, real, z_d, z_g, alpha)):
            # The sphere prior changed after the capture.  The captured kernels hold the kNN graph, its CSR and the
            # "every shape carries the same prior" decision of the OLD x (Generator caches them per tensor and version, so the
            # capture contains no kNN launch): replaying would silently use stale neighbours.  Drop the graph, run this step
            # eagerly (rebuilds the caches for the new x) and capture again on the next one; a caller whose prior changes every
            # step (sphere_generator(static=False)) gets eager issue for good after a few re-captures.
            self._graph = None
            self._recaptures += 1
            if self._recaptures > 3:
                import warnings
                warnings.warn("TrainStep(graph=True): the sphere prior x keeps changing between steps; the captured graph depends on "
                              "its kNN graph, so steps are issued eagerly from now on")
                self.use_graph = False
                return self._eager_step(x, real, z_d, z_g, alpha)
            self._eager_calls = max(self.graph_warmup - 1, 0)
        if self._graph is None and self._eager_calls < self.graph_warmup:
            # eager warm-up on a side stream (allocator / autograd state as the capture will see it)
            if self._side is None:
                self._side = torch.cuda.Stream()
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                info = self._eager_step(*self._static)
            torch.cuda.current_stream().wait_stream(self._side)
            self._eager_calls += 1
            return info
        if self._graph is None:
            before = self._bn_snapshot()
            tG, tD = self.optG.t, self.optD.t
            # weight-derived host caches (transposes, permuted conv_out weights) must be derived INSIDE the capture: an eager forward
            # since the last optimiser step (a sample dump between two steps) may have filled them
            from . import nets
            nets.drop_weight_caches()
            self.G.__dict__["_ec1_twin"] = None
            try:
                if self.dpD is None or self.single_graph_dp:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        self._static_info = self._eager_step(*self._static)
                    graphs = [g]
                else:
                    # data parallel: three graphs, the two flat all-reduces are issued eagerly between them (RCCL stays outside
                    # the capture); all three share one memory pool
                    sx, sreal, szd, szg, salpha = self._static
                    info: Dict[str, torch.Tensor] = {}
                    g1, gf, g2, g3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                    w = 1.0 / self.dpD.world_size
                    with torch.cuda.graph(g1, capture_error_mode="thread_local"):   # other threads (RCCL watchdog) stay free to call HIP
                        real_t = self._seg_d(sx, sreal, szd, salpha, False, info)
                    g_fake = None
                    if self.overlap_g_forward:          # replayed while D's all-reduce is in flight
                        with torch.cuda.graph(gf, pool=g1.pool(), capture_error_mode="thread_local"):
                            g_fake = self._seg_gfwd(sx, szg, pm=isinstance(real_t, tuple))
                    with torch.cuda.graph(g2, pool=g1.pool(), capture_error_mode="thread_local"):
                        self._seg_g(sx, real_t, szg, w, False, info, g_fake=g_fake)
                    with torch.cuda.graph(g3, pool=g1.pool(), capture_error_mode="thread_local"):
                        self._seg_opt_g(w, False, info)
                    self._static_info = info
                    graphs = [g1, gf if self.overlap_g_forward else None, g2, g3]
            except Exception as e:                                   # noqa: BLE001
                # a failed capture must not cost the run: undo the host-side bookkeeping of the aborted attempt and issue this and all
                # later steps eagerly (data parallel: the collectives are eager in both modes, so ranks stay in lockstep)
                import warnings
                warnings.warn("hipGraph capture of the train step failed (%s: %s); falling back to eager issue" % (type(e).__name__, e))
                after = self._bn_snapshot()
                for m, b, a in zip(self._bn_modules(), before, after):
                    store = m.__dict__.setdefault("_bn_pending", {})
                    for pre, pend in a.items():
                        for k, n in pend.items():
                            store[pre][k] = b.get(pre, {}).get(k, 0)
                self.optG.t, self.optD.t = tG, tD
                if self.optG.capturable:
                    self.optG.dev_state[:1].view(torch.int32).fill_(tG); self.optD.dev_state[:1].view(torch.int32).fill_(tD)
                # the aborted capture recorded Adam launches that never ran: their "the step left the gradient buffer zeroed" flags are
                # false, and whatever the warm-up steps left in the buffers is unknown to the host -- fill them
                self.optG.invalidate_grads(); self.optD.invalidate_grads()
                self._sinkD.clear(); self._sinkG.clear(); self._pair_gfake = None
                self.use_graph = False
                torch.cuda.synchronize()
                # cache entries created during the aborted capture carry current stamps but live in the aborted graph's pool and were
                # never written: the eager fallback must re-derive them
                nets.drop_weight_caches()
                self.G.__dict__["_ec1_twin"] = None
                return self._eager_step(*self._static)
            after = self._bn_snapshot()
            # nothing ran during the capture: take the host-side bookkeeping of that step back, keep it as the per-replay delta
            self._bn_delta = []
            for m, b, a in zip(self._bn_modules(), before, after):
                d = {pre: {k: n - b.get(pre, {}).get(k, 0) for k, n in pend.items()} for pre, pend in a.items()}
                self._bn_delta.append(d)
                store = m.__dict__.setdefault("_bn_pending", {})
                for pre, pend in d.items():
                    for k, n in pend.items():
                        store[pre][k] -= n
            self.optG.t, self.optD.t = tG, tD
            self._graph = graphs
        if len(self._graph) == 1:
            self._graph[0].replay()
        else:
            g1, gf, g2, g3 = self._graph
            g1.replay()
            self.dpD.allreduce_grads_begin()
            if gf is not None:
                gf.replay()                          # G(x, z_g) of the G step, under D's all-reduce
            self.dpD.allreduce_grads_end()
            g2.replay()
            self.dpG.allreduce_grads()
            g3.replay()
        ops.bump_weights_epoch(self.optD.fp.flat); ops.bump_weights_epoch(self.optG.fp.flat)   # both networks were updated by the replayed Adam kernels: host-side weight caches are stale
        for m, d in zip(self._bn_modules(), self._bn_delta):
            store = m.__dict__.setdefault("_bn_pending", {})
            for pre, pend in d.items():
                tgt = store.setdefault(pre, {})
                for k, n in pend.items():
                    tgt[k] = tgt.get(k, 0) + n
        self.optG.t += 1; self.optD.t += 1
        return self._static_info

    def step(self, x: torch.Tensor, real: torch.Tensor, z_d: torch.Tensor, z_g: torch.Tensor, alpha: Optional[torch.Tensor] = None,
             keep_grads: bool = False) -> Dict[str, torch.Tensor]:
        """x: sphere [B,N,3]; real [B,N,3]; z_d, z_g [B,N,nz] (noise for the D- and the G-step).
        In graph mode the returned tensors are static buffers that the next step overwrites."""
        if self.use_graph and not keep_grads:
            return self._graph_step(x, real, z_d, z_g, alpha)
        return self._eager_step(x, real, z_d, z_g, alpha, keep_grads)

    # The iteration in three segments, split where the data-parallel all-reduces sit (they stay outside the captured graphs).
    def _seg_d(self, x, real, z_d, alpha, keep_grads, info, z_g=None):
        """D step up to lossD.backward() (model.py:240-258).  z_g given (single process): the G step's generator forward is evaluated here,
        together with the D step's (Generator.forward_pair), and waits in self._pair_gfake for _seg_g."""
        # the nodes built here add their parameter gradients straight into the flat .grad buffers -- through the step's sink: one split-sum
        # reduction and one accumulation launch for the whole backward instead of a pair per node
        self._sinkD.clear(); self._sinkG.clear()      # a step starts here: nothing of an aborted earlier backward may be left
        with fused_grad_accumulation(self._sinkD):
            out = self._seg_d_body(x, real, z_d, alpha, keep_grads, info, z_g)
        self._sinkD.flush()
        return out

    def _seg_d_body(self, x, real, z_d, alpha, keep_grads, info, z_g=None):
        G, D = self.G, self.D
        B, N, _ = real.shape
        requires_grad(G, False); requires_grad(D, True)
        self.optD.zero_grad()
        self._pair_gfake = None
        # Point-major internal route: the batched conv stacks take the generator's output [B*N,3] as it leaves its last GEMM and the real
        # cloud as the loader delivers it ([B,N,3] IS point-major) -- no [B,3,N] round trips (layout kernels, cat + transpose, and their
        # adjoints in the penalty's double backward); the same values into the same kernels (only the penalty's per-shape norm sums its
        # 3N squares in the other memory order: last-bit differences).
        # (the route builds x_hat itself as real + alpha*(fake - real), gradient_penalty.py:24-25: only for the penalty's "common" mixing
        # rule -- a GradientPenalty(mix="loss_utils") keeps the [B,3,N] route, which calls GradientPenalty.interpolate)
        pm = (self.point_major and self.batch_d_forwards and D.training and N % ops.ROW_TILE == 0 and not getattr(G, "off", False)
              and tuple(x.shape) == tuple(real.shape) and (not self.use_gp or self.gp.mix == "common"))
        pair = (pm and z_g is not None and self.pair_g_forwards and self.twin_g_forwards and self.dpD is None and hasattr(G, "forward_pair")
                and G.pair_ok(x, z_d, z_g))
        if pair:
            # both generator forwards of the step as one pipeline (same prior, same weights; G(x, z_g) depends on nothing this D step produces):
            # the per-point / per-shape stages run once on the rows of both passes.  Its autograd nodes leave their gradients in the G step's sink.
            requires_grad(G, True)
            self.optG.zero_grad()
            with fused_grad_accumulation(self._sinkG):
                fake, self._pair_gfake = G.forward_pair(x, z_d, z_g)
        else:
            G.twin_forward = "first" if self.twin_g_forwards else None
            try:
                fake = G(x, z_d, pm_out=True).detach() if pm else G(x, z_d).detach()
            finally:
                G.twin_forward = None
        if pm:
            M = B * N
            real_pm = real.reshape(M, 3).contiguous()
            ins = [real_pm, fake]
            x_hat = None
            if self.use_gp:
                a_ = alpha if alpha is not None else torch.rand(B, 1, 1, device=real.device)      # gradient_penalty.py:24
                x_hat = ops.lerp_rows(real_pm.view(B, N * 3), fake.view(B, N * 3), a_.reshape(B)).view(M, 3)     # real + alpha*(fake - real)
                ins.append(x_hat)
            pre = D.forward_stacks_grouped(ins, pm_shape=(B, N))
            if self.joint_d_backward and hasattr(D, "stacks_joint") and D.joint_ok(pre):
                # one node for the conv stacks of all passes: their backward work (real, fake, the penalty's double backward) runs in lock
                # step, every layer's launch issued once for the three of them (nets.d_backward_joint)
                outs = D.stacks_joint(pre[:2], pre[2] if self.use_gp else None)
                d_real, d_fake = D.forward_heads(outs[:2])
                out5, g_real, g_fake = dis_loss_with_grads(d_real, d_fake, self.gan, self.flip_d)
                roots, seeds = [d_real, d_fake], [g_real, g_fake]
                loss_d = out5[0]
                if self.use_gp:
                    pen, v, tot = self.gp.from_input_gradient(outs[2], B, loss_add=out5[0:1])
                    roots.append(outs[2]); seeds.append(v)
                    loss_d = tot[0]
            else:
                d_real, d_fake = D.forward_heads([D.forward_stack(real_pm, pre=pre[0]), D.forward_stack(fake, pre=pre[1])])
                out5, g_real, g_fake = dis_loss_with_grads(d_real, d_fake, self.gan, self.flip_d)
                roots, seeds = [d_real, d_fake], [g_real, g_fake]
                loss_d = out5[0]
                if self.use_gp:
                    pen, gx, v = self.gp.with_grads_pm(D, x_hat, pre[2], B)
                    roots.append(gx); seeds.append(v)
                    loss_d = loss_d + pen[0]
            torch.autograd.backward(roots, seeds)
            if keep_grads:
                info["fake_d"] = ops.pm_to_cm(fake, B, N)
            info.update(loss_d=loss_d.detach(), real_acc=out5[3], fake_acc=out5[4])
            return ("pm", real_pm, B, N)
        real_t = ops.pm_to_cm(real.reshape(B * N, 3), B, N)                      # real_points.transpose(2,1)
        pre_hat = x_hat = None
        if self.batch_d_forwards and D.training and N % ops.ROW_TILE == 0 and tuple(fake.shape) == tuple(real_t.shape):
            ins = [real_t, fake]
            if self.use_gp:
                x_hat = self.gp.interpolate(real_t, fake, alpha)
                ins.append(x_hat)
            pre = D.forward_stacks_grouped(ins)
            d_real, d_fake = D.forward_heads([D.forward_stack(real_t, pre=pre[0]), D.forward_stack(fake, pre=pre[1])])
            if self.use_gp:
                pre_hat = pre[2]
        elif self.reference_schedule or not hasattr(D, "forward_many"):
            d_real = D(real_t)
            d_fake = D(fake)
        else:
            d_real, d_fake = D.forward_many(real_t, fake)        # same two passes, the BatchNorm-free head of both as one batch
        # lossD.backward() with the seeds written out: the loss kernel returns d loss / d logits, the penalty kernel d penalty / d
        # (input gradient); one backward pass from those three roots (no loss clones, no sum node, no multiplication by the seed 1)
        out5, g_real, g_fake = dis_loss_with_grads(d_real, d_fake, self.gan, self.flip_d)
        roots, seeds = [d_real, d_fake], [g_real, g_fake]
        loss_d = out5[0]
        if self.use_gp:
            pen, gx, v = self.gp.with_grads(D, real_t, fake, alpha=alpha, interpolates=x_hat, pre=pre_hat)
            roots.append(gx); seeds.append(v)
            loss_d = loss_d + pen[0]
        torch.autograd.backward(roots, seeds)
        if keep_grads:
            info["fake_d"] = fake
        info.update(loss_d=loss_d.detach(), real_acc=out5[3], fake_acc=out5[4])
        return real_t

    def _seg_gfwd(self, x, z_g, pm: bool = False):
        """The generator's forward of the G step (model.py:264-271 without D's half of the requires_grad toggle).  It reads neither
        D's weights nor D's gradients, so the data-parallel schedule issues it BEFORE optimizerD.step() (model.py:260), while D's
        gradient all-reduce is in flight; every tensor it produces is what the reference's order produces (G is untouched by D's
        update), and the order of all updates of D's and G's running statistics is unchanged."""
        G = self.G
        requires_grad(G, True)
        self.optG.zero_grad()
        G.twin_forward = "second" if self.twin_g_forwards else None
        try:
            with fused_grad_accumulation(self._sinkG):
                return G(x, z_g, pm_out=True) if pm else G(x, z_g)
        finally:
            G.twin_forward = None

    def _seg_g(self, x, real_t, z_g, scale_d, keep_grads, info, g_fake=None):
        """optimizerD.step(), then the G step up to lossG.backward() (model.py:259-277).  g_fake: the generator's forward when the
        caller already issued it (_seg_gfwd)."""
        with fused_grad_accumulation(self._sinkG):
            self._seg_g_body(x, real_t, z_g, scale_d, keep_grads, info, g_fake)
        self._sinkG.flush()

    def _seg_g_body(self, x, real_t, z_g, scale_d, keep_grads, info, g_fake=None):
        G, D = self.G, self.D
        if keep_grads:
            info["d_grads"] = {n: p.grad.detach().clone() * scale_d for n, p in D.named_parameters()}
        self.optD.step(scale_d)
        requires_grad(G, True); requires_grad(D, False)
        pm = isinstance(real_t, tuple)                      # ("pm", real_pm, B, N): the D step ran the point-major route
        if g_fake is None:
            g_fake = self._seg_gfwd(x, z_g, pm=pm)
        if pm:
            _, real_pm, B_, N_ = real_t
            g_fake_logit = D(g_fake, pre=D.forward_stack_after_stats_pass(real_pm, g_fake, pm_shape=(B_, N_)))
            out5, seed = gen_loss_with_grads(g_fake_logit, self.gan, self.flip_g)
            torch.autograd.backward([g_fake_logit], [seed])
            if keep_grads:
                info["fake_g"] = ops.pm_to_cm(g_fake.detach(), B_, N_)
            info["loss_g"] = out5[0]
            return
        # model.py:272-274: d_real = D(real) is computed but gen_loss ignores it (loss_utils.py:727-802) -- what lasts of that call
        # are D's BatchNorm running statistics, advanced here without the 1024-wide layer, the pool and the head
        g_real_logit = None
        if self.reference_schedule:
            g_real_logit = D(real_t)
            g_fake_logit = D(g_fake)
        elif (self.batch_d_forwards and hasattr(D, "forward_stack_after_stats_pass") and D.training and g_fake.shape[2] % ops.ROW_TILE == 0
              and tuple(g_fake.shape) == tuple(real_t.shape)):
            # the statistics pass of D(real) and the conv stack of D(G(z)) share their first three layers as one batch
            g_fake_logit = D(g_fake, pre=D.forward_stack_after_stats_pass(real_t, g_fake))
        else:
            D.advance_running_stats(real_t)
            g_fake_logit = D(g_fake)
        out5, seed = gen_loss_with_grads(g_fake_logit, self.gan, self.flip_g)      # gen_loss ignores d_real (loss_utils.py:727-802)
        torch.autograd.backward([g_fake_logit], [seed])
        if keep_grads:
            info["fake_g"] = g_fake.detach()
        info["loss_g"] = out5[0]

    def _seg_opt_g(self, scale_g, keep_grads, info):
        if keep_grads:
            info["g_grads"] = {n: p.grad.detach().clone() * scale_g for n, p in self.G.named_parameters()}
        self.optG.step(scale_g)

    def _eager_step(self, x, real, z_d, z_g, alpha=None, keep_grads: bool = False) -> Dict[str, torch.Tensor]:
        info: Dict[str, torch.Tensor] = {}
        real_t = self._seg_d(x, real, z_d, alpha, keep_grads, info, z_g=z_g if self.dpD is None else None)
        g_fake, scale = self._pair_gfake, 1.0
        self._pair_gfake = None
        if self.dpD is not None:
            # data parallel: the generator's forward of the G step is issued under D's gradient all-reduce (it depends on neither)
            self.dpD.allreduce_grads_begin()
            if self.overlap_g_forward:
                g_fake = self._seg_gfwd(x, z_g, pm=isinstance(real_t, tuple))
            scale = self.dpD.allreduce_grads_end()
        self._seg_g(x, real_t, z_g, scale, keep_grads, info, g_fake=g_fake)
        scale = self.dpG.allreduce_grads() if self.dpG is not None else 1.0
        self._seg_opt_g(scale, keep_grads, info)
        return info
