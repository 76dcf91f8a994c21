"""GPU: the hipGraph-captured train step replays exactly the eager step (same kernels, same order: bit-identical parameters,
buffers and optimiser state), including the host-side BatchNorm call counts and Adam's step count."""
import pytest
import torch

from spgan import fixture_rng as fr
from oracle import spgan_oracle as orc
from test_parity_gpu import Opts, _load, sp  # noqa: F401  (sp is a fixture)

pytestmark = pytest.mark.gpu


def _run(sp, graph, steps, B=4, N=256, flags=None, lr_change_at=None, sample_dump_at=()):
    flags = flags or {}
    o = type("O", (Opts,), flags)()
    G = _load(sp.Generator(o), fr.init_params(orc.generator_shapes(**flags), salt=8))
    D = _load(sp.Discriminator(o), fr.init_params(orc.discriminator_shapes(), salt=8))
    tr = sp.TrainStep(G, D, gan="wgan", use_gp=True, lambda_gp=10.0, lr_g=1e-4, lr_d=1e-4, graph=graph, graph_warmup=2)
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
    real = [fr.synthetic_real(B, N, seed=90 + i).cuda() for i in range(2)]
    zs = [fr.latent(B, N, seed=70 + i).cuda() for i in range(3)]
    alpha = fr.uniform("graph.alpha", (B, 1, 1), 0.0, 1.0).cuda()
    losses = []
    for i in range(steps):
        if lr_change_at is not None and i == lr_change_at:
            tr.optD.set_lr(5e-5); tr.optG.set_lr(2.5e-5)      # a schedule step (model.py:309-312) after the graph was captured
        if i in sample_dump_at:
            with torch.no_grad():                             # model.py:385-392: a sample dump between two iterations (train mode, no_grad)
                G(x, zs[2])
        info = tr.step(x, real[i % 2], zs[i % 3], zs[(i + 1) % 3], alpha=alpha)
        losses.append((info["loss_d"].item(), info["loss_g"].item()))
    torch.cuda.synchronize()
    return G, D, tr, losses


def test_graph_replay_equals_eager(sp):
    steps = 6                                    # 2 eager warm-up steps, capture + 4 replays
    Ge, De, tre, le = _run(sp, False, steps)
    Gg, Dg, trg, lg = _run(sp, True, steps)
    assert trg._graph is not None, "the step was never captured"
    assert le == lg, (le, lg)
    for (n, a), (_, b) in zip(list(Ge.state_dict().items()) + list(De.state_dict().items()),
                              list(Gg.state_dict().items()) + list(Dg.state_dict().items())):
        assert torch.equal(a, b), n
    assert (tre.optG.t, tre.optD.t) == (trg.optG.t, trg.optD.t) == (steps, steps)
    assert torch.equal(tre.optD.m, trg.optD.m) and torch.equal(tre.optG.v, trg.optG.v)
    assert int(trg.optD.dev_state[:1].view(torch.int32).item()) == steps


def test_graph_capture_after_an_eager_generator_call(sp):
    """An eager G(x, z) between the last warm-up step and the capture step (a periodic sample dump) fills the weight-derived host
    caches (permuted conv_out weights, transposes): the capture must still record the kernels that derive them -- otherwise every
    replay reads weights frozen at capture time.  Also one dump between replays."""
    steps = 7
    Ge, De, tre, le = _run(sp, False, steps, sample_dump_at=(2, 5))       # step 2 is the capture step (graph_warmup=2)
    Gg, Dg, trg, lg = _run(sp, True, steps, sample_dump_at=(2, 5))
    assert trg._graph is not None
    assert le == lg, (le, lg)
    for (n, a), (_, b) in zip(list(Ge.state_dict().items()) + list(De.state_dict().items()),
                              list(Gg.state_dict().items()) + list(Dg.state_dict().items())):
        assert torch.equal(a, b), n


def test_two_model_pairs_replaying_in_one_process(sp):
    """The weight-derived host caches (nets._T_CACHE / _WO_CACHE, ops.WEIGHTS_EPOCH_OF) are process-global and keyed by addresses:
    two independent (G, D, TrainStep(graph=True)) triples stepping ALTERNATELY in one process -- each capturing while the other
    already replays -- must end exactly where each ends when it runs alone."""
    B, N, steps = 4, 256, 6
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
    alpha = fr.uniform("graph.alpha", (B, 1, 1), 0.0, 1.0).cuda()

    def make(salt):
        o = Opts()
        G = _load(sp.Generator(o), fr.init_params(orc.generator_shapes(), salt=salt))
        D = _load(sp.Discriminator(o), fr.init_params(orc.discriminator_shapes(), salt=salt))
        return G, D, sp.TrainStep(G, D, gan="wgan", use_gp=True, lambda_gp=10.0, graph=True, graph_warmup=2 if salt == 8 else 3)

    def inputs(salt, i):
        return (x, fr.synthetic_real(B, N, seed=90 + salt + i % 2).cuda(), fr.latent(B, N, seed=70 + salt + i % 3).cuda(),
                fr.latent(B, N, seed=71 + salt + i % 3).cuda())

    def state(G, D):
        G.flush_bn_counts(); D.flush_bn_counts()
        return [v.detach().clone() for v in list(G.state_dict().values()) + list(D.state_dict().values())]
    solo = {}
    for salt in (8, 9):
        G, D, tr = make(salt)
        for i in range(steps):
            tr.step(*inputs(salt, i), alpha=alpha)
        torch.cuda.synchronize()
        assert tr._graph is not None
        solo[salt] = state(G, D)
    pairs = {salt: make(salt) for salt in (8, 9)}
    for i in range(steps):
        for salt in (8, 9):                                   # pair 8 captures at step 2 while pair 9 is still warming up, pair 9 at
            pairs[salt][2].step(*inputs(salt, i), alpha=alpha)   # step 3 while pair 8 already replays
    torch.cuda.synchronize()
    for salt in (8, 9):
        G, D, tr = pairs[salt]
        assert tr._graph is not None
        for a, b in zip(solo[salt], state(G, D)):
            assert torch.equal(a, b), "pair %d differs from its solo run" % salt


def test_graph_replay_follows_lr_schedule(sp):
    """A learning-rate change after capture reaches the replayed Adam kernels (device-side multiplier) bit-exactly."""
    steps = 7
    Ge, De, tre, le = _run(sp, False, steps, lr_change_at=4)
    Gg, Dg, trg, lg = _run(sp, True, steps, lr_change_at=4)
    Gn, Dn, trn, ln = _run(sp, True, steps)
    assert trg._graph is not None and le == lg and le != ln
    for (n, a), (_, b) in zip(list(Ge.state_dict().items()) + list(De.state_dict().items()),
                              list(Gg.state_dict().items()) + list(Dg.state_dict().items())):
        assert torch.equal(a, b), n
    assert trg.optD.get_lr() == 5e-5 and trg.optG.state_dict()["lr"] == 2.5e-5
    from spgan.optim import StepLR
    sch = StepLR(trg.optD, step_size=2, gamma=0.5)
    for _ in range(4):
        sch.step()
    assert abs(sch.get_last_lr()[0] - 5e-5 * 0.25) < 1e-12


def test_graph_replay_attn_eql_variant(sp):
    """--attn --eql --use_head through the whole train step: the captured step equals the eager one bit for bit (ScaleFn, the
    per-shape attention launches and the 0-d gate parameter all live inside the graph), and the gate actually trains."""
    flags = dict(attn=True, eql=True, use_head=True)
    steps = 5
    Ge, De, tre, le = _run(sp, False, steps, flags=flags)
    Gg, Dg, trg, lg = _run(sp, True, steps, flags=flags)
    assert trg._graph is not None, "the step was never captured"
    assert le == lg, (le, lg)
    assert all(torch.isfinite(torch.tensor(l)).all() for l in le)
    for (n, a), (_, b) in zip(list(Ge.state_dict().items()) + list(De.state_dict().items()),
                              list(Gg.state_dict().items()) + list(Dg.state_dict().items())):
        assert torch.equal(a, b), n
    assert Gg.attn.gamma.item() != 0.7 and abs(Gg.attn.gamma.item() - 0.7) < 1e-2       # moved by <= steps * lr


def test_graph_replay_data_parallel_segments(sp):
    """Data-parallel mode captures four graphs (D step | G forward of the G step, replayed under D's all-reduce | Adam(D) + rest of the G
    step | Adam(G)) with the RCCL all-reduces issued eagerly between them (one rank here): bit-identical to the sequential eager step."""
    import os
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not dist.is_initialized():
        dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1)
    try:
        steps = 5
        Ge, De, tre, le = _run(sp, False, steps)

        def run_dp():
            o = Opts()
            G = _load(sp.Generator(o), fr.init_params(orc.generator_shapes(), salt=8))
            D = _load(sp.Discriminator(o), fr.init_params(orc.discriminator_shapes(), salt=8))
            tr = sp.TrainStep(G, D, gan="wgan", use_gp=True, lambda_gp=10.0, lr_g=1e-4, lr_d=1e-4, graph=True, graph_warmup=2, distributed=True)
            B, N = 4, 256
            x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
            real = [fr.synthetic_real(B, N, seed=90 + i).cuda() for i in range(2)]
            zs = [fr.latent(B, N, seed=70 + i).cuda() for i in range(3)]
            alpha = fr.uniform("graph.alpha", (B, 1, 1), 0.0, 1.0).cuda()
            for i in range(steps):
                tr.step(x, real[i % 2], zs[i % 3], zs[(i + 1) % 3], alpha=alpha)
            torch.cuda.synchronize()
            return G, D, tr
        Gg, Dg, trg = run_dp()
        assert len(trg._graph) == 4 and trg._graph[1] is not None
        for (n, a), (_, b) in zip(list(Ge.state_dict().items()) + list(De.state_dict().items()),
                                  list(Gg.state_dict().items()) + list(Dg.state_dict().items())):
            assert torch.equal(a, b), n
    finally:
        dist.destroy_process_group()


def test_graph_capture_failure_falls_back_to_eager(sp, monkeypatch):
    """A capture that throws must not cost the run: the step falls back to eager issue with identical results."""
    steps = 5
    Ge, De, tre, le = _run(sp, False, steps)

    class Boom:
        def __init__(self, *a, **k):
            raise RuntimeError("simulated capture failure")
    monkeypatch.setattr(torch.cuda, "CUDAGraph", Boom)
    with pytest.warns(UserWarning, match="falling back to eager"):
        Gg, Dg, trg, lg = _run(sp, True, steps)
    assert trg.use_graph is False and trg._graph is None
    assert le == lg
    for (n, a), (_, b) in zip(list(Ge.state_dict().items()) + list(De.state_dict().items()),
                              list(Gg.state_dict().items()) + list(Dg.state_dict().items())):
        assert torch.equal(a, b), n
    assert (trg.optG.t, trg.optD.t) == (steps, steps)


def _state(G, D):
    return list(G.state_dict().items()) + list(D.state_dict().items())


def test_graph_replay_with_fresh_temporaries_every_step(sp):
    """The caller hands over freshly allocated tensors on every step and frees them afterwards (examples/train.py does): the
    caching allocator returns the just-freed blocks, so the new tensors recur at old addresses with version 0.  The graph's
    static buffers must nevertheless hold the NEW content on every replay."""
    steps, B, N = 8, 4, 256

    def run(graph):
        G = _load(sp.Generator(Opts()), fr.init_params(orc.generator_shapes(), salt=8))
        D = _load(sp.Discriminator(Opts()), fr.init_params(orc.discriminator_shapes(), salt=8))
        tr = sp.TrainStep(G, D, gan="wgan", use_gp=True, lambda_gp=10.0, graph=graph, graph_warmup=2)
        x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
        losses, ptrs = [], []
        for i in range(steps):
            real = fr.synthetic_real(B, N, seed=300 + i).cuda() * 1.0          # temporaries: products of an op, freed after the step
            z_d = fr.latent(B, N, seed=400 + i).cuda() * 1.0
            z_g = fr.latent(B, N, seed=500 + i).cuda() * 1.0
            alpha = fr.uniform("tmp.alpha.%d" % i, (B, 1, 1), 0.0, 1.0).cuda() * 1.0
            ptrs.append((real.data_ptr(), z_d.data_ptr(), z_g.data_ptr(), real._version))
            info = tr.step(x, real, z_d, z_g, alpha=alpha)
            losses.append((info["loss_d"].item(), info["loss_g"].item()))
            del real, z_d, z_g, alpha
        torch.cuda.synchronize()
        return G, D, tr, losses, ptrs

    Ge, De, _, le, _ = run(False)
    Gg, Dg, trg, lg, ptrs = run(True)
    assert trg._graph is not None
    assert len({p[:3] for p in ptrs[3:]}) < len(ptrs[3:]), "the scenario needs recurring addresses to mean anything"
    assert le == lg, (le, lg)
    for (n, a), (_, b) in zip(_state(Ge, De), _state(Gg, Dg)):
        assert torch.equal(a, b), n


def test_graph_mode_survives_a_changed_sphere_prior(sp):
    """The capture depends on the kNN graph of x (cached per tensor and version, so no kNN launch is inside the graph).  A new x
    after the capture must not replay stale neighbours: the harness steps eagerly once, re-captures, and stays equal to eager."""
    steps, B, N = 9, 4, 256

    def run(graph):
        G = _load(sp.Generator(Opts()), fr.init_params(orc.generator_shapes(), salt=8))
        D = _load(sp.Discriminator(Opts()), fr.init_params(orc.discriminator_shapes(), salt=8))
        tr = sp.TrainStep(G, D, gan="ls", graph=graph, graph_warmup=2)
        xa = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
        perm = torch.randperm(N, generator=torch.Generator().manual_seed(1))
        xb = (fr.sphere_template(N)[perm] * 0.9)[None].repeat(B, 1, 1).cuda()            # a different prior (other points, other graph)
        real = fr.synthetic_real(B, N, seed=90).cuda()
        zs = [fr.latent(B, N, seed=70 + i).cuda() for i in range(2)]
        losses = []
        for i in range(steps):
            x = xa if i < 5 else xb
            if i == 7:
                xb.mul_(1.05)                                                             # ... and an in-place change of the same tensor
            info = tr.step(x, real, zs[0], zs[1])
            losses.append((info["loss_d"].item(), info["loss_g"].item()))
        torch.cuda.synchronize()
        return G, D, tr, losses

    Ge, De, _, le = run(False)
    Gg, Dg, trg, lg = run(True)
    assert trg._recaptures == 2 and trg._graph is not None and trg.use_graph
    assert le == lg, (le, lg)
    for (n, a), (_, b) in zip(_state(Ge, De), _state(Gg, Dg)):
        assert torch.equal(a, b), n


def test_graph_mode_with_noisy_labels(sp):
    """flip_d / flip_g (noise_label=True in dis_loss / gen_loss): the labels are drawn on the device inside the captured step, so the
    capture succeeds and every replay draws new ones (the loss of identical inputs differs from replay to replay)."""
    B, N = 4, 256
    G = _load(sp.Generator(Opts()), fr.init_params(orc.generator_shapes(), salt=8))
    D = _load(sp.Discriminator(Opts()), fr.init_params(orc.discriminator_shapes(), salt=8))
    tr = sp.TrainStep(G, D, gan="ls", flip_d=True, flip_g=True, lr_g=0.0, lr_d=0.0, graph=True, graph_warmup=2)
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
    real = fr.synthetic_real(B, N, seed=90).cuda()
    z = fr.latent(B, N, seed=70).cuda()
    vals = []
    for i in range(7):
        info = tr.step(x, real, z, z)
        vals.append(info["loss_d"].item())
    assert tr._graph is not None and tr.use_graph, "capture must succeed with device-side label draws"
    # lr = 0: parameters never move, BatchNorm is in train mode -> the only thing that changes between replays is the label draw
    assert len(set(vals[3:])) == len(vals[3:]), vals


def test_native_rccl_allreduce_flat_one_rank_and_in_a_graph(sp):
    """spgan_allreduce_flat (include/spgan_hip.h): the library's own RCCL communicator (dlopen'ed librccl, id drawn
    by spgan_comm_unique_id) -- exercised with the one rank this box has: the sum over one rank is the buffer itself, eagerly and
    captured into a hipGraph together with a kernel before and after it; and a data-parallel TrainStep with collective='rccl'
    equals the single-process step (world size 1: scale 1)."""
    import ctypes as C
    import os
    import torch.distributed as dist
    from spgan import _lib
    lib = _lib.load()
    assert lib.spgan_comm_available() == 1, "librccl not loadable on the GPU box"
    ident = (C.c_ubyte * 128)()
    assert lib.spgan_comm_unique_id(ident) == 0
    comm = C.c_void_p()
    assert lib.spgan_comm_init(ident, 0, 1, C.byref(comm)) == 0, lib.spgan_comm_last_error()
    assert lib.spgan_comm_world(comm) == 1
    buf = fr.normal("rccl.buf", (980353 + 3,)).cuda()
    want = buf.clone()
    assert lib.spgan_allreduce_flat(comm, buf.data_ptr(), buf.numel(), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(buf, want)
    # captured: scale -> all-reduce -> scale, replayed three times
    static = buf.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        static.mul_(1.0)
        assert lib.spgan_allreduce_flat(comm, static.data_ptr(), static.numel(), side.cuda_stream) == 0
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        static.mul_(2.0)
        assert lib.spgan_allreduce_flat(comm, static.data_ptr(), static.numel(), torch.cuda.current_stream().cuda_stream) == 0
        static.add_(1.0)
    static.copy_(want)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    ref = want.clone()
    for _ in range(3):
        ref = ref * 2.0 + 1.0
    assert torch.equal(static, ref)
    assert lib.spgan_comm_destroy(comm) == 0
    assert lib.spgan_allreduce_flat(None, buf.data_ptr(), 4, None) != 0            # argument validation, not a crash
    # the data-parallel harness over it
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not dist.is_initialized():
        dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29534", rank=0, world_size=1)
    os.environ["SPGAN_DP_COLLECTIVE"] = "rccl"
    try:
        steps = 4
        Ge, De, tre, le = _run(sp, False, steps)
        o = Opts()
        G = _load(sp.Generator(o), fr.init_params(orc.generator_shapes(), salt=8))
        D = _load(sp.Discriminator(o), fr.init_params(orc.discriminator_shapes(), salt=8))
        tr = sp.TrainStep(G, D, gan="wgan", use_gp=True, lambda_gp=10.0, lr_g=1e-4, lr_d=1e-4, distributed=True)
        assert tr.dpD.collective == "rccl"
        B, N = 4, 256
        x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
        real = [fr.synthetic_real(B, N, seed=90 + i).cuda() for i in range(2)]
        zs = [fr.latent(B, N, seed=70 + i).cuda() for i in range(3)]
        alpha = fr.uniform("graph.alpha", (B, 1, 1), 0.0, 1.0).cuda()
        for i in range(steps):
            tr.step(x, real[i % 2], zs[i % 3], zs[(i + 1) % 3], alpha=alpha)
            tr.dpD._allreduce_native(); tr.dpG._allreduce_native()       # world 1: allreduce_grads() skips the call; issue it explicitly (identity)
        torch.cuda.synchronize()
        for (n, a), (_, b) in zip(list(Ge.state_dict().items()) + list(De.state_dict().items()),
                                  list(G.state_dict().items()) + list(D.state_dict().items())):
            assert torch.equal(a, b), n
    finally:
        os.environ.pop("SPGAN_DP_COLLECTIVE", None)
        dist.destroy_process_group()


def test_point_major_route_equals_the_channel_major_one():
    """TrainStep's internal point-major route (generator output and real cloud handed to the Discriminator as [B*N,3], no [B,3,N] round trips)
    feeds the same values into the same kernels; the one difference is the gradient penalty's per-shape norm, whose 3N squares are summed in
    the other memory order (a last-bit difference that reaches D's gradients).  Compared on what the routes compute -- losses, every
    gradient of both networks, the generated clouds -- at rounding level relative to each tensor's own scale.  (Not on the weights after
    Adam: its first updates are +-lr * sign-like for entries whose gradient is itself rounding noise, e.g. the columns of tail.0.weight
    that only see the penalty's last bits -- tools/exp/pm_route_diff.py shows up to 0.4 lr there on equal gradients.)"""
    import spgan
    B, N = 4, 256
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
    out = []
    for pm in (True, False):
        o = Opts()
        G = _load(spgan.Generator(o), fr.init_params(orc.generator_shapes(), salt=31))
        D = _load(spgan.Discriminator(o), fr.init_params(orc.discriminator_shapes(), salt=31))
        tr = spgan.TrainStep(G, D, gan="wgan", use_gp=True)
        tr.point_major = pm
        torch.manual_seed(1234)
        info = tr.step(x, fr.synthetic_real(B, N, seed=40).cuda(), fr.latent(B, N, seed=50).cuda(), fr.latent(B, N, seed=60).cuda(), keep_grads=True)
        torch.cuda.synchronize()
        out.append(info)
    a, b = out
    assert a["loss_d"].item() == pytest.approx(b["loss_d"].item(), rel=1e-6) and a["loss_g"].item() == pytest.approx(b["loss_g"].item(), rel=1e-6)
    assert torch.equal(a["fake_d"], b["fake_d"]), "the D step's generated clouds come from the same kernels"
    for which in ("d_grads", "g_grads"):
        assert a[which].keys() == b[which].keys()
        for k in a[which]:
            ga, gb = a[which][k].double(), b[which][k].double()
            # G's gradients are taken through the UPDATED discriminator (model.py:259-277), so they carry D's Adam amplification: 1e-3
            tol = 1e-5 if which == "d_grads" else 1e-3
            assert (ga - gb).abs().max().item() <= tol * gb.abs().max().item() + 1e-12, (which, k)


def test_gradients_dirtied_between_replays_need_invalidate_grads(sp):
    """Advisor (round 4): with zero_grad folded into the Adam kernel a captured step records no gradient fill, so anything written into
    the flat `.grad` buffers between two replays (a diagnostic backward on D, a caller's regulariser) is added to the next step.  The
    contract: p.grad reads as zero after step(); whoever writes there calls `optimizer.invalidate_grads()` before the next step -- then
    the trajectory is the undisturbed one, bit for bit."""
    steps = 6
    Ge, De, tre, le = _run(sp, True, steps)
    B, N = 4, 256
    o = Opts()
    G = _load(sp.Generator(o), fr.init_params(orc.generator_shapes(), salt=8))
    D = _load(sp.Discriminator(o), fr.init_params(orc.discriminator_shapes(), salt=8))
    tr = sp.TrainStep(G, D, gan="wgan", use_gp=True, lambda_gp=10.0, lr_g=1e-4, lr_d=1e-4, graph=True, graph_warmup=2)
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
    real = [fr.synthetic_real(B, N, seed=90 + i).cuda() for i in range(2)]
    zs = [fr.latent(B, N, seed=70 + i).cuda() for i in range(3)]
    alpha = fr.uniform("graph.alpha", (B, 1, 1), 0.0, 1.0).cuda()
    losses = []
    for i in range(steps):
        if i == 4:                                            # between two replays
            assert not tr.optD.fp.grad.any().item() and not tr.optG.fp.grad.any().item(), "p.grad must read as zero after step()"
            tr.optD.fp.grad.add_(1.0); tr.optG.fp.grad.add_(-1.0)
            tr.optD.invalidate_grads(); tr.optG.invalidate_grads()
        info = tr.step(x, real[i % 2], zs[i % 3], zs[(i + 1) % 3], alpha=alpha)
        losses.append((info["loss_d"].item(), info["loss_g"].item()))
    assert tr._graph is not None and losses == le
    for (n, a), (_, b) in zip(list(Ge.state_dict().items()) + list(De.state_dict().items()), list(G.state_dict().items()) + list(D.state_dict().items())):
        assert torch.equal(a, b), n


def test_point_major_route_is_only_taken_for_the_common_mixing_rule(sp):
    """Advisor (round 4): TrainStep's point-major D step builds x_hat itself as real + alpha*(fake - real) (gradient_penalty.py:24-25); a
    GradientPenalty with mix="loss_utils" (alpha*real + (1-alpha)*fake, loss_utils.py:1108) must keep the route that calls its
    interpolate(): the same TrainStep with either flag setting then penalises the same points."""
    B, N = 4, 256
    res = []
    for pm in (True, False):
        o = Opts()
        G = _load(sp.Generator(o), fr.init_params(orc.generator_shapes(), salt=8))
        D = _load(sp.Discriminator(o), fr.init_params(orc.discriminator_shapes(), salt=8))
        tr = sp.TrainStep(G, D, gan="wgan", use_gp=True, lambda_gp=10.0)
        tr.gp = sp.GradientPenalty(10.0, gamma=1, mix="loss_utils")
        tr.point_major = pm
        x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
        alpha = fr.uniform("graph.alpha", (B, 1, 1), 0.0, 1.0).cuda()
        info = tr.step(x, fr.synthetic_real(B, N, seed=90).cuda(), fr.latent(B, N, seed=70).cuda(), fr.latent(B, N, seed=71).cuda(), alpha=alpha,
                       keep_grads=True)
        res.append(info)
    assert res[0]["loss_d"].item() == res[1]["loss_d"].item()
    for n in res[0]["d_grads"]:
        assert torch.equal(res[0]["d_grads"][n], res[1]["d_grads"][n]), n
