"""GPU: the hipGraph-captured train step replays exactly the eager step (same kernels, same order: bit-identical parameters,
buffers and optimiser state), including the host-side BatchNorm call counts and Adam's step count."""
import pytest
import torch

from spgan import fixture_rng as fr
from oracle import spgan_oracle as orc
from test_parity_gpu import Opts, _load, sp  # noqa: F401  (sp is a fixture)

pytestmark = pytest.mark.gpu


def _run(sp, graph, steps, B=4, N=256, flags=None, lr_change_at=None):
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
    """Data-parallel mode captures three graphs with the RCCL all-reduces issued eagerly between them (one rank here)."""
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
        assert len(trg._graph) == 3
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
