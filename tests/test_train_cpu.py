"""CPU: losses, GradientPenalty and the full D-step/G-step harness (spgan.train.TrainStep) with HIP ops replaced by
their kernel models, against the golden vectors captured from the real reference (G6, G8)."""
import numpy as np
import pytest
import torch

from helpers import check, golden
from oracle import spgan_oracle as orc
from spgan import fixture_rng as fr
from test_host_cpu import Opts, ZERO_GRAD_BIASES, spgan_cpu, _load   # noqa: F401  (fixture import)


def test_losses_golden(spgan_cpu):
    import spgan
    d = golden("g6_losses.npz")
    for gan in ("ls", "wgan", "hinge", "gan"):
        dr = torch.from_numpy(d["d_real"]).requires_grad_(True)
        df = torch.from_numpy(d["d_fake"]).requires_grad_(True)
        l, info = spgan.dis_loss(dr, df, gan=gan)
        l.backward()
        np.testing.assert_allclose(l.item(), float(d["dis|%s|loss" % gan]), rtol=1e-5)
        np.testing.assert_allclose(dr.grad.numpy(), d["dis|%s|g_real" % gan], rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(df.grad.numpy(), d["dis|%s|g_fake" % gan], rtol=1e-4, atol=1e-7)
        df2 = torch.from_numpy(d["d_fake"]).requires_grad_(True)
        l, _ = spgan.gen_loss(dr.detach(), df2, gan=gan)
        l.backward()
        np.testing.assert_allclose(l.item(), float(d["gen|%s|loss" % gan]), rtol=1e-5)
        np.testing.assert_allclose(df2.grad.numpy(), d["gen|%s|g_fake" % gan], rtol=1e-4, atol=1e-7)
    dr = torch.from_numpy(d["d_real"]).requires_grad_(True)
    df = torch.from_numpy(d["d_fake"]).requires_grad_(True)
    l, _ = spgan.dis_loss(dr, df, gan="ls", real_label=torch.from_numpy(d["dis|ls_noisy|real_label"]))
    l.backward()
    np.testing.assert_allclose(l.item(), float(d["dis|ls_noisy|loss"]), rtol=1e-5)
    np.testing.assert_allclose(dr.grad.numpy(), d["dis|ls_noisy|g_real"], rtol=1e-4, atol=1e-7)
    with pytest.raises(NotImplementedError):
        spgan.dis_loss(dr, df, gan="nope")


@pytest.mark.parametrize("tag,gan,use_gp,B,N", [("ls", "ls", False, 4, 512), ("wgangp", "wgan", True, 4, 256)])
def test_train_step_golden(spgan_cpu, tag, gan, use_gp, B, N):
    import spgan
    d = golden("g8_train_step_%s.npz" % tag)
    G = _load(spgan.Generator(Opts), fr.init_params(orc.generator_shapes(), salt=8))
    D = _load(spgan.Discriminator(Opts), fr.init_params(orc.discriminator_shapes(), salt=8))
    tr = spgan.TrainStep(G, D, gan=gan, use_gp=use_gp, lambda_gp=10.0)
    x = fr.sphere_template(N)[None].repeat(B, 1, 1)
    real = fr.synthetic_real(B, N, seed=81)
    z_d, z_g = fr.latent(B, N, seed=82), fr.latent(B, N, seed=83)
    info = tr.step(x, real, z_d, z_g, alpha=torch.from_numpy(d["alpha"]), keep_grads=True)
    np.testing.assert_allclose(info["loss_d"].item(), float(d["lossD"]), rtol=2e-3 if use_gp else 1e-4)   # GP is a function of a kink-limited gradient
    np.testing.assert_allclose(info["loss_g"].item(), float(d["lossG"]), rtol=2e-3)
    check(d, "fake_d", info["fake_d"], rtol=2e-4)
    for n, g in info["d_grads"].items():
        check(d, "dgrad|" + n, g, rtol=3e-2, atol=2e-3 if n.endswith(ZERO_GRAD_BIASES) else 1e-7)
    for n, g in info["g_grads"].items():
        # after D's Adam step (+-lr per element) and through D's kinks: loose by nature
        check(d, "ggrad|" + n, g, rtol=1.5e-1, atol=2e-3 if n.endswith(ZERO_GRAD_BIASES) else 1e-7)
    for n, p in D.named_parameters():
        if not n.endswith(ZERO_GRAD_BIASES):
            check(d, "dparam|" + n, p, rtol=1e-3, atol=2.5e-4)      # one Adam step moves an element by <= lr; sign noise => 2*lr
    for n, p in G.named_parameters():
        if not n.endswith(ZERO_GRAD_BIASES):
            check(d, "gparam|" + n, p, rtol=1e-3, atol=2.5e-4)      # one Adam step moves an element by <= lr; sign noise => 2*lr
    for n, b in [(k, v) for k, v in G.state_dict().items() if k in dict(G.named_buffers())]:
        np.testing.assert_allclose(b.numpy(), d["gbuf|" + n], rtol=2e-3, atol=2e-4)
    for n, b in [(k, v) for k, v in D.state_dict().items() if k in dict(D.named_buffers())]:
        np.testing.assert_allclose(b.numpy(), d["dbuf|" + n], rtol=2e-3, atol=2e-4)
    # second step runs (optimizer state, re-bound flat gradients)
    tr.step(x, real, z_d, z_g, alpha=torch.from_numpy(d["alpha"]))
    assert tr.optD.t == 2 and tr.optG.t == 2


def test_checkpoint_interchange(spgan_cpu):
    """state_dict written by our modules loads into a fresh one after flattening (views keep names/shapes)."""
    import spgan
    G = spgan.Generator(Opts)
    spgan.flatten_module(G)
    sd = {k: v.clone() for k, v in G.state_dict().items()}
    G2 = spgan.Generator(Opts)
    G2.load_state_dict(sd)
    for (n, a), (_, b) in zip(G.named_parameters(), G2.named_parameters()):
        assert torch.equal(a, b), n
