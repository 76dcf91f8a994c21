"""CPU: losses, GradientPenalty and the full D-step/G-step harness (spgan.train.TrainStep) with HIP ops replaced by
their kernel models, against the golden vectors captured from the real reference (G6, G8)."""
import numpy as np
import pytest
import torch

from helpers import check, check_adam_updates, golden, load_mid_state, measured_grad_errors
from oracle import spgan_oracle as orc
from spgan import fixture_rng as fr
from test_host_cpu import Opts, ZERO_GRAD_BIASES, spgan_cpu, _load   # noqa: F401  (fixture import)
import kernel_model as km


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
    # batch 40: labels that really flipped, D side and G side
    dr = torch.from_numpy(d["b40|d_real"]).requires_grad_(True)
    df = torch.from_numpy(d["b40|d_fake"]).requires_grad_(True)
    l, _ = spgan.dis_loss(dr, df, gan="ls", real_label=torch.from_numpy(d["b40|dis|ls_noisy|real_label"]))
    l.backward()
    np.testing.assert_allclose(l.item(), float(d["b40|dis|ls_noisy|loss"]), rtol=1e-5)
    np.testing.assert_allclose(dr.grad.numpy(), d["b40|dis|ls_noisy|g_real"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(df.grad.numpy(), d["b40|dis|ls_noisy|g_fake"], rtol=1e-4, atol=1e-7)
    df2 = torch.from_numpy(d["b40|d_fake"]).requires_grad_(True)
    l, _ = spgan.gen_loss(None, df2, gan="ls", fake_label=torch.from_numpy(d["b40|gen|ls_noisy|fake_label"]))
    l.backward()
    np.testing.assert_allclose(l.item(), float(d["b40|gen|ls_noisy|loss"]), rtol=1e-5)
    np.testing.assert_allclose(df2.grad.numpy(), d["b40|gen|ls_noisy|g_fake"], rtol=1e-4, atol=1e-7)


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
    # post-Adam parameters as UPDATES p - p0 on above-noise elements (helpers.check_adam_updates)
    for kind, net, shapes in (("d", D, orc.discriminator_shapes()), ("g", G, orc.generator_shapes())):
        check_adam_updates(d, kind, net.named_parameters(), fr.init_params(shapes, salt=8), [kind + "grad|"],
                           measured_grad_errors(d, kind + "grad|", info[kind + "_grads"], skip=ZERO_GRAD_BIASES), skip=ZERO_GRAD_BIASES, what=tag)
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


def test_autograd_contract_outside_trainstep(spgan_cpu):
    """Outside TrainStep parameter gradients travel through autograd as usual: torch.autograd.grad returns them and leaves
    `.grad` alone even when the module was flattened (p.grad pre-bound), tensor hooks fire; only inside
    `fused_grad_accumulation()` are they added straight into the flat buffer."""
    import spgan
    from spgan.functions import fused_grad_accumulation
    D = _load(spgan.Discriminator(Opts), fr.init_params(orc.discriminator_shapes(), salt=21)).train()
    spgan.flatten_module(D)
    x = fr.synthetic_real(2, 128, seed=22).transpose(2, 1).contiguous()
    params = list(D.parameters())
    before = [p.grad.clone() for p in params]
    seen = []
    h = params[0].register_hook(lambda g: seen.append(g.clone()))
    grads = torch.autograd.grad(D(x).sum(), params)
    assert all(g is not None for g in grads) and len(seen) == 1
    assert all(torch.equal(p.grad, b) for p, b in zip(params, before)), "autograd.grad must not touch .grad"
    h.remove()
    D(x).sum().backward()                                             # plain backward: AccumulateGrad adds into the bound slices
    flat = D._spgan_flat
    assert all(p.grad.data_ptr() == flat.grad.data_ptr() + 4 * off for p, off in zip(flat.params, flat.offsets))
    for p, g in zip(params, grads):
        assert torch.allclose(p.grad, g, rtol=1e-6, atol=1e-8)
    flat.zero_grad()
    with fused_grad_accumulation():
        D(x).sum().backward()
    for p, g in zip(params, grads):
        assert torch.allclose(p.grad, g, rtol=1e-6, atol=1e-8)


def test_backward_modes_are_thread_local_and_recorded_per_graph(spgan_cpu):
    """SURVEY 8(b) "no global mutable state": `fused_grad_accumulation()` / `input_grad_only()` held open by ANOTHER thread do not change
    what this thread's graphs do, and a graph keeps the mode it was BUILT under (autograd runs its backward on an engine thread)."""
    import threading
    import spgan
    from spgan.functions import fused_grad_accumulation, input_grad_only
    D = _load(spgan.Discriminator(Opts), fr.init_params(orc.discriminator_shapes(), salt=21)).train()
    spgan.flatten_module(D)
    x = fr.synthetic_real(2, 128, seed=22).transpose(2, 1).contiguous()
    params = list(D.parameters())
    entered, release = threading.Event(), threading.Event()

    def other():
        with fused_grad_accumulation(), input_grad_only():
            entered.set()
            release.wait(30)
    t = threading.Thread(target=other)
    t.start()
    assert entered.wait(30)
    try:
        before = [p.grad.clone() for p in params]
        grads = torch.autograd.grad(D(x).sum(), params)                  # plain semantics: every parameter gradient is returned ...
        assert all(g is not None for g in grads)
        assert all(torch.equal(p.grad, b) for p, b in zip(params, before))   # ... and .grad is untouched
    finally:
        release.set(); t.join()
    # built outside, differentiated inside the context: the graph keeps the mode it was built under (plain)
    y = D(x).sum()
    with fused_grad_accumulation():
        g2 = torch.autograd.grad(y, params)
    assert all(g is not None for g in g2)
    # built inside: fused (nothing returned to autograd for leaf parameters with a bound .grad; the sums land in .grad)
    D._spgan_flat.zero_grad()
    with fused_grad_accumulation():
        y = D(x).sum()
    y.backward()
    for p, g in zip(params, grads):
        assert torch.allclose(p.grad, g, rtol=1e-6, atol=1e-8)


def test_load_state_dict_discards_pending_bn_counts(spgan_cpu):
    import spgan
    D = spgan.Discriminator(Opts).train()
    x = fr.synthetic_real(2, 128, seed=23).transpose(2, 1).contiguous()
    D(x); D(x)
    sd = {k: v.clone() for k, v in D.state_dict().items()}            # flushes: 2 calls recorded
    assert int(sd["mlps.1.num_batches_tracked"]) == 2
    D(x)                                                              # one more, pending on the host
    sd["mlps.1.num_batches_tracked"] = torch.tensor(7)
    D.load_state_dict(sd)
    assert int(D.state_dict()["mlps.1.num_batches_tracked"]) == 7    # not 8: the pre-load call is not added afterwards
    D(x)
    assert int(D.state_dict()["mlps.1.num_batches_tracked"]) == 8


def test_noisy_labels_semantics(spgan_cpu):
    """loss_utils.py:698-725: smooth labels in [0.9,1), int(0.05*B) positions flipped to 1-y; drawn with torch's generator."""
    from spgan import losses
    torch.manual_seed(5)
    y = losses._smooth_labels(64, torch.device("cpu"))
    assert y.shape == (64,) and float(y.min()) >= 0.9 and float(y.max()) < 1.0
    y2 = losses._noisy_labels(y.clone())
    changed = (y2 != y).nonzero().flatten()
    assert 1 <= changed.numel() <= 3                                   # int(0.05*64) = 3 draws with replacement
    assert torch.allclose(y2[changed], 1 - y[changed])
    assert torch.equal(losses._noisy_labels(torch.ones(8)), torch.ones(8))   # int(0.05*8) = 0: nothing flips


@pytest.mark.parametrize("tag,gan,use_gp,B,N", [("ls", "ls", False, 4, 512), ("wgangp", "wgan", True, 4, 256)])
def test_literal_reference_loop_body_golden(spgan_cpu, tag, gan, use_gp, B, N):
    """CPU twin of tests/test_literal_loop_gpu.py: the reference's loop body statement for statement (examples/reference_loop.py,
    Generation/model.py:239-279) with torch.optim.Adam over the host pipelines (kernel-model doubles) against golden G8."""
    import spgan
    from reference_loop import LoopState, reference_loop_body
    d = golden("g8_train_step_%s.npz" % tag)
    G = _load(spgan.Generator(Opts), fr.init_params(orc.generator_shapes(), salt=8)).train()
    D = _load(spgan.Discriminator(Opts), fr.init_params(orc.discriminator_shapes(), salt=8)).train()
    optG = torch.optim.Adam(filter(lambda p: p.requires_grad, G.parameters()), lr=1e-4, betas=(0.5, 0.99))
    optD = torch.optim.Adam(filter(lambda p: p.requires_grad, D.parameters()), lr=1e-4, betas=(0.5, 0.99))
    alpha = torch.from_numpy(d["alpha"])
    gp = (lambda netD, real, fake: spgan.GradientPenalty(10.0, gamma=1)(netD, real, fake, alpha=alpha)) if use_gp else None
    s = LoopState(G, D, optG, optD, gan=gan, gp=gp)
    s.keep = {}
    x = fr.sphere_template(N)[None].repeat(B, 1, 1)
    lossD, lossG, _ = reference_loop_body(s, x, fr.synthetic_real(B, N, seed=81), fr.latent(B, N, seed=82), fr.latent(B, N, seed=83))
    np.testing.assert_allclose(lossD.item(), float(d["lossD"]), rtol=2e-3 if use_gp else 1e-4)
    np.testing.assert_allclose(lossG.item(), float(d["lossG"]), rtol=2e-3)
    check(d, "fake_d", s.keep["fake_d"], rtol=2e-4)
    for n, g in s.keep["d_grads"].items():
        check(d, "dgrad|" + n, g, rtol=3e-2, atol=2e-3 if n.endswith(ZERO_GRAD_BIASES) else 1e-7)
    for kind, net, shapes in (("d", D, orc.discriminator_shapes()), ("g", G, orc.generator_shapes())):
        check_adam_updates(d, kind, net.named_parameters(), fr.init_params(shapes, salt=8), [kind + "grad|"],
                           measured_grad_errors(d, kind + "grad|", s.keep[kind + "_grads"], skip=ZERO_GRAD_BIASES), skip=ZERO_GRAD_BIASES)
    for n, b in [(k, v) for k, v in D.state_dict().items() if k in dict(D.named_buffers())]:
        np.testing.assert_allclose(b.numpy(), d["dbuf|" + n], rtol=2e-3, atol=2e-4)


def test_step_from_mid_training_state_g20(spgan_cpu):
    """CPU twin of tests/test_multistep_golden_gpu.py::test_one_step_from_a_mid_training_state (C1): the host side of state carry --
    BatchNorm buffers and call counts loaded through load_state_dict, spgan.optim.Adam.load_state_dict, one TrainStep from there --
    over the kernel models, against the reference started from the same fixture state (golden G20)."""
    import spgan
    tag, B, N, salt = "c1_ls", 4, 512, 21
    d = golden("g20_mid_state_step_%s.npz" % tag)

    class O(Opts):
        np = N
    init_g, init_d = fr.init_params(orc.generator_shapes(), salt=salt), fr.init_params(orc.discriminator_shapes(), salt=salt)
    G = _load(spgan.Generator(O), init_g)
    D = _load(spgan.Discriminator(O, num_point=N), init_d)
    tr = spgan.TrainStep(G, D, gan="ls", use_gp=False)
    load_mid_state(D, tr.optD, orc.discriminator_shapes(), salt, 21)
    load_mid_state(G, tr.optG, orc.generator_shapes(), salt, 14)
    x = fr.sphere_template(N)[None].repeat(B, 1, 1)
    G.inject_graph2([torch.from_numpy(d[w].astype(np.int64)).view(B, N * 10) for w in ("idx2_d", "idx2_g")])
    info = tr.step(x, fr.synthetic_real(B, N, seed=2001), fr.latent(B, N, seed=2002), fr.latent(B, N, seed=2003), keep_grads=True)
    np.testing.assert_allclose(info["loss_d"].item(), float(d["lossD"]), rtol=1e-4)
    np.testing.assert_allclose(info["loss_g"].item(), float(d["lossG"]), rtol=2e-3)
    check(d, "fake_g", info["fake_g"], rtol=2e-4)
    for n, g in info["d_grads"].items():
        check(d, "dgrad|" + n, g, rtol=3e-2, atol=2e-3 if n.endswith(ZERO_GRAD_BIASES) else 1e-7)
    assert tr.optD.t == 8 and tr.optG.t == 8
    for kind, net, init in (("d", D, init_d), ("g", G, init_g)):
        check_adam_updates(d, kind, net.named_parameters(), init, [], None, skip=ZERO_GRAD_BIASES, select_by_gradient=False, atol=2e-6,
                           min_selected=0.5, what="kernel models, step 8")
    for kind, net, calls in (("d", D, 25), ("g", G, 16)):
        names = dict(net.named_buffers())
        for n, b in [(k_, v_) for k_, v_ in net.state_dict().items() if k_ in names]:
            if n.endswith("num_batches_tracked"):
                assert int(b.item()) == int(d["%sbuf|%s" % (kind, n)]) == calls, n
            else:
                np.testing.assert_allclose(b.numpy(), d["%sbuf|%s" % (kind, n)], rtol=2e-3, atol=2e-4, err_msg=n)


@pytest.mark.parametrize("gan,use_gp,B,N", [("ls", False, 4, 256), ("wgan", True, 4, 256), ("hinge", True, 2, 512)])
def test_joint_d_backward_equals_one_node_per_pass(spgan_cpu, monkeypatch, gan, use_gp, B, N):
    """TrainStep's D step with the conv stacks of all passes behind ONE autograd node (Discriminator.stacks_joint -> nets.d_backward_joint: the
    backward work of the real pass, the fake pass and the penalty's double backward in lock step) against one node per pass: same losses,
    same gradients (the per-parameter sums are formed in another order: 1e-6), and the joint route is the one that ran."""
    import spgan
    from spgan import nets
    calls = []
    real_joint = nets.d_backward_joint
    monkeypatch.setattr(nets, "d_backward_joint", lambda P, firsts, dbl=None: (calls.append((len(firsts), dbl is not None)), real_joint(P, firsts, dbl))[1])
    x = fr.sphere_template(N)[None].repeat(B, 1, 1)
    real = fr.synthetic_real(B, N, seed=91)
    z_d, z_g = fr.latent(B, N, seed=92), fr.latent(B, N, seed=93)
    alpha = fr.uniform("joint.alpha", (B, 1, 1), 0.0, 1.0)
    outs = []
    for joint in (True, False):
        G = _load(spgan.Generator(Opts), fr.init_params(orc.generator_shapes(), salt=9))
        D = _load(spgan.Discriminator(Opts), fr.init_params(orc.discriminator_shapes(), salt=9))
        tr = spgan.TrainStep(G, D, gan=gan, use_gp=use_gp, lambda_gp=10.0)
        tr.joint_d_backward = joint
        outs.append((tr.step(x, real, z_d, z_g, alpha=alpha, keep_grads=True), {k: v.clone() for k, v in D.state_dict().items()}))
    assert calls == [(2, use_gp)], calls
    (a, sa), (b, sb) = outs
    np.testing.assert_allclose(a["loss_d"].item(), b["loss_d"].item(), rtol=1e-6)
    np.testing.assert_allclose(a["loss_g"].item(), b["loss_g"].item(), rtol=1e-5)
    for n in a["d_grads"]:
        ga, gb = a["d_grads"][n], b["d_grads"][n]
        assert (ga - gb).norm().item() <= 1e-6 * gb.norm().item() + 1e-12, n
    for k in sa:      # (the G step's D passes run on the UPDATED weights, which carry the other summation order: not bit-equal)
        if "running" in k or "num_batches" in k:
            assert torch.allclose(sa[k].float(), sb[k].float(), rtol=1e-5, atol=1e-6), k


@pytest.mark.parametrize("joint", [True, False])
def test_collapsed_layer_as_a_pair_of_launches_equals_the_fused_launch(spgan_cpu, monkeypatch, joint):
    """The split-bf16 mode's route through the collapsed 256 -> 1024 layer backward (ops.collapsed_pair_preferred: gemm_tn + gemm_nt_bnbwd with the
    phase-B tail and the stored X = xbarA + gamma*g, instead of ONE gemm_dual launch): same step, in the joint D-step node and with one node per
    pass; the double backward's lazy phase B stays on (the pair returns the coefficient tail)."""
    import spgan
    B, N = 4, 256
    x = fr.sphere_template(N)[None].repeat(B, 1, 1)
    real = fr.synthetic_real(B, N, seed=91)
    z_d, z_g = fr.latent(B, N, seed=92), fr.latent(B, N, seed=93)
    alpha = fr.uniform("joint.alpha", (B, 1, 1), 0.0, 1.0)
    seen = []
    real_bnbwd = spgan_cpu.ops.gemm_nt_bnbwd
    monkeypatch.setattr(spgan_cpu.ops, "gemm_nt_bnbwd", lambda *a, **k: (seen.append((k.get("phaseb") is not None, k.get("gout") is not None)), real_bnbwd(*a, **k))[1])
    outs = []
    for pair in (False, True):
        monkeypatch.setattr(km, "SPLIT_PAIR", [pair])
        seen.clear()
        G = _load(spgan.Generator(Opts), fr.init_params(orc.generator_shapes(), salt=9))
        D = _load(spgan.Discriminator(Opts), fr.init_params(orc.discriminator_shapes(), salt=9))
        tr = spgan.TrainStep(G, D, gan="wgan", use_gp=True, lambda_gp=10.0)
        tr.joint_d_backward = joint
        outs.append(tr.step(x, real, z_d, z_g, alpha=alpha, keep_grads=True))
        if pair:      # the penalty pass's launch carried phase B's sums and the stored adjoint
            assert (True, True) in seen, seen
    a, b = outs
    np.testing.assert_allclose(a["loss_d"].item(), b["loss_d"].item(), rtol=1e-6)
    for n in a["d_grads"]:
        ga, gb = a["d_grads"][n], b["d_grads"][n]
        assert (ga - gb).norm().item() <= 2e-6 * gb.norm().item() + 1e-12, n


@pytest.mark.parametrize("gan,use_gp", [("wgan", True), ("ls", False)])
def test_joint_d_backward_with_two_launch_layers(spgan_cpu, monkeypatch, gan, use_gp):
    """The joint D-step node where no fused layer-backward kernel exists (the "f16" operand mode: nets._joint_two_launch): every layer as gemm_tn +
    gemm_nt_bnbwd per pass, phase B's sums from the finalize tail, the grouped pool / collapse launches kept -- against one node per pass."""
    import spgan
    from spgan import nets
    monkeypatch.setattr(km, "GEMM_DUAL", [False])
    monkeypatch.setattr(nets, "_joint_two_launch", lambda: True)
    calls = []
    real_joint = nets.d_backward_joint
    monkeypatch.setattr(nets, "d_backward_joint", lambda P, firsts, dbl=None: (calls.append((len(firsts), dbl is not None)), real_joint(P, firsts, dbl))[1])
    B, N = 4, 256
    x = fr.sphere_template(N)[None].repeat(B, 1, 1)
    real = fr.synthetic_real(B, N, seed=91)
    z_d, z_g = fr.latent(B, N, seed=92), fr.latent(B, N, seed=93)
    alpha = fr.uniform("joint.alpha", (B, 1, 1), 0.0, 1.0)
    outs = []
    for joint in (True, False):
        G = _load(spgan.Generator(Opts), fr.init_params(orc.generator_shapes(), salt=9))
        D = _load(spgan.Discriminator(Opts), fr.init_params(orc.discriminator_shapes(), salt=9))
        tr = spgan.TrainStep(G, D, gan=gan, use_gp=use_gp, lambda_gp=10.0)
        tr.joint_d_backward = joint
        outs.append(tr.step(x, real, z_d, z_g, alpha=alpha, keep_grads=True))
    assert calls == [(2, use_gp)], calls
    a, b = outs
    np.testing.assert_allclose(a["loss_d"].item(), b["loss_d"].item(), rtol=1e-6)
    for n in a["d_grads"]:
        ga, gb = a["d_grads"][n], b["d_grads"][n]
        assert (ga - gb).norm().item() <= 2e-6 * gb.norm().item() + 1e-12, n


@pytest.mark.parametrize("gan,use_gp", [("wgan", True), ("ls", False)])
def test_paired_generator_forwards_equal_separate_forwards(spgan_cpu, monkeypatch, gan, use_gp):
    """TrainStep with the step's two generator forwards evaluated as ONE pipeline (Generator.forward_pair -> nets.g_pair_forward: the per-point /
    per-shape stages once on the rows of both passes, the BatchNorm stages per pass) against two separate forwards: two steps, same losses,
    gradients, parameters and BatchNorm buffers; and the paired route is the one that ran."""
    import spgan
    from spgan import nets
    calls = []
    real_pair = nets.g_pair_forward
    monkeypatch.setattr(nets, "g_pair_forward", lambda *a, **k: (calls.append(1), real_pair(*a, **k))[1])
    B, N = 4, 256
    x = fr.sphere_template(N)[None].repeat(B, 1, 1)
    outs = []
    for pair in (True, False):
        G = _load(spgan.Generator(Opts), fr.init_params(orc.generator_shapes(), salt=9))
        D = _load(spgan.Discriminator(Opts), fr.init_params(orc.discriminator_shapes(), salt=9))
        tr = spgan.TrainStep(G, D, gan=gan, use_gp=use_gp, lambda_gp=10.0)
        tr.pair_g_forwards = pair
        infos = []
        for step in range(2):
            real = fr.synthetic_real(B, N, seed=61 + step)
            z_d, z_g = fr.latent(B, N, seed=62 + 2 * step)[:, :1, :].contiguous(), fr.latent(B, N, seed=63 + 2 * step)[:, :1, :].contiguous()
            alpha = fr.uniform("pair.alpha.%d" % step, (B, 1, 1), 0.0, 1.0)
            infos.append(tr.step(x, real, z_d, z_g, alpha=alpha, keep_grads=True))
        outs.append((infos, {k: v.clone() for k, v in G.state_dict().items()}, {k: v.clone() for k, v in D.state_dict().items()}))
    assert len(calls) == 2, calls
    (ia, ga, da), (ib, gb, db) = outs
    for step, (a, b) in enumerate(zip(ia, ib)):
        if step == 1:      # behind the first Adam step (+-lr on every element: the sign of a noise-level gradient element may differ): loose
            np.testing.assert_allclose(a["fake_d"].numpy(), b["fake_d"].numpy(), rtol=1e-3, atol=1e-3)
            np.testing.assert_allclose(a["loss_d"].item(), b["loss_d"].item(), rtol=2e-2, atol=1e-3)
            continue
        np.testing.assert_allclose(a["loss_d"].item(), b["loss_d"].item(), rtol=2e-3 if use_gp else 1e-5)     # the penalty is a function of a kink-limited gradient
        np.testing.assert_allclose(a["loss_g"].item(), b["loss_g"].item(), rtol=2e-3, atol=1e-5)
        # (the CPU doubles' matrix products block differently for M and 2M rows: last-bit differences per row, which reach the gradients through
        # LeakyReLU / arg-max kinks; the GPU test compares the HIP kernels, whose rows are independent of the row count)
        np.testing.assert_allclose(a["fake_d"].numpy(), b["fake_d"].numpy(), rtol=1e-5, atol=5e-6)
        np.testing.assert_allclose(a["fake_g"].numpy(), b["fake_g"].numpy(), rtol=1e-5, atol=5e-6)
        for n in a["g_grads"]:
            assert (a["g_grads"][n] - b["g_grads"][n]).norm().item() <= 2e-3 * b["g_grads"][n].norm().item() + 1e-9, n
    for k in ga:       # parameters after two Adam steps move by <= 2e-4; buffers by the batch statistics' rounding
        assert torch.allclose(ga[k].float(), gb[k].float(), rtol=1e-4, atol=4.1e-4 if "num_batches" not in k else 0), k
    for k in da:
        assert torch.allclose(da[k].float(), db[k].float(), rtol=1e-4, atol=4.1e-4 if "num_batches" not in k else 0), k


def test_delivery_sink_forgets_an_aborted_backward(spgan_cpu):
    """A backward pass that raised leaves its gradients in the step's DeliverySink; the next step must not add them (TrainStep clears both sinks
    where a step starts)."""
    import spgan
    from spgan.functions import DeliverySink
    s = DeliverySink()
    d, g = torch.zeros(8), torch.ones(8)
    s.add([(d, g)])
    s.clear()
    s.flush()
    assert float(d.sum()) == 0.0
    s.add([(d, g), (d, 2 * g)])
    s.flush()
    assert torch.equal(d, torch.full((8,), 3.0))
    B, N = 4, 256
    x = fr.sphere_template(N)[None].repeat(B, 1, 1)
    outs = []
    for dirty in (True, False):
        G = _load(spgan.Generator(Opts), fr.init_params(orc.generator_shapes(), salt=9))
        D = _load(spgan.Discriminator(Opts), fr.init_params(orc.discriminator_shapes(), salt=9))
        tr = spgan.TrainStep(G, D, gan="ls", use_gp=False)
        if dirty:      # what an aborted backward would have left
            tr._sinkD.add([(tr.optD.fp.grad, torch.ones_like(tr.optD.fp.grad))])
            tr._sinkG.add([(tr.optG.fp.grad, torch.ones_like(tr.optG.fp.grad))])
        outs.append(tr.step(x, fr.synthetic_real(B, N, seed=3), fr.latent(B, N, seed=4)[:, :1].contiguous(), fr.latent(B, N, seed=5)[:, :1].contiguous(),
                            keep_grads=True))
    for kind in ("d_grads", "g_grads"):
        for n in outs[0][kind]:
            assert torch.equal(outs[0][kind][n], outs[1][kind][n]), (kind, n)


def test_delivery_sink_orders_overlapping_destinations(spgan_cpu):
    """A CatCols column block of a parameter's .grad and the whole .grad of the SAME parameter (two nodes differentiating one generator
    forward each) are different destinations sharing elements: they must not meet in one grouped launch (functions.DeliverySink.flush sends
    them through stream-ordered single launches); disjoint column blocks of one matrix stay on the grouped path."""
    from spgan import functions as F
    from spgan import ops
    grad = torch.zeros(6, 10)
    left, right = grad.view(6, -1)[:, :4], grad.view(6, -1)[:, 4:]
    assert not F._share_elements(left, right) and F._share_elements(left, grad) and F._share_elements(right, grad)
    assert not F._share_elements(grad[:3], grad[3:]) and F._share_elements(grad[:4], grad[3:])
    assert F._overlapping_keys([left, right]) == set() and F._overlapping_keys([left, right, grad]) == {0, 1, 2}
    other = torch.zeros(7)
    launches = []
    real_add = ops.multi_add
    def spy(dsts, srcs):
        launches.append([d.data_ptr() for d in dsts])
        return real_add(dsts, srcs)
    ops.multi_add = spy
    try:
        s = F.DeliverySink()
        a, b, w, o = torch.full((6, 4), 1.0), torch.full((6, 6), 2.0), torch.full((6, 10), 4.0), torch.full((7,), 8.0)
        s.add([(left, a), (right, b), (other, o)])       # node 1: column blocks
        s.add([(grad, w)])                               # node 2: the whole gradient of the same parameter
        s.flush()
    finally:
        ops.multi_add = real_add
    want = torch.cat([torch.full((6, 4), 5.0), torch.full((6, 6), 6.0)], 1)
    assert torch.equal(grad, want) and torch.equal(other, torch.full((7,), 8.0))
    clash_ptrs = {left.data_ptr(), right.data_ptr(), grad.data_ptr()}
    for l in launches:       # no launch holds two of the clashing destinations
        assert len([p for p in l if p in clash_ptrs]) <= 1 or l == [other.data_ptr()], l
    assert [other.data_ptr()] in launches
