"""CPU: the oracle (oracle/spgan_oracle.py) against the vectors captured from the real
reference (tests/golden/make_golden.py).  This is what pins the oracle."""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import check, check_adam_updates, golden, measured_grad_errors, params_from, worst_reference_noise
from oracle import spgan_oracle as orc
from spgan import fixture_rng as fr


# conv/linear biases that feed a train-mode BatchNorm have mathematically zero gradient; fp32
# yields rounding noise there -> compared with an absolute bound only.
ZERO_GRAD_BIASES = ("conv_w.0.bias", "conv_w.3.bias", "conv_x.0.bias", "global_conv.0.bias", "global_conv.3.bias",
                    "mlps.0.bias", "mlps.3.bias", "mlps.6.bias", "fc2.0.bias")


def _sub(params, prefix):
    return {k: v for k, v in params.items() if k.startswith(prefix)}


# ---------------------------------------------------------------- G1
@pytest.mark.parametrize("N", [256, 512, 1024, 2048, 4096])
def test_sphere_knn_exact(N):
    d = golden("g1_edge_features.npz")
    x = fr.sphere_template(N)[None].transpose(2, 1).contiguous()
    ref = d["sphere%d|idx" % N]
    if N <= 2048:
        assert np.array_equal(orc.knn_sorted(x, 10)[0].numpy(), ref)
    # the fp64 direct-difference order reproduces the reference's fp32 order on every template
    assert np.array_equal(orc.knn_sorted_fp64_direct(x, 10)[0].numpy(), ref)
    if N <= 512:
        ee = orc.get_edge_features(x, 10)
        assert np.array_equal(ee.numpy(), d["sphere%d|ee" % N])


@pytest.mark.parametrize("N,C", [(256, 64), (512, 64), (300, 5)])
def test_feature_knn(N, C):
    d = golden("g1_edge_features.npz")
    x = fr.normal("g1.feat.%d.%d" % (N, C), (2, C, N), 0.5)
    ee, idx = orc.get_edge_features(x, 10, return_idx=True)
    assert np.array_equal(idx.view(2, N, 10).numpy(), d["feat%d_%d|idx" % (N, C)])
    check(d, "feat%d_%d|ee" % (N, C), ee, rtol=0, atol=0)


# ---------------------------------------------------------------- G2
@pytest.mark.parametrize("tag,fin,fout", [("ec1", 3, 64), ("ec2", 64, 128)])
def test_edgeblock(tag, fin, fout):
    d = golden("g2_edgeblock.npz")
    B, N = 2, 256
    pref = "EdgeConv1" if fin == 3 else "EdgeConv2"
    p = params_from(_sub(orc.generator_shapes(), pref + "."), 2, requires_grad=True)
    buf = orc.bn_buffers({k: tuple(v.shape) for k, v in p.items()})
    if fin == 3:
        x = fr.sphere_template(N)[None].repeat(B, 1, 1).transpose(2, 1).contiguous()
        x = x + 0.01 * fr.normal("g2.jit", x.shape)
    else:
        x = fr.normal("g2.x.%s" % tag, (B, fin, N), 0.7)
    x.requires_grad_(True)
    idx = torch.from_numpy(d[tag + "|idx"].astype(np.int64))
    y, idx_own = orc.edge_block(p, pref, x, 10, training=True, buffers=buf, return_idx=True)
    assert np.array_equal(idx_own.view(B, N, 10).numpy(), idx.numpy())
    check(d, tag + "|y", y, rtol=2e-6)
    dy = fr.normal("g2.dy.%s" % tag, y.shape)
    names = list(p.keys())
    grads = torch.autograd.grad(y, [x] + [p[n] for n in names], dy)
    check(d, tag + "|dx", grads[0], rtol=2e-5)
    for n, g in zip(names, grads[1:]):
        check(d, tag + "|grad|" + n[len(pref) + 1:], g, rtol=5e-5, atol=1e-3 if n.endswith(ZERO_GRAD_BIASES) else 1e-7)
    for k in buf:
        np.testing.assert_allclose(buf[k].numpy(), d[tag + "|buf|" + k[len(pref) + 1:]], rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------- G3
def test_adain():
    d = golden("g3_adain.npz")
    B, C, N = 2, 64, 256
    p = params_from({"a.style.weight": (2 * C, 128, 1), "a.style.bias": (2 * C,)}, 3)
    p = {k: v.clone().requires_grad_(True) for k, v in fr.init_params({"style.weight": (2 * C, 128, 1), "style.bias": (2 * C,)}, salt=3).items()}
    p = {"a." + k: v for k, v in p.items()}
    x = fr.normal("g3.x", (B, C, N)).requires_grad_(True)
    s = fr.normal("g3.s", (B, 128, N), 0.3).requires_grad_(True)
    y = orc.adaptive_point_norm(p, "a", x, s)
    dy = fr.normal("g3.dy", y.shape)
    gx, gs, gw, gb = torch.autograd.grad(y, [x, s, p["a.style.weight"], p["a.style.bias"]], dy)
    for n, t in (("y", y), ("dx", gx), ("dstyle", gs), ("dw", gw), ("db", gb)):
        check(d, n, t, rtol=5e-6)


# ---------------------------------------------------------------- G4 / G5
def test_discriminator():
    d = golden("g5_discriminator.npz")
    B, N = 4, 256
    p = params_from(orc.discriminator_shapes(), 4, requires_grad=True)
    buf = orc.bn_buffers(orc.discriminator_shapes())
    real = fr.synthetic_real(B, N, seed=5).transpose(2, 1).contiguous().requires_grad_(True)
    logit = orc.discriminator_forward(p, real, True, buf)
    check(d, "logit", logit, rtol=2e-6)
    loss = ((logit - 1.0) ** 2).mean()
    names = list(p.keys())
    grads = torch.autograd.grad(loss, [real] + [p[n] for n in names])
    check(d, "dx", grads[0], rtol=2e-5)
    for n, g in zip(names, grads[1:]):
        check(d, "grad|" + n, g, rtol=5e-5, atol=1e-3 if n.endswith(ZERO_GRAD_BIASES) else 1e-7)
    for k in buf:
        np.testing.assert_allclose(buf[k].numpy(), d["buf|" + k], rtol=1e-5, atol=1e-6)


def test_generator():
    d = golden("g4_generator.npz")
    B, N = 4, 256
    gp = params_from(orc.generator_shapes(), 4, requires_grad=True)
    dp = params_from(orc.discriminator_shapes(), 4)
    gbuf = orc.bn_buffers(orc.generator_shapes())
    dbuf = orc.bn_buffers(orc.discriminator_shapes())
    x = fr.sphere_template(N)[None].repeat(B, 1, 1)
    z = fr.latent(B, N, seed=44)
    st = {}
    out = orc.generator_forward(gp, x, z, training=True, buffers=gbuf, stages=st)
    assert np.array_equal(st["idx1"].view(B, N, 10).numpy(), d["idx1"])
    assert np.array_equal(st["idx2"].view(B, N, 10).numpy(), d["idx2"])
    for n in ("style", "x1", "x2"):
        check(d, "stage|" + n, st[n], rtol=5e-6)
    # BatchNorm1d over only B samples amplifies rounding differences
    check(d, "stage|feat_global", st["feat_global"], rtol=1e-4)
    check(d, "out", out, rtol=1e-4)
    logit = orc.discriminator_forward(dp, out, True, dbuf)
    loss = ((logit - 1.0) ** 2).mean()
    np.testing.assert_allclose(loss.item(), float(d["loss"]), rtol=1e-5)
    names = list(gp.keys())
    dy = fr.normal("g4.dy", out.shape)
    grads = torch.autograd.grad(out, [gp[n] for n in names], dy)
    for n, g in zip(names, grads):
        # pre-BN conv biases have mathematically zero gradient: fp32 noise only
        # whole-network gradients are kink-limited: 2 LeakyReLU sign flips out of 65k units (forward
        # diff 8e-6) already move them by 1e-2 rel-L2, the reference in fp32 vs itself in fp64
        # included.  Tight gradient pins are the block goldens G2/G3/G5.
        check(d, "grad|" + n, g, rtol=3e-2, atol=1e-3 if n.endswith(ZERO_GRAD_BIASES) else 1e-7)
    for k in gbuf:
        np.testing.assert_allclose(gbuf[k].numpy(), d["buf|" + k], rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------- G6
def test_losses():
    d = golden("g6_losses.npz")
    dr = torch.from_numpy(d["d_real"]).requires_grad_(True)
    df = torch.from_numpy(d["d_fake"]).requires_grad_(True)
    for gan in ("ls", "wgan", "hinge", "gan"):
        l = orc.dis_loss(dr, df, gan)
        gr, gf = torch.autograd.grad(l, [dr, df])
        np.testing.assert_allclose(l.item(), float(d["dis|%s|loss" % gan]), rtol=1e-6)
        np.testing.assert_allclose(gr.numpy(), d["dis|%s|g_real" % gan], rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(gf.numpy(), d["dis|%s|g_fake" % gan], rtol=1e-5, atol=1e-8)
        l = orc.gen_loss(dr, df, gan)
        gf, = torch.autograd.grad(l, [df])
        np.testing.assert_allclose(l.item(), float(d["gen|%s|loss" % gan]), rtol=1e-6)
        np.testing.assert_allclose(gf.numpy(), d["gen|%s|g_fake" % gan], rtol=1e-5, atol=1e-8)
    rl = torch.from_numpy(d["dis|ls_noisy|real_label"])
    l = orc.dis_loss(dr, df, "ls", real_label=rl)
    gr, gf = torch.autograd.grad(l, [dr, df])
    np.testing.assert_allclose(l.item(), float(d["dis|ls_noisy|loss"]), rtol=1e-6)
    np.testing.assert_allclose(gr.numpy(), d["dis|ls_noisy|g_real"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(gf.numpy(), d["dis|ls_noisy|g_fake"], rtol=1e-5, atol=1e-8)
    # batch 40: labels that really flipped, on the D side (:897-901) and on the G side (:753-755)
    dr = torch.from_numpy(d["b40|d_real"]).requires_grad_(True)
    df = torch.from_numpy(d["b40|d_fake"]).requires_grad_(True)
    l = orc.dis_loss(dr, df, "ls", real_label=torch.from_numpy(d["b40|dis|ls_noisy|real_label"]))
    gr, gf = torch.autograd.grad(l, [dr, df])
    np.testing.assert_allclose(l.item(), float(d["b40|dis|ls_noisy|loss"]), rtol=1e-6)
    np.testing.assert_allclose(gr.numpy(), d["b40|dis|ls_noisy|g_real"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(gf.numpy(), d["b40|dis|ls_noisy|g_fake"], rtol=1e-5, atol=1e-8)
    l = orc.gen_loss(dr, df, "ls", fake_label=torch.from_numpy(d["b40|gen|ls_noisy|fake_label"]))
    gf, = torch.autograd.grad(l, [df])
    np.testing.assert_allclose(l.item(), float(d["b40|gen|ls_noisy|loss"]), rtol=1e-6)
    np.testing.assert_allclose(gf.numpy(), d["b40|gen|ls_noisy|g_fake"], rtol=1e-5, atol=1e-8)


# ---------------------------------------------------------------- G7
def test_gradient_penalty():
    d = golden("g7_gradient_penalty.npz")
    B, N = 3, 256
    p = params_from(orc.discriminator_shapes(), 7, requires_grad=True)
    buf = orc.bn_buffers(orc.discriminator_shapes())
    real = fr.synthetic_real(B, N, seed=71).transpose(2, 1).contiguous()
    fake = (0.8 * fr.synthetic_real(B, N, seed=72) + 0.05 * fr.normal("g7.n", (B, N, 3))).transpose(2, 1).contiguous()
    alpha = torch.from_numpy(d["alpha"])
    gp = orc.gradient_penalty(lambda t: orc.discriminator_forward(p, t, True, buf), real, fake, alpha, 10.0, 1.0)
    np.testing.assert_allclose(gp.item(), float(d["gp"]), rtol=2e-5)
    names = list(p.keys())
    grads = torch.autograd.grad(gp, [p[n] for n in names], allow_unused=True)
    for n, g in zip(names, grads):
        g = torch.zeros_like(p[n]) if g is None else g
        check(d, "grad|" + n, g, rtol=2e-4, atol=1e-3 if n.endswith(ZERO_GRAD_BIASES) else 1e-7)
    xh = (real + alpha * (fake - real)).requires_grad_(True)
    gin, = torch.autograd.grad(orc.discriminator_forward(p, xh, True, None).sum(), xh)
    check(d, "input_grad", gin, rtol=2e-5)


# ---------------------------------------------------------------- G8
@pytest.mark.parametrize("tag,gan,use_gp,B,N", [("ls", "ls", False, 4, 512), ("wgangp", "wgan", True, 4, 256)])
def test_train_step(tag, gan, use_gp, B, N):
    d = golden("g8_train_step_%s.npz" % tag)
    gp_ = params_from(orc.generator_shapes(), 8, requires_grad=True)
    dp_ = params_from(orc.discriminator_shapes(), 8, requires_grad=True)
    gbuf = orc.bn_buffers(orc.generator_shapes()); dbuf = orc.bn_buffers(orc.discriminator_shapes())
    optG, optD = orc.AdamState(gp_), orc.AdamState(dp_)
    x = fr.sphere_template(N)[None].repeat(B, 1, 1)
    real = fr.synthetic_real(B, N, seed=81)
    z_d, z_g = fr.latent(B, N, seed=82), fr.latent(B, N, seed=83)
    alpha = torch.from_numpy(d["alpha"])
    out = orc.train_step(gp_, gbuf, dp_, dbuf, optG, optD, x, real, z_d, z_g, gan=gan, use_gp=use_gp, alpha=alpha)
    np.testing.assert_allclose(out["loss_d"].item(), float(d["lossD"]), rtol=2e-5)
    check(d, "fake_d", out["fake_d"], rtol=1e-4)
    for n, g in out["d_grads"].items():
        check(d, "dgrad|" + n, g, rtol=3e-2, atol=1e-3 if n.endswith(ZERO_GRAD_BIASES) else 1e-7)
    # G-step: runs after D's Adam update (each element moves by ~+-lr, sign-noise for ~0 grads) and
    # is kink-limited end-to-end: loose bounds here, tight ones in the block tests
    check(d, "fake_g", out["fake_g"], rtol=2e-4)
    np.testing.assert_allclose(out["loss_g"].item(), float(d["lossG"]), rtol=2e-3)
    for n, g in out["g_grads"].items():
        check(d, "ggrad|" + n, g, rtol=5e-2, atol=1e-3 if n.endswith(ZERO_GRAD_BIASES) else 1e-7)
    # post-Adam parameters as UPDATES p - p0 on the elements whose golden gradient is above the noise floor (helpers.check_adam_updates;
    # the zero-gradient biases random-walk under Adam)
    for kind, params, grads, salt_shapes in (("d", dp_, out["d_grads"], orc.discriminator_shapes()), ("g", gp_, out["g_grads"], orc.generator_shapes())):
        check_adam_updates(d, kind, params.items(), fr.init_params(salt_shapes, salt=8), [kind + "grad|"],
                           measured_grad_errors(d, kind + "grad|", grads, skip=ZERO_GRAD_BIASES), skip=ZERO_GRAD_BIASES, what="oracle")
    for k, v in gbuf.items():
        np.testing.assert_allclose(v.numpy(), d["gbuf|" + k], rtol=2e-3, atol=2e-4)
    for k, v in dbuf.items():
        np.testing.assert_allclose(v.numpy(), d["dbuf|" + k], rtol=2e-3, atol=2e-4)


# ---------------------------------------------------------------- G19: three consecutive steps (state carry)
def test_three_steps_c1_g19():
    """The oracle over THREE reference steps at C1 (B=4, N=512, LS; golden G19: fresh inputs per step, the reference's EdgeConv2 graphs
    injected): Adam at step >= 2, the running statistics' later updates in call order, num_batches_tracked.  No float32 implementation
    reproduces a multi-step trajectory to rounding (Adam's first updates are +-lr for noise-level gradients, D's kinks amplify): every
    bound is 3 x the reference's OWN float32-vs-float64 distance of that quantity at that step (golden `noise|...`; for gradient and
    moment tensors the worst tensor of the network at that step), floored at the one-step tolerances of G8 above."""
    tag, B, N, salt = "c1_ls", 4, 512, 19
    d = golden("g19_three_steps_%s.npz" % tag)
    bound = lambda key, floor: max(3.0 * float(d["noise|" + key]), floor)
    nbound = lambda prefix, floor: max(3.0 * worst_reference_noise(d, prefix, ZERO_GRAD_BIASES), floor)
    gp_ = params_from(orc.generator_shapes(), salt, requires_grad=True)
    dp_ = params_from(orc.discriminator_shapes(), salt, requires_grad=True)
    gbuf = orc.bn_buffers(orc.generator_shapes()); dbuf = orc.bn_buffers(orc.discriminator_shapes())
    optG, optD = orc.AdamState(gp_), orc.AdamState(dp_)
    x = fr.sphere_template(N)[None].repeat(B, 1, 1)
    derr, gerr = {}, {}
    for k in range(3):
        real, z_d, z_g = fr.synthetic_real(B, N, seed=1900 + 10 * k), fr.latent(B, N, seed=1901 + 10 * k), fr.latent(B, N, seed=1902 + 10 * k)
        pre = "s%d|" % k
        graphs = tuple(torch.from_numpy(d[pre + w].astype(np.int64)).view(B, N * 10) for w in ("idx2_d", "idx2_g"))
        out = orc.train_step(gp_, gbuf, dp_, dbuf, optG, optD, x, real, z_d, z_g, gan="ls", use_gp=False, graphs=graphs)
        np.testing.assert_allclose(out["loss_d"].item(), float(d[pre + "lossD"]), rtol=bound(pre + "lossD", 2e-5))
        np.testing.assert_allclose(out["loss_g"].item(), float(d[pre + "lossG"]), rtol=bound(pre + "lossG", 2e-3))
        check(d, pre + "fake_g", out["fake_g"], rtol=bound(pre + "fake_g", 2e-4))
        for kind, grads, floor, errs in (("dgrad", out["d_grads"], 3e-2, derr), ("ggrad", out["g_grads"], 5e-2, gerr)):
            for n, g in grads.items():
                check(d, pre + kind + "|" + n, g, rtol=nbound(pre + kind + "|", floor), atol=1e-3 if n.endswith(ZERO_GRAD_BIASES) else 1e-7)
            for n, e in measured_grad_errors(d, pre + kind + "|", grads, skip=ZERO_GRAD_BIASES).items():
                errs.setdefault(n, []).append(e)
    assert optD.step == 3 and optG.step == 3
    for kind, params, opt, errs, shapes, floor in (("d", dp_, optD, derr, orc.discriminator_shapes(), 3e-2), ("g", gp_, optG, gerr, orc.generator_shapes(), 5e-2)):
        for n in params:
            if n.endswith(ZERO_GRAD_BIASES):
                continue
            check(d, "%sm|%s" % (kind, n), opt.m[n], rtol=nbound("%sm|" % kind, floor), atol=1e-9)
            check(d, "%sv|%s" % (kind, n), opt.v[n], rtol=nbound("%sv|" % kind, 2 * floor), atol=1e-16)
        check_adam_updates(d, kind, params.items(), fr.init_params(shapes, salt=salt), ["s%d|%sgrad|" % (k, kind) for k in range(3)], errs,
                           skip=ZERO_GRAD_BIASES, what="oracle, 3 steps", ref_noise_prefix="noise_rms|", min_selected=0.0, min_ok=0.95)      # min_selected 0: G's gradients are 26-118 % noise after three C1 steps -- nothing stands 20 x above it (logged)
    for kind, bufs, calls in (("d", dbuf, 12), ("g", gbuf, 6)):
        for n, b in bufs.items():
            ref = d["%sbuf|%s" % (kind, n)]
            if n.endswith("num_batches_tracked"):
                assert int(b) == int(ref) == calls, n
            else:
                tol_abs = 3.0 * np.abs(ref.astype(np.float64) - d["%sbuf64|%s" % (kind, n)]).max() + 2e-3 * np.abs(ref) + 2e-4      # worst channel of the buffer
                assert (np.abs(b.numpy() - ref) <= tol_abs).all(), n


# ---------------------------------------------------------------- G20: one step from a mid-training state
def test_step_from_mid_training_state_c1_g20():
    """Adam at step 8 and running statistics advanced from non-initial buffers, against the reference started from the same fixture
    state (make_golden.py::g20) -- one-step tolerances."""
    tag, B, N, salt = "c1_ls", 4, 512, 21
    d = golden("g20_mid_state_step_%s.npz" % tag)
    gp_ = params_from(orc.generator_shapes(), salt, requires_grad=True)
    dp_ = params_from(orc.discriminator_shapes(), salt, requires_grad=True)
    gbuf = orc.bn_buffers(orc.generator_shapes()); dbuf = orc.bn_buffers(orc.discriminator_shapes())
    optG, optD = orc.AdamState(gp_), orc.AdamState(dp_)
    for opt, bufs, shapes, batches in ((optD, dbuf, orc.discriminator_shapes(), 21), (optG, gbuf, orc.generator_shapes(), 14)):
        st = fr.mid_training_state(shapes, list(bufs.keys()), salt=salt, batches=batches)
        opt.m, opt.v, opt.step = {k: v.clone() for k, v in st["m"].items()}, {k: v.clone() for k, v in st["v"].items()}, st["step"]
        for k in bufs:
            bufs[k] = st["buffers"][k].clone()
    x = fr.sphere_template(N)[None].repeat(B, 1, 1)
    real, z_d, z_g = fr.synthetic_real(B, N, seed=2001), fr.latent(B, N, seed=2002), fr.latent(B, N, seed=2003)
    graphs = tuple(torch.from_numpy(d[w].astype(np.int64)).view(B, N * 10) for w in ("idx2_d", "idx2_g"))
    out = orc.train_step(gp_, gbuf, dp_, dbuf, optG, optD, x, real, z_d, z_g, gan="ls", use_gp=False, graphs=graphs)
    np.testing.assert_allclose(out["loss_d"].item(), float(d["lossD"]), rtol=2e-5)
    np.testing.assert_allclose(out["loss_g"].item(), float(d["lossG"]), rtol=2e-3)
    check(d, "fake_d", out["fake_d"], rtol=1e-4); check(d, "fake_g", out["fake_g"], rtol=2e-4)
    for n, g in out["d_grads"].items():
        check(d, "dgrad|" + n, g, rtol=3e-2, atol=1e-3 if n.endswith(ZERO_GRAD_BIASES) else 1e-7)
    for n, g in out["g_grads"].items():
        check(d, "ggrad|" + n, g, rtol=5e-2, atol=1e-3 if n.endswith(ZERO_GRAD_BIASES) else 1e-7)
    assert optD.step == 8 and optG.step == 8
    for kind, params, opt, shapes, tol in (("d", dp_, optD, orc.discriminator_shapes(), 3e-2), ("g", gp_, optG, orc.generator_shapes(), 5e-2)):
        for n in params:
            if not n.endswith(ZERO_GRAD_BIASES):
                check(d, "%sm|%s" % (kind, n), opt.m[n], rtol=tol, atol=1e-9); check(d, "%sv|%s" % (kind, n), opt.v[n], rtol=tol, atol=1e-16)
        check_adam_updates(d, kind, params.items(), fr.init_params(shapes, salt=salt), [], None, skip=ZERO_GRAD_BIASES, select_by_gradient=False,
                           atol=2e-6, min_selected=0.5, what="oracle, step 8")
    for kind, bufs, calls in (("d", dbuf, 25), ("g", gbuf, 16)):
        for n, b in bufs.items():
            if n.endswith("num_batches_tracked"):
                assert int(b) == int(d["%sbuf|%s" % (kind, n)]) == calls, n
            else:
                np.testing.assert_allclose(b.numpy(), d["%sbuf|%s" % (kind, n)], rtol=2e-3, atol=2e-4, err_msg=n)


# ---------------------------------------------------------------- G9
def test_ball_group_family():
    d = golden("g9_ball_group.npz")
    B, N, S = 2, 256, 32
    xyz = fr.synthetic_real(B, N, seed=91)
    feat = fr.normal("g9.feat", (B, N, 5))
    new_xyz = xyz[:, ::N // S][:, :S].contiguous()
    np.testing.assert_allclose(orc.square_distance(new_xyz, xyz).numpy(), d["square_distance"], rtol=0, atol=1e-6)
    for r, ns in ((0.3, 16), (0.15, 32), (0.02, 8)):
        assert np.array_equal(orc.query_ball_point(r, ns, xyz, new_xyz).numpy(), d["query_ball|%g|%d" % (r, ns)])
    idx = orc.query_ball_point(0.3, 16, xyz, new_xyz)
    assert np.array_equal(orc.index_points(feat, idx).numpy(), d["index_points3"])
    assert np.array_equal(orc.index_points(feat, idx[:, :, 0]).numpy(), d["index_points2"])
    start = torch.from_numpy(d["fps_start"].astype(np.int64))
    assert np.array_equal(orc.farthest_point_sample(xyz, 24, start).numpy(), d["fps"])
    assert np.array_equal(orc.farthest_point_sample(xyz, 24).numpy(), d["fps0"])
    nx, npts = orc.sample_and_group(24, 0.3, 16, xyz, feat, start)
    assert np.array_equal(nx.numpy(), d["sag|new_xyz"]) and np.array_equal(npts.numpy(), d["sag|new_points"])
    knn = orc.knn_point(10, xyz, xyz)
    assert np.array_equal(torch.sort(knn, dim=-1)[0].numpy(), d["knn_point_sorted"])
    gi = torch.from_numpy(d["group|idx"].astype(np.int64))
    np_, gx = orc.group(10, xyz, feat, idx=gi)
    assert np.array_equal(np_.numpy(), d["group|new_points"]) and np.array_equal(gx.numpy(), d["group|xyz_norm"])


# ---------------------------------------------------------------- G10: eval-mode generation + interpolate
def _eval_setup():
    d = golden("g10_eval_interpolate.npz")
    B, N = 2, 256
    pg = fr.init_params(orc.generator_shapes(), salt=10)
    pd = fr.init_params(orc.discriminator_shapes(), salt=10)
    bg = {k[5:]: torch.from_numpy(np.asarray(v)) for k, v in d.items() if k.startswith("gbuf|")}
    bd = {k[5:]: torch.from_numpy(np.asarray(v)) for k, v in d.items() if k.startswith("dbuf|")}
    x = fr.sphere_template(N)[None].repeat(B, 1, 1)
    z1, z2 = fr.latent(B, N, seed=110), fr.latent(B, N, seed=111)
    sel = torch.from_numpy(d["selection"].astype(np.int64))
    return d, pg, pd, bg, bd, x, z1, z2, sel, float(d["alpha"])


@pytest.mark.parametrize("tag", ["fwd", "interp_z", "interp_style"])
def test_eval_generation_and_interpolate(tag):
    d, pg, pd, bg, bd, x, z1, z2, sel, alpha = _eval_setup()
    B, N = x.shape[:2]
    idx2 = torch.from_numpy(d[tag + "|idx2"].astype(np.int64)).view(B, N * 10)      # the reference's own feature graph (tie-aware protocol)
    st = {}
    with torch.no_grad():
        if tag == "fwd":
            out = orc.generator_forward(pg, x, z1, training=False, buffers=bg, idx2=idx2, stages=st)
        else:
            out = orc.generator_interpolate(pg, x, z1, z2, sel, alpha, use_latent=(tag == "interp_style"), training=False, buffers=bg,
                                            idx2=idx2, stages=st)
        logit = orc.discriminator_forward(pd, out, training=False, buffers=bd)
    check(d, tag + "|x1", st["x1"], rtol=2e-5)
    check(d, tag + "|out", out, rtol=1e-4)
    check(d, tag + "|logit", logit, rtol=1e-3)


# ---------------------------------------------------------------- G11: Chamfer-based evaluation metrics
def _metric_sets():
    S, R, N = 6, 5, 128
    smp = torch.stack([fr.synthetic_real(1, N, seed=300 + i)[0] for i in range(S)])
    ref = torch.stack([fr.synthetic_real(1, N, seed=400 + i)[0] * (0.8 + 0.05 * i) for i in range(R)])
    return smp, ref


def test_chamfer_metrics():
    d = golden("g11_chamfer_metrics.npz")
    smp, ref = _metric_sets()
    dl, dr = orc.dist_chamfer(smp[:5].contiguous(), ref)
    check(d, "dl", dl, rtol=1e-5); check(d, "dr", dr, rtol=1e-5)
    M_rs, M_rr, M_ss = orc.pairwise_cd(ref, smp), orc.pairwise_cd(ref, ref), orc.pairwise_cd(smp, smp)
    np.testing.assert_allclose(M_rs.numpy(), d["M_rs"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(M_rr.numpy(), d["M_rr"], rtol=1e-5, atol=1e-6)
    for k, v in orc.lgan_mmd_cov(M_rs.t()).items():
        np.testing.assert_allclose(float(v), float(d["mmdcov|" + k]), rtol=1e-5)
    for k, v in orc.one_nn_accuracy(M_rr, M_rs, M_ss, 1).items():
        np.testing.assert_allclose(float(v), float(d["1nn|" + k]), rtol=1e-6)
    # the direct-difference form (the CUDA kernel's) agrees with the expanded form up to rounding
    d1, d2, i1, i2 = orc.nn_distance(smp[:5], ref)
    np.testing.assert_allclose(d1.numpy(), dr.numpy(), atol=2e-6); np.testing.assert_allclose(d2.numpy(), dl.numpy(), atol=2e-6)


# ---------------------------------------------------------------- G12: non-default flags
def test_variant_flags():
    d = golden("g12_variants.npz")
    B, N = 4, 256
    x = fr.sphere_template(N)[None].repeat(B, 1, 1)
    z = fr.latent(B, N, seed=120)
    # --use_head: graphs of both EdgeConvs live in feature space -> inject the reference's (tie-aware protocol)
    p = {k: v.clone().requires_grad_(True) for k, v in fr.init_params(orc.generator_shapes(use_head=True), salt=20).items()}
    i1 = torch.from_numpy(d["head|idx1"].astype(np.int64)).view(B, N * 10)
    i2 = torch.from_numpy(d["head|idx2"].astype(np.int64)).view(B, N * 10)
    out = orc.generator_forward(p, x, z, training=True, buffers=orc.bn_buffers(orc.generator_shapes(use_head=True)), idx1=i1, idx2=i2)
    check(d, "head|out", out, rtol=2e-5)
    names = list(p.keys())
    grads = torch.autograd.grad(out, [p[n] for n in names], fr.normal("g12.dy", out.shape))
    for n, g in zip(names, grads):
        check(d, "head|grad|" + n, g, rtol=3e-3, atol=1e-3 if n.endswith(ZERO_GRAD_BIASES) else 1e-6)
    # --off --z_norm
    p = fr.init_params(orc.generator_shapes(), salt=21)
    i2 = torch.from_numpy(d["off|idx2"].astype(np.int64)).view(B, N * 10)
    out = orc.generator_forward(p, x, z, training=True, buffers=orc.bn_buffers(orc.generator_shapes()), idx2=i2, off=True, z_norm=True)
    check(d, "off|out", out, rtol=2e-5)
    # --small_d
    p = {k: v.clone().requires_grad_(True) for k, v in fr.init_params(orc.discriminator_shapes(small_d=True), salt=22).items()}
    real = fr.synthetic_real(4, N, seed=23).transpose(2, 1).contiguous().requires_grad_(True)
    logit = orc.discriminator_forward(p, real, True, orc.bn_buffers(orc.discriminator_shapes(small_d=True)))
    check(d, "small|logit", logit, rtol=2e-5)
    names = list(p.keys())
    grads = torch.autograd.grad(((logit - 1.0) ** 2).mean(), [real] + [p[n] for n in names])
    check(d, "small|dx", grads[0], rtol=2e-3)
    for n, g in zip(names, grads[1:]):
        check(d, "small|grad|" + n, g, rtol=5e-3, atol=1e-3 if n.endswith(ZERO_GRAD_BIASES) else 1e-7)


# ---------------------------------------------------------------- G13: --attn / --eql
@pytest.mark.parametrize("tag,flags,salt", [("attn", dict(attn=True), 30), ("eql", dict(eql=True), 31),
                                            ("both", dict(attn=True, eql=True, use_head=True), 32)])
def test_attn_eql_variants(tag, flags, salt):
    d = golden("g13_attn_eql.npz")
    B, N = 4, 256
    x = fr.sphere_template(N)[None].repeat(B, 1, 1)
    z = fr.latent(B, N, seed=130)
    shapes = orc.generator_shapes(**flags)
    p = {k: v.clone().requires_grad_(True) for k, v in fr.init_params(shapes, salt=salt).items()}
    i1 = torch.from_numpy(d[tag + "|idx1"].astype(np.int64)).view(B, N * 10) if flags.get("use_head") else None
    i2 = torch.from_numpy(d[tag + "|idx2"].astype(np.int64)).view(B, N * 10)
    out = orc.generator_forward(orc.eql_effective_params(p), x, z, training=True, buffers=orc.bn_buffers(shapes), idx1=i1, idx2=i2)
    check(d, tag + "|out", out, rtol=2e-5)
    names = list(p.keys())
    grads = torch.autograd.grad(out, [p[n] for n in names], fr.normal("g13.dy." + tag, out.shape))
    for n, g in zip(names, grads):
        plain = n.replace(".linear.", ".").replace(".conv.", ".")
        # "both": N(0,1) equalised weights in pc_head + BatchNorm over 4 shapes: the gradients are ill-conditioned
        check(d, tag + "|grad|" + n, g, rtol=1e-2 if tag == "both" else 3e-3, atol=1e-3 if plain.endswith(ZERO_GRAD_BIASES) else 1e-6)


# ---------------------------------------------------------------- G14: JSD between occupancy grids
def _jsd_sets():
    S, R, N = 12, 10, 512
    smp = np.stack([fr.synthetic_real(1, N, seed=500 + i)[0].numpy() * 0.5 for i in range(S)])
    ref = np.stack([fr.synthetic_real(1, N, seed=600 + i)[0].numpy() * (0.35 + 0.015 * i) for i in range(R)])
    return smp, ref


def test_jsd_occupancy_grid():
    d = golden("g14_jsd.npz")
    smp, ref = _jsd_sets()
    np.testing.assert_array_equal(orc.unit_cube_grid_point_cloud(6, False)[0], d["grid6_full"])
    for res in (16, 28):
        grid, spacing = orc.unit_cube_grid_point_cloud(res, True)
        np.testing.assert_array_equal(grid, d["grid%d" % res]); assert spacing == float(d["spacing%d" % res])
        ent, cnt = orc.entropy_of_occupancy_grid(smp, res, True)
        np.testing.assert_array_equal(cnt.astype(np.int32), d["cnt%d" % res])
        assert abs(ent - float(d["ent%d" % res])) <= 1e-12
        assert abs(orc.jsd_between_point_cloud_sets(smp, ref, res) - float(d["jsd%d" % res])) <= 1e-12


# ---------------------------------------------------------------- auction EMD (unpinned by reference outputs)
def _emd_sets(B=3, n=128):
    a = np.stack([(fr.synthetic_real(1, n, seed=800 + i)[0].numpy() * 0.5 + 0.5) for i in range(B)]).astype(np.float32)
    b = np.stack([(fr.synthetic_real(1, n, seed=900 + i)[0].numpy() * 0.45 + 0.5) for i in range(B)]).astype(np.float32)
    return a, b


def test_emd_auction_against_optimal_assignment():
    """The auction restatement against the exact optimum (Hungarian method): once every point is assigned the matching is a
    permutation whose cost is within n*eps of the optimal cost (Bertsekas' eps-complementary-slackness bound)."""
    from scipy.optimize import linear_sum_assignment
    a, b = _emd_sets()
    eps, n = 0.002, a.shape[1]
    dist, assign = orc.emd_auction(a, b, eps=eps, iters=3000)
    for i in range(a.shape[0]):
        assert sorted(assign[i].tolist()) == list(range(n))                  # a bijection
        cost = np.linalg.norm(a[i][:, None, :] - b[i][None, :, :], axis=-1).astype(np.float64)
        r, c = linear_sum_assignment(cost)
        opt = cost[r, c].sum()
        got = np.sqrt(dist[i].astype(np.float64)).sum()
        assert opt - 1e-4 <= got <= opt + n * eps + 1e-4, (got, opt)
    # identical clouds: every point keeps itself, distance 0
    dist, assign = orc.emd_auction(a, a, eps=0.005, iters=50)
    assert (assign == np.arange(n)[None]).all() and (dist == 0).all()
    # short runs (the module's default 50 iterations) leave a valid, possibly non-bijective assignment
    dist, assign = orc.emd_auction(a, b, eps=0.005, iters=5)
    assert assign.min() >= 0 and assign.max() < n


# ---------------------------------------------------------------- G17: the benchmarked size (C2: B=32, N=2048)
def test_oracle_at_the_benchmarked_size():
    """The oracle is also the CPU baseline bench.py times at C2: its Discriminator logits and its Generator output (the reference's
    EdgeConv2 graph injected) against the reference's at B=32, N=2048 (golden G17)."""
    B, N = 32, 2048
    d = golden("g17_fullsize_c2.npz")
    torch.set_num_threads(8)
    with torch.no_grad():
        dp = fr.init_params(orc.discriminator_shapes(), salt=17)
        real = fr.synthetic_real(B, N, seed=171).transpose(2, 1).contiguous()
        logit = orc.discriminator_forward(dp, real, True, None)
        np.testing.assert_allclose(logit.numpy(), d["d|logit"], rtol=2e-4, atol=2e-6)
        gp_ = fr.init_params(orc.generator_shapes(), salt=17)
        x = fr.sphere_template(N)[None].repeat(B, 1, 1)
        z = fr.latent(B, N, seed=173)
        idx2 = torch.from_numpy(d["g|idx2"].astype(np.int64)).view(B, N * 10)
        out = orc.generator_forward(gp_, x, z, training=True, buffers=None, idx2=idx2)
    check(d, "g|out", out, rtol=2e-5)


def test_step_noise_tables_g18():
    """Golden G18 (the reference's own float32-vs-float64 and tie-flip movements of the benchmarked step) is present, covers every
    gradient tensor of the step golden and yields finite, monotone bounds -- the GPU step tests derive their tolerances from it."""
    from helpers import StepNoise, golden
    sn, step = StepNoise(), golden("g17_step_c2.npz")
    names_d = [k[len("dgrad|"):].split("|")[0] for k in step.files if k.startswith("dgrad|")]
    names_g = [k[len("ggrad|"):].split("|")[0] for k in step.files if k.startswith("ggrad|")]
    assert names_d and names_g
    for kind, names in (("dgrad", names_d), ("ggrad", names_g)):
        for n in set(names):
            b0, b20, b100 = sn.tensor_bound(kind, n, 0, 2.0), sn.tensor_bound(kind, n, 20, 2.0), sn.tensor_bound(kind, n, 100, 2.0)
            assert np.isfinite([b0, b20, b100]).all() and b0 > 0 and b20 >= b0 * 0.999, (kind, n, b0, b20, b100)
    w = [sn.whole_g_bound(n, 1.0) for n in (0, 1, 5, 20, 100)]
    assert w[0] < 0.05 and w[0] < w[2] < w[4] < 0.5, w          # float32 noise 1.4e-2; 20 flipped ties move the whole G gradient by 0.18


# ---------------------------------------------------------------- G21: the auction against a sequential emulation of emd_cuda.cu
def emd_trace_statement(assign_fn, what):
    """Shared by the CPU (oracle) and the GPU (HIP kernel) test.  assign_fn(a, b, eps, T) -> (dist [B,n], assignment [B,n]) numpy.
    Fixture: tests/golden/make_emd_trace.py -- the reference's CUDA kernels (metrics/emd/emd_cuda.cu:93-236) emulated sequentially in
    the file's own arithmetic, its one data race (GetMax: the last writer among bidders within 1e-6 of the maximum keeps the object)
    resolved by a fixed thread order, ascending and descending, with and without nvcc's fma contraction.  The statement, round by round:
      * rounds 1-3: the assignment equals the emulation's in EVERY variant, row for row;
      * rounds <= 10: it equals the variant in which the lowest near-tied bidder wins (= the build's rule: exact maximum, lowest index);
      * later: the rows that differ from the nearest variant are fewer than 1.5 x the rows the emulation's own two thread orders differ
        in (0.5 % at 20 rounds, 8 % at 50, 41 % at 100, 71 % at 1000: the reference's assignment IS its race), and the matching cost lies
        within 0.5 % of the variants' own range at every round count; the exact optimum bounds it as in test_emd_auction_against_optimal_assignment."""
    sys_path_golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    if sys_path_golden not in sys.path:
        sys.path.insert(0, sys_path_golden)
    import make_emd_trace as mt
    d = golden("g21_emd_trace.npz")
    a, b = mt.trace_inputs()
    eps = float(d["eps"])
    report = []
    for T in [int(t) for t in d["iters"]]:
        dist, asg = assign_fn(a, b, eps, T)
        cost = np.sqrt(np.asarray(dist, dtype=np.float64)).sum(1)
        diff = {v: float((np.asarray(asg) != d["assign|%s|%s|%d" % (v + (T,))]).mean()) for v in mt.VARIANTS}
        own = float((d["assign|ascending|fma|%d" % T] != d["assign|descending|fma|%d" % T]).mean())
        report.append((T, min(diff.values()), own))
        if T <= 3:
            assert max(diff.values()) == 0.0, (what, T, diff)
        if T <= 10:
            assert diff[("descending", "fma")] == 0.0 and diff[("descending", "none")] == 0.0, (what, T, diff)
        assert min(diff.values()) <= max(1.5 * own, 0.003), (what, T, diff, own)
        refs = np.stack([d["cost|%s|%s|%d" % (v + (T,))] for v in mt.VARIANTS])            # the variants themselves are up to 0.8 % apart (300 rounds)
        assert np.all(cost >= refs.min(0) * (1 - 5e-3)) and np.all(cost <= refs.max(0) * (1 + 5e-3)), (what, T, cost, refs)
    return report


def test_emd_trace_fixture_is_what_the_script_produces():
    import os as _os, sys as _sys
    _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden"))
    import make_emd_trace as mt
    d = golden("g21_emd_trace.npz")
    a, b = mt.trace_inputs()
    for order, contract, T in (("ascending", "fma", 5), ("descending", "none", 20)):
        for i in range(a.shape[0]):
            _, asg, _ = mt.emulate_emd_cuda(a[i], b[i], float(d["eps"]), T, order, contract)
            assert np.array_equal(asg.astype(np.int16), d["assign|%s|%s|%d" % (order, contract, T)][i])


def test_emd_trace_oracle_against_cuda_emulation():
    rep = emd_trace_statement(lambda a, b, eps, T: orc.emd_auction(a, b, eps, T), "oracle")
    assert rep[0][1] == 0.0
