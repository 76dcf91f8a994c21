"""GPU parity: the HIP-backed modules against (a) the golden vectors captured from the real
reference and (b) the CPU oracle on the same seeded inputs.  Tolerances are set to <= ~3x the errors MEASURED
on MI355X (profiles/r02_parity.json lists every comparison: rel-L2 and max-abs): block tests with injected upstream land at 5e-7 ..
1.4e-6 rel-L2 (fp32 rounding class; SURVEY H1b planned ~1e-5), the generated cloud at 8e-6, kNN is judged tie-aware, and only the
end-to-end gradients of a whole train step are kink-limited (1e-3 .. 8e-3 measured: a LeakyReLU / arg-max flip moves them discretely)."""
import numpy as np
import pytest
import torch

from helpers import check, check_adam_updates, golden, measured_grad_errors, rel_l2
from oracle import spgan_oracle as orc
from spgan import fixture_rng as fr

pytestmark = pytest.mark.gpu

ZERO_GRAD_BIASES = ("conv_w.0.bias", "conv_w.3.bias", "conv_x.0.bias", "global_conv.0.bias", "global_conv.3.bias",
                    "mlps.0.bias", "mlps.3.bias", "mlps.6.bias", "fc2.0.bias")


class Opts:
    np = 256; nk = 20; nz = 128; softmax = True; off = False; attn = False
    use_head = False; eql = False; z_norm = False; small_d = False


@pytest.fixture(scope="module")
def sp():
    import spgan
    from spgan import _lib, modules, ops
    _lib.load()
    return spgan


def _load(module, params):
    sd = module.state_dict()
    module.load_state_dict({**sd, **{k: v.detach().clone() for k, v in params.items()}})
    return module.cuda()


def _atol(n):
    return 2e-3 if n.endswith(ZERO_GRAD_BIASES) else 1e-7


# ---------------------------------------------------------------- get_edge_features (G1)
@pytest.mark.parametrize("N", [256, 512])
def test_get_edge_features_sphere_bit_exact(sp, N):
    d = golden("g1_edge_features.npz")
    x = fr.sphere_template(N)[None].transpose(2, 1).contiguous().cuda()
    ee, idx = sp.get_edge_features(x, 10, return_idx=True)
    assert np.array_equal(idx.view(N, 10).cpu().numpy(), d["sphere%d|idx" % N])
    assert np.array_equal(ee.cpu().numpy(), d["sphere%d|ee" % N])               # pure gather/subtract: bit equal
    ee2 = sp.get_edge_features(x, 10, idx=idx)
    assert torch.equal(ee, ee2)


# ---------------------------------------------------------------- EdgeBlock (G2)
@pytest.mark.parametrize("tag,fin,fout", [("ec1", 3, 64), ("ec2", 64, 128)])
def test_edgeblock_golden(sp, tag, fin, fout):
    d = golden("g2_edgeblock.npz")
    B, N = 2, 256
    pref = "EdgeConv1." if fin == 3 else "EdgeConv2."
    shapes = {k: v for k, v in orc.generator_shapes().items() if k.startswith(pref)}
    params = {k[len(pref):]: v for k, v in fr.init_params(shapes, salt=2).items()}
    blk = _load(sp.EdgeBlock(fin, fout, 10), params).train()
    if fin == 3:
        x = fr.sphere_template(N)[None].repeat(B, 1, 1).transpose(2, 1).contiguous()
        x = x + 0.01 * fr.normal("g2.jit", x.shape)
    else:
        x = fr.normal("g2.x.%s" % tag, (B, fin, N), 0.7)
    x = x.cuda().requires_grad_(True)
    idx = torch.from_numpy(d[tag + "|idx"].astype(np.int64)).view(B, N * 10).cuda()
    y = blk(x, idx=idx)                                                          # reference graph injected (tie-aware protocol)
    check(d, tag + "|y", y, rtol=3e-6)                                          # measured 5.9e-7 / 7.3e-7
    dy = fr.normal("g2.dy.%s" % tag, y.shape).cuda()
    (y * dy).sum().backward()
    check(d, tag + "|dx", x.grad, rtol=3e-6)                                    # measured 6.8e-7
    for n, p in blk.named_parameters():
        check(d, tag + "|grad|" + n, p.grad, rtol=5e-6, atol=_atol(n))             # measured <= 1.3e-6
    for n, b in [(k, v) for k, v in blk.state_dict().items() if k in dict(blk.named_buffers())]:
        np.testing.assert_allclose(b.cpu().numpy(), d[tag + "|buf|" + n], rtol=1e-5, atol=1e-6)
    # own graph: the block's kNN agrees with the reference's except at near-ties
    y2 = blk(x.detach())
    own = sp.ops.idx_to_local64(blk.last_idx, B, N)
    agree = (own.view(B * N, 10) == idx.view(B * N, 10)).all(dim=1).float().mean().item()
    assert agree >= 0.995, agree


# ---------------------------------------------------------------- AdaptivePointNorm (G3)
def test_adain_golden(sp):
    d = golden("g3_adain.npz")
    B, C, N = 2, 64, 256
    m = _load(sp.AdaptivePointNorm(C, 128), fr.init_params({"style.weight": (2 * C, 128, 1), "style.bias": (2 * C,)}, salt=3))
    x = fr.normal("g3.x", (B, C, N)).cuda().requires_grad_(True)
    s = fr.normal("g3.s", (B, 128, N), 0.3).cuda().requires_grad_(True)
    y = m(x, s)
    (y * fr.normal("g3.dy", y.shape).cuda()).sum().backward()
    for n, t in (("y", y), ("dx", x.grad), ("dstyle", s.grad), ("dw", m.style.weight.grad), ("db", m.style.bias.grad)):
        check(d, n, t, rtol=1.5e-6)                                              # measured <= 4.0e-7


# ---------------------------------------------------------------- Discriminator (G5, G7)
@pytest.mark.parametrize("hint", [0, 2])
def test_discriminator_golden(sp, hint):
    """hint 2: every eligible product (the 128->256 and the fused 256->1024 layer, their input gradients) through the 256 x 256-tile
    kernel of csrc/gemm_wide.hip -- at this size the automatic rule would not pick it; same golden, same tolerances."""
    d = golden("g5_discriminator.npz")
    B, N = 4, 256
    D = _load(sp.Discriminator(Opts), fr.init_params(orc.discriminator_shapes(), salt=4)).train()
    real = fr.synthetic_real(B, N, seed=5).transpose(2, 1).contiguous().cuda().requires_grad_(True)
    with sp.ops.nt_tile_hint(hint):
        logit = D(real)
        check(d, "logit", logit, rtol=2e-6)                                     # measured 4.4e-7
        ((logit - 1.0) ** 2).mean().backward()
    check(d, "dx", real.grad, rtol=3e-6)                                         # measured 7.6e-7
    for n, p in D.named_parameters():
        check(d, "grad|" + n, p.grad, rtol=5e-6, atol=_atol(n))                  # measured <= 1.4e-6
    for n, b in [(k, v) for k, v in D.state_dict().items() if k in dict(D.named_buffers())]:
        np.testing.assert_allclose(b.cpu().numpy(), d["buf|" + n], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("hint", [0, 2])
def test_gradient_penalty_golden(sp, hint):
    """hint 2: the double backward with the 256 x 256-tile kernel wherever a product is eligible (M = 768 = three row tiles)."""
    d = golden("g7_gradient_penalty.npz")
    B, N = 3, 256
    D = _load(sp.Discriminator(Opts), fr.init_params(orc.discriminator_shapes(), salt=7)).train()
    real = fr.synthetic_real(B, N, seed=71).transpose(2, 1).contiguous().cuda()
    fake = (0.8 * fr.synthetic_real(B, N, seed=72) + 0.05 * fr.normal("g7.n", (B, N, 3))).transpose(2, 1).contiguous().cuda()
    alpha = torch.from_numpy(d["alpha"]).cuda()
    with sp.ops.nt_tile_hint(hint):
        gp = sp.GradientPenalty(10.0, gamma=1)(D, real, fake, alpha=alpha)
        np.testing.assert_allclose(gp.item(), float(d["gp"]), rtol=1e-5)
        gp.backward()
    for n, p in D.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        check(d, "grad|" + n, g, rtol=5e-6, atol=2e-3 if n.endswith(ZERO_GRAD_BIASES) else 1e-7)    # double backward: measured <= 1.1e-6
    xh = (real + alpha * (fake - real)).requires_grad_(True)
    D2 = _load(sp.Discriminator(Opts), fr.init_params(orc.discriminator_shapes(), salt=7)).train()
    gin, = torch.autograd.grad(D2(xh).sum(), xh)
    check(d, "input_grad", gin, rtol=3e-6)                                      # measured 6.9e-7


# ---------------------------------------------------------------- Generator (G4)
def test_generator_golden(sp):
    d = golden("g4_generator.npz")
    B, N = 4, 256
    G = _load(sp.Generator(Opts), fr.init_params(orc.generator_shapes(), salt=4)).train()
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
    z = fr.latent(B, N, seed=44).cuda()
    out = G(x, z)
    i1 = sp.ops.idx_to_local64(G.EdgeConv1.last_idx, B, N).view(B, N, 10).cpu().numpy()
    i2 = sp.ops.idx_to_local64(G.EdgeConv2.last_idx, B, N).view(B, N, 10).cpu().numpy()
    assert np.array_equal(i1, d["idx1"]), "sphere graph must be bit-exact (SURVEY H1a)"
    rows2 = (i2 == d["idx2"]).all(axis=2).mean()
    assert rows2 >= 0.995, "EdgeConv2 kNN row agreement %.4f" % rows2
    # every disagreeing EdgeConv2 row must be a near-tie in the reference's own distances (tie-aware protocol):
    x1 = torch.from_numpy(d["stage|x1|full"])
    srt = torch.sort(orc.pairwise_sqdist(x1), dim=2)[0][:, :, :12]
    gaps = srt.diff(dim=2).abs().min(dim=2)[0].numpy()
    bad = ~(i2 == d["idx2"]).all(axis=2)
    assert (gaps[bad] < 1e-4).all(), "a non-tie EdgeConv2 row disagrees with the reference"
    # A single flipped row moves ~all outputs by >1e-3 through the batch-of-4 BatchNorm1d + global max coupling
    # (measured on the reference against itself), so the raw output is only comparable when the
    # graphs coincide; otherwise test_generator_vs_oracle_with_injected_graph carries the comparison.
    if rows2 == 1.0:
        check(d, "out", out, rtol=3e-5)                                          # measured 7.6e-6
        for n, b in [(k, v) for k, v in G.state_dict().items() if k in dict(G.named_buffers())]:
            np.testing.assert_allclose(b.cpu().numpy(), d["buf|" + n], rtol=2e-3, atol=1e-4)
    # the stage feeding EdgeConv2's graph is tie-independent and must be tight (<= 1e-5 class)
    check(d, "stage|x1", sp.ops.pm_to_cm(G.last_x1, B, N), rtol=3e-6)           # measured 9.7e-7


def test_generator_vs_oracle_with_injected_graph(sp):
    """Oracle run on the CPU with OUR kNN graphs injected: isolates everything but tie-breaking."""
    B, N = 4, 256
    p = fr.init_params(orc.generator_shapes(), salt=31)
    G = _load(sp.Generator(Opts), p).train()
    x = fr.synthetic_real(B, N, seed=32); z = fr.latent(B, N, seed=33)
    out = G(x.cuda(), z.cuda())
    idx1 = sp.ops.idx_to_local64(G.EdgeConv1.last_idx, B, N).cpu()
    idx2 = sp.ops.idx_to_local64(G.EdgeConv2.last_idx, B, N).cpu()
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    buf = orc.bn_buffers(orc.generator_shapes())
    ref = orc.generator_forward(po, x, z, training=True, buffers=buf, idx1=idx1, idx2=idx2)
    assert rel_l2(out.detach().cpu().numpy(), ref.detach().numpy()) <= 2e-5         # measured 4.8e-6
    dy = fr.normal("pg.dy", out.shape)
    (out * dy.cuda()).sum().backward()
    names = list(po.keys())
    grads = torch.autograd.grad((ref * dy).sum(), [po[n] for n in names])
    gsd = dict(G.named_parameters())
    for n, g in zip(names, grads):
        e = rel_l2(gsd[n].grad.cpu().numpy(), g.numpy())
        mx = (gsd[n].grad.cpu() - g).abs().max().item()
        assert e <= 1.5e-3 or mx <= _atol(n), "%s: rel-L2 %.3e max-abs %.3e" % (n, e, mx)      # kink-limited: measured <= 3.9e-4
    flat_a = torch.cat([gsd[n].grad.cpu().reshape(-1) for n in names if not n.endswith(ZERO_GRAD_BIASES)])
    flat_b = torch.cat([g.reshape(-1) for n, g in zip(names, grads) if not n.endswith(ZERO_GRAD_BIASES)])
    cos = torch.dot(flat_a, flat_b) / (flat_a.norm() * flat_b.norm())
    assert cos.item() >= 0.99999, cos.item()


# ---------------------------------------------------------------- one full train step (G8)
@pytest.mark.parametrize("tag,gan,use_gp,B,N", [("ls", "ls", False, 4, 512), ("wgangp", "wgan", True, 4, 256)])
def test_train_step_golden(sp, tag, gan, use_gp, B, N):
    d = golden("g8_train_step_%s.npz" % tag)
    o = Opts()
    G = _load(sp.Generator(o), fr.init_params(orc.generator_shapes(), salt=8))
    D = _load(sp.Discriminator(o), fr.init_params(orc.discriminator_shapes(), salt=8))
    tr = sp.TrainStep(G, D, gan=gan, use_gp=use_gp, lambda_gp=10.0, lr_g=1e-4, lr_d=1e-4)
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
    real = fr.synthetic_real(B, N, seed=81).cuda()
    z_d, z_g = fr.latent(B, N, seed=82).cuda(), fr.latent(B, N, seed=83).cuda()
    alpha = torch.from_numpy(d["alpha"]).cuda()
    info = tr.step(x, real, z_d, z_g, alpha=alpha, keep_grads=True)
    np.testing.assert_allclose(info["loss_d"].item(), float(d["lossD"]), rtol=3e-3)
    np.testing.assert_allclose(info["loss_g"].item(), float(d["lossG"]), rtol=5e-3)
    check(d, "fake_d", info["fake_d"], rtol=6e-5)                               # measured 1.8e-5 / 5.9e-6
    for n, g in info["d_grads"].items():
        check(d, "dgrad|" + n, g, rtol=4e-3, atol=_atol(n))                     # kink-limited end-to-end gradients: measured <= 1.2e-3
    for n, g in info["g_grads"].items():
        check(d, "ggrad|" + n, g, rtol=2.5e-2, atol=_atol(n))   # after D's Adam step and through D's kinks: measured <= 7.8e-3
    # post-Adam parameters, compared as UPDATES p - p0 on the elements whose golden gradient is above the noise floor (a tolerance of the
    # size of lr on p itself would hold for a wrong-sign or an absent update: helpers.check_adam_updates)
    for kind, net, shapes in (("d", D, orc.discriminator_shapes()), ("g", G, orc.generator_shapes())):
        check_adam_updates(d, kind, net.named_parameters(), fr.init_params(shapes, salt=8), [kind + "grad|"],
                           measured_grad_errors(d, kind + "grad|", info[kind + "_grads"], skip=ZERO_GRAD_BIASES), skip=ZERO_GRAD_BIASES, what=tag)
    for n, b in [(k, v) for k, v in D.state_dict().items() if k in dict(D.named_buffers())]:
        np.testing.assert_allclose(b.cpu().numpy(), d["dbuf|" + n], rtol=2e-3, atol=2e-4)


# ---------------------------------------------------------------- eval-mode generation + interpolate (G10)
@pytest.mark.parametrize("tag", ["fwd", "interp_z", "interp_style"])
def test_eval_generation_and_interpolate_golden(sp, tag):
    """model_test.py:63-64: G.eval() + G(x, z) / G.interpolate(...): BatchNorm on running statistics, no autograd."""
    d = golden("g10_eval_interpolate.npz")
    B, N = 2, 256
    G = _load(sp.Generator(Opts), fr.init_params(orc.generator_shapes(), salt=10))
    D = _load(sp.Discriminator(Opts), fr.init_params(orc.discriminator_shapes(), salt=10))
    G.load_state_dict({k[5:]: torch.from_numpy(np.asarray(v)) for k, v in d.items() if k.startswith("gbuf|")}, strict=False)
    D.load_state_dict({k[5:]: torch.from_numpy(np.asarray(v)) for k, v in d.items() if k.startswith("dbuf|")}, strict=False)
    G.eval(); D.eval()
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
    z1, z2 = fr.latent(B, N, seed=110).cuda(), fr.latent(B, N, seed=111).cuda()
    sel = torch.from_numpy(d["selection"].astype(np.int64)).cuda()
    alpha = float(d["alpha"])
    with torch.no_grad():
        if tag == "fwd":
            out = G(x, z1.clone())
        else:
            out = G.interpolate(x, z1.clone(), z2.clone(), sel, alpha, use_latent=(tag == "interp_style"))
        logit = D(out)
    check(d, tag + "|x1", sp.ops.pm_to_cm(G.last_x1, B, N), rtol=3e-6)          # tie-independent stage: measured 8.0e-7
    i2 = sp.ops.idx_to_local64(G.EdgeConv2.last_idx, B, N).view(B, N, 10).cpu().numpy()
    same = (i2 == d[tag + "|idx2"]).all(axis=2)
    assert same.mean() >= 0.995, "EdgeConv2 kNN row agreement %.4f" % same.mean()
    if same.all():
        check(d, tag + "|out", out, rtol=2e-4)
        check(d, tag + "|logit", logit, rtol=2e-3)
    else:
        # a flipped near-tie row: compare with the oracle run on OUR graph instead (tie-aware protocol)
        pg = fr.init_params(orc.generator_shapes(), salt=10)
        bg = {k[5:]: torch.from_numpy(np.asarray(v)) for k, v in d.items() if k.startswith("gbuf|")}
        idx2 = sp.ops.idx_to_local64(G.EdgeConv2.last_idx, B, N).cpu()
        xc, z1c, z2c = x.cpu(), z1.cpu(), z2.cpu()
        with torch.no_grad():
            if tag == "fwd":
                ref = orc.generator_forward(pg, xc, z1c, training=False, buffers=bg, idx2=idx2)
            else:
                ref = orc.generator_interpolate(pg, xc, z1c, z2c, sel.cpu(), alpha, use_latent=(tag == "interp_style"), training=False,
                                                buffers=bg, idx2=idx2)
        assert rel_l2(out.cpu().numpy(), ref.numpy()) <= 2e-4


def test_shared_sphere_edgeconv1_equals_per_shape_evaluation(sp, monkeypatch):
    """The tiled sphere prior lets EdgeConv1 run on ONE copy (Generator._body): outputs, parameter gradients and BatchNorm
    buffers must equal the per-shape evaluation of all B copies (up to summation order)."""
    B, N = 4, 256
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
    z = fr.latent(B, N, seed=77).cuda()
    dy = fr.normal("shared.dy", (B, 3, N)).cuda()
    res = []
    for shared in (True, False):
        G = _load(sp.Generator(Opts), fr.init_params(orc.generator_shapes(), salt=12)).train()
        if not shared:
            monkeypatch.setattr(torch, "equal", lambda a, b: False)
        out = G(x.clone(), z)
        monkeypatch.undo()
        assert G._sphere_graph["shared"] == shared
        (out * dy).sum().backward()
        res.append((out.detach(), {n: p.grad.detach().clone() for n, p in G.named_parameters()},
                    {n: b.detach().clone() for n, b in G.state_dict().items() if n in dict(G.named_buffers())},
                    G.EdgeConv1.last_idx.clone()))
    assert torch.equal(res[0][3], res[1][3])
    assert rel_l2(res[0][0].cpu().numpy(), res[1][0].cpu().numpy()) <= 1e-5
    for n in res[0][1]:
        a, b = res[0][1][n].cpu(), res[1][1][n].cpu()
        # BatchNorm weight gradients are sums of g*xhat with heavy cancellation: the two evaluations sum them (and the statistics
        # behind xhat) in different orders -- measured up to 2.2e-4
        assert rel_l2(a.numpy(), b.numpy()) <= 4e-4 or (a - b).abs().max().item() <= _atol(n), n
    for n in res[0][2]:
        np.testing.assert_allclose(res[0][2][n].cpu().numpy(), res[1][2][n].cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=n)


@pytest.mark.parametrize("shifted", [False, True])
def test_discriminator_advance_running_stats_equals_forward(sp, shifted):
    """D.advance_running_stats(x) leaves exactly the buffers a train-mode D(x) leaves (fc2.1 statistics via the covariance of
    its input instead of the 1024-wide GEMM).  shifted: BatchNorm 3 with a large offset and a small scale, i.e. activations whose
    mean^2 is ~10^4 times their variance -- the regime in which a Gram/M - mu mu^T covariance would lose its digits; the operand
    is centred on load instead, and the running variance must stay accurate (and non-negative)."""
    B, N = 4, 512
    x = fr.synthetic_real(B, N, seed=55).transpose(2, 1).contiguous().cuda()
    params = fr.init_params(orc.discriminator_shapes(), salt=14)
    if shifted:
        params["mlps.7.bias"] = params["mlps.7.bias"] + 4.0
        params["mlps.7.weight"] = params["mlps.7.weight"] * 0.04
    bufs = []
    for fast in (False, True):
        D = _load(sp.Discriminator(Opts), params).train()
        with torch.no_grad():
            D(x * 0.9)                                   # move the running statistics away from (0, 1) first
            if fast:
                D.advance_running_stats(x)
            else:
                D(x)
        bufs.append({n: b.detach().clone().cpu() for n, b in D.state_dict().items() if n in dict(D.named_buffers())})
    for n in bufs[0]:
        rel_l2(bufs[1][n].float().numpy(), bufs[0][n].float().numpy(), "advance_running_stats%s|%s" % ("|shifted" if shifted else "", n))
        np.testing.assert_allclose(bufs[1][n].numpy(), bufs[0][n].numpy(), rtol=1e-4 if shifted else 2e-5, atol=1e-6, err_msg=n)
    assert int(bufs[1]["fc2.1.num_batches_tracked"]) == 2
    assert float(bufs[1]["fc2.1.running_var"].min()) >= 0.0


# ---------------------------------------------------------------- non-default flags (G12)
def _tie_aware_out(sp, G, d, tag, out, x, z, B, N, shapes_kw, salt, **fkw):
    """Compare with the golden output when both feature-space graphs coincide with the reference's, otherwise with the oracle
    run on OUR graphs (tie-aware protocol)."""
    i1 = sp.ops.idx_to_local64(G.EdgeConv1.last_idx, B, N).view(B, N, 10).cpu()
    i2 = sp.ops.idx_to_local64(G.EdgeConv2.last_idx, B, N).view(B, N, 10).cpu()
    same = np.array_equal(i2.numpy(), d[tag + "|idx2"]) and ((tag + "|idx1") not in d or np.array_equal(i1.numpy(), d[tag + "|idx1"]))
    if same:
        check(d, tag + "|out", out, rtol=6e-5)                                     # measured 2.4e-6 .. 1.9e-5
    agree = (i2.numpy() == d[tag + "|idx2"]).all(axis=2).mean()
    assert agree >= 0.99, agree
    p = fr.init_params(orc.generator_shapes(**shapes_kw), salt=salt)
    ref = orc.generator_forward(orc.eql_effective_params(p), x.cpu(), z.cpu(), training=True,
                                buffers=orc.bn_buffers(orc.generator_shapes(**shapes_kw)), idx1=i1.view(B, -1), idx2=i2.view(B, -1), **fkw)
    assert rel_l2(out.detach().cpu().numpy(), ref.detach().numpy()) <= 6e-5     # measured <= 1.6e-5
    return same


def test_generator_use_head_golden(sp):
    d = golden("g12_variants.npz")
    B, N = 4, 256

    class OH(Opts):
        use_head = True
    G = _load(sp.Generator(OH), fr.init_params(orc.generator_shapes(use_head=True), salt=20)).train()
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
    z = fr.latent(B, N, seed=120).cuda()
    out = G(x, z)
    same = _tie_aware_out(sp, G, d, "head", out, x, z, B, N, dict(use_head=True), 20)
    dy = fr.normal("g12.dy", out.shape).cuda()
    (out * dy).sum().backward()
    if same:
        for n, p in G.named_parameters():
            check(d, "head|grad|" + n, p.grad, rtol=1.5e-2, atol=_atol(n))            # whole-G gradients (kink-limited): measured <= 4.9e-3


def test_generator_off_znorm_golden(sp):
    d = golden("g12_variants.npz")
    B, N = 4, 256

    class OO(Opts):
        off = True; z_norm = True
    G = _load(sp.Generator(OO), fr.init_params(orc.generator_shapes(), salt=21)).train()
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
    z = fr.latent(B, N, seed=120).cuda()
    out = G(x, z)
    _tie_aware_out(sp, G, d, "off", out, x, z, B, N, {}, 21, off=True, z_norm=True)


def test_discriminator_small_d_golden(sp):
    d = golden("g12_variants.npz")
    N = 256

    class OS(Opts):
        small_d = True
    D = _load(sp.Discriminator(OS), fr.init_params(orc.discriminator_shapes(small_d=True), salt=22)).train()
    real = fr.synthetic_real(4, N, seed=23).transpose(2, 1).contiguous().cuda().requires_grad_(True)
    logit = D(real)
    check(d, "small|logit", logit, rtol=3e-6)
    ((logit - 1.0) ** 2).mean().backward()
    check(d, "small|dx", real.grad, rtol=5e-6)
    for n, p in D.named_parameters():
        check(d, "small|grad|" + n, p.grad, rtol=5e-6, atol=_atol(n))                # measured <= 7.3e-7


@pytest.mark.parametrize("B,N,small", [(3, 300, False), (2, 200, True)])
def test_discriminator_ragged_n_vs_oracle(sp, B, N, small):
    """N % 128 != 0: the pooling partials cannot be used (tiles would straddle shapes), the output of fc2.0 is stored and
    pooled by the stand-alone kernel -- forward, WGAN loss + gradient penalty gradients against the oracle's autograd."""
    class O(Opts):
        small_d = small
    shapes = orc.discriminator_shapes(small_d=small)
    p = fr.init_params(shapes, salt=15)
    D = _load(sp.Discriminator(O), p).train()
    real = fr.synthetic_real(B, N, seed=151).transpose(2, 1).contiguous()
    fake = (0.8 * fr.synthetic_real(B, N, seed=152) + 0.05 * fr.normal("rag.n", (B, N, 3))).transpose(2, 1).contiguous()
    alpha = fr.uniform("rag.alpha", (B, 1, 1), 0.0, 1.0)
    loss = D(fake.cuda()).mean() - D(real.cuda()).mean() + sp.GradientPenalty(10.0, gamma=1)(D, real.cuda(), fake.cuda(), alpha=alpha.cuda())
    loss.backward()
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    f = lambda t: orc.discriminator_forward(po, t, True, None)
    ref = f(fake).mean() - f(real).mean() + orc.gradient_penalty(f, real, fake, alpha, 10.0, 1.0)
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=2e-4)
    names = list(po.keys())
    gref = torch.autograd.grad(ref, [po[n] for n in names], allow_unused=True)
    own = dict(D.named_parameters())
    for n, g in zip(names, gref):
        g = torch.zeros_like(po[n]) if g is None else g
        mine = own[n].grad if own[n].grad is not None else torch.zeros_like(own[n])
        e = rel_l2(mine.cpu().numpy(), g.numpy())
        assert e <= 6e-6 or (mine.cpu() - g).abs().max().item() <= (2e-3 if n.endswith(ZERO_GRAD_BIASES) else 1e-7), "%s %.3e" % (n, e)    # measured <= 1.5e-6


# ---------------------------------------------------------------- --attn / --eql (G13)
@pytest.mark.parametrize("tag,flags,salt", [("attn", dict(attn=True), 30), ("eql", dict(eql=True), 31),
                                            ("both", dict(attn=True, eql=True, use_head=True), 32)])
def test_generator_attn_eql_golden(sp, tag, flags, salt):
    d = golden("g13_attn_eql.npz")
    B, N = 4, 256
    O = type("O_" + tag, (Opts,), flags)
    shapes = orc.generator_shapes(**flags)
    params = fr.init_params(shapes, salt=salt)
    G = sp.Generator(O)
    assert list(G.state_dict().keys()) and {k: tuple(v.shape) for k, v in G.named_parameters()} == {k: tuple(v) for k, v in shapes.items()}
    _load(G, params).train()
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
    z = fr.latent(B, N, seed=130).cuda()
    out = G(x, z)
    same = _tie_aware_out(sp, G, d, tag, out, x, z, B, N, flags, salt)
    dy = fr.normal("g13.dy." + tag, out.shape)
    (out * dy.cuda()).sum().backward()
    # gradients against the oracle on OUR graphs (always), and against the reference's when the graphs coincide
    i1 = sp.ops.idx_to_local64(G.EdgeConv1.last_idx, B, N).cpu(); i2 = sp.ops.idx_to_local64(G.EdgeConv2.last_idx, B, N).cpu()
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = orc.generator_forward(orc.eql_effective_params(po), x.cpu(), z.cpu(), training=True, buffers=orc.bn_buffers(shapes), idx1=i1, idx2=i2)
    grads = dict(zip(po.keys(), torch.autograd.grad((ref * dy).sum(), list(po.values()))))
    rtol = {"attn": 1e-4, "eql": 1.5e-3, "both": 1e-2}[tag]     # whole-G gradients, kink-limited: measured 2.4e-5 / 4.3e-4 / 3.0e-3
    for n, p in G.named_parameters():
        g = grads[n]
        plain = n.replace(".linear.", ".").replace(".conv.", ".")
        e = rel_l2(p.grad.cpu().numpy(), g.numpy())
        assert e <= rtol or (p.grad.cpu() - g).abs().max().item() <= _atol(plain), (n, e)
        if same:
            check(d, tag + "|grad|" + n, p.grad, rtol=rtol, atol=_atol(plain))


def test_attention_module_fullsize(sp):
    """Attention(640) at the C2 shape count per shape (N=2048): against the oracle for two shapes, forward and backward."""
    B, N, ch = 2, 2048, 640
    A = sp.modules.Attention(ch).cuda()
    names = ["theta.weight", "phi.weight", "g.weight", "o.weight", "gamma"]
    params = fr.init_params({"attn." + n: tuple(dict(A.named_parameters())[n].shape) for n in names}, salt=33)
    A.load_state_dict({n: params["attn." + n] for n in names})
    x = fr.normal("attn.x", (B, ch, N), 0.5)
    xg = x.cuda().requires_grad_(True)
    y = A(xg)
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    xo = x.clone().requires_grad_(True)
    ref = orc.attention(po, "attn", xo)
    assert rel_l2(y.detach().cpu().numpy(), ref.detach().numpy()) <= 1e-5
    dy = fr.normal("attn.dy", y.shape)
    (y * dy.cuda()).sum().backward()
    grads = torch.autograd.grad((ref * dy).sum(), [xo] + list(po.values()))
    assert rel_l2(xg.grad.cpu().numpy(), grads[0].numpy()) <= 1e-4
    for n, g in zip(names, grads[1:]):
        assert rel_l2(dict(A.named_parameters())[n].grad.cpu().numpy(), g.numpy()) <= 1e-4, n


# ---------------------------------------------------------------- the second GradientPenalty (Common/loss_utils.py:1087-1131)
@pytest.mark.parametrize("mapping", [False, True])
def test_gradient_penalty_loss_utils_variant(sp, mapping):
    B, N = 3, 256
    params = fr.init_params(orc.discriminator_shapes(), salt=7)
    D = _load(sp.Discriminator(Opts), params).train()
    real = (fr.synthetic_real(B, N, seed=71) * 0.5 + 0.5).transpose(2, 1).contiguous()
    fake = (0.4 * fr.synthetic_real(B, N, seed=72) + 0.5 + 0.02 * fr.normal("g7.n", (B, N, 3))).transpose(2, 1).contiguous()
    alpha = fr.uniform("gpv.alpha", (B, 1, 1), 0.0, 1.0)
    gp = sp.GradientPenalty(10.0, gamma=1, mix="loss_utils")(D, real.cuda(), fake.cuda(), alpha=alpha.cuda(), mapping=mapping)
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    bufs = orc.bn_buffers(orc.discriminator_shapes())
    ref = orc.gradient_penalty_loss_utils(lambda x: orc.discriminator_forward(po, x, True, bufs), real, fake, alpha, 10.0, 1.0, mapping=mapping)
    np.testing.assert_allclose(gp.item(), ref.item(), rtol=2e-4)
    gp.backward()
    grads = torch.autograd.grad(ref, list(po.values()), allow_unused=True)
    for (n, p), g in zip(D.named_parameters(), grads):
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        want = g if g is not None else torch.zeros(p.shape)
        e = rel_l2(got.cpu().numpy(), want.numpy())
        assert e <= 5e-3 or (got.cpu() - want).abs().max().item() <= (2e-3 if n.endswith(ZERO_GRAD_BIASES) else 2e-6), (n, e)


def test_gradient_penalty_through_eval_mode_discriminator(sp):
    """Common/gradient_penalty.py:28-33 on a D.eval(): double backward through BatchNorm with running statistics (a fixed affine) --
    value, parameter gradients (zero for biases and BatchNorm shifts) against torch autograd through the oracle; the reference's own
    formula (plain torch around our Discriminator) and spgan.GradientPenalty agree."""
    B, N = 3, 256
    params = fr.init_params(orc.discriminator_shapes(), salt=7)
    bufs = orc.bn_buffers(orc.discriminator_shapes())
    for k in bufs:
        if k.endswith("running_mean"):
            bufs[k] = fr.normal("gpe.m." + k, bufs[k].shape, 0.05)
        elif k.endswith("running_var"):
            bufs[k] = fr.uniform("gpe.v." + k, bufs[k].shape, 0.5, 1.5)
    D = _load(sp.Discriminator(Opts), {**params, **bufs}).eval()
    real = fr.synthetic_real(B, N, seed=71).transpose(2, 1).contiguous()
    fake = (0.8 * fr.synthetic_real(B, N, seed=72) + 0.05 * fr.normal("g7.n", (B, N, 3))).transpose(2, 1).contiguous()
    alpha = fr.uniform("gpe.alpha", (B, 1, 1), 0.0, 1.0)
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = orc.gradient_penalty(lambda x: orc.discriminator_forward(po, x, False, bufs), real, fake, alpha, 10.0, 1.0)
    grads = torch.autograd.grad(ref, list(po.values()), allow_unused=True)
    for impl in ("spgan", "caller"):
        D.zero_grad(set_to_none=True)
        if impl == "spgan":
            gp = sp.GradientPenalty(10.0, gamma=1)(D, real.cuda(), fake.cuda(), alpha=alpha.cuda())
        else:
            gp = orc.gradient_penalty(D, real.cuda(), fake.cuda(), alpha.cuda(), 10.0, 1.0)
        np.testing.assert_allclose(gp.item(), ref.item(), rtol=2e-4)
        gp.backward()
        for (n, p), g in zip(D.named_parameters(), grads):
            got = p.grad if p.grad is not None else torch.zeros_like(p)
            want = g if g is not None else torch.zeros(p.shape)
            e = rel_l2(got.cpu().numpy(), want.numpy(), "gp eval %s|%s" % (impl, n))
            assert e <= 2e-3 or (got.cpu() - want).abs().max().item() <= 2e-6, (impl, n, e)


# ---------------------------------------------------------------- one latent per shape, passed un-tiled
def test_generator_per_shape_latent_matches_tiled_and_oracle(sp):
    """z handed over un-tiled [B,1,nz] (HeadFn: latent half of head.0 once per shape, coordinate half as a K = 3 product) against
    the tiled [B,N,nz] input (one K = 131 MFMA product).  The two evaluations round differently in the last bits, so each run may
    build its own EdgeConv2 graph at near-ties (tie-aware protocol): every run is checked against the oracle on ITS OWN
    graphs, the runs against each other only when their graphs coincide."""
    B, N = 4, 256
    params = fr.init_params(orc.generator_shapes(), salt=4)
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
    zt = fr.latent(B, N, seed=44)                                            # tiled, as the reference builds it
    outs, grads, graphs, x1s = [], [], [], []
    dy = fr.normal("psl.dy", (B, 3, N)).cuda()
    for z in (zt[:, :1, :].contiguous().cuda(), zt.cuda()):
        G = _load(sp.Generator(Opts), params).train()
        out = G(x, z)
        (out * dy).sum().backward()
        outs.append(out.detach()); grads.append({n: p.grad.clone() for n, p in G.named_parameters()})
        graphs.append((sp.ops.idx_to_local64(G.EdgeConv1.last_idx, B, N).cpu(), sp.ops.idx_to_local64(G.EdgeConv2.last_idx, B, N).cpu()))
        x1s.append(G.last_x1.clone())
    assert rel_l2(x1s[0].cpu().numpy(), x1s[1].cpu().numpy(), "per-shape vs tiled latent|x1") <= 3e-6   # the stage in front of the graph: tie-independent
    assert torch.equal(graphs[0][0], graphs[1][0])
    # This configuration is ill-conditioned (train-mode BatchNorm of the global feature over 4 shapes): the reference arithmetic
    # in fp32 is itself ~3e-2 away from its fp64 evaluation on the same graphs, and a 1e-7 relative change of z moves the fp32
    # gradients by 3e-4.  So the yardstick is the fp64 oracle, and the HIP path has to be no further from it than a small multiple
    # of the fp32 oracle's own distance.
    for r in range(2):
        og = {}
        for dt in (torch.float32, torch.float64):
            po = {k: v.clone().to(dt).requires_grad_(True) for k, v in params.items()}
            bufs = {k: v.to(dt) for k, v in orc.bn_buffers(orc.generator_shapes()).items()}
            ref = orc.generator_forward(po, x.cpu().to(dt), zt.to(dt), training=True, buffers=bufs, idx1=graphs[r][0], idx2=graphs[r][1])
            og[dt] = dict(zip(po.keys(), torch.autograd.grad((ref * dy.cpu().to(dt)).sum(), list(po.values()))))
            if dt == torch.float32:
                assert rel_l2(outs[r].cpu().numpy(), ref.detach().numpy(), "latent run %d|out vs oracle on its own graphs" % r) <= 2e-5
        for n in grads[r]:
            want = og[torch.float64][n].numpy()
            own = rel_l2(og[torch.float32][n].numpy(), want)                 # the reference arithmetic's own fp32 error
            e = rel_l2(grads[r][n].cpu().double().numpy(), want, "latent run %d|grad|%s vs fp64 oracle on its own graphs" % (r, n))
            assert e <= 3.0 * own + 1e-4 or (grads[r][n].cpu().double() - og[torch.float64][n]).abs().max().item() <= _atol(n), (r, n, e, own)
    same = torch.equal(graphs[0][1], graphs[1][1])
    agree = (graphs[0][1].view(B * N, -1) == graphs[1][1].view(B * N, -1)).all(dim=1).float().mean().item()
    assert agree >= 0.995, agree
    if same:
        assert rel_l2(outs[0].cpu().numpy(), outs[1].cpu().numpy()) <= 2e-5
    for n in grads[0]:
        # the two evaluations round differently; with this conditioning (see above) that alone moves the gradients by percents
        e = rel_l2(grads[0][n].cpu().numpy(), grads[1][n].cpu().numpy())
        assert e <= 1e-1 or (grads[0][n] - grads[1][n]).abs().max().item() <= _atol(n), (n, e)


def test_per_shape_latent_with_eql_and_znorm(sp):
    """The un-tiled latent path under --eql (scaled, non-leaf head weights) and --z_norm: same output and gradients as the tiled input."""
    B, N = 4, 256
    flags = dict(eql=True, z_norm=True)
    O = type("O_pe", (Opts,), flags)
    params = fr.init_params(orc.generator_shapes(eql=True), salt=61)
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
    zt = fr.latent(B, N, seed=62)
    dy = fr.normal("pe.dy", (B, 3, N)).cuda()
    res = []
    for z in (zt[:, :1, :].contiguous().cuda(), zt.cuda()):
        G = _load(sp.Generator(O), params).train()
        out = G(x, z)
        (out * dy).sum().backward()
        res.append((out.detach(), {n: p.grad.clone() for n, p in G.named_parameters()}))
    assert rel_l2(res[0][0].cpu().numpy(), res[1][0].cpu().numpy()) <= 2e-5
    for n in ("head.0.conv.weight_orig", "head.0.conv.bias", "head.2.conv.weight_orig", "head.2.conv.bias"):
        e = rel_l2(res[0][1][n].cpu().numpy(), res[1][1][n].cpu().numpy())
        assert e <= 6e-2, (n, e)


def test_discriminator_forward_many_equals_separate_calls_gpu(sp):
    """D.forward_many(real, fake) (what TrainStep uses for the D step): the BatchNorm-free head of both passes as one batch; logits,
    input gradients, parameter gradients and buffers equal those of two separate calls."""
    B, N = 4, 512
    p = fr.init_params(orc.discriminator_shapes(), salt=41)
    xa = fr.synthetic_real(B, N, seed=42).transpose(2, 1).contiguous().cuda()
    xb = (0.7 * fr.synthetic_real(B, N, seed=43)).transpose(2, 1).contiguous().cuda()
    w = fr.normal("fm.w", (2 * B, 1)).cuda()
    res = []
    for many in (False, True):
        D = _load(sp.Discriminator(Opts), p).train()
        a, b = xa.clone().requires_grad_(True), xb.clone().requires_grad_(True)
        la, lb = D.forward_many(a, b) if many else (D(a), D(b))
        (torch.cat([la, lb]) * w).sum().backward()
        res.append(([la.detach(), lb.detach(), a.grad, b.grad], {n: q.grad.clone() for n, q in D.named_parameters()},
                    {k: v.clone() for k, v in D.state_dict().items() if k not in dict(D.named_parameters())}))
    for i in range(4):
        assert rel_l2(res[1][0][i].cpu().numpy(), res[0][0][i].cpu().numpy(), "forward_many|tensor%d" % i) <= 1e-6
    for n in res[0][1]:
        a_, b_ = res[1][1][n].cpu(), res[0][1][n].cpu()
        assert rel_l2(a_.numpy(), b_.numpy(), "forward_many|grad|" + n) <= 3e-6 or (a_ - b_).abs().max().item() <= 1e-7, n
    for k in res[0][2]:
        assert torch.equal(res[1][2][k], res[0][2][k]), k


def test_discriminator_grouped_passes_equal_separate_calls(sp):
    """Discriminator.forward_stacks_grouped (the D step's D(real), D(fake), D(x_hat) conv stacks as one batch) and
    forward_stack_after_stats_pass (the G step's statistics pass + D(G(z))): logits, every gradient and all BatchNorm buffers are those
    of the separate calls, bit for bit."""
    B, N = 4, 256
    params = fr.init_params(orc.discriminator_shapes(), salt=11)
    xs = [(fr.synthetic_real(B, N, seed=90 + i) * (1.0 + 0.2 * i)).transpose(2, 1).contiguous().cuda() for i in range(3)]
    seeds = [fr.normal("grp.seed%d" % i, (B, 1)).cuda() for i in range(3)]

    def run(grouped):
        D = _load(sp.Discriminator(Opts), params).train()
        ins = [x.clone().requires_grad_(True) for x in xs]
        if grouped:
            pre = D.forward_stacks_grouped(ins)
            logits = D.forward_heads([D.forward_stack(ins[0], pre=pre[0]), D.forward_stack(ins[1], pre=pre[1])]) + [D(ins[2], pre=pre[2])]
        else:
            logits = D.forward_heads([D.forward_stack(ins[0]), D.forward_stack(ins[1])]) + [D(ins[2])]
        torch.autograd.backward(logits, seeds)
        return ([l.detach().clone() for l in logits], [i.grad.clone() for i in ins], {n: p.grad.clone() for n, p in D.named_parameters()},
                {k: v.clone() for k, v in D.state_dict().items() if k in dict(D.named_buffers())})

    a, b = run(False), run(True)
    for la, lb in zip(a[0], b[0]):
        assert torch.equal(la, lb)
    for ga, gb in zip(a[1], b[1]):
        assert torch.equal(ga, gb)
    for n in a[2]:
        assert torch.equal(a[2][n], b[2][n]), n
    for k in a[3]:
        assert torch.equal(a[3][k], b[3][k]), k

    def run_g(grouped):
        D = _load(sp.Discriminator(Opts), params).train()
        x = xs[1].clone().requires_grad_(True)
        if grouped:
            logit = D(x, pre=D.forward_stack_after_stats_pass(xs[0], x))
        else:
            D.advance_running_stats(xs[0])
            logit = D(x)
        sp.requires_grad(D, False)
        (logit * seeds[0]).sum().backward()
        return logit.detach().clone(), x.grad.clone(), {k: v.clone() for k, v in D.state_dict().items() if k in dict(D.named_buffers())}

    a, b = run_g(False), run_g(True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), k


def test_twin_generator_forwards_reuse_edgeconv1(sp):
    """TrainStep evaluates EdgeConv1 once for the generator's two forwards of a step (same sphere prior, same weights) and advances
    its BatchNorm running statistics twice in one update.  Against the step that evaluates it twice: every output, gradient and
    parameter bit for bit; the three running-statistics pairs to rounding (one combined update vs two), the call counters exactly."""
    B, N = 4, 256
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
    real = fr.synthetic_real(B, N, seed=81).cuda()
    z_d, z_g = fr.latent(B, N, seed=82)[:, :1, :].contiguous().cuda(), fr.latent(B, N, seed=83)[:, :1, :].contiguous().cuda()
    alpha = fr.uniform("twin.alpha", (B, 1, 1), 0.0, 1.0).cuda()
    res = []
    for twin in (False, True):
        G = _load(sp.Generator(Opts), fr.init_params(orc.generator_shapes(), salt=8))
        D = _load(sp.Discriminator(Opts), fr.init_params(orc.discriminator_shapes(), salt=8))
        tr = sp.TrainStep(G, D, gan="wgan", use_gp=True, lambda_gp=10.0)
        tr.twin_g_forwards = twin
        infos = [tr.step(x, real, z_d, z_g, alpha=alpha, keep_grads=True) for _ in range(2)]
        res.append((infos, {k: v.clone() for k, v in G.state_dict().items()}, {k: v.clone() for k, v in D.state_dict().items()}))
    (ia, ga, da), (ib, gb, db) = res
    for s in range(2):
        assert torch.equal(ia[s]["fake_d"], ib[s]["fake_d"]) and torch.equal(ia[s]["fake_g"], ib[s]["fake_g"])
        assert torch.equal(ia[s]["loss_d"], ib[s]["loss_d"]) and torch.equal(ia[s]["loss_g"], ib[s]["loss_g"])
        for n in ia[s]["g_grads"]:
            assert torch.equal(ia[s]["g_grads"][n], ib[s]["g_grads"][n]), n
        for n in ia[s]["d_grads"]:
            assert torch.equal(ia[s]["d_grads"][n], ib[s]["d_grads"][n]), n
    for k in da:
        assert torch.equal(da[k], db[k]), k
    for k in ga:
        if k.startswith("EdgeConv1.") and ("running_mean" in k or "running_var" in k):
            assert torch.allclose(ga[k], gb[k], rtol=2e-6, atol=1e-8), k
        else:
            assert torch.equal(ga[k], gb[k]), k                   # parameters, the other buffers and every num_batches_tracked


def test_twin_second_without_a_reusable_first_does_not_advance_statistics_again(sp):
    """The advisor's twin-forward case: the forward announced as "second" cannot reuse its twin (EdgeConv1's weights changed in
    between) -- it must evaluate EdgeConv1 WITHOUT advancing the running statistics a third time ("first" already accounted for
    two updates); a forward outside the protocol (twin_forward None: an eval call / sample dump) in between leaves a pending twin alone."""
    B, N = 4, 256
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
    z = fr.latent(B, N, seed=82)[:, :1, :].contiguous().cuda()
    G = _load(sp.Generator(Opts), fr.init_params(orc.generator_shapes(), salt=8)).train()
    stats = lambda: {k: v.clone() for k, v in G.state_dict().items() if k.startswith("EdgeConv1.") and ("running" in k or "tracked" in k)}
    with torch.no_grad():
        G.twin_forward = "first"
        G(x, z)
        G.twin_forward = None
        after_first = stats()
        assert G.__dict__.get("_ec1_twin") is not None
        G(x, z)                                                   # a call outside the protocol: advances the statistics once, keeps the twin
        assert G.__dict__.get("_ec1_twin") is not None
        between = stats()
        assert any(not torch.equal(after_first[k], between[k]) for k in between if "running" in k)
        G.EdgeConv1.conv_x[0].weight.mul_(1.0)                    # in-place touch: the version stamp of the twin no longer matches
        G.twin_forward = "second"
        out = G(x, z)
        G.twin_forward = None
        G.flush_bn_counts()
        for k, v in stats().items():
            if "running" in k:
                assert torch.equal(v, between[k]), k              # not advanced a third time
    assert torch.isfinite(out).all() and G.__dict__.get("_ec1_twin") is None
