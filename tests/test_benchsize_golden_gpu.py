"""GPU: numeric parity with the REFERENCE at the sizes bench.py times -- BASELINE configs[1] (C2: B=32, N=2048, WGAN-GP) and
configs[3]'s per-GPU shape (C4: B=16, N=4096) -- against golden G17 (tests/golden/make_golden.py::g17, the real reference run in
the build container; large tensors as l2 / strided samples, logits and BatchNorm buffers in full, the reference's own EdgeConv2
graphs as int16).  Before round 3 these sizes were covered by finite / deterministic properties only (test_fullsize_gpu.py).

  * Discriminator forward / backward and the gradient penalty's double backward (Discriminator.py:97-115,
    gradient_penalty.py:19-37): no discrete choice besides the arg-max of the pool -> block tolerances;
  * Generator forward / backward (Generator.py:160-198) with the reference's EdgeConv2 graph injected (tie-aware protocol), and
    with its own graph (row agreement with the reference's);
  * the whole benchmarked WGAN-GP train step (model.py:239-279): losses, logits, generated clouds, every gradient, post-Adam
    parameters, BatchNorm buffers;
  * the dominant kernel itself -- the fused 256 -> 1024 GEMM + BatchNorm + LeakyReLU + max-pool at M = 65536 and the three-pass
    grouped launch at M = 3 x 65536 (1024 / 3072 workgroups of the 256 x 256-tile kernel, the XCD-aware tile map) -- against a
    float64 model.
Tolerances are <= ~3x the errors measured on MI355X (profiles/r03_parity.json)."""
import numpy as np
import pytest
import torch

from helpers import (StepNoise, check, check_adam_updates, check_bounded_by_reference_noise as check64, check_step_gradients_bounded,
                     check_whole_gradient_bounded, golden, measured_grad_errors, rel_l2)
from oracle import spgan_oracle as orc
from spgan import fixture_rng as fr

pytestmark = pytest.mark.gpu

ZERO_GRAD_BIASES = ("conv_w.0.bias", "conv_w.3.bias", "conv_x.0.bias", "global_conv.0.bias", "global_conv.3.bias",
                    "mlps.0.bias", "mlps.3.bias", "mlps.6.bias", "fc2.0.bias")
CFGS = {"c2": (32, 2048), "c4": (16, 4096)}


def _opts(N):
    class O:
        np = N; nk = 20; nz = 128; softmax = True; off = False; attn = False
        use_head = False; eql = False; z_norm = False; small_d = False
    return O


@pytest.fixture(scope="module")
def sp():
    import spgan
    from spgan import _lib
    _lib.load()
    return spgan


def _load(module, params):
    sd = module.state_dict()
    module.load_state_dict({**sd, **{k: v.detach().clone() for k, v in params.items()}})
    return module.cuda()


def _atol(n):
    return 2e-3 if n.endswith(ZERO_GRAD_BIASES) else 1e-7


def _buffers(m):
    names = dict(m.named_buffers())
    return [(k, v) for k, v in m.state_dict().items() if k in names]


# ---------------------------------------------------------------- Discriminator at the benchmarked sizes
@pytest.mark.parametrize("tag", ["c2", "c4"])
def test_discriminator_benchsize_golden(sp, tag):
    B, N = CFGS[tag]
    d = golden("g17_fullsize_%s.npz" % tag)
    D = _load(sp.Discriminator(_opts(N), num_point=N), fr.init_params(orc.discriminator_shapes(), salt=17)).train()
    real = fr.synthetic_real(B, N, seed=171).transpose(2, 1).contiguous().cuda().requires_grad_(True)
    logit = D(real)
    assert rel_l2(logit.detach().cpu().numpy(), d["d|logit"], "d|logit") <= 3e-6
    ((logit - 1.0) ** 2).mean().backward()
    # 1024 x B arg-max choices over N points sit between the logits and the gradients below the pool: a near-tie that float32 and
    # float64 resolve differently moves them discretely (at C4 the reference's own float32 gradients are 8e-4 off its float64 ones
    # below the pool, 5e-7 above it) -> bounded by the reference's own float32 error against float64, floor = block tolerance
    check64(d, "d|dx", "d|dx64", real.grad, floor=8e-6)
    for n, p in D.named_parameters():
        check64(d, "d|grad|" + n, "d|grad64|" + n, p.grad, floor=8e-6, atol=_atol(n))
    for n, b in _buffers(D):
        np.testing.assert_allclose(b.cpu().numpy(), d["d|buf|" + n], rtol=1e-5, atol=1e-6, err_msg=n)


@pytest.mark.parametrize("tag", ["c2", "c4"])
def test_gradient_penalty_benchsize_golden(sp, tag):
    B, N = CFGS[tag]
    d = golden("g17_fullsize_%s.npz" % tag)
    D = _load(sp.Discriminator(_opts(N), num_point=N), fr.init_params(orc.discriminator_shapes(), salt=17)).train()
    real = fr.synthetic_real(B, N, seed=171).transpose(2, 1).contiguous().cuda()
    fake = (0.8 * fr.synthetic_real(B, N, seed=172) + 0.05 * fr.normal("g17.n.%s" % tag, (B, N, 3))).transpose(2, 1).contiguous().cuda()
    alpha = torch.from_numpy(d["gp|alpha"]).cuda()
    gp = sp.GradientPenalty(10.0, gamma=1)(D, real, fake, alpha=alpha)
    np.testing.assert_allclose(gp.item(), float(d["gp|value"]), rtol=1e-5)
    gp.backward()
    for n, p in D.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        check64(d, "gp|grad|" + n, "gp|grad64|" + n, g, floor=8e-6, atol=2e-3 if n.endswith(ZERO_GRAD_BIASES) else 1e-7)


# ---------------------------------------------------------------- Generator at the benchmarked sizes
@pytest.mark.parametrize("tag", ["c2", "c4"])
def test_generator_benchsize_golden(sp, tag):
    B, N = CFGS[tag]
    d = golden("g17_fullsize_%s.npz" % tag)
    G = _load(sp.Generator(_opts(N)), fr.init_params(orc.generator_shapes(), salt=17)).train()
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
    z = fr.latent(B, N, seed=173).cuda()
    ref_idx2 = torch.from_numpy(d["g|idx2"].astype(np.int64)).view(B, N * 10)
    # (1) own graph: the stage in front of EdgeConv2's graph is tie-independent; the graph agrees with the reference's row for row
    # except at near-ties (every differing row checked in exact arithmetic), and the generated cloud deviates by no more than the
    # REFERENCE's own output moves when that many near-tied rows are resolved the other way (golden g|tie_*)
    with torch.no_grad():
        out_own = G(x, z)
    check(d, "g|x1", sp.ops.pm_to_cm(G.last_x1, B, N), rtol=3e-6)
    n_diff = _tie_aware_graph_check(sp, G, ref_idx2.view(B * N, 10), B, N, min_rows=0.995)
    _check_within_tie_sensitivity(d, "g|out", out_own, n_diff)
    # (2) the reference's graph injected: everything behind the discrete choice, forward and backward
    G = _load(sp.Generator(_opts(N)), fr.init_params(orc.generator_shapes(), salt=17)).train()
    G.inject_graph2([ref_idx2])
    out = G(x, z)
    check(d, "g|x1", sp.ops.pm_to_cm(G.last_x1, B, N), rtol=3e-6)
    check(d, "g|x2", sp.ops.pm_to_cm(G.last_x2, B, N), rtol=1e-5)
    check(d, "g|out", out, rtol=3e-5)
    dy = fr.normal("g17.dy.%s" % tag, out.shape).cuda()
    (out * dy).sum().backward()
    # global_conv's BatchNorm1d normalises over the B shapes of the batch, whose global features are nearly equal: the reference's
    # own float32 gradients are only good to ~2e-3 (measured against its float64 pass on the same graph, golden g|grad64|*)
    for n, p in G.named_parameters():
        check64(d, "g|grad|" + n, "g|grad64|" + n, p.grad, floor=3e-5, atol=_atol(n))
    for n, b in _buffers(G):
        np.testing.assert_allclose(b.cpu().numpy(), d["g|buf|" + n], rtol=2e-4, atol=2e-5, err_msg=n)


# ---------------------------------------------------------------- the benchmarked train step
@pytest.mark.parametrize("inject", [True, False])
def test_train_step_benchsize_golden(sp, inject):
    """C2: one WGAN-GP D-step + G-step at B=32, N=2048 -- the configuration bench.py reports -- against the reference's step.
    inject=True: the reference's two EdgeConv2 graphs (D step, G step) are handed to the generator, so that every float is
    compared tightly; inject=False: the build's own graphs (what bench.py runs), end-to-end tolerances of the small-size G8 test."""
    B, N = 32, 2048
    d = golden("g17_step_c2.npz")
    o = _opts(N)
    G = _load(sp.Generator(o), fr.init_params(orc.generator_shapes(), salt=18))
    D = _load(sp.Discriminator(o, num_point=N), fr.init_params(orc.discriminator_shapes(), salt=18))
    tr = sp.TrainStep(G, D, gan="wgan", use_gp=True, lambda_gp=10.0, lr_g=1e-4, lr_d=1e-4)
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
    real = fr.synthetic_real(B, N, seed=181).cuda()
    z_d, z_g = fr.latent(B, N, seed=182).cuda(), fr.latent(B, N, seed=183).cuda()
    alpha = torch.from_numpy(d["alpha"]).cuda()
    n_diff = 0
    noise = StepNoise()      # golden G18: the reference's own float32-vs-float64 and tie-flip movements of exactly this step
    if inject:
        G.inject_graph2([torch.from_numpy(d["idx2_d"].astype(np.int64)).view(B, N * 10), torch.from_numpy(d["idx2_g"].astype(np.int64)).view(B, N * 10)])
    info = tr.step(x, real, z_d, z_g, alpha=alpha, keep_grads=True)
    tight = inject
    np.testing.assert_allclose(info["loss_d"].item(), float(d["lossD"]), rtol=2e-4 if tight else 3e-3)
    # lossG = -mean(D(G(z))) of logits of magnitude 0.1 that cancel to 0.009: absolute bound in units of the logits
    np.testing.assert_allclose(info["loss_g"].item(), float(d["lossG"]), rtol=0, atol=(2e-4 if tight else 2e-2) * float(np.abs(d["d_gfake"]).max()))
    # own graphs: the few near-tie rows that differ from the reference's graph (checked below) reach every output through
    # global_conv's BatchNorm1d over 32 nearly equal global features (ill-conditioned: the reference's own float32 gradients are
    # only good to 2e-3 there, golden G17 g|grad64) -- measured 8.4e-3 on the generated cloud
    if tight:
        check(d, "fake_d", info["fake_d"], rtol=3e-5)
        check(d, "fake_g", info["fake_g"], rtol=2e-4)
    else:
        n_diff = _tie_aware_graph_check(sp, G, torch.from_numpy(d["idx2_g"].astype(np.int64)).view(B * N, 10), B, N)
        # the D step's graph (first generator forward) is not kept by the module; its stage tensor is the same function of the same
        # weights on another latent -- the flip tables are indexed by the larger of the two counts, taken >= the G step's
        n_diff = max(n_diff, 1)
        table = golden("g17_fullsize_c2.npz")
        _check_within_tie_sensitivity(d, "fake_d", info["fake_d"], max(n_diff, 1), table=table)
        _check_within_tie_sensitivity(d, "fake_g", info["fake_g"], max(n_diff, 1), factor=6.0, table=table)      # also behind D's and nothing else's update: G's weights are the same
    # Gradient bounds DERIVED from the reference (golden G18), not asserted.  Same graphs: every tensor within 2.5 x the reference's own
    # float32-vs-float64 movement of that tensor (two float32 evaluations differ by up to 2 x one's distance from the exact value; D's
    # gradients 2e-4 .. 3.5e-3, G's -- behind D's Adam step and its kinks -- 8e-3 .. 1.9e-2).  Own graphs (n_diff near-tie rows differ):
    # every D tensor and the whole G gradient within 2 x what the reference moves when it resolves >= n_diff ties the other way.
    if tight:
        check_step_gradients_bounded(d, noise, "dgrad", info["d_grads"], 0, 2.5, skip=ZERO_GRAD_BIASES)
        check_step_gradients_bounded(d, noise, "ggrad", info["g_grads"], 0, 2.5, skip=ZERO_GRAD_BIASES)
        check_whole_gradient_bounded(d, noise, "ggrad|", info["g_grads"], 0, 2.5, skip=ZERO_GRAD_BIASES)
    else:
        check_step_gradients_bounded(d, noise, "dgrad", info["d_grads"], n_diff, 2.0, skip=ZERO_GRAD_BIASES)
        check_whole_gradient_bounded(d, noise, "ggrad|", info["g_grads"], n_diff, 2.0, skip=ZERO_GRAD_BIASES)
    # post-Adam parameters as UPDATES p - p0 on the elements whose golden gradient is above the noise floor (helpers.check_adam_updates)
    for kind, net, shapes in (("d", D, orc.discriminator_shapes()), ("g", G, orc.generator_shapes())):
        check_adam_updates(d, kind, net.named_parameters(), fr.init_params(shapes, salt=18), [kind + "grad|"],
                           measured_grad_errors(d, kind + "grad|", info[kind + "_grads"], skip=ZERO_GRAD_BIASES), skip=ZERO_GRAD_BIASES,
                           what="injected graphs" if inject else "own graphs",
                           min_selected=0.2 if (inject or kind == "d") else 0.0)   # own graphs: G's gradient tensors move by 1e-1 under ~15 tie flips -- little stands 20 x above that
    for n, b in _buffers(D):          # own graphs: D's running statistics saw a generated cloud that differs by ~1e-2 (above)
        np.testing.assert_allclose(b.cpu().numpy(), d["dbuf|" + n], rtol=2e-3 if tight else 2e-2, atol=2e-4 if tight else 2e-3, err_msg=n)
    for n, b in _buffers(G):
        np.testing.assert_allclose(b.cpu().numpy(), d["gbuf|" + n], rtol=2e-3 if tight else 2e-2, atol=2e-4 if tight else 2e-3, err_msg=n)


def _check_within_tie_sensitivity(d, name, t, n_diff, factor=3.0, table=None):
    """rel-L2 of `t` against golden `name` <= factor x what the reference's own output moves when >= n_diff near-tied kNN rows flip."""
    from helpers import _entry
    table = d if table is None else table
    ns, rels = table["g|tie_nflip"], table["g|tie_out_rel"]
    pick = [r for n, r in zip(ns, rels) if n >= n_diff]
    bound = factor * (pick[0] if pick else rels[-1] * n_diff / ns[-1])
    a = t.detach().cpu().numpy()
    ref, got = _entry(d, name, a)
    err = rel_l2(got, ref, name + " (own graph, %d tie rows differ; reference tie sensitivity bound %.2e)" % (n_diff, bound))
    assert err <= max(bound, 3e-5), "%s: rel-L2 %.3e with %d differing near-tie rows; the reference itself moves %.3e per such flip set" % (name, err, n_diff, bound / factor)
    return err


def _tie_aware_graph_check(sp, G, ref, B, N, min_rows=0.99, tol=2e-5):
    """SURVEY 8(c) tie-aware protocol at full size: the build's EdgeConv2 graph agrees with the reference's row for row except at
    near-ties -- for every differing row BOTH neighbour lists must carry the k smallest distances (rank by rank, within `tol` of the
    row's distance scale) in exact float64 arithmetic on the build's own stage tensor (which matches the reference's to 1e-6)."""
    own = sp.ops.idx_to_local64(G.EdgeConv2.last_idx, B, N).view(B * N, 10).cpu()
    same = (own == ref).all(dim=1)
    assert same.float().mean().item() >= min_rows, "EdgeConv2 kNN row agreement %.5f" % same.float().mean().item()
    x1 = G.last_x1.double().cpu().view(B, N, -1)
    bad = (~same).nonzero().flatten().tolist()
    for r in bad[:512]:
        b, i = divmod(r, N)
        dist = ((x1[b] - x1[b, i]) ** 2).sum(1)
        srt = torch.sort(dist)[0][1:11]
        for lst in (own[r], ref[r]):
            got = dist[lst]
            assert (got - srt).abs().max().item() <= tol * max(srt[-1].item(), 1e-12) + 1e-9, \
                "row %d: a neighbour list that is not the k nearest (gap %.3e of scale %.3e)" % (r, (got - srt).abs().max().item(), srt[-1].item())
    return len(bad)


# ---------------------------------------------------------------- the dominant kernel at the launch geometry the bench times
def _f64_layer(A, W, b, gamma, beta, psc, psh, rows, slope=0.01):
    """float64 model of fc2.0 + BatchNorm1d(train) + LeakyReLU + max over the points of a shape, on the BN+LeakyReLU'd operand."""
    a = A.double() * psc.double() + psh.double()
    a = torch.where(a > 0, a, a * slope)
    y = a @ W.double().t() + b.double()
    mean, var = y.mean(0), y.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    z = (y - mean) * invstd * gamma.double() + beta.double()
    z = torch.where(z > 0, z, z * slope)
    Bn = y.shape[0] // rows
    pooled, arg = z.view(Bn, rows, -1).max(dim=1)
    return y, mean, var, invstd, pooled, arg + (torch.arange(Bn, device=y.device) * rows).view(Bn, 1)


@pytest.mark.parametrize("groups", [1, 3])
def test_dominant_kernel_at_bench_geometry_vs_float64(sp, groups):
    """M = 65536 (one pass: 1024 workgroups) and 3 x 65536 (the D step's three passes as one launch: 3072 workgroups), N = 1024,
    K = 256, automatic tile selection => gemm_nt_wide_kernel with the XCD-aware tile map: batch statistics, pooled values, arg-max
    rows and running statistics against a float64 evaluation of the same layer."""
    ops = sp.ops
    rows, Bn, K, C = 2048, 32, 256, 1024
    Mg = rows * Bn
    M = groups * Mg
    A = fr.normal("dom.A.%d" % groups, (M, K)).cuda()
    A = A + torch.arange(groups, device="cuda").repeat_interleave(Mg).view(M, 1) * 0.25
    W = fr.uniform("dom.W", (C, K), -1.0 / 16, 1.0 / 16).cuda()
    b = fr.uniform("dom.b", (C,), -1.0 / 16, 1.0 / 16).cuda()
    gamma, beta = fr.uniform("dom.g", (C,), 0.5, 1.5).cuda(), fr.uniform("dom.be", (C,), -0.2, 0.2).cuda()
    gamma[::7] *= -1.0                                                        # both branches of the pool (max of pre-BN / min of pre-BN)
    sc, sh = fr.uniform("dom.sc", (groups, K), 0.5, 1.5).cuda(), fr.uniform("dom.sh", (groups, K), -0.3, 0.3).cuda()
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    if groups == 1:
        _, st, pooled, arg, yarg = ops.gemm_bn_pool(A, W, b, (gamma, beta, rm, rv), rows, 0.01, pro=(sc[0].contiguous(), sh[0].contiguous(), 0.01))
        st = torch.stack(list(st)).view(4, 1, C)
    else:
        st, pooled, arg, yarg = ops.gemm_bn_groups(A, W, b, (gamma, beta, rm, rv), groups, pro=(sc, sh, 0.01), rows=rows, slope=0.01)
    rm64, rv64 = torch.zeros(C, dtype=torch.float64, device="cuda"), torch.ones(C, dtype=torch.float64, device="cuda")
    for g in range(groups):
        y, mean, var, invstd, p64, a64 = _f64_layer(A[g * Mg:(g + 1) * Mg], W, b, gamma, beta, sc[g], sh[g], rows)
        scale64 = gamma.double() * invstd
        assert rel_l2(st[0, g].cpu().numpy(), scale64.cpu().numpy(), "dom.scale") <= 2e-6
        assert rel_l2(st[1, g].cpu().numpy(), (beta.double() - mean * scale64).cpu().numpy(), "dom.shift") <= 5e-6
        assert rel_l2(st[2, g].cpu().numpy(), invstd.cpu().numpy(), "dom.invstd") <= 2e-6
        assert rel_l2(st[3, g].cpu().numpy(), mean.cpu().numpy(), "dom.mean") <= 2e-6
        pg, ag, yg = pooled[g * Bn:(g + 1) * Bn], arg[g * Bn:(g + 1) * Bn].long(), yarg[g * Bn:(g + 1) * Bn]
        assert rel_l2(pg.cpu().numpy(), p64.cpu().numpy(), "dom.pooled") <= 3e-6
        # the arg-max row carries the maximum (index equality up to rounding ties: compare the float64 activation at OUR row)
        cols = torch.arange(C, device="cuda").view(1, C).expand(Bn, C)
        assert ((ag // rows) == torch.arange(Bn, device="cuda").view(Bn, 1)).all()
        z64 = (y[ag, cols] - mean) * scale64 + beta.double()
        z64 = torch.where(z64 > 0, z64, z64 * 0.01)
        assert (z64 - p64).abs().max().item() <= 2e-5 * p64.abs().max().item()
        assert (ag == a64).float().mean().item() >= 0.999
        assert rel_l2(yg.cpu().numpy(), y[ag, cols].cpu().numpy(), "dom.yarg") <= 2e-6
        rm64 = 0.9 * rm64 + 0.1 * mean
        rv64 = 0.9 * rv64 + 0.1 * var * (Mg / (Mg - 1.0))
    assert rel_l2(rm.cpu().numpy(), rm64.cpu().numpy(), "dom.running_mean") <= 2e-6
    assert rel_l2(rv.cpu().numpy(), rv64.cpu().numpy(), "dom.running_var") <= 2e-6
