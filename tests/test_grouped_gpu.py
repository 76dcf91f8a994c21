"""GPU: the grouped launches of the D step's lock-step backward (round 5): every `ops.*_multi` entry runs the body of its stand-alone kernel on
one workgroup range per problem, so each problem's results must be BIT-IDENTICAL to its separate launch; nets.d_backward_joint against the
per-pass calls it stands for (Generation/Discriminator.py:97-115 three times per D step, Common/gradient_penalty.py:19-37)."""
import pytest
import torch

from spgan import fixture_rng as fr
from test_kernels_gpu import close, ops, rnd  # noqa: F401  (ops is a fixture)

pytestmark = pytest.mark.gpu


def _same(a, b, what):
    if isinstance(a, (tuple, list)):
        assert len(a) == len(b), what
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, "%s[%d]" % (what, i))
    elif isinstance(a, torch.Tensor):
        assert torch.equal(a, b), "%s: grouped launch differs from the separate one (max %.3e)" % (what, (a - b).abs().max().item())


def _dual_spec(ops, tag, M, Nb, mode, tail):
    Na = 256 if Nb >= 128 else 128
    sc, sh = rnd(tag + ".sc", (Nb,)), rnd(tag + ".sh", (Nb,), 0.3)
    mu, iv = rnd(tag + ".mu", (Nb,), 0.2), rnd(tag + ".iv", (Nb,)).abs() + 0.5
    sp = dict(scale=sc, shift=sh, mean=mu, invstd=iv, slope=0.01)
    if mode == "act":
        y3 = rnd(tag + ".y3", (M, Na), 1.5)
        Wm = rnd(tag + ".G", (Na, Na), 0.05); Wm = (Wm + Wm.t()).contiguous()
        sp.update(dy=ops.ActOperand(y3, sc, sh, 0.01), W=Wm, y_ref=y3, bias=rnd(tag + ".cv", (Nb,), 0.2), rowadd=rnd(tag + ".E", (M, Nb), 0.3), with_colsum=True)
    else:
        g, y = rnd(tag + ".g", (M, Na)), rnd(tag + ".y", (M, Na), 2.0) + 0.3
        mean, inv = y.mean(0), 1.0 / torch.sqrt(y.var(0, unbiased=False) + 1e-5)
        gamma = rnd(tag + ".ga", (Na,)).abs() + 0.5
        sums = torch.cat([g.sum(0), (g * ((y - mean) * inv)).sum(0)])
        dy = ops.bn_bwd_lazy(g, y, mean, inv, gamma, sums, M) if mode == "lazy" else ops.bn_bwd_apply(g, y, mean, inv, gamma, sums, M)
        sp.update(dy=dy, W=rnd(tag + ".W", (Na, Nb), 0.1), y_ref=rnd(tag + ".prev", (M, Nb), 1.5))
    if tail == "coef":
        sp["coef_bn"] = (rnd(tag + ".cg", (Nb,)).abs() + 0.5, M)
    elif tail == "phaseb":
        v = lambda n: rnd(tag + "." + n, (Nb,))
        sp["phaseb"] = ((v("U0"), v("U1"), v("Ugz"), v("S0"), v("S1"), M), v("pg").abs() + 0.5, v("pinv").abs() + 0.5)
        if mode == "dense":
            sp["out"] = rnd(tag + ".acc", (Na, Nb)); sp["beta"] = 1.0
    return sp


@pytest.mark.parametrize("M,Nb,mode", [(16384, 256, "act"), (16384, 128, "dense"), (8192, 64, "lazy")])
def test_gemm_dual_stored_adjoint_and_phase_b_coefficients(ops, M, Nb, mode):
    """gout = (add, scale): the stored tile is add + scale*g with the statistics of g; a 4-element phaseb also yields the coefficients of the lazy
    operand p*X + q*y + r that equals bn_bwd_apply(add, y, mean, invstd, None, sums, M, add=(g, scale)) -- phase B of the double backward without
    its BatchNorm-backward pass (nets._phaseb_below)."""
    sp = _dual_spec(ops, "gsa%d.%d" % (Nb, M), M, Nb, mode, "phaseb")
    sp.pop("out", None); sp.pop("beta", None)
    plain = ops.gemm_dual(defer=False, **sp)
    g, s0, s1 = plain[1], plain[2], plain[3]
    sums, dg = plain[-2], plain[-1]
    add, scale = rnd("gsa.add%d" % Nb, (M, Nb)), rnd("gsa.scale%d" % Nb, (Nb,)).abs() + 0.5
    ymean = rnd("gsa.ymean%d" % Nb, (Nb,), 0.2)
    sp2 = dict(sp, phaseb=tuple(sp["phaseb"]) + (ymean,), gout=(add, scale))
    res = ops.gemm_dual(defer=False, **sp2)
    X, coef = res[1], res[-1]
    assert torch.equal(res[0], plain[0]) and torch.equal(res[2], s0) and torch.equal(res[3], s1) and torch.equal(res[-3], sums) and torch.equal(res[-2], dg)
    close(X, add.double() + scale.double() * g.double(), rtol=1e-6, atol=1e-6, what="stored adjoint")
    y = sp["y_ref"]
    inv = sp["phaseb"][2]
    ref = ops.bn_bwd_apply(add, y, ymean, inv, None, sums, M, add=(g, scale))
    close(ops.Affine2(X, y, coef).dense(), ref, rtol=2e-6, atol=2e-6 * float(ref.abs().max()), what="lazy phase-B operand vs bn_bwd_apply")


def test_collapsed_layer_pair_of_split_launches_equals_the_fused_launch(ops):
    """The split-bf16 mode's route through the collapsed 256 -> 1024 layer backward (ops.collapsed_pair_preferred): gemm_tn + gemm_nt_bnbwd with the
    phase-B tail and the stored tile X = add + scale*g (spgan_gemm_nt_args.gout_add, an epilogue of csrc/gemm_wide3.hip) against the ONE exact-fp32
    gemm_dual launch it replaces -- every result at fp32-product tolerance; the stored-tile epilogue is refused where that kernel does not run."""
    M, Nb = 32768, 256
    sp = _dual_spec(ops, "pair%d" % M, M, Nb, "act", "phaseb")
    add, scale = rnd("pair.add", (M, Nb)), rnd("pair.scale", (Nb,)).abs() + 0.5
    ymean = rnd("pair.ymean", (Nb,), 0.2)
    sp = dict(sp, phaseb=tuple(sp["phaseb"]) + (ymean,), gout=(add, scale))
    gram, X, s0, s1, cs, sums, dg, coef = ops.gemm_dual(defer=False, **sp)
    y3, pro3 = sp["y_ref"], (sp["scale"], sp["shift"], sp["slope"])
    kw = dict(pro=pro3, bias=sp["bias"], rowadd=sp["rowadd"], phaseb=sp["phaseb"], gout=sp["gout"])
    with pytest.raises(Exception):      # fp32 operands: no kernel with that epilogue
        ops.gemm_nt_bnbwd(y3, sp["W"], y3, sp["scale"], sp["shift"], sp["mean"], sp["invstd"], sp["slope"], **kw)
    ops.set_mfma_operands("bf16x3")
    try:
        assert ops.collapsed_pair_preferred(M, Nb) == ops.SPLIT_PAIR[0]
        gram2, cs2 = ops.gemm_tn(y3, y3, a_pro=pro3, pro=pro3, with_colsum=True)
        X2, s02, s12, sums2, dg2, coef2 = ops.gemm_nt_bnbwd(y3, sp["W"], y3, sp["scale"], sp["shift"], sp["mean"], sp["invstd"], sp["slope"], **kw)
        plain = ops.gemm_nt_bnbwd(y3, sp["W"], y3, sp["scale"], sp["shift"], sp["mean"], sp["invstd"], sp["slope"], pro=pro3, bias=sp["bias"], rowadd=sp["rowadd"])
    finally:
        ops.set_mfma_operands("f32")
    assert torch.equal(s02, plain[1]) and torch.equal(s12, plain[2])            # the statistics are those of g, not of the stored tile
    close(X2, add.double() + scale.double() * plain[0].double(), rtol=1e-6, atol=1e-6, what="stored tile = add + scale*g")
    for a, b, what in ((gram2, gram, "gram"), (cs2, cs, "colsum"), (X2, X, "stored tile"), (s02, s0, "s0"), (s12, s1, "s1"), (sums2, sums, "phase-B sums"),
                       (dg2, dg, "dgamma"), (coef2, coef, "lazy coefficients")):
        close(a, b, rtol=2e-5, atol=2e-5 * float(b.abs().max()), what=what)


@pytest.mark.parametrize("M,Nb,modes", [(16384, 256, ("act", "act", "act")), (65536, 128, ("lazy", "lazy", "dense")), (16384, 64, ("lazy", "lazy", "dense")),
                                        (8192, 128, ("lazy", "dense"))])
def test_gemm_dual_multi_equals_separate_launches(ops, M, Nb, modes):
    """The shapes and operand mixes of the D step: the collapsed fc2.0 (three activation operands with bias + row addend + colsum; finalize tails
    coef / coef / phase B), mlps.6 and mlps.3 (two lazy operands and phase B's dense one with beta = 1 accumulation)."""
    tails = ["coef"] * (len(modes) - 1) + ["phaseb"]
    specs = [_dual_spec(ops, "gdm%d.%d.%d" % (Nb, M, i), M, Nb, m, t) for i, (m, t) in enumerate(zip(modes, tails))]
    if M == 65536:     # the last problem as the double backward issues it: stored adjoint + lazy coefficients
        specs[-1]["phaseb"] = tuple(specs[-1]["phaseb"]) + (rnd("gdm.ymean", (Nb,), 0.2),)
        specs[-1]["gout"] = (rnd("gdm.add", (M, Nb)), rnd("gdm.gscale", (Nb,)).abs() + 0.5)
    acc0 = [sp["out"].clone() if "out" in sp else None for sp in specs]
    got = ops.gemm_dual_multi(specs, defer=False)
    got = [tuple(t.clone() for t in r) for r in got]
    for sp, a0 in zip(specs, acc0):
        if a0 is not None:
            sp["out"].copy_(a0)
    for i, sp in enumerate(specs):
        ref = ops.gemm_dual(defer=False, **sp)
        _same(got[i], ref, "problem %d" % i)
    # and the switch that issues them one by one
    for sp, a0 in zip(specs, acc0):
        if a0 is not None:
            sp["out"].copy_(a0)
    ops.GROUPED[0] = False
    try:
        one = ops.gemm_dual_multi(specs, defer=False)
    finally:
        ops.GROUPED[0] = True
    _same(got, [tuple(r) for r in one], "GROUPED = False")


def test_collapse_prep_and_wgrad_collapse_grouped(ops):
    from test_collapse_gpu import _sparse
    C, K, B, rows = 1024, 256, 4, 256
    W = rnd("gcp.W", (C, K), 0.1)
    v = lambda n, s=1.0: rnd("gcp." + n, (C,), s)
    problems = [(v("a0"), v("be0", 0.3), v("b", 0.2)), (v("a1"), v("be1", 0.3), v("b", 0.2)), (v("c1"), None, None), (v("c2"), v("c3", 0.3), v("b", 0.2))]
    sets = [_sparse("gcp.s%d" % i, B, rows, C) for i in range(3)]
    outs, Es = ops.collapse_prep(W, problems, [s[0] for s in sets], [s[1] for s in sets], rows)
    for i, pr in enumerate(problems):
        _same(outs[i], ops.wt_diag_w(W, *pr), "weight problem %d" % i)
    for i, (val, arg) in enumerate(sets):
        _same(Es[i], ops.sparse_rows_nt(val, arg, rows, W), "sparse set %d" % i)
    (o1,), E1 = ops.collapse_prep(W, problems[:1], sets[0][0], sets[0][1], rows)          # the single-set form still returns a tensor
    _same((o1, E1), (outs[0], Es[0]), "single form")
    # wgrad_collapse: two single-product problems and phase B's two-product accumulating one
    N = K
    X = [rnd("gcp.X%d" % i, (N, K)) for i in range(3)]
    X2 = rnd("gcp.X2", (K, N))
    Bm = [rnd("gcp.Bm%d" % i, (B * rows, N)) for i in range(3)]
    pro = (rnd("gcp.psc", (N,)), rnd("gcp.psh", (N,), 0.3), 0.01)
    acc = rnd("gcp.acc", (C, N))
    specs = [dict(W=W, X1=X[i], a1=v("wa%d" % i), b1=v("b", 0.2), d1=v("wd%d" % i), v1=rnd("gcp.v%d" % i, (N,)), sparse=(sets[i][0], sets[i][1], rows, Bm[i], pro))
             for i in range(2)]
    specs.append(dict(W=W, X1=X[2], a1=v("wa2"), b1=v("b", 0.2), d1=v("wd2"), v1=rnd("gcp.v2", (N,)), X2=X2, x2_t=True, a2=v("wa3"),
                      sparse=(sets[2][0], sets[2][1], rows, Bm[2], pro), out=acc.clone(), accumulate=True))
    got = [t.clone() for t in ops.wgrad_collapse_multi(specs)]
    specs[2]["out"] = acc.clone()
    for i, sp in enumerate(specs):
        _same(got[i], ops.wgrad_collapse(**sp), "wgrad problem %d" % i)


def test_small_grouped_launches(ops):
    """pool_bwd_stats_multi, gemm_tn_narrow_multi (lazy + dense operand, beta = 1 accumulation) and multi_addn."""
    B, C, M = 8, 1024, 16384
    specs = []
    for i in range(2):
        t = "sgl%d." % i
        g = torch.Generator().manual_seed(50 + i)
        arg = (torch.randint(0, M // B, (B, C), generator=g) + torch.arange(B)[:, None] * (M // B)).int().cuda()
        specs.append(dict(gpool=rnd(t + "gp", (B, C)), pooled=rnd(t + "po", (B, C)), argmax=arg, y=rnd(t + "ya", (B, C)), mean=rnd(t + "mu", (C,), 0.2),
                          invstd=rnd(t + "iv", (C,)).abs() + 0.5, slope=0.01, prep=(rnd(t + "ga", (C,)).abs() + 0.5, M, None, M // B)))
    got = ops.pool_bwd_stats_multi(specs)
    for i, sp in enumerate(specs):
        gval, sums, dy = ops.pool_bwd_stats(**sp)
        _same((got[i][0], got[i][1], got[i][2].alpha, got[i][2].beta, got[i][2].sp_val), (gval, sums, dy.alpha, dy.beta, dy.sp_val), "pool_bwd pass %d" % i)
        assert got[i][2].sp_arg is sp["argmax"] and got[i][2].rows == M // B
    # L0 weight gradient: [M,64]^T [M,3]
    x = [rnd("sgl.x%d" % i, (M, 3)) for i in range(3)]
    A = []
    for i in range(2):
        g_, y_ = rnd("sgl.g%d" % i, (M, 64)), rnd("sgl.y%d" % i, (M, 64), 2.0)
        A.append(ops.Affine2(g_, y_, rnd("sgl.coef%d" % i, (3, 64))))
    A.append(rnd("sgl.ybar", (M, 64)))
    acc = rnd("sgl.acc", (64, 3))
    tsp = [dict(A=A[0], Bm=x[0]), dict(A=A[1], Bm=x[1]), dict(A=A[2], Bm=x[2], out=acc.clone(), beta=1.0)]
    got = ops.gemm_tn_narrow_multi(tsp)
    ops.flush_tn()
    got = [t.clone() for t in got]
    for i in range(3):
        ref = ops.gemm_tn(A[i], x[i], out=acc.clone() if i == 2 else None, beta=1.0 if i == 2 else 0.0, defer=True)
        ops.flush_tn()
        _same(got[i], ref, "narrow product %d" % i)
        ref2 = ops.gemm_tn(A[i], x[i], out=acc.clone() if i == 2 else None, beta=1.0 if i == 2 else 0.0)      # the non-deferred call the double backward used
        _same(got[i], ref2, "narrow product %d (own reduction)" % i)
    # multi_addn = successive multi_add launches
    dst = [rnd("sgl.d%d" % i, (n,)) for i, n in enumerate((1024, 262144, 77, 256 * 128))]
    src = [[rnd("sgl.s%d.%d" % (i, j), tuple(d.shape)) for j in range(k)] for i, (d, k) in enumerate(zip(dst, (3, 2, 3, 1)))]
    ref = [d.clone() for d in dst]
    for j in range(3):
        pairs = [(r, s[j]) for r, s in zip(ref, src) if len(s) > j]
        ops.multi_add([p[0] for p in pairs], [p[1] for p in pairs])
    ops.multi_addn(dst, src)
    _same(dst, ref, "multi_addn")


def _d_setup(B, N, salt):
    import spgan
    from oracle import spgan_oracle as orc
    from spgan import nets

    class O:
        np = N; nk = 20; nz = 128; softmax = True; off = False; attn = False; use_head = False; eql = False; z_norm = False; small_d = False
    D = spgan.Discriminator(O)
    D.load_state_dict({**D.state_dict(), **fr.init_params(orc.discriminator_shapes(), salt=salt)})
    D.cuda().train()
    P = {k: v.detach() for k, v in D.named_parameters()}
    xs = [fr.synthetic_real(B, N, seed=salt + i).cuda().reshape(B * N, 3).contiguous() for i in range(3)]
    return nets, P, xs


@pytest.mark.parametrize("mode", ["f32", "f16"])
@pytest.mark.parametrize("B,N,with_dbl", [(8, 2048, True), (32, 2048, True), (16, 1024, False)])
def test_d_backward_joint_equals_separate_calls(ops, B, N, with_dbl, mode):
    """nets.d_backward_joint (the real pass, the fake pass and the penalty's double backward in lock step, every layer's launch issued once) gives,
    per pass, the gradients of nets.d_backward / nets.d_double_backward bit for bit.  mode "f16": no fused layer-backward kernel -- the joint node
    issues each layer's two launches per pass (nets._layer_backward_multi) and groups the rest."""
    ops.set_mfma_operands(mode)
    try:
        _joint_equals_separate(ops, B, N, with_dbl)
    finally:
        ops.set_mfma_operands("f32")


def _joint_equals_separate(ops, B, N, with_dbl):
    nets, P, xs = _d_setup(B, N, 31)
    M = B * N

    def run(joint):
        with torch.no_grad():
            pres = nets.d_forward_groups(P, None, xs, update_running=False, pm_shape=(B, N))
            gpools = [rnd("dbj.gp%d" % i, (B, 1024), 1e-2) for i in range(2)]
            hat = None
            if with_dbl:
                pooled_h, hctx = pres[2]
                logits, hctx["hs"] = nets.d_head_forward(P, pooled_h)
                dx, _, saved = nets.d_backward(P, hctx, torch.ones_like(logits), True, False, keep_for_double=True)
                v = rnd("dbj.v", (M, 3), 1e-3)
                hat = (hctx, saved, v)
            if joint:
                assert nets.d_joint_ok(P, [c for _, c in pres])
                fg, hg = nets.d_backward_joint(P, [(pres[0][1], gpools[0]), (pres[1][1], gpools[1])], hat)
                ops.flush_tn()
                return list(fg) + ([hg] if hg is not None else [])
            out = []
            for i in range(2):
                out.append(nets.d_backward(P, pres[i][1], None, False, True, gpool=gpools[i])[1])
                ops.flush_tn()
            if with_dbl:
                out.append(nets.d_double_backward(P, hat[0], hat[1], hat[2])[0])
                ops.flush_tn()
            return out
    a, b = run(True), run(False)
    assert len(a) == len(b) == (3 if with_dbl else 2)
    for ci, (ga, gb) in enumerate(zip(a, b)):
        assert set(ga) == set(gb), (ci, set(ga) ^ set(gb))
        for n in ga:
            if isinstance(ga[n], int) or isinstance(gb[n], int):
                assert isinstance(ga[n], int) and isinstance(gb[n], int), n
                continue
            assert torch.equal(ga[n].reshape(-1), gb[n].reshape(-1)), "pass %d, %s: max diff %.3e" % (ci, n, (ga[n].reshape(-1) - gb[n].reshape(-1)).abs().max().item())


@pytest.mark.parametrize("gan,use_gp,small_d", [("wgan", True, False), ("ls", False, False), ("hinge", True, True)])
def test_train_step_joint_d_backward_equals_one_node_per_pass(ops, gan, use_gp, small_d):
    """TrainStep with the joint D-step node (default) against one autograd node per pass: D's gradients differ only by the order in which the
    three per-pass gradients are added into .grad (1e-6 of each tensor), losses agree."""
    import spgan
    from oracle import spgan_oracle as orc

    class O:
        np = 2048; nk = 20; nz = 128; softmax = True; off = False; attn = False; use_head = False; eql = False; z_norm = False
    O.small_d = small_d            # --small_d: a 512-wide top layer (Discriminator.py:52)
    B, N = 8, 2048
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
    real = fr.synthetic_real(B, N, seed=71).cuda()
    z_d, z_g = fr.latent(B, N, seed=72).cuda(), fr.latent(B, N, seed=73).cuda()
    alpha = fr.uniform("tsj.alpha", (B, 1, 1), 0.0, 1.0).cuda()
    outs = []
    for joint in (True, False):
        G, D = spgan.Generator(O), spgan.Discriminator(O)
        G.load_state_dict({**G.state_dict(), **fr.init_params(orc.generator_shapes(), salt=7)})
        if not small_d:
            D.load_state_dict({**D.state_dict(), **fr.init_params(orc.discriminator_shapes(), salt=7)})
        else:
            torch.manual_seed(5)
            D.load_state_dict({k: (torch.randn_like(v) * 0.05 if v.dtype.is_floating_point and "running" not in k else v) for k, v in D.state_dict().items()})
            with torch.no_grad():
                for k, v in D.named_parameters():
                    if k.endswith("1.weight") or k.endswith("4.weight") or k.endswith("7.weight"):      # BatchNorm scales around one
                        v.add_(1.0)
        G.cuda().train(); D.cuda().train()
        tr = spgan.TrainStep(G, D, gan=gan, use_gp=use_gp)
        tr.joint_d_backward = joint
        if joint:
            calls = []
            from spgan import nets
            real_joint = nets.d_backward_joint
            nets.d_backward_joint = lambda P, firsts, dbl=None: (calls.append(len(firsts)), real_joint(P, firsts, dbl))[1]
            try:
                outs.append(tr.step(x, real, z_d, z_g, alpha=alpha, keep_grads=True))
            finally:
                nets.d_backward_joint = real_joint
            assert calls == [2], "the joint route did not run"
            continue
        outs.append(tr.step(x, real, z_d, z_g, alpha=alpha, keep_grads=True))
    a, b = outs
    assert abs(a["loss_d"].item() - b["loss_d"].item()) <= 1e-6 * abs(b["loss_d"].item()) + 1e-7
    for n in a["d_grads"]:
        ga, gb = a["d_grads"][n], b["d_grads"][n]
        assert (ga - gb).norm().item() <= 2e-6 * gb.norm().item() + 1e-12, (n, (ga - gb).norm().item(), gb.norm().item())


def test_gp_penalty_fwd_bwd_equals_separate_calls(ops):
    B, L = 32, 3 * 2048
    g = rnd("gpf.g", (B, L), 1e-3)
    la = rnd("gpf.la", (1,))
    loss, norms = ops.gp_penalty_fwd(g, 1.0, 10.0)
    v = ops.gp_penalty_bwd(g, norms, 1.0, 10.0, None)
    l2, n2, v2, tot = ops.gp_penalty_fwd_bwd(g, 1.0, 10.0, la)
    assert torch.equal(loss, l2) and torch.equal(norms, n2) and torch.equal(v, v2) and torch.equal(tot, la + loss)
    assert ops.gp_penalty_fwd_bwd(g, 1.0, 10.0)[3] is None


@pytest.mark.parametrize("gan,use_gp,graph", [("wgan", True, False), ("ls", False, False), ("wgan", True, True)])
def test_paired_generator_forwards_equal_separate_forwards(ops, gan, use_gp, graph):
    """TrainStep with the step's two generator forwards as ONE pipeline (Generator.forward_pair / nets.g_pair_forward) against two separate
    forwards, eagerly and replayed as a hipGraph: the HIP kernels compute every row independently of the row count, so clouds, losses and
    BatchNorm buffers of the first step agree to rounding of the kernels whose tile geometry depends on M, and the trajectories stay together."""
    import spgan
    from oracle import spgan_oracle as orc
    from spgan import nets

    class O:
        np = 2048; nk = 20; nz = 128; softmax = True; off = False; attn = False; use_head = False; eql = False; z_norm = False; small_d = False
    B, N = 8, 2048
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
    outs, calls = [], []
    real_pair = nets.g_pair_forward
    nets.g_pair_forward = lambda *a, **k: (calls.append(1), real_pair(*a, **k))[1]
    try:
        for pair in (True, False):
            G, D = spgan.Generator(O), spgan.Discriminator(O)
            G.load_state_dict({**G.state_dict(), **fr.init_params(orc.generator_shapes(), salt=7)})
            D.load_state_dict({**D.state_dict(), **fr.init_params(orc.discriminator_shapes(), salt=7)})
            G.cuda().train(); D.cuda().train()
            tr = spgan.TrainStep(G, D, gan=gan, use_gp=use_gp, graph=graph, graph_warmup=1)
            tr.pair_g_forwards = pair
            rec = []
            for step in range(4 if graph else 2):
                real = fr.synthetic_real(B, N, seed=81 + step).cuda()
                z_d = fr.latent(B, N, seed=82 + 2 * step)[:, :1, :].contiguous().cuda()
                z_g = fr.latent(B, N, seed=83 + 2 * step)[:, :1, :].contiguous().cuda()
                alpha = fr.uniform("pairg.alpha.%d" % step, (B, 1, 1), 0.0, 1.0).cuda()
                info = tr.step(x, real, z_d, z_g, alpha=alpha, keep_grads=not graph)
                rec.append({k: v.detach().clone() if isinstance(v, torch.Tensor) else {n: g.clone() for n, g in v.items()} for k, v in info.items()})
            outs.append((rec, {k: v.clone() for k, v in G.state_dict().items()}, {k: v.clone() for k, v in D.state_dict().items()}))
    finally:
        nets.g_pair_forward = real_pair
    assert len(calls) >= 2, "the paired route did not run"
    (ra, ga, da), (rb, gb, db) = outs
    a, b = ra[0], rb[0]
    assert abs(a["loss_d"].item() - b["loss_d"].item()) <= 2e-6 * abs(b["loss_d"].item()) + 1e-7
    assert abs(a["loss_g"].item() - b["loss_g"].item()) <= 2e-5 * abs(b["loss_g"].item()) + 1e-6
    if not graph:
        close(a["fake_d"], b["fake_d"], rtol=1e-6, atol=1e-6, what="D-step cloud")
        close(a["fake_g"], b["fake_g"], rtol=1e-6, atol=1e-6, what="G-step cloud")
        for n in a["g_grads"]:
            close(a["g_grads"][n], b["g_grads"][n], rtol=2e-4, atol=1e-9, what="G gradient " + n)
        for n in a["d_grads"]:
            close(a["d_grads"][n], b["d_grads"][n], rtol=2e-5, atol=1e-9, what="D gradient " + n)
    for k in ga:       # after the last step: parameters within a few Adam steps' rounding, BatchNorm buffers and call counts together
        if "num_batches" in k:
            assert torch.equal(ga[k], gb[k]), k
        else:
            close(ga[k].float(), gb[k].float(), rtol=2e-3, atol=1e-3, what=k)
    for k in da:
        if "num_batches" in k:
            assert torch.equal(da[k], db[k]), k
        else:
            close(da[k].float(), db[k].float(), rtol=2e-3, atol=1e-3, what=k)
