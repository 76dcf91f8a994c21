#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (read-only at
/root/reference) on deterministic inputs.  Runs only in the build container; the
GPU box never sees the reference, only these vectors.

    python tests/golden/make_golden.py

What is executed from the reference (never copied into this repo):
  * Generation.Generator.{Generator,EdgeBlock,AdaptivePointNorm}, Generation.Discriminator.Discriminator,
    Generation.modules.get_edge_features, Common.pointnet_util.*, Common.gradient_penalty.GradientPenalty
    -- imported normally;
  * Common/loss_utils.py {dis_loss,gen_loss,...} and Common/pointconv_util.py {knn_point,group,...}
    -- those modules do not import here (CUDA extensions / removed sklearn API), so the
    individual function definitions are compiled straight from the reference file with `ast`
    and executed with `.cuda()` patched to the identity.
"""
import ast
import functools
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "sp-gan_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from spgan import fixture_rng as fr                                   # noqa: E402
from oracle import spgan_oracle as orc                                # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self                        # reference hard-codes .cuda()
torch.set_num_threads(8)

from Generation.Generator import Generator, EdgeBlock, AdaptivePointNorm   # noqa: E402
from Generation.Discriminator import Discriminator                         # noqa: E402
from Generation.modules import get_edge_features                           # noqa: E402
import Common.pointnet_util as pnu                                         # noqa: E402
from Common.gradient_penalty import GradientPenalty                        # noqa: E402


def extract_functions(path, names, extra_globals):
    """Compile selected top-level defs from a reference source file."""
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    mod = ast.Module(body=body, type_ignores=[])
    ns = dict(extra_globals)
    exec(compile(mod, path, "exec"), ns)
    return types.SimpleNamespace(**{n: ns[n] for n in names})


from torch.autograd import Variable                                        # noqa: E402
LU = extract_functions(os.path.join(REF, "Common/loss_utils.py"),
                       ["dis_loss", "gen_loss", "BCEloss", "BCEfakeloss", "smooth_labels", "noisy_labels"],
                       dict(torch=torch, nn=nn, F=F, np=np, functools=functools, Variable=Variable))
PCU = extract_functions(os.path.join(REF, "Common/pointconv_util.py"),
                        ["square_distance", "index_points", "knn_point", "group", "farthest_point_sample"],
                        dict(torch=torch, nn=nn, F=F, np=np))


class Opts:
    np = 2048; nk = 20; nz = 128; softmax = True; off = False; attn = False
    use_head = False; eql = False; z_norm = False; small_d = False


def summarize(t, full_limit=4096, nsamp=1024):
    """Full tensor when small, else (l2, sum, strided samples)."""
    a = t.detach().cpu().numpy()
    if a.size <= full_limit:
        return {"full": a}
    flat = a.reshape(-1)
    stride = max(1, flat.size // nsamp)
    return {"l2": np.float64(np.sqrt((flat.astype(np.float64) ** 2).sum())),
            "sum": np.float64(flat.astype(np.float64).sum()),
            "stride": np.int64(stride), "samples": flat[::stride][:nsamp].copy()}


def put(d, name, t, **kw):
    for k, v in summarize(t, **kw).items():
        d["%s|%s" % (name, k)] = v


def save(fname, d):
    path = os.path.join(HERE, fname)
    np.savez_compressed(path, **d)
    print("%-28s %7.1f KB  (%d arrays)" % (fname, os.path.getsize(path) / 1024, len(d)))


def load_into(module, params):
    sd = module.state_dict()
    for k, v in params.items():
        assert sd[k].shape == v.shape, (k, sd[k].shape, v.shape)
    module.load_state_dict({**sd, **params})
    return module


# ---------------------------------------------------------------- G1: kNN + edge features
def g1():
    d = {}
    for N in (256, 512, 1024, 2048, 4096):
        x = fr.sphere_template(N)[None].transpose(2, 1).contiguous()       # [1,3,N]
        ee, idx = get_edge_features(x, 10, return_idx=True)
        d["sphere%d|idx" % N] = idx.view(N, 10).numpy().astype(np.int32)
        if N <= 512:
            d["sphere%d|ee" % N] = ee.numpy()
    for N, C in ((256, 64), (512, 64), (300, 5)):
        x = fr.normal("g1.feat.%d.%d" % (N, C), (2, C, N), 0.5)
        ee, idx = get_edge_features(x, 10, return_idx=True)
        dist = orc.pairwise_sqdist(x)
        srt = torch.sort(dist, dim=2)[0][:, :, :12]
        d["feat%d_%d|idx" % (N, C)] = idx.view(2, N, 10).numpy().astype(np.int32)
        d["feat%d_%d|sorted_dist" % (N, C)] = srt.numpy()                   # for the tie-aware protocol
        put(d, "feat%d_%d|ee" % (N, C), ee, full_limit=0, nsamp=2048)
    save("g1_edge_features.npz", d)


# ---------------------------------------------------------------- G2: EdgeBlock
def g2():
    d = {}
    for tag, fin, fout, B, N in (("ec1", 3, 64, 2, 256), ("ec2", 64, 128, 2, 256)):
        blk = EdgeBlock(fin, fout, 10)
        pref = "EdgeConv1." if fin == 3 else "EdgeConv2."
        shapes = {k: v for k, v in orc.generator_shapes().items() if k.startswith(pref)}
        params = {k[len(pref):]: v for k, v in fr.init_params(shapes, salt=2).items()}
        load_into(blk, params).train()
        if fin == 3:
            x = fr.sphere_template(N)[None].repeat(B, 1, 1).transpose(2, 1).contiguous()
            x = x + 0.01 * fr.normal("g2.jit", x.shape)                     # distinct clouds per sample
        else:
            x = fr.normal("g2.x.%s" % tag, (B, fin, N), 0.7)
        x.requires_grad_(True)
        _, idx = get_edge_features(x.detach(), 10, return_idx=True)
        y = blk(x)
        dy = fr.normal("g2.dy.%s" % tag, y.shape)
        grads = torch.autograd.grad(y, [x] + list(blk.parameters()), dy)
        d[tag + "|idx"] = idx.view(B, N, 10).numpy().astype(np.int32)
        put(d, tag + "|y", y, full_limit=1 << 20)
        put(d, tag + "|dx", grads[0], full_limit=1 << 20)
        for (n, _), g in zip(blk.named_parameters(), grads[1:]):
            put(d, tag + "|grad|" + n, g, full_limit=1 << 15)
        for n, b in blk.named_buffers():
            d[tag + "|buf|" + n] = b.numpy()
    save("g2_edgeblock.npz", d)


# ---------------------------------------------------------------- G3: AdaptivePointNorm
def g3():
    d = {}
    B, C, N = 2, 64, 256
    m = AdaptivePointNorm(C, 128)
    params = fr.init_params({"style.weight": (2 * C, 128, 1), "style.bias": (2 * C,)}, salt=3)
    load_into(m, params)
    x = fr.normal("g3.x", (B, C, N)).requires_grad_(True)
    s = fr.normal("g3.s", (B, 128, N), 0.3).requires_grad_(True)
    y = m(x, s)
    dy = fr.normal("g3.dy", y.shape)
    gx, gs, gw, gb = torch.autograd.grad(y, [x, s, m.style.weight, m.style.bias], dy)
    for n, t in (("y", y), ("dx", gx), ("dstyle", gs), ("dw", gw), ("db", gb)):
        put(d, n, t, full_limit=1 << 20)
    save("g3_adain.npz", d)


# ---------------------------------------------------------------- G4/G5: Generator, Discriminator
def make_gd(salt=4):
    G = load_into(Generator(Opts), fr.init_params(orc.generator_shapes(), salt=salt)).train()
    D = load_into(Discriminator(Opts), fr.init_params(orc.discriminator_shapes(), salt=salt)).train()
    return G, D


def g4_g5():
    B, N = 4, 256
    G, D = make_gd()
    x = fr.sphere_template(N)[None].repeat(B, 1, 1)
    z = fr.latent(B, N, seed=44)
    # --- D alone, on a synthetic real cloud
    real = fr.synthetic_real(B, N, seed=5).transpose(2, 1).contiguous().requires_grad_(True)
    d = {}
    logit = D(real)
    loss = ((logit - 1.0) ** 2).mean()
    grads = torch.autograd.grad(loss, [real] + list(D.parameters()))
    put(d, "logit", logit); put(d, "dx", grads[0], full_limit=1 << 20)
    for (n, _), g in zip(D.named_parameters(), grads[1:]):
        put(d, "grad|" + n, g)
    for n, b in D.named_buffers():
        d["buf|" + n] = b.numpy()
    save("g5_discriminator.npz", d)
    # --- G forward + stages + grads of mse(D(G),1)
    G, D = make_gd()
    stages = {}
    hooks = [G.head.register_forward_hook(lambda m, i, o: stages.__setitem__("style", o.detach().clone())),
             G.adain1.register_forward_hook(lambda m, i, o: stages.__setitem__("x1", o.detach().clone())),
             G.adain2.register_forward_hook(lambda m, i, o: stages.__setitem__("x2", o.detach().clone())),
             G.global_conv.register_forward_hook(lambda m, i, o: stages.__setitem__("feat_global", o.detach().clone()))]
    out = G(x, z)
    for h in hooks:
        h.remove()
    _, idx1 = get_edge_features(x.transpose(2, 1).contiguous(), 10, return_idx=True)
    # EdgeConv2's graph is built on adain1's output
    _, idx2 = get_edge_features(stages["x1"], 10, return_idx=True)
    logit = D(out)
    loss = ((logit - 1.0) ** 2).mean()
    # gradient goldens use an injected dL/d(out): end-to-end gradients through D are
    # kink-limited and would not pin anything tighter than ~1e-2
    dy = fr.normal("g4.dy", out.shape)
    grads = torch.autograd.grad(out, list(G.parameters()), dy)
    d = {"idx1": idx1.view(B, N, 10).numpy().astype(np.int32), "idx2": idx2.view(B, N, 10).numpy().astype(np.int32)}
    put(d, "out", out, full_limit=1 << 20)
    for n in ("style", "x1", "x2", "feat_global"):
        put(d, "stage|" + n, stages[n], full_limit=1 << 17)
    put(d, "logit", logit); d["loss"] = loss.detach().numpy()
    for (n, _), g in zip(G.named_parameters(), grads):
        put(d, "grad|" + n, g)
    for n, b in G.named_buffers():
        d["buf|" + n] = b.numpy()
    save("g4_generator.npz", d)


# ---------------------------------------------------------------- G6: losses
def g6():
    d = {}
    B = 6
    dr = fr.normal("g6.dreal", (B, 1)).requires_grad_(True)
    df = fr.normal("g6.dfake", (B, 1)).requires_grad_(True)
    d["d_real"] = dr.detach().numpy(); d["d_fake"] = df.detach().numpy()
    for gan in ("ls", "wgan", "hinge", "gan"):
        l, _ = LU.dis_loss(dr, df, gan=gan)
        gr, gf = torch.autograd.grad(l, [dr, df], allow_unused=True)
        d["dis|%s|loss" % gan] = l.detach().numpy()
        d["dis|%s|g_real" % gan] = gr.numpy(); d["dis|%s|g_fake" % gan] = gf.numpy()
        l, _ = LU.gen_loss(dr, df, gan=gan)
        gf, = torch.autograd.grad(l, [df])
        d["gen|%s|loss" % gan] = l.detach().numpy(); d["gen|%s|g_fake" % gan] = gf.numpy()
    # the [B,1] x [B] broadcast quirk with non-constant labels (noise_label=True path):
    np.random.seed(7)
    l, _ = LU.dis_loss(dr, df, gan="ls", noise_label=True)
    np.random.seed(7)
    rl = LU.noisy_labels(LU.smooth_labels(B, ran=[0.9, 1.0]), 0.05)        # same draw order as loss_utils.py:897-901
    d["dis|ls_noisy|real_label"] = rl.astype(np.float32)
    d["dis|ls_noisy|loss"] = l.detach().numpy()
    gr, gf = torch.autograd.grad(l, [dr, df])
    d["dis|ls_noisy|g_real"] = gr.numpy(); d["dis|ls_noisy|g_fake"] = gf.numpy()
    # batch 40: int(0.05*40) = 2 labels really flip (loss_utils.py:718-725), on both sides (D: :897-901, G: :753-755)
    B2 = 40
    dr2 = fr.normal("g6.dreal40", (B2, 1)).requires_grad_(True)
    df2 = fr.normal("g6.dfake40", (B2, 1)).requires_grad_(True)
    d["b40|d_real"] = dr2.detach().numpy(); d["b40|d_fake"] = df2.detach().numpy()
    np.random.seed(11)
    l, _ = LU.dis_loss(dr2, df2, gan="ls", noise_label=True)
    np.random.seed(11)
    rl = LU.noisy_labels(LU.smooth_labels(B2, ran=[0.9, 1.0]), 0.05)
    assert (rl < 0.5).sum() >= 1, "no label flipped"
    d["b40|dis|ls_noisy|real_label"] = rl.astype(np.float32)
    d["b40|dis|ls_noisy|loss"] = l.detach().numpy()
    gr, gf = torch.autograd.grad(l, [dr2, df2])
    d["b40|dis|ls_noisy|g_real"] = gr.numpy(); d["b40|dis|ls_noisy|g_fake"] = gf.numpy()
    np.random.seed(12)
    l, _ = LU.gen_loss(dr2, df2, gan="ls", noise_label=True)
    np.random.seed(12)
    fl = LU.noisy_labels(np.ones((B2,)), 0.05)
    assert (fl < 0.5).sum() >= 1
    d["b40|gen|ls_noisy|fake_label"] = fl.astype(np.float32)
    d["b40|gen|ls_noisy|loss"] = l.detach().numpy()
    gf, = torch.autograd.grad(l, [df2])
    d["b40|gen|ls_noisy|g_fake"] = gf.numpy()
    save("g6_losses.npz", d)


# ---------------------------------------------------------------- G7: WGAN-GP through the reference D
def g7():
    d = {}
    B, N = 3, 256
    _, D = make_gd(salt=7)
    real = fr.synthetic_real(B, N, seed=71).transpose(2, 1).contiguous()
    fake = (0.8 * fr.synthetic_real(B, N, seed=72) + 0.05 * fr.normal("g7.n", (B, N, 3))).transpose(2, 1).contiguous()
    alpha = fr.uniform("g7.alpha", (B, 1, 1), 0.0, 1.0)
    orig = torch.rand
    torch.rand = lambda *a, **k: alpha.clone().requires_grad_(k.get("requires_grad", False))
    try:
        gp = GradientPenalty(10.0, gamma=1)(D, real, fake)
    finally:
        torch.rand = orig
    grads = torch.autograd.grad(gp, list(D.parameters()), allow_unused=True)
    d["alpha"] = alpha.numpy(); d["gp"] = gp.detach().numpy()
    for (n, p), g in zip(D.named_parameters(), grads):
        put(d, "grad|" + n, g if g is not None else torch.zeros_like(p))
    # also the first-order input gradient (what the penalty is a function of)
    xh = (real + alpha * (fake - real)).requires_grad_(True)
    _, D2 = make_gd(salt=7)
    gin, = torch.autograd.grad(D2(xh).sum(), xh)
    put(d, "input_grad", gin, full_limit=1 << 20)
    save("g7_gradient_penalty.npz", d)


# ---------------------------------------------------------------- G8: one full D-step + G-step
def g8():
    for tag, gan, use_gp, B, N in (("ls", "ls", False, 4, 512), ("wgangp", "wgan", True, 4, 256)):
        d = {}
        G, D = make_gd(salt=8)
        optG = torch.optim.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.99))
        optD = torch.optim.Adam(D.parameters(), lr=1e-4, betas=(0.5, 0.99))
        x = fr.sphere_template(N)[None].repeat(B, 1, 1)
        real = fr.synthetic_real(B, N, seed=81)
        z_d, z_g = fr.latent(B, N, seed=82), fr.latent(B, N, seed=83)
        alpha = fr.uniform("g8.alpha", (B, 1, 1), 0.0, 1.0)

        def req(m, f):
            for p in m.parameters():
                p.requires_grad = f
        # D step (Generation/model.py:240-260)
        req(G, False); req(D, True); optD.zero_grad()
        fake = G(x, z_d).detach()
        real_t = real.transpose(2, 1).contiguous()
        lossD, _ = LU.dis_loss(D(real_t), D(fake), gan=gan)
        if use_gp:
            orig = torch.rand
            torch.rand = lambda *a, **k: alpha.clone().requires_grad_(k.get("requires_grad", False))
            try:
                lossD = lossD + GradientPenalty(10.0, gamma=1)(D, real_t, fake)
            finally:
                torch.rand = orig
        lossD.backward()
        for n, p in D.named_parameters():
            put(d, "dgrad|" + n, p.grad)
        optD.step()
        # G step (model.py:264-279)
        req(G, True); req(D, False); optG.zero_grad()
        g_fake = G(x, z_g)
        g_real_logit = D(real_t)
        lossG, _ = LU.gen_loss(g_real_logit, D(g_fake), gan=gan)
        lossG.backward()
        for n, p in G.named_parameters():
            put(d, "ggrad|" + n, p.grad)
        optG.step()
        d["lossD"] = lossD.detach().numpy(); d["lossG"] = lossG.detach().numpy(); d["alpha"] = alpha.numpy()
        put(d, "fake_d", fake, full_limit=1 << 20); put(d, "fake_g", g_fake, full_limit=1 << 20)
        for n, p in G.named_parameters():
            put(d, "gparam|" + n, p)
        for n, p in D.named_parameters():
            put(d, "dparam|" + n, p)
        for n, b in G.named_buffers():
            d["gbuf|" + n] = b.numpy()
        for n, b in D.named_buffers():
            d["dbuf|" + n] = b.numpy()
        save("g8_train_step_%s.npz" % tag, d)


# ---------------------------------------------------------------- G9: ball query / grouping family
def g9():
    d = {}
    B, N, S = 2, 256, 32
    xyz = fr.synthetic_real(B, N, seed=91)
    feat = fr.normal("g9.feat", (B, N, 5))
    new_xyz = xyz[:, ::N // S][:, :S].contiguous()
    d["square_distance"] = pnu.square_distance(new_xyz, xyz).numpy()
    for r, ns in ((0.3, 16), (0.15, 32), (0.02, 8)):
        d["query_ball|%g|%d" % (r, ns)] = pnu.query_ball_point(r, ns, xyz, new_xyz).numpy().astype(np.int32)
    idx = pnu.query_ball_point(0.3, 16, xyz, new_xyz)
    d["index_points3"] = pnu.index_points(feat, idx).numpy()
    d["index_points2"] = pnu.index_points(feat, idx[:, :, 0]).numpy()
    start = torch.tensor([3, 100])
    orig = torch.randint
    torch.randint = lambda *a, **k: start.clone()
    try:
        d["fps"] = pnu.farthest_point_sample(xyz, 24).numpy().astype(np.int32)
        nx, npts = pnu.sample_and_group(24, 0.3, 16, xyz, feat)
    finally:
        torch.randint = orig
    d["fps_start"] = start.numpy().astype(np.int32)
    d["sag|new_xyz"] = nx.numpy(); d["sag|new_points"] = npts.numpy()
    d["fps0"] = PCU.farthest_point_sample(xyz, 24).numpy().astype(np.int32)          # start index 0 variant
    knn = PCU.knn_point(10, xyz, xyz)
    d["knn_point_sorted"] = torch.sort(knn, dim=-1)[0].numpy().astype(np.int32)       # order unspecified -> compare as sets
    np_, gx = PCU.group(10, xyz, feat)
    # `group` inherits knn_point's unspecified order: store with the idx it used
    d["group|idx"] = knn.numpy().astype(np.int32)
    d["group|new_points"] = np_.numpy(); d["group|xyz_norm"] = gx.numpy()
    save("g9_ball_group.npz", d)


# ---------------------------------------------------------------- G10: eval-mode generation + interpolate
def g10():
    """model_test.py:63-64 calls G.eval() before generating: BatchNorm uses running statistics (advanced here by two
    train-mode forwards), InstanceNorm is unchanged.  Generator.forward, Generator.interpolate (both modes) and the
    Discriminator in eval mode on the generated clouds."""
    B, N = 2, 256
    G, D = make_gd(salt=10)
    x = fr.sphere_template(N)[None].repeat(B, 1, 1)
    with torch.no_grad():
        for s in (0, 1):
            D(G(x, fr.latent(B, N, seed=100 + s)))            # train mode: running statistics move away from (0, 1)
    G.eval(); D.eval()
    d = {}
    for n, b in G.named_buffers():
        d["gbuf|" + n] = b.numpy().copy()
    for n, b in D.named_buffers():
        d["dbuf|" + n] = b.numpy().copy()
    z1, z2 = fr.latent(B, N, seed=110), fr.latent(B, N, seed=111)
    selection = (fr.uniform("g10.sel", (N,), 0.0, 1.0) < 0.4).to(torch.int64)
    alpha = 0.3
    stage = {}
    hook = G.adain1.register_forward_hook(lambda m, i, o: stage.__setitem__("x1", o.detach().clone()))
    with torch.no_grad():
        for tag, fn in (("fwd", lambda: G(x, z1.clone())),
                        ("interp_z", lambda: G.interpolate(x, z1.clone(), z2.clone(), selection, alpha, use_latent=False)),
                        ("interp_style", lambda: G.interpolate(x, z1.clone(), z2.clone(), selection, alpha, use_latent=True))):
            out = fn()
            _, idx2 = get_edge_features(stage["x1"], 10, return_idx=True)
            put(d, tag + "|out", out, full_limit=1 << 20)
            put(d, tag + "|x1", stage["x1"], full_limit=1 << 17)
            d[tag + "|idx2"] = idx2.view(B, N, 10).numpy().astype(np.int32)
            put(d, tag + "|logit", D(out))
    hook.remove()
    d["selection"] = selection.numpy().astype(np.int32)
    d["alpha"] = np.float32(alpha)
    save("g10_eval_interpolate.npz", d)


# ---------------------------------------------------------------- G11: Chamfer-based evaluation metrics
def g11():
    """metrics/evaluation_metrics.py does not import here (its CUDA extensions are absent): distChamfer, lgan_mmd_cov and knn
    are compiled from the reference file with `ast` (they are plain torch) and the CD half of _pairwise_EMD_CD_ is replayed
    with the reference's distChamfer."""
    EM = extract_functions(os.path.join(REF, "metrics/evaluation_metrics.py"), ["distChamfer", "lgan_mmd_cov", "knn"],
                           {"torch": torch, "np": np})
    S, R, N = 6, 5, 128
    smp = torch.stack([fr.synthetic_real(1, N, seed=300 + i)[0] for i in range(S)])
    ref = torch.stack([fr.synthetic_real(1, N, seed=400 + i)[0] * (0.8 + 0.05 * i) for i in range(R)])
    d = {}
    dl, dr = EM.distChamfer(smp[:R].contiguous(), ref)
    put(d, "dl", dl, full_limit=1 << 20); put(d, "dr", dr, full_limit=1 << 20)

    def pairwise(a, b):
        rows = []
        for i in range(a.shape[0]):
            x, y = EM.distChamfer(a[i].view(1, -1, 3).expand(b.shape[0], -1, -1).contiguous(), b)
            rows.append((x.mean(dim=1) + y.mean(dim=1)).view(1, -1))
        return torch.cat(rows, dim=0)
    M_rs, M_rr, M_ss = pairwise(ref, smp), pairwise(ref, ref), pairwise(smp, smp)
    d["M_rs"], d["M_rr"], d["M_ss"] = M_rs.numpy(), M_rr.numpy(), M_ss.numpy()
    for k, v in EM.lgan_mmd_cov(M_rs.t()).items():
        d["mmdcov|" + k] = np.float32(v.item())
    for k, v in EM.knn(M_rr, M_rs, M_ss, 1, sqrt=False).items():
        if "acc" in k:
            d["1nn|" + k] = np.float32(v.item())
    save("g11_chamfer_metrics.npz", d)


# ---------------------------------------------------------------- G12: non-default flags
def g12():
    """--use_head (pc_head + 128-wide EdgeConv1), --off + --z_norm, --small_d: train-mode forward, and gradients for an
    injected upstream (as in G4)."""
    B, N = 4, 256                    # 4 shapes: BatchNorm1d over the batch in global_conv is too ill-conditioned at 2
    x = fr.sphere_template(N)[None].repeat(B, 1, 1)
    z = fr.latent(B, N, seed=120)
    d = {}

    class OH(Opts):
        use_head = True
    G = load_into(Generator(OH), fr.init_params(orc.generator_shapes(use_head=True), salt=20)).train()
    stage = {}
    hook = G.adain1.register_forward_hook(lambda m, i, o: stage.__setitem__("x1", o.detach().clone()))
    hook0 = G.pc_head.register_forward_hook(lambda m, i, o: stage.__setitem__("feat", o.detach().clone()))
    out = G(x, z)
    hook.remove(); hook0.remove()
    _, idx1 = get_edge_features(stage["feat"], 10, return_idx=True)
    _, idx2 = get_edge_features(stage["x1"], 10, return_idx=True)
    dy = fr.normal("g12.dy", out.shape)
    grads = torch.autograd.grad(out, list(G.parameters()), dy)
    put(d, "head|out", out, full_limit=1 << 20)
    d["head|idx1"] = idx1.view(B, N, 10).numpy().astype(np.int32); d["head|idx2"] = idx2.view(B, N, 10).numpy().astype(np.int32)
    for (n, _), g in zip(G.named_parameters(), grads):
        put(d, "head|grad|" + n, g)

    class OO(Opts):
        off = True; z_norm = True
    G = load_into(Generator(OO), fr.init_params(orc.generator_shapes(), salt=21)).train()
    stage = {}
    hook = G.adain1.register_forward_hook(lambda m, i, o: stage.__setitem__("x1", o.detach().clone()))
    out = G(x, z)
    hook.remove()
    _, idx2 = get_edge_features(stage["x1"], 10, return_idx=True)
    put(d, "off|out", out, full_limit=1 << 20)
    d["off|idx2"] = idx2.view(B, N, 10).numpy().astype(np.int32)

    class OS(Opts):
        small_d = True
    D = load_into(Discriminator(OS), fr.init_params(orc.discriminator_shapes(small_d=True), salt=22)).train()
    real = fr.synthetic_real(4, N, seed=23).transpose(2, 1).contiguous().requires_grad_(True)
    logit = D(real)
    loss = ((logit - 1.0) ** 2).mean()
    grads = torch.autograd.grad(loss, [real] + list(D.parameters()))
    put(d, "small|logit", logit); put(d, "small|dx", grads[0], full_limit=1 << 20)
    for (n, _), g in zip(D.named_parameters(), grads[1:]):
        put(d, "small|grad|" + n, g)
    save("g12_variants.npz", d)


def g13():
    """--attn (Attention(640) between the concat and the tail) and --eql (equalised-LR head / global_conv): train-mode forward
    and gradients for an injected upstream, as G12."""
    B, N = 4, 256
    x = fr.sphere_template(N)[None].repeat(B, 1, 1)
    z = fr.latent(B, N, seed=130)
    d = {}
    for tag, flags, salt in (("attn", dict(attn=True), 30), ("eql", dict(eql=True), 31), ("both", dict(attn=True, eql=True, use_head=True), 32)):
        O = type("O_" + tag, (Opts,), flags)
        G = Generator(O)
        shapes = orc.generator_shapes(use_head=flags.get("use_head", False), attn=flags.get("attn", False), eql=flags.get("eql", False))
        assert {k: tuple(v.shape) for k, v in G.named_parameters()} == {k: tuple(v) for k, v in shapes.items()}, tag
        load_into(G, fr.init_params(shapes, salt=salt)).train()
        stage = {}
        hooks = [G.adain1.register_forward_hook(lambda m, i, o: stage.__setitem__("x1", o.detach().clone()))]
        if flags.get("use_head"):
            hooks.append(G.pc_head.register_forward_hook(lambda m, i, o: stage.__setitem__("feat", o.detach().clone())))
        out = G(x, z)
        for h in hooks:
            h.remove()
        if flags.get("use_head"):
            _, idx1 = get_edge_features(stage["feat"], 10, return_idx=True)
            d[tag + "|idx1"] = idx1.view(B, N, 10).numpy().astype(np.int32)
        _, idx2 = get_edge_features(stage["x1"], 10, return_idx=True)
        d[tag + "|idx2"] = idx2.view(B, N, 10).numpy().astype(np.int32)
        dy = fr.normal("g13.dy." + tag, out.shape)
        grads = torch.autograd.grad(out, list(G.parameters()), dy)
        put(d, tag + "|out", out, full_limit=1 << 20)
        for (n, _), g in zip(G.named_parameters(), grads):
            put(d, tag + "|grad|" + n, g)
    save("g13_attn_eql.npz", d)


def g14():
    """JSD between occupancy grids: the reference functions of metrics/evaluation_metrics.py:210-322 (compiled with `ast`, like
    G11; they use sklearn's NearestNeighbors and scipy's entropy) on two small sets of clouds scaled into the radius-0.5 sphere."""
    import warnings
    from numpy.linalg import norm
    from scipy.stats import entropy
    from sklearn.neighbors import NearestNeighbors
    EM = extract_functions(os.path.join(REF, "metrics/evaluation_metrics.py"),
                           ["unit_cube_grid_point_cloud", "jsd_between_point_cloud_sets", "entropy_of_occupancy_grid",
                            "jensen_shannon_divergence", "_jsdiv"],
                           {"np": np, "norm": norm, "entropy": entropy, "NearestNeighbors": NearestNeighbors, "warnings": warnings})
    S, R, N = 12, 10, 512
    smp = np.stack([fr.synthetic_real(1, N, seed=500 + i)[0].numpy() * 0.5 for i in range(S)])
    ref = np.stack([fr.synthetic_real(1, N, seed=600 + i)[0].numpy() * (0.35 + 0.015 * i) for i in range(R)])
    d = {}
    for res in (16, 28):
        grid, spacing = EM.unit_cube_grid_point_cloud(res, True)
        d["grid%d" % res] = grid.astype(np.float32); d["spacing%d" % res] = np.float64(spacing)
        ent, cnt = EM.entropy_of_occupancy_grid(smp, res, True)
        d["ent%d" % res] = np.float64(ent); d["cnt%d" % res] = cnt.astype(np.int32)
        d["jsd%d" % res] = np.float64(EM.jsd_between_point_cloud_sets(smp, ref, res))
    full, _ = EM.unit_cube_grid_point_cloud(6, False)
    d["grid6_full"] = full.astype(np.float32)
    save("g14_jsd.npz", d)


def extract_methods(path, cls, names, extra_globals):
    """Compile selected methods of a reference class into a bare stand-in class (the class body around them -- trainer set-up,
    CUDA calls -- does not run here)."""
    tree = ast.parse(open(path).read())
    cdef = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls][0]
    body = [n for n in cdef.body if isinstance(n, ast.FunctionDef) and n.name in names]
    mod = ast.Module(body=[ast.ClassDef(name=cls, bases=[], keywords=[], body=body, decorator_list=[])], type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = dict(extra_globals)
    exec(compile(mod, path, "exec"), ns)
    return ns[cls]


def g15():
    """The deterministic halves of the input samplers (Generation/model.py:46-52, 122-180): pc_normalize, the normalised sphere
    prior, the `ball_dist` region ordering, and which points a region-mixed latent covers for a given (centre, size)."""
    import random
    MP = os.path.join(REF, "Generation/model.py")
    PN = extract_functions(MP, ["pc_normalize"], dict(np=np))
    d = {}
    pc = (fr.normal("g15.pc", (300, 3)) * 2.5 + 4.0).double().numpy()
    d["pc_normalize|in"] = pc
    d["pc_normalize|out"] = PN.pc_normalize(pc.copy())
    Model = extract_methods(MP, "Model", ["noise_generator", "sphere_generator"],
                            dict(np=np, random=random, torch=torch, Variable=Variable, pc_normalize=PN.pc_normalize))
    cwd = os.getcwd()
    os.chdir(REF)                                                   # 'template/balls/%d.xyz' is a relative path (model.py:159)
    try:
        for n_pts in (256, 2048):
            m = Model()
            m.opts = types.SimpleNamespace(np=n_pts, nz=128, nv=0.2, n_rand=False, n_mix=True)
            m.ball = None
            ball = m.sphere_generator(bs=2, static=True)
            assert torch.equal(ball[0], ball[1])
            d["N%d|ball" % n_pts] = ball[0].numpy()                  # fp32, as the Generator receives it
            ids = [0, 17, n_pts // 2, n_pts - 1]
            d["N%d|order_ids" % n_pts] = np.array(ids)
            d["N%d|order" % n_pts] = np.stack([np.argsort(m.ball_dist[i])[::1] for i in ids]).astype(np.int32)
            # region mixing: record the (centre, size) the reference draws and the points that received the second latent
            bs = 6
            rec = {"ids": [], "u": []}
            real_randint, real_random = np.random.randint, random.random

            def randint(*a, **k):
                v = real_randint(*a, **k); rec["ids"].append(int(v)); return v

            def rnd():
                v = real_random(); rec["u"].append(v); return v
            np.random.seed(100 + n_pts); random.seed(3)                # seed 3: the coin flip (first random.random()) is < 0.5
            np.random.randint, random.random = randint, rnd
            try:
                noise = m.noise_generator(bs=bs)
            finally:
                np.random.randint, random.random = real_randint, real_random
            assert rec["u"][0] < 0.5 and len(rec["ids"]) == bs and len(rec["u"]) == bs + 1
            noise = noise.numpy()                                      # [bs, np, nz]
            masks = np.zeros((bs, n_pts), dtype=np.uint8)
            for i in range(bs):
                first = noise[i, np.argsort(m.ball_dist[rec["ids"][i]])[0]]      # the region's latent (its closest point is always inside)
                masks[i] = (noise[i] == first).all(axis=1)
            nums = np.array([int(max(u, 0.1) * n_pts) for u in rec["u"][1:]])
            assert (masks.sum(1) == nums).all(), (masks.sum(1), nums)
            d["N%d|mix_ids" % n_pts] = np.array(rec["ids"]); d["N%d|mix_num" % n_pts] = nums
            d["N%d|mix_mask" % n_pts] = np.packbits(masks, axis=1)
    finally:
        os.chdir(cwd)
    save("g15_samplers.npz", d)


def g16():
    """The per-item data path of Generation/H5DataLoader.py:107,113-118 over Generation/point_operation.py:144-163,207-235,292-307:
    set normalisation, row shuffle, rotation about the up axis, random scale -- run with numpy's global RNG seeded, with the
    random draws it consumed recorded next to the outputs (the draws are inputs of the deterministic transforms)."""
    PO = extract_functions(os.path.join(REF, "Generation/point_operation.py"),
                           ["normalize_point_cloud", "rotate_point_cloud_and_gt", "random_scale_point_cloud_and_gt"], dict(np=np))
    d = {}
    raw = (fr.normal("g16.raw", (5, 300, 3)) * fr.uniform("g16.s", (5, 1, 3), 0.5, 3.0) + fr.normal("g16.c", (5, 1, 3)) * 2.0).numpy().astype(np.float32)
    d["raw"] = raw
    data = 0.9 * PO.normalize_point_cloud(raw)                         # H5DataLoader.py:107 with opts.scale = 0.9
    d["normalized"] = data.astype(np.float32)
    raw6 = np.concatenate([raw[:2], fr.normal("g16.nor", (2, 300, 3)).numpy()], axis=-1)
    d["raw6"] = raw6; d["normalized6"] = PO.normalize_point_cloud(raw6).astype(np.float32)
    num_points = 256
    items, perms, angles, scales = [], [], [], []
    for index in range(5):
        np.random.seed(1000 + index)
        point_set = data[index][:num_points, :3].copy()               # __getitem__, augment=True
        np.random.shuffle(point_set)
        point_set = PO.rotate_point_cloud_and_gt(point_set)
        point_set = PO.random_scale_point_cloud_and_gt(point_set)
        items.append(point_set.astype(np.float32))
        np.random.seed(1000 + index)                                  # replay the same stream to learn the draws
        perm = np.arange(num_points); np.random.shuffle(perm)
        ang = np.random.uniform(size=(3)) * 2 * np.pi
        sc = np.random.uniform(0.8, 1.25, 1)
        perms.append(perm); angles.append(ang[1]); scales.append(sc[0])
        assert np.array_equal(data[index][:num_points][perm], (lambda a: (np.random.seed(1000 + index), np.random.shuffle(a), a)[2])(data[index][:num_points].copy()))
    d["items"] = np.stack(items); d["perm"] = np.stack(perms).astype(np.int64)
    d["angle_y"] = np.array(angles, dtype=np.float64); d["scale"] = np.array(scales, dtype=np.float64)
    save("g16_data_path.npz", d)

# ---------------------------------------------------------------- G17: the BENCHMARKED sizes (BASELINE configs[1] and configs[3] per GPU)
def _g17_cfgs():
    return (("c2", 32, 2048), ("c4", 16, 4096))


def g17():
    """Round-2 review item 1: numeric pins at the sizes bench.py times -- C2 (B=32, N=2048) and C4's per-GPU shape (B=16, N=4096).
    (a) Discriminator.forward / backward and the gradient penalty (Discriminator.py:97-115, gradient_penalty.py:19-37): logits in
        full, summaries (l2, sum, strided samples) of the input gradient and every parameter gradient, BatchNorm buffers in full;
    (b) Generator.forward (Generator.py:160-198) with the reference's own EdgeConv2 graph stored as int16 (the tie-aware protocol:
        the graph is injected into the build under test), stage summaries x1 / x2 / out and parameter gradients for an injected
        upstream;
    (c) at C2 one full WGAN-GP D-step + G-step (model.py:239-279 with the build contract of SURVEY 8(a)7), both EdgeConv2 graphs
        stored, losses, generated clouds, gradients, post-Adam parameters and buffers as summaries."""
    for tag, B, N in _g17_cfgs():
        class O(Opts):
            np = N
        d = {}
        # ---- (a) D
        D = load_into(Discriminator(O, num_point=N), fr.init_params(orc.discriminator_shapes(), salt=17)).train()
        real = fr.synthetic_real(B, N, seed=171).transpose(2, 1).contiguous().requires_grad_(True)
        logit = D(real)
        loss = ((logit - 1.0) ** 2).mean()
        grads = torch.autograd.grad(loss, [real] + list(D.parameters()))
        d["d|logit"] = logit.detach().numpy()
        put(d, "d|dx", grads[0], nsamp=4096)
        for (n, _), g in zip(D.named_parameters(), grads[1:]):
            put(d, "d|grad|" + n, g)
        for n, b in D.named_buffers():
            d["d|buf|" + n] = b.numpy().copy()
        # The same pass in float64: 1024 x B arg-max choices over N points sit between the logits and the lower layers' gradients,
        # and a near-tie that float32 and float64 resolve differently moves those gradients discretely (observed at C4: the
        # reference's own float32 gradients are 8e-4 off its float64 ones below the pool, 5e-7 above it).  A build may land on
        # either side of such a tie: the tests accept agreement with the float32 OR the float64 reference at block tolerance.
        D64 = load_into(Discriminator(O, num_point=N), fr.init_params(orc.discriminator_shapes(), salt=17)).train().double()
        real64 = real.detach().double().requires_grad_(True)
        grads = torch.autograd.grad(((D64(real64) - 1.0) ** 2).mean(), [real64] + list(D64.parameters()))
        put(d, "d|dx64", grads[0].float(), nsamp=4096)
        for (n, _), g in zip(D64.named_parameters(), grads[1:]):
            put(d, "d|grad64|" + n, g.float())
        print(tag, "D done", flush=True)
        # ---- (a) gradient penalty on a fresh D (same weights, untouched running statistics)
        D = load_into(Discriminator(O, num_point=N), fr.init_params(orc.discriminator_shapes(), salt=17)).train()
        realc = real.detach()
        fake = (0.8 * fr.synthetic_real(B, N, seed=172) + 0.05 * fr.normal("g17.n.%s" % tag, (B, N, 3))).transpose(2, 1).contiguous()
        alpha = fr.uniform("g17.alpha", (B, 1, 1), 0.0, 1.0)
        orig = torch.rand
        torch.rand = lambda *a, **k: alpha.clone().requires_grad_(k.get("requires_grad", False))
        try:
            gp = GradientPenalty(10.0, gamma=1)(D, realc, fake)
        finally:
            torch.rand = orig
        grads = torch.autograd.grad(gp, list(D.parameters()), allow_unused=True)
        d["gp|alpha"] = alpha.numpy(); d["gp|value"] = gp.detach().numpy()
        for (n, p), g in zip(D.named_parameters(), grads):
            put(d, "gp|grad|" + n, g if g is not None else torch.zeros_like(p))
        D64 = load_into(Discriminator(O, num_point=N), fr.init_params(orc.discriminator_shapes(), salt=17)).train().double()
        alpha64 = alpha.double()
        torch.rand = lambda *a, **k: alpha64.clone().requires_grad_(k.get("requires_grad", False))
        torch.set_default_dtype(torch.float64)                  # gradient_penalty.py:32 builds its seed with torch.ones(...)
        try:
            gp64 = GradientPenalty(10.0, gamma=1)(D64, realc.double(), fake.double())
        finally:
            torch.rand = orig
            torch.set_default_dtype(torch.float32)
        grads = torch.autograd.grad(gp64, list(D64.parameters()), allow_unused=True)
        d["gp|value64"] = gp64.detach().numpy()
        for (n, p), g in zip(D64.named_parameters(), grads):
            put(d, "gp|grad64|" + n, (g if g is not None else torch.zeros_like(p)).float())
        print(tag, "GP done", flush=True)
        # ---- (b) G with its own graphs recorded
        G = load_into(Generator(O), fr.init_params(orc.generator_shapes(), salt=17)).train()
        x = fr.sphere_template(N)[None].repeat(B, 1, 1)
        z = fr.latent(B, N, seed=173)
        stages = {}
        hooks = [G.adain1.register_forward_hook(lambda m, i, o: stages.__setitem__("x1", o.detach().clone())),
                 G.adain2.register_forward_hook(lambda m, i, o: stages.__setitem__("x2", o.detach().clone()))]
        out = G(x, z)
        for h in hooks:
            h.remove()
        _, idx2 = get_edge_features(stages["x1"], 10, return_idx=True)
        assert N <= 32767
        d["g|idx2"] = idx2.view(B, N, 10).numpy().astype(np.int16)
        put(d, "g|x1", stages["x1"], nsamp=4096); put(d, "g|x2", stages["x2"], nsamp=4096); put(d, "g|out", out, nsamp=8192)
        dy = fr.normal("g17.dy.%s" % tag, out.shape)
        grads = torch.autograd.grad(out, list(G.parameters()), dy)
        for (n, _), g in zip(G.named_parameters(), grads):
            put(d, "g|grad|" + n, g)
        for n, b in G.named_buffers():
            d["g|buf|" + n] = b.numpy().copy()
        # The same pass in float64 on the SAME EdgeConv2 graph (injected into the reference through its module-level
        # get_edge_features): global_conv's BatchNorm1d normalises over the 32 (16) shapes of the batch, whose global features are
        # nearly equal -- the reference's own float32 gradients are only good to ~2e-3 there (measured against this pass).  The
        # tests bound the build's error against the float64 result by the reference's own float32 error.
        import Generation.Generator as GG
        G64 = load_into(Generator(O), fr.init_params(orc.generator_shapes(), salt=17)).train().double()
        calls = [0]
        orig_gef = GG.get_edge_features

        def injected(x_, k_, num=-1, idx=None, return_idx=False):
            calls[0] += 1
            return orig_gef(x_, k_, num, idx2 if calls[0] == 2 else idx, return_idx)
        GG.get_edge_features = injected
        try:
            out64 = G64(x.double(), z.double())
        finally:
            GG.get_edge_features = orig_gef
        assert calls[0] == 2
        grads = torch.autograd.grad(out64, list(G64.parameters()), dy.double())
        put(d, "g|out64", out64.float(), nsamp=8192)
        for (n, _), g in zip(G64.named_parameters(), grads):
            put(d, "g|grad64|" + n, g.float())
        # How much the REFERENCE's own output moves when near-tied kNN rows are resolved the other way: for the n rows with the
        # smallest relative gap between their k-th and (k+1)-th neighbour distance, the k-th neighbour is replaced by the (k+1)-th
        # and the generator is run again on that graph.  (One exactly tied row moves the cloud by 8e-4: global_conv's BatchNorm1d
        # over B nearly equal global features amplifies it.)  The own-graph tests bound the build's deviation by this table.
        with torch.no_grad():
            dist = orc.pairwise_sqdist(stages["x1"])
            srt, order = torch.sort(dist, dim=2)
            gap = ((srt[:, :, 11] - srt[:, :, 10]) / srt[:, :, 10]).reshape(-1)
            ns, rels, gaps = [1, 5, 20, 50, 100, 200], [], []
            for nflip in ns:
                rows = torch.topk(-gap, nflip)[1]
                alt = order[:, :, 1:11].clone().reshape(B * N, 10)
                for r in rows.tolist():
                    alt[r, 9] = order[r // N, r % N, 11]
                alt = alt.view(B, N * 10)
                calls = [0]

                def flipped(x_, k_, num=-1, idx=None, return_idx=False):
                    calls[0] += 1
                    return orig_gef(x_, k_, num, alt if calls[0] == 2 else idx, return_idx)
                GG.get_edge_features = flipped
                try:
                    out2 = G(x, z)
                finally:
                    GG.get_edge_features = orig_gef
                rels.append(((out2 - out).norm() / out.norm()).item()); gaps.append(gap[rows].max().item())
            del dist, srt, order
        d["g|tie_nflip"] = np.array(ns); d["g|tie_out_rel"] = np.array(rels); d["g|tie_gap"] = np.array(gaps)
        print(tag, "G done; tie sensitivity", list(zip(ns, rels)), flush=True)
        save("g17_fullsize_%s.npz" % tag, d)
    # ---- (c) the benchmarked train step
    B, N = 32, 2048
    d = {}
    G, D = make_gd(salt=18)
    optG = torch.optim.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.99))
    optD = torch.optim.Adam(D.parameters(), lr=1e-4, betas=(0.5, 0.99))
    x = fr.sphere_template(N)[None].repeat(B, 1, 1)
    real = fr.synthetic_real(B, N, seed=181)
    z_d, z_g = fr.latent(B, N, seed=182), fr.latent(B, N, seed=183)
    alpha = fr.uniform("g17.step.alpha", (B, 1, 1), 0.0, 1.0)
    stages = {}
    hook = G.adain1.register_forward_hook(lambda m, i, o: stages.__setitem__("x1", o.detach().clone()))

    def req(m, f):
        for p in m.parameters():
            p.requires_grad = f
    req(G, False); req(D, True); optD.zero_grad()
    fake = G(x, z_d).detach()
    _, idx2 = get_edge_features(stages["x1"], 10, return_idx=True)
    d["idx2_d"] = idx2.view(B, N, 10).numpy().astype(np.int16)
    real_t = real.transpose(2, 1).contiguous()
    d_real, d_fake = D(real_t), D(fake)
    lossD, _ = LU.dis_loss(d_real, d_fake, gan="wgan")
    d["d_real"] = d_real.detach().numpy(); d["d_fake"] = d_fake.detach().numpy()
    orig = torch.rand
    torch.rand = lambda *a, **k: alpha.clone().requires_grad_(k.get("requires_grad", False))
    try:
        gpv = GradientPenalty(10.0, gamma=1)(D, real_t, fake)
    finally:
        torch.rand = orig
    d["gp"] = gpv.detach().numpy()
    lossD = lossD + gpv
    lossD.backward()
    for n, p in D.named_parameters():
        put(d, "dgrad|" + n, p.grad)
    optD.step()
    print("step: D done", flush=True)
    req(G, True); req(D, False); optG.zero_grad()
    g_fake = G(x, z_g)
    _, idx2 = get_edge_features(stages["x1"], 10, return_idx=True)
    d["idx2_g"] = idx2.view(B, N, 10).numpy().astype(np.int16)
    hook.remove()
    g_real_logit = D(real_t)
    d_gfake = D(g_fake)
    lossG, _ = LU.gen_loss(g_real_logit, d_gfake, gan="wgan")
    d["d_gfake"] = d_gfake.detach().numpy()
    lossG.backward()
    for n, p in G.named_parameters():
        put(d, "ggrad|" + n, p.grad)
    optG.step()
    d["lossD"] = lossD.detach().numpy(); d["lossG"] = lossG.detach().numpy(); d["alpha"] = alpha.numpy()
    put(d, "fake_d", fake, nsamp=8192); put(d, "fake_g", g_fake, nsamp=8192)
    for n, p in G.named_parameters():
        put(d, "gparam|" + n, p)
    for n, p in D.named_parameters():
        put(d, "dparam|" + n, p)
    for n, b in G.named_buffers():
        d["gbuf|" + n] = b.numpy().copy()
    for n, b in D.named_buffers():
        d["dbuf|" + n] = b.numpy().copy()
    save("g17_step_c2.npz", d)


# ---------------------------------------------------------------- G18: the reference's OWN noise on the benchmarked step
ZERO_GRAD_BIASES = ("EdgeConv1.conv_w.0.bias", "EdgeConv1.conv_w.3.bias", "EdgeConv1.conv_x.0.bias", "EdgeConv2.conv_w.0.bias",
                    "EdgeConv2.conv_w.3.bias", "EdgeConv2.conv_x.0.bias", "global_conv.0.bias", "global_conv.3.bias",
                    "mlps.0.bias", "mlps.3.bias", "mlps.6.bias", "fc2.0.bias")


def _ref_wgangp_step(dtype, graphs=None):
    """One WGAN-GP D-step + G-step of the imported reference at C2 (the run g17 (c) stores), in `dtype`, optionally with both
    EdgeConv2 graphs injected.  Returns the two graphs, the stage tensors in front of them, the clouds and every gradient."""
    import Generation.Generator as GG
    B, N = 32, 2048
    G, D = make_gd(salt=18)
    G, D = G.to(dtype), D.to(dtype)
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).to(dtype)
    real = fr.synthetic_real(B, N, seed=181).to(dtype)
    z_d, z_g = fr.latent(B, N, seed=182).to(dtype), fr.latent(B, N, seed=183).to(dtype)
    alpha = fr.uniform("g17.step.alpha", (B, 1, 1), 0.0, 1.0).to(dtype)
    stages, rec = {}, {}
    hook = G.adain1.register_forward_hook(lambda m, i, o: stages.__setitem__("x1", o.detach().clone()))
    orig_gef, calls, inject = GG.get_edge_features, [0], [None]

    def patched(x_, k_, num=-1, idx=None, return_idx=False):
        calls[0] += 1
        return orig_gef(x_, k_, num, inject[0] if (calls[0] % 2 == 0 and inject[0] is not None) else idx, return_idx)

    def req(m, f):
        for p in m.parameters():
            p.requires_grad = f
    GG.get_edge_features = patched
    orig_rand = torch.rand
    torch.rand = lambda *a, **k: alpha.clone().requires_grad_(k.get("requires_grad", False))
    torch.set_default_dtype(dtype)                              # gradient_penalty.py:32 builds its seed with torch.ones(...)
    try:
        optD = torch.optim.Adam(D.parameters(), lr=1e-4, betas=(0.5, 0.99))
        req(G, False); req(D, True); optD.zero_grad()
        inject[0] = None if graphs is None else graphs[0]
        fake = G(x, z_d).detach()
        rec["x1_d"] = stages["x1"]
        rec["idx2_d"] = orig_gef(stages["x1"], 10, return_idx=True)[1] if graphs is None else graphs[0]
        real_t = real.transpose(2, 1).contiguous()
        d_real, d_fake = D(real_t), D(fake)
        lossD, _ = LU.dis_loss(d_real, d_fake, gan="wgan")
        lossD = lossD + GradientPenalty(10.0, gamma=1)(D, real_t, fake)
        lossD.backward()
        rec["dgrad"] = {n: p.grad.detach().clone() for n, p in D.named_parameters()}
        optD.step()
        req(G, True); req(D, False)
        inject[0] = None if graphs is None else graphs[1]
        g_fake = G(x, z_g)
        rec["x1_g"] = stages["x1"]
        rec["idx2_g"] = orig_gef(stages["x1"], 10, return_idx=True)[1] if graphs is None else graphs[1]
        D(real_t)
        lossG, _ = LU.gen_loss(None, D(g_fake), gan="wgan")
        lossG.backward()
        rec["ggrad"] = {n: p.grad.detach().clone() for n, p in G.named_parameters()}
        rec["fake_d"], rec["fake_g"] = fake, g_fake.detach()
    finally:
        GG.get_edge_features = orig_gef
        torch.rand = orig_rand
        torch.set_default_dtype(torch.float32)
        hook.remove()
    return rec


def _rel(a, b):
    return float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-300))


def _whole(grads, skip):
    return torch.cat([g.double().reshape(-1) for n, g in grads.items() if not n.endswith(skip)])


def g18():
    """Round-3 review item 3(a): bounds for the own-graph step tests DERIVED from the reference instead of asserted.
    (a) the benchmarked C2 step (g17 (c)) once more in float64 on the float32 run's two EdgeConv2 graphs: what the reference's own
        float32 rounding does to every D and G gradient of the step with the discrete kNN choice held fixed;
    (b) the float32 step with the n most nearly tied kNN rows of BOTH graphs resolved the other way (n = 1 .. 100): how far every D
        gradient, the whole G gradient (cosine, norm ratio) and the clouds move when the reference itself lands on the other side of
        n ties -- the build's own graphs differ from the reference's in 15-20 such rows."""
    B, N = 32, 2048
    d = {}
    base = _ref_wgangp_step(torch.float32)
    ref = np.load(os.path.join(HERE, "g17_step_c2.npz"))
    assert np.array_equal(ref["idx2_d"], base["idx2_d"].view(B, N, 10).numpy().astype(np.int16)), "not the run g17 (c) stored"
    graphs = (base["idx2_d"], base["idx2_g"])
    print("g18: float32 step done", flush=True)
    r64 = _ref_wgangp_step(torch.float64, graphs=graphs)
    for n, g in r64["dgrad"].items():
        put(d, "dgrad64|" + n, g.float())
        d["noise32|dgrad|" + n] = np.float64(_rel(base["dgrad"][n], g))
    for n, g in r64["ggrad"].items():
        put(d, "ggrad64|" + n, g.float())
        d["noise32|ggrad|" + n] = np.float64(_rel(base["ggrad"][n], g))
    put(d, "fake_d64", r64["fake_d"].float(), nsamp=8192); put(d, "fake_g64", r64["fake_g"].float(), nsamp=8192)
    a, b = _whole(base["ggrad"], ZERO_GRAD_BIASES), _whole(r64["ggrad"], ZERO_GRAD_BIASES)
    d["noise32|ggrad_cos"] = np.float64(a @ b / (a.norm() * b.norm())); d["noise32|ggrad_ratio"] = np.float64(a.norm() / b.norm())
    print("g18: float64 step done; whole-G cosine %.7f ratio %.5f; worst D tensor %.2e" % (
        d["noise32|ggrad_cos"], d["noise32|ggrad_ratio"], max(float(v) for k, v in d.items() if k.startswith("noise32|dgrad|"))), flush=True)
    del r64
    # (b) tie flips in both graphs
    ns = [1, 5, 20, 50, 100]
    alts = {}
    for which in ("d", "g"):
        with torch.no_grad():
            dist = orc.pairwise_sqdist(base["x1_" + which])
            srt, order = torch.sort(dist, dim=2)
            gap = ((srt[:, :, 11] - srt[:, :, 10]) / srt[:, :, 10]).reshape(-1)
            alts[which] = (gap, order[:, :, 1:12].clone())
            del dist, srt, order
    tab = {k: [] for k in ("ggrad_cos", "ggrad_ratio", "fake_d", "fake_g", "gap")}
    dtab = {n: [] for n in base["dgrad"]}
    gtab = {n: [] for n in base["ggrad"]}
    for nflip in ns:
        gs = []
        for which in ("d", "g"):
            gap, order = alts[which]
            rows = torch.topk(-gap, nflip)[1]
            alt = order[:, :, :10].clone().reshape(B * N, 10)
            for r in rows.tolist():
                alt[r, 9] = order[r // N, r % N, 10]
            gs.append(alt.view(B, N * 10))
            tab["gap"].append(gap[rows].max().item())
        fl = _ref_wgangp_step(torch.float32, graphs=tuple(gs))
        for n in dtab:
            dtab[n].append(_rel(fl["dgrad"][n], base["dgrad"][n]))
        for n in gtab:
            gtab[n].append(_rel(fl["ggrad"][n], base["ggrad"][n]))
        a, b = _whole(fl["ggrad"], ZERO_GRAD_BIASES), _whole(base["ggrad"], ZERO_GRAD_BIASES)
        tab["ggrad_cos"].append(float(a @ b / (a.norm() * b.norm()))); tab["ggrad_ratio"].append(float(a.norm() / b.norm()))
        tab["fake_d"].append(_rel(fl["fake_d"], base["fake_d"])); tab["fake_g"].append(_rel(fl["fake_g"], base["fake_g"]))
        print("g18: %3d flips per graph: cloud %.2e / %.2e, whole-G cosine %.5f ratio %.4f, worst D tensor %.2e" % (
            nflip, tab["fake_d"][-1], tab["fake_g"][-1], tab["ggrad_cos"][-1], tab["ggrad_ratio"][-1], max(v[-1] for v in dtab.values())), flush=True)
    d["tie|nflip"] = np.array(ns)
    for k, v in tab.items():
        d["tie|" + k] = np.array(v, dtype=np.float64)
    for n, v in dtab.items():
        d["tie|dgrad|" + n] = np.array(v, dtype=np.float64)
    for n, v in gtab.items():
        d["tie|ggrad|" + n] = np.array(v, dtype=np.float64)
    save("g18_step_noise_c2.npz", d)


# ---------------------------------------------------------------- G19: THREE consecutive reference steps (state carry)
def _g19_inputs(tag, B, N, k):
    """Inputs of step k (fresh real batch, fresh latents, fresh mixing factors: every step sees other data, as in model.py:239-279)."""
    return (fr.synthetic_real(B, N, seed=1900 + 10 * k), fr.latent(B, N, seed=1901 + 10 * k), fr.latent(B, N, seed=1902 + 10 * k),
            fr.uniform("g19.%s.alpha.%d" % (tag, k), (B, 1, 1), 0.0, 1.0))


class _NP64:
    """numpy with float32 -> float64: the reference's LS losses build their labels with `.astype(np.float32)` (loss_utils.py:902-903),
    which a float64 run of the same functions cannot take."""
    def __getattr__(self, k):
        return np.float64 if k == "float32" else getattr(np, k)


LU64 = extract_functions(os.path.join(REF, "Common/loss_utils.py"),
                         ["dis_loss", "gen_loss", "BCEloss", "BCEfakeloss", "smooth_labels", "noisy_labels"],
                         dict(torch=torch, nn=nn, F=F, np=_NP64(), functools=functools, Variable=Variable))


def _ref_three_steps(tag, gan, use_gp, B, N, salt, dtype, graphs=None):
    """Three consecutive iterations of the imported reference in `dtype`; graphs = per step (idx2_d, idx2_g) to inject into EdgeConv2
    (None: built, and returned).  Returns per-step records and the final state."""
    import Generation.Generator as GG
    lu = LU64 if dtype == torch.float64 else LU

    class O(Opts):
        np = N
    G = load_into(Generator(O), fr.init_params(orc.generator_shapes(), salt=salt)).train().to(dtype)
    D = load_into(Discriminator(O, num_point=N), fr.init_params(orc.discriminator_shapes(), salt=salt)).train().to(dtype)
    init = {"g": {n: p.detach().clone() for n, p in G.named_parameters()}, "d": {n: p.detach().clone() for n, p in D.named_parameters()}}
    optG = torch.optim.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.99))
    optD = torch.optim.Adam(D.parameters(), lr=1e-4, betas=(0.5, 0.99))
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).to(dtype)
    stages = {}
    hook = G.adain1.register_forward_hook(lambda m, i, o: stages.__setitem__("x1", o.detach().clone()))
    orig_gef, calls, inject = GG.get_edge_features, [0], [None]

    def patched(x_, k_, num=-1, idx=None, return_idx=False):
        calls[0] += 1
        return orig_gef(x_, k_, num, inject[0] if (calls[0] % 2 == 0 and inject[0] is not None) else idx, return_idx)

    def req(m, f):
        for p in m.parameters():
            p.requires_grad = f
    GG.get_edge_features = patched
    orig_rand = torch.rand
    torch.set_default_dtype(dtype)                              # gradient_penalty.py:32 builds its seed with torch.ones(...)
    steps = []
    try:
        for k in range(3):
            real, z_d, z_g, alpha = (t.to(dtype) for t in _g19_inputs(tag, B, N, k))
            rec = {}
            # D step (model.py:240-260)
            req(G, False); req(D, True); optD.zero_grad()
            inject[0] = None if graphs is None else graphs[k][0]
            fake = G(x, z_d).detach()
            rec["idx2_d"] = orig_gef(stages["x1"], 10, return_idx=True)[1] if graphs is None else graphs[k][0]
            real_t = real.transpose(2, 1).contiguous()
            lossD, _ = lu.dis_loss(D(real_t), D(fake), gan=gan)
            if use_gp:
                torch.rand = lambda *a, **kw: alpha.clone().requires_grad_(kw.get("requires_grad", False))
                try:
                    gpv = GradientPenalty(10.0, gamma=1)(D, real_t, fake)
                finally:
                    torch.rand = orig_rand
                rec["gp"] = gpv.detach()
                lossD = lossD + gpv
            lossD.backward()
            rec["dgrad"] = {n: p.grad.detach().clone() for n, p in D.named_parameters()}
            optD.step()
            # G step (model.py:264-279)
            req(G, True); req(D, False); optG.zero_grad()
            inject[0] = None if graphs is None else graphs[k][1]
            g_fake = G(x, z_g)
            rec["idx2_g"] = orig_gef(stages["x1"], 10, return_idx=True)[1] if graphs is None else graphs[k][1]
            g_real_logit = D(real_t)
            lossG, _ = lu.gen_loss(g_real_logit, D(g_fake), gan=gan)
            lossG.backward()
            rec["ggrad"] = {n: p.grad.detach().clone() for n, p in G.named_parameters()}
            optG.step()
            rec.update(lossD=lossD.detach(), lossG=lossG.detach(), alpha=alpha, fake_g=g_fake.detach())
            steps.append(rec)
            print("g19 %s %s: step %d lossD %.6f lossG %.6f" % (tag, str(dtype)[6:], k, float(lossD.detach()), float(lossG.detach())), flush=True)
    finally:
        GG.get_edge_features = orig_gef
        torch.rand = orig_rand
        torch.set_default_dtype(torch.float32)
        hook.remove()
    final = {}
    for kind, net, opt in (("d", D, optD), ("g", G, optG)):
        final[kind] = dict(param={n: p.detach() for n, p in net.named_parameters()},
                           upd={n: p.detach() - init[kind][n] for n, p in net.named_parameters()},
                           m={n: opt.state[p]["exp_avg"] for n, p in net.named_parameters()},
                           v={n: opt.state[p]["exp_avg_sq"] for n, p in net.named_parameters()},
                           buf={n: b.detach().clone() for n, b in net.named_buffers()})
        assert all(int(opt.state[p]["step"]) == 3 for p in net.parameters())
    return steps, final


def g19():
    """Round-4 review item 2(b): state carry across steps against the REFERENCE.  Three consecutive D-step + G-step iterations of the
    imported reference (model.py:239-279 with torch.optim.Adam(lr=1e-4, betas=(0.5, 0.99)), model.py:94-97) at C1 (B=4, N=512, LS)
    and at C2 (B=32, N=2048, WGAN-GP), from fixture weights, fresh inputs every step.  Stored per step: both EdgeConv2 graphs (the
    GPU test injects them), losses, every D / G gradient (summaries); after step 3: parameters, the UPDATE param - init (what Adam
    did -- the parameters themselves move by <= 3e-4 and pin nothing), Adam's exp_avg / exp_avg_sq, every BatchNorm buffer incl.
    num_batches_tracked (G's 8 layers are advanced by both G forwards of a step, D's 4 by all four / five D forwards).

    A multi-step trajectory is NOT reproducible to rounding by anybody: Adam's first updates are +-lr whatever the gradient's size, so
    every element whose gradient is rounding noise moves with a sign that depends on the summation order, and D's kinks amplify the
    difference from step to step.  The bounds of the tests are therefore DERIVED, as for G18: the same three steps are run once more in
    float64 on the float32 run's graphs, and `noise|...` holds how far the reference's own float32 trajectory is from it (per step:
    losses, cloud, every gradient tensor, each gradient's rms element difference; at the end: moments, buffers)."""
    for tag, gan, use_gp, B, N, salt in (("c1_ls", "ls", False, 4, 512, 19), ("c2_wgangp", "wgan", True, 32, 2048, 20)):
        d = {}
        steps, final = _ref_three_steps(tag, gan, use_gp, B, N, salt, torch.float32)
        graphs = [(r["idx2_d"], r["idx2_g"]) for r in steps]
        s64, f64 = _ref_three_steps(tag, gan, use_gp, B, N, salt, torch.float64, graphs=graphs)
        for k, (r, r64) in enumerate(zip(steps, s64)):
            pre = "s%d|" % k
            d[pre + "idx2_d"] = r["idx2_d"].view(B, N, 10).numpy().astype(np.int16)
            d[pre + "idx2_g"] = r["idx2_g"].view(B, N, 10).numpy().astype(np.int16)
            d[pre + "lossD"] = r["lossD"].numpy(); d[pre + "lossG"] = r["lossG"].numpy(); d[pre + "alpha"] = r["alpha"].numpy()
            if use_gp:
                d[pre + "gp"] = r["gp"].numpy()
            put(d, pre + "fake_g", r["fake_g"], nsamp=4096)
            for kind in ("dgrad", "ggrad"):
                for n, g in r[kind].items():
                    put(d, pre + kind + "|" + n, g)
                    d["noise|" + pre + kind + "|" + n] = np.float64(_rel(g, r64[kind][n]))
                    d["noise_rms|" + pre + kind + "|" + n] = np.float64((g.double() - r64[kind][n]).pow(2).mean().sqrt())
            d["noise|" + pre + "lossD"] = np.float64(abs(float(r["lossD"]) - float(r64["lossD"])) / abs(float(r64["lossD"])))
            d["noise|" + pre + "lossG"] = np.float64(abs(float(r["lossG"]) - float(r64["lossG"])) / abs(float(r64["lossG"])))
            d["noise|" + pre + "fake_g"] = np.float64(_rel(r["fake_g"], r64["fake_g"]))
            print("g19 %s step %d: float32 vs float64: lossD %.2e lossG %.2e cloud %.2e worst D grad %.2e worst G grad %.2e" % (
                tag, k, d["noise|" + pre + "lossD"], d["noise|" + pre + "lossG"], d["noise|" + pre + "fake_g"],
                max(_rel(g, r64["dgrad"][n]) for n, g in r["dgrad"].items() if not n.endswith(ZERO_GRAD_BIASES)),
                max(_rel(g, r64["ggrad"][n]) for n, g in r["ggrad"].items() if not n.endswith(ZERO_GRAD_BIASES))), flush=True)
        for kind in ("d", "g"):
            for what in ("param", "upd", "m", "v"):
                for n, t in final[kind][what].items():
                    put(d, "%s%s|%s" % (kind, what, n), t)
                    if what != "param":
                        d["noise|%s%s|%s" % (kind, what, n)] = np.float64(_rel(t, f64[kind][what][n]))
            for n, b in final[kind]["buf"].items():
                d["%sbuf|%s" % (kind, n)] = b.numpy().copy()
                d["%sbuf64|%s" % (kind, n)] = f64[kind]["buf"][n].numpy().copy()
        save("g19_three_steps_%s.npz" % tag, d)


# ---------------------------------------------------------------- G20: ONE step from a mid-training state (tight state-carry pin)
def g20():
    """The tight half of round-4 review item 2(b).  G19's free-running trajectory is chaos-limited (the reference's own float32 and
    float64 runs are tens of per cent apart in the gradients after three steps), so it cannot pin "Adam at step >= 2" or "the n-th
    running-statistics update" to better than that.  Here the state that such a step STARTS from is a fixture both sides can build
    (spgan.fixture_rng.mid_training_state: Adam exp_avg / exp_avg_sq of the gradients' magnitude at step 7, non-trivial BatchNorm
    running statistics with 21 / 14 tracked batches), loaded into the imported reference's torch.optim.Adam and modules, and ONE
    iteration of model.py:239-279 is run from it: C1 (B=4, N=512, LS) and C2 (B=32, N=2048, WGAN-GP).  Stored: both EdgeConv2 graphs,
    losses, gradients, and after the step the parameters, Adam's moments (step 8: bias corrections 1-0.5^8, 1-0.99^8) and every
    BatchNorm buffer incl. num_batches_tracked -- all at one-step noise."""
    for tag, gan, use_gp, B, N, salt in (("c1_ls", "ls", False, 4, 512, 21), ("c2_wgangp", "wgan", True, 32, 2048, 22)):
        class O(Opts):
            np = N
        d = {}
        G = load_into(Generator(O), fr.init_params(orc.generator_shapes(), salt=salt)).train()
        D = load_into(Discriminator(O, num_point=N), fr.init_params(orc.discriminator_shapes(), salt=salt)).train()
        optG = torch.optim.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.99))
        optD = torch.optim.Adam(D.parameters(), lr=1e-4, betas=(0.5, 0.99))
        for net, opt, shapes, batches in ((D, optD, orc.discriminator_shapes(), 21), (G, optG, orc.generator_shapes(), 14)):
            st = fr.mid_training_state(shapes, [n for n, _ in net.named_buffers()], salt=salt, batches=batches)
            for n, p in net.named_parameters():
                opt.state[p] = {"step": torch.tensor(float(st["step"])), "exp_avg": st["m"][n].clone(), "exp_avg_sq": st["v"][n].clone()}
            net.load_state_dict({**net.state_dict(), **st["buffers"]})
        x = fr.sphere_template(N)[None].repeat(B, 1, 1)
        real, z_d, z_g = fr.synthetic_real(B, N, seed=2001), fr.latent(B, N, seed=2002), fr.latent(B, N, seed=2003)
        alpha = fr.uniform("g20.%s.alpha" % tag, (B, 1, 1), 0.0, 1.0)
        stages = {}
        hook = G.adain1.register_forward_hook(lambda m, i, o: stages.__setitem__("x1", o.detach().clone()))

        def req(m, f):
            for p in m.parameters():
                p.requires_grad = f
        req(G, False); req(D, True); optD.zero_grad()
        fake = G(x, z_d).detach()
        d["idx2_d"] = get_edge_features(stages["x1"], 10, return_idx=True)[1].view(B, N, 10).numpy().astype(np.int16)
        real_t = real.transpose(2, 1).contiguous()
        lossD, _ = LU.dis_loss(D(real_t), D(fake), gan=gan)
        if use_gp:
            orig = torch.rand
            torch.rand = lambda *a, **kw: alpha.clone().requires_grad_(kw.get("requires_grad", False))
            try:
                lossD = lossD + GradientPenalty(10.0, gamma=1)(D, real_t, fake)
            finally:
                torch.rand = orig
        lossD.backward()
        for n, p in D.named_parameters():
            put(d, "dgrad|" + n, p.grad)
        optD.step()
        req(G, True); req(D, False); optG.zero_grad()
        g_fake = G(x, z_g)
        d["idx2_g"] = get_edge_features(stages["x1"], 10, return_idx=True)[1].view(B, N, 10).numpy().astype(np.int16)
        hook.remove()
        g_real_logit = D(real_t)
        lossG, _ = LU.gen_loss(g_real_logit, D(g_fake), gan=gan)
        lossG.backward()
        for n, p in G.named_parameters():
            put(d, "ggrad|" + n, p.grad)
        optG.step()
        d["lossD"] = lossD.detach().numpy(); d["lossG"] = lossG.detach().numpy(); d["alpha"] = alpha.numpy()
        put(d, "fake_d", fake, nsamp=4096); put(d, "fake_g", g_fake, nsamp=4096)
        for kind, net, opt in (("d", D, optD), ("g", G, optG)):
            for n, p in net.named_parameters():
                put(d, kind + "param|" + n, p)
                st = opt.state[p]
                assert int(st["step"]) == 8
                put(d, kind + "m|" + n, st["exp_avg"]); put(d, kind + "v|" + n, st["exp_avg_sq"])
            for n, b in net.named_buffers():
                d[kind + "buf|" + n] = b.numpy().copy()
        print("g20 %s: lossD %.6f lossG %.6f" % (tag, float(lossD.detach()), float(lossG.detach())), flush=True)
        save("g20_mid_state_step_%s.npz" % tag, d)


if __name__ == "__main__":
    torch.manual_seed(0)
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4_g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15", "g16", "g17", "g18", "g19", "g20"]
    for name in which:
        globals()[name]()
