"""CPU: the host composition (spgan.nets / functions / modules: forward, backward, WGAN-GP double
backward, state_dict surface) checked against the oracle, with every HIP op replaced by its
plain-PyTorch model from tests/kernel_model.py (test double, injected here only)."""
import inspect
import types

import numpy as np
import pytest
import torch

import kernel_model as km
from helpers import rel_l2
from oracle import spgan_oracle as orc
from spgan import fixture_rng as fr


class Opts:
    np = 256; nk = 20; nz = 128; softmax = True; off = False; attn = False
    use_head = False; eql = False; z_norm = False; small_d = False


ZERO_GRAD_BIASES = ("conv_w.0.bias", "conv_w.3.bias", "conv_x.0.bias", "global_conv.0.bias", "global_conv.3.bias",
                    "mlps.0.bias", "mlps.3.bias", "mlps.6.bias", "fc2.0.bias")


@pytest.fixture()
def spgan_cpu(monkeypatch):
    """spgan with ops -> kernel models, GPU guard off."""
    import spgan.ops as ops
    import spgan.modules as modules
    for name, fn in inspect.getmembers(km, inspect.isfunction):
        if name.startswith("_"):
            continue
        assert hasattr(ops, name), "kernel_model.%s has no counterpart in spgan.ops" % name
        monkeypatch.setattr(ops, name, fn)
    monkeypatch.setattr(ops, "SparseAffine", km.SparseAffine)          # isinstance checks in nets.py select the collapsed paths
    monkeypatch.setattr(ops, "Affine2", km.Affine2)
    monkeypatch.setattr(ops, "ActOperand", km.ActOperand)
    monkeypatch.setattr(modules, "_require_gpu", lambda t, what: None)
    return types.SimpleNamespace(ops=ops, modules=modules)


def test_every_op_has_a_model():
    import spgan.ops as ops
    settings = {"set_mfma_operands", "get_mfma_operands", "bump_weights_epoch", "weights_epoch_of", "index_check_flag", "index_check_raise"}   # switches / host bookkeeping, not arithmetic
    public = [n for n, f in inspect.getmembers(ops, inspect.isfunction)
              if not n.startswith("_") and f.__module__ == ops.__name__ and n not in settings]
    missing = [n for n in public if not hasattr(km, n)]
    assert not missing, "ops without a kernel model: %s" % missing


def _load(module, params):
    sd = module.state_dict()
    module.load_state_dict({**sd, **{k: v.detach().clone() for k, v in params.items()}})
    return module


def _cmp(name, got, ref, rtol, atol=1e-6):
    e = rel_l2(got.detach().numpy(), ref.detach().numpy())
    mx = (got - ref).abs().max().item()
    assert e <= rtol or mx <= atol, "%s: rel-L2 %.3e max-abs %.3e" % (name, e, mx)


def test_state_dict_surface(spgan_cpu):
    G = spgan_cpu.modules.Generator(Opts)
    D = spgan_cpu.modules.Discriminator(Opts)
    gs, ds = orc.generator_shapes(), orc.discriminator_shapes()
    gsd, dsd = G.state_dict(), D.state_dict()
    for k, shp in gs.items():
        assert tuple(gsd[k].shape) == shp, k
    for k, shp in ds.items():
        assert tuple(dsd[k].shape) == shp, k
    assert len([k for k in gsd if k not in gs]) == 24 and len([k for k in dsd if k not in ds]) == 12     # BN buffers
    assert sum(p.numel() for p in G.parameters()) == 585155 and sum(p.numel() for p in D.parameters()) == 980353
    assert torch.equal(G.adain1.style.bias.detach(), torch.cat([torch.ones(64), torch.zeros(64)]))


def test_discriminator_fwd_bwd(spgan_cpu):
    B, N = 3, 128
    p = fr.init_params(orc.discriminator_shapes(), salt=11)
    D = _load(spgan_cpu.modules.Discriminator(Opts), p).train()
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    buf = orc.bn_buffers(orc.discriminator_shapes())
    x = fr.synthetic_real(B, N, seed=12).transpose(2, 1).contiguous()
    x1 = x.clone().requires_grad_(True); x2 = x.clone().requires_grad_(True)
    out = D(x1)
    ref = orc.discriminator_forward(po, x2, True, buf)
    _cmp("logit", out, ref, 1e-5)
    w = fr.normal("hd.w", out.shape)
    (out * w).sum().backward()
    names = list(po.keys())
    grads = torch.autograd.grad((ref * w).sum(), [x2] + [po[n] for n in names])
    _cmp("dx", x1.grad, grads[0], 2e-4)
    for n, g in zip(names, grads[1:]):
        _cmp("grad " + n, dict(D.named_parameters())[n].grad, g, 5e-4, atol=1e-3 if n.endswith(ZERO_GRAD_BIASES) else 1e-7)
    for k, v in buf.items():
        np.testing.assert_allclose(D.state_dict()[k].numpy(), v.numpy(), rtol=1e-5, atol=1e-6)
    # frozen D (G-step): input gradient only
    for q in D.parameters():
        q.requires_grad_(False)
    x3 = x.clone().requires_grad_(True)
    (D(x3) * w).sum().backward()
    buf2 = orc.bn_buffers(orc.discriminator_shapes())
    _cmp("dx frozen", x3.grad, grads[0], 2e-4)


def test_discriminator_gradient_penalty(spgan_cpu):
    """autograd.grad(create_graph=True) + backward through our Function pair == oracle autograd double backward."""
    B, N = 3, 128
    p = fr.init_params(orc.discriminator_shapes(), salt=13)
    D = _load(spgan_cpu.modules.Discriminator(Opts), p).train()
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    real = fr.synthetic_real(B, N, seed=14).transpose(2, 1).contiguous()
    fake = (0.8 * fr.synthetic_real(B, N, seed=15) + 0.05 * fr.normal("hgp.n", (B, N, 3))).transpose(2, 1).contiguous()
    alpha = fr.uniform("hgp.alpha", (B, 1, 1), 0.0, 1.0)
    gp_ref = orc.gradient_penalty(lambda t: orc.discriminator_forward(po, t, True, None), real, fake, alpha, 10.0, 1.0)
    names = list(po.keys())
    gref = torch.autograd.grad(gp_ref, [po[n] for n in names], allow_unused=True)
    gp = orc.gradient_penalty(D, real, fake, alpha, 10.0, 1.0)         # same formula, our module as netD
    np.testing.assert_allclose(gp.item(), gp_ref.item(), rtol=1e-4)
    gp.backward()
    for n, g in zip(names, gref):
        mine = dict(D.named_parameters())[n].grad
        g = torch.zeros_like(po[n]) if g is None else g
        mine = torch.zeros_like(g) if mine is None else mine
        _cmp("gp grad " + n, mine, g, 2e-3, atol=1e-3 if n.endswith(ZERO_GRAD_BIASES) else 2e-6)


def test_discriminator_gradient_penalty_eval_mode(spgan_cpu):
    """The same double backward through a Discriminator in eval() mode (running statistics: BatchNorm is a fixed affine;
    Common/gradient_penalty.py:28-33 works in either mode) == oracle autograd."""
    B, N = 3, 128
    p = fr.init_params(orc.discriminator_shapes(), salt=13)
    D = _load(spgan_cpu.modules.Discriminator(Opts), p)
    buf = orc.bn_buffers(orc.discriminator_shapes())
    for k in buf:                                       # non-trivial running statistics
        if k.endswith("running_mean"):
            buf[k] = fr.normal("hgpe.m." + k, buf[k].shape, 0.05)
        elif k.endswith("running_var"):
            buf[k] = fr.uniform("hgpe.v." + k, buf[k].shape, 0.5, 1.5)
    D.load_state_dict({**D.state_dict(), **buf})
    D.eval()
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    real = fr.synthetic_real(B, N, seed=14).transpose(2, 1).contiguous()
    fake = (0.8 * fr.synthetic_real(B, N, seed=15) + 0.05 * fr.normal("hgp.n", (B, N, 3))).transpose(2, 1).contiguous()
    alpha = fr.uniform("hgp.alpha", (B, 1, 1), 0.0, 1.0)
    gp_ref = orc.gradient_penalty(lambda t: orc.discriminator_forward(po, t, False, buf), real, fake, alpha, 10.0, 1.0)
    names = list(po.keys())
    gref = torch.autograd.grad(gp_ref, [po[n] for n in names], allow_unused=True)
    gp = orc.gradient_penalty(D, real, fake, alpha, 10.0, 1.0)
    np.testing.assert_allclose(gp.item(), gp_ref.item(), rtol=1e-4)
    gp.backward()
    for n, g in zip(names, gref):
        mine = dict(D.named_parameters())[n].grad
        g = torch.zeros_like(po[n]) if g is None else g
        mine = torch.zeros_like(g) if mine is None else mine
        _cmp("gp(eval) grad " + n, mine, g, 2e-3, atol=2e-6)


@pytest.mark.parametrize("fin,fout", [(3, 64), (64, 128)])
def test_edgeblock(spgan_cpu, fin, fout):
    B, N, k = 2, 96, 10
    pref = "EdgeConv1" if fin == 3 else "EdgeConv2"
    shapes = {kk: v for kk, v in orc.generator_shapes().items() if kk.startswith(pref + ".")}
    p = fr.init_params(shapes, salt=21)
    blk = _load(spgan_cpu.modules.EdgeBlock(fin, fout, k), {kk[len(pref) + 1:]: v for kk, v in p.items()}).train()
    po = {kk: v.clone().requires_grad_(True) for kk, v in p.items()}
    buf = orc.bn_buffers({kk: tuple(v.shape) for kk, v in p.items()})
    x = fr.normal("heb.x%d" % fin, (B, fin, N), 0.7)
    x1 = x.clone().requires_grad_(True); x2 = x.clone().requires_grad_(True)
    y = blk(x1)
    idx = spgan_cpu.ops.idx_to_local64(blk.last_idx, B, N)
    yr = orc.edge_block(po, pref, x2, k, idx=idx, training=True, buffers=buf)
    _cmp("y", y, yr, 2e-5)
    dy = fr.normal("heb.dy%d" % fin, y.shape)
    (y * dy).sum().backward()
    names = list(po.keys())
    grads = torch.autograd.grad((yr * dy).sum(), [x2] + [po[n] for n in names])
    _cmp("dx", x1.grad, grads[0], 2e-4)
    for n, g in zip(names, grads[1:]):
        _cmp("grad " + n, dict(blk.named_parameters())[n[len(pref) + 1:]].grad, g, 5e-4, atol=2e-3 if n.endswith(ZERO_GRAD_BIASES) else 1e-7)
    for kk, v in buf.items():
        np.testing.assert_allclose(blk.state_dict()[kk[len(pref) + 1:]].numpy(), v.numpy(), rtol=1e-5, atol=1e-6)


def test_adain(spgan_cpu):
    B, C, N = 2, 64, 96
    p = fr.init_params({"a.style.weight": (2 * C, 128, 1), "a.style.bias": (2 * C,)}, salt=3)
    m = _load(spgan_cpu.modules.AdaptivePointNorm(C, 128), {"style.weight": p["a.style.weight"], "style.bias": p["a.style.bias"]})
    po = {kk: v.clone().requires_grad_(True) for kk, v in p.items()}
    x = fr.normal("had.x", (B, C, N)); s = fr.normal("had.s", (B, 128, N), 0.3)
    x1, s1 = x.clone().requires_grad_(True), s.clone().requires_grad_(True)
    x2, s2 = x.clone().requires_grad_(True), s.clone().requires_grad_(True)
    y, yr = m(x1, s1), orc.adaptive_point_norm(po, "a", x2, s2)
    _cmp("y", y, yr, 1e-5)
    dy = fr.normal("had.dy", y.shape)
    (y * dy).sum().backward()
    gx, gs, gw, gb = torch.autograd.grad((yr * dy).sum(), [x2, s2, po["a.style.weight"], po["a.style.bias"]])
    _cmp("dx", x1.grad, gx, 1e-4); _cmp("dstyle", s1.grad, gs, 1e-4)
    _cmp("dw", m.style.weight.grad, gw, 1e-4); _cmp("db", m.style.bias.grad, gb, 1e-4)


def test_generator(spgan_cpu):
    B, N = 4, 128
    p = fr.init_params(orc.generator_shapes(), salt=31)
    G = _load(spgan_cpu.modules.Generator(Opts), p).train()
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    buf = orc.bn_buffers(orc.generator_shapes())
    x = fr.synthetic_real(B, N, seed=32)
    z = fr.latent(B, N, seed=33)
    out = G(x, z)
    idx1 = spgan_cpu.ops.idx_to_local64(G.EdgeConv1.last_idx, B, N)
    idx2 = spgan_cpu.ops.idx_to_local64(G.EdgeConv2.last_idx, B, N)
    st = {}
    ref = orc.generator_forward(po, x, z, training=True, buffers=buf, idx1=idx1, idx2=idx2, stages=st)
    # kNN agreement with the oracle's own graph construction (tie-aware: rows may differ only at near-ties)
    own1 = orc.knn_sorted(x.transpose(2, 1).contiguous(), 10)
    assert (own1.reshape(B, -1) == idx1).float().mean().item() > 0.999
    _cmp("out", out, ref, 5e-4)
    dy = fr.normal("hg.dy", out.shape)
    (out * dy).sum().backward()
    names = list(po.keys())
    grads = torch.autograd.grad((ref * dy).sum(), [po[n] for n in names])
    for n, g in zip(names, grads):
        # kink-limited end to end (see tests/test_oracle_golden.py::test_generator)
        _cmp("grad " + n, dict(G.named_parameters())[n].grad, g, 3e-2, atol=2e-3 if n.endswith(ZERO_GRAD_BIASES) else 1e-7)
    for kk, v in buf.items():
        np.testing.assert_allclose(G.state_dict()[kk].numpy(), v.numpy(), rtol=2e-4, atol=1e-5)


def test_sphere_graph_cache(spgan_cpu, monkeypatch):
    """EdgeConv1's kNN graph is rebuilt only when the sphere tensor object or its version changes."""
    B, N = 2, 96
    G = _load(spgan_cpu.modules.Generator(Opts), fr.init_params(orc.generator_shapes(), salt=51)).train()
    calls = []
    real_knn = spgan_cpu.ops.knn
    monkeypatch.setattr(spgan_cpu.ops, "knn", lambda x, B_, N_, k, mode=0: (calls.append(x.shape[1]), real_knn(x, B_, N_, k, mode))[1])
    x = fr.synthetic_real(B, N, seed=52); z = fr.latent(B, N, seed=53)
    o1 = G(x, z); o2 = G(x, z)
    assert calls.count(3) == 1 and calls.count(64) == 2              # sphere graph once, feature graph every call
    assert torch.equal(G.EdgeConv1.last_idx, G._sphere_graph["idx"])
    x.mul_(1.0)                                                       # in-place write -> version bump -> rebuild
    G(x, z)
    assert calls.count(3) == 2
    x2 = x.clone()                                                    # different tensor object -> rebuild
    G(x2, z)
    assert calls.count(3) == 3


def test_generator_no_grad_and_eval(spgan_cpu):
    B, N = 2, 96
    p = fr.init_params(orc.generator_shapes(), salt=41)
    G = _load(spgan_cpu.modules.Generator(Opts), p).train()
    x = fr.synthetic_real(B, N, seed=42); z = fr.latent(B, N, seed=43)
    for q in G.parameters():
        q.requires_grad_(False)
    out = G(x, z)                                  # D-step: frozen G, no graph
    assert not out.requires_grad
    G.eval()
    buf = {k: v.clone() for k, v in G.state_dict().items() if "running" in k or "num_batches" in k}
    oe = G(x, z)
    idx1 = spgan_cpu.ops.idx_to_local64(G.EdgeConv1.last_idx, B, N)
    idx2 = spgan_cpu.ops.idx_to_local64(G.EdgeConv2.last_idx, B, N)
    ref = orc.generator_forward(p, x, z, training=False, buffers=buf, idx1=idx1, idx2=idx2)
    _cmp("eval out", oe, ref, 1e-4)


@pytest.mark.parametrize("flags", [dict(attn=True), dict(eql=True), dict(attn=True, eql=True, use_head=True)])
def test_generator_attn_eql(spgan_cpu, flags):
    """--attn / --eql host logic (parameter surface, EqualLR scaling, attention forward/backward composition) on the kernel models."""
    B, N = 4, 64
    O = type("O", (Opts,), flags)
    shapes = orc.generator_shapes(**flags)
    p = fr.init_params(shapes, salt=41)
    G = spgan_cpu.modules.Generator(O)
    assert {k: tuple(v.shape) for k, v in G.named_parameters()} == {k: tuple(v) for k, v in shapes.items()}
    assert [k for k, _ in G.named_parameters()] == list(shapes.keys()) or set(dict(G.named_parameters())) == set(shapes)
    _load(G, p).train()
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    x = fr.sphere_template(256)[None, :N].repeat(B, 1, 1).contiguous()
    z = fr.latent(B, N, seed=43)
    out = G(x, z)
    idx1 = spgan_cpu.ops.idx_to_local64(G.EdgeConv1.last_idx, B, N)
    idx2 = spgan_cpu.ops.idx_to_local64(G.EdgeConv2.last_idx, B, N)
    ref = orc.generator_forward(orc.eql_effective_params(po), x, z, training=True, buffers=orc.bn_buffers(shapes), idx1=idx1, idx2=idx2)
    _cmp("out", out, ref, 5e-4)
    dy = fr.normal("hga.dy", out.shape)
    (out * dy).sum().backward()
    names = list(po.keys())
    grads = torch.autograd.grad((ref * dy).sum(), [po[n] for n in names])
    for n, g in zip(names, grads):
        plain = n.replace(".linear.", ".").replace(".conv.", ".")
        _cmp("grad " + n, dict(G.named_parameters())[n].grad, g, 3e-2, atol=2e-3 if plain.endswith(ZERO_GRAD_BIASES) else 1e-7)


def test_generator_per_shape_latent(spgan_cpu):
    """z [B,1,nz] (one latent per shape, what noise_generator tiles over N) == the tiled [B,N,nz] input: forward and all gradients."""
    B, N = 3, 64
    p = fr.init_params(orc.generator_shapes(), salt=51)
    x = fr.sphere_template(256)[None, :N].repeat(B, 1, 1).contiguous()
    z1 = fr.latent(B, N, seed=53)[:, :1, :].contiguous()
    dy = None
    res = []
    for z in (z1, z1.expand(B, N, -1).contiguous()):
        G = _load(spgan_cpu.modules.Generator(Opts), p).train()
        out = G(x, z)
        dy = fr.normal("hps.dy", out.shape) if dy is None else dy
        (out * dy).sum().backward()
        res.append((out.detach(), {n: q.grad.clone() for n, q in G.named_parameters()}))
    _cmp("out", res[0][0], res[1][0], 1e-5)
    for n in res[0][1]:
        _cmp("grad " + n, res[0][1][n], res[1][1][n], 2e-3, atol=2e-3 if n.endswith(ZERO_GRAD_BIASES) else 1e-6)


def test_discriminator_forward_many_equals_separate_calls(spgan_cpu):
    """D.forward_many(a, b): the conv stacks run separately (own BatchNorm statistics, running statistics in call order), the
    BatchNorm-free head once on the stacked pooled features -- same logits, gradients and buffers as D(a), D(b)."""
    B, N = 3, 128
    p = fr.init_params(orc.discriminator_shapes(), salt=41)
    xa = fr.synthetic_real(B, N, seed=42).transpose(2, 1).contiguous()
    xb = (0.7 * fr.synthetic_real(B, N, seed=43)).transpose(2, 1).contiguous()
    w = fr.normal("fm.w", (2 * B, 1))
    res = []
    for many in (False, True):
        D = _load(spgan_cpu.modules.Discriminator(Opts), p).train()
        a, b = xa.clone().requires_grad_(True), xb.clone().requires_grad_(True)
        la, lb = D.forward_many(a, b) if many else (D(a), D(b))
        (torch.cat([la, lb]) * w).sum().backward()
        res.append((la.detach(), lb.detach(), a.grad, b.grad, {n: q.grad.clone() for n, q in D.named_parameters()},
                    {k: v.clone() for k, v in D.state_dict().items() if k not in dict(D.named_parameters())}))
    for i in range(4):
        _cmp("forward_many tensor %d" % i, res[1][i], res[0][i], 1e-6)
    for n in res[0][4]:
        _cmp("forward_many grad " + n, res[1][4][n], res[0][4][n], 2e-6, atol=1e-7)
    for k in res[0][5]:
        assert torch.allclose(res[1][5][k].float(), res[0][5][k].float(), rtol=1e-6, atol=1e-7), k


def test_multi_add_stride_merging_is_host_logic():
    """ops._strided3: how a pair of equal-shape views becomes the <= 3 strided dimensions of spgan_multi_add3 (pure host code: runs on CPU tensors)."""
    from spgan import ops as real_ops
    import importlib
    o = importlib.reload(real_ops) if not hasattr(real_ops, "_strided3") else real_ops
    F_, k = 8, 5
    dst = torch.zeros(F_, F_, 1, k)
    g = torch.arange(F_ * k * F_, dtype=torch.float32).view(F_, k * F_)
    src = g.view(F_, k, F_).permute(0, 2, 1).unsqueeze(2)                       # the conv_out gradient as a view of the parameter's shape
    n1, n2, ds, ss = o._strided3(dst, src)
    assert (n1, n2) == (F_, k) and ds == [F_ * k, k, 1] and ss == [k * F_, 1, F_]
    assert o._strided3(dst, dst.clone()) is None                                # contiguous pair: the plain kernel
    full = torch.zeros(6, 10, 1)
    n1, n2, ds, ss = o._strided3(full.view(6, 10)[:, 4:], torch.ones(6, 6))     # a column block of the destination
    assert (n1, n2) == (6, 6) and ds[1:] == [10, 1] and ss[1:] == [6, 1]
    with pytest.raises(ValueError):
        o._strided3(torch.zeros(2, 3, 4, 5).permute(3, 2, 1, 0), torch.zeros(5, 4, 3, 2))


def test_stacked_rows_is_a_view_only_for_adjacent_blocks():
    """ops.stacked_rows (host logic): consecutive row blocks of one buffer come back as a view of it, anything else through torch.cat."""
    import importlib
    real_ops = importlib.import_module("spgan.ops")
    import inspect
    src = inspect.getsource(real_ops)                      # the fixture swaps the public ops for their models: take the real helper from source
    ns = {"torch": torch, "Tensor": torch.Tensor}
    start = src.index("def stacked_rows(")
    exec(src[start:src.index("\ndef ", start + 10)], ns)
    stacked = ns["stacked_rows"]
    base = torch.arange(24.0).view(6, 4)
    a, b = base[:2], base[2:5]
    v = stacked([a, b])
    assert v.data_ptr() == base.data_ptr() and torch.equal(v, base[:5])
    w = stacked([b, a])                                    # not adjacent in this order: a copy
    assert w.data_ptr() != base.data_ptr() and torch.equal(w, torch.cat([b, a]))
    assert stacked([a]).data_ptr() == a.data_ptr()
    c = torch.zeros(3, 4)
    assert torch.equal(stacked([a, c]), torch.cat([a, c]))
