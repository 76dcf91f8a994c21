"""GPU: fp16-operand mode of the MFMA contractions (BASELINE configs[4] "fp16 MFMA MLPs"): every gemm_nt flavour against its
fp32 model at fp16-operand accuracy, and one full train step against the fp32 step."""
import numpy as np
import pytest
import torch

import kernel_model as km
from spgan import fixture_rng as fr
from oracle import spgan_oracle as orc
from test_kernels_gpu import close, ops, rnd  # noqa: F401  (ops is a fixture)
from test_parity_gpu import Opts, _load, sp   # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture()
def f16(ops):
    ops.set_mfma_operands("f16")
    yield ops
    ops.set_mfma_operands("f32")


@pytest.mark.parametrize("M,N,K", [(2048, 256, 128), (4096, 128, 1280), (1024, 1024, 256), (700, 64, 64), (512, 40, 36)])
def test_gemm_nt_f16(f16, M, N, K):
    ops = f16
    A, W, b = rnd("h.A%d" % K, (M, K)), rnd("h.W%d.%d" % (N, K), (N, K), 0.1), rnd("h.b%d" % N, (N,))
    ref = km.gemm_nt(A, W, b)
    close(ops.gemm_nt(A, W, b), ref, rtol=1e-3, what="plain")
    exact = km.gemm_nt(A.half().float(), W.half().float(), b)            # fp16-rounded operands, exact products, fp32 sums
    close(ops.gemm_nt(A, W, b), exact, rtol=2e-5, atol=1e-5, what="fp16-operand arithmetic")
    sc, sh = rnd("h.sc%d" % K, (K,)).abs() + 0.5, rnd("h.sh%d" % K, (K,), 0.3)
    y, m, v = ops.gemm_nt(A, W, b, pro=(sc, sh, 0.01), stats=True)
    y2, m2, v2 = km.gemm_nt(A, W, b, pro=(sc, sh, 0.01), stats=True)
    close(y, y2, rtol=1e-3, what="affine"); close(m, m2, rtol=1e-3, atol=1e-4); close(v, v2, rtol=2e-3)
    refm = rnd("h.ref%d.%d" % (M, N), (M, N))
    close(ops.gemm_nt_maskout(A, W, refm, 0.01), km.gemm_nt_maskout(A, W, refm, 0.01), rtol=1e-3, what="maskout")
    bsc, bsh, mu, inv = rnd("h.bsc%d" % N, (N,)), rnd("h.bsh%d" % N, (N,), 0.3), rnd("h.mu%d" % N, (N,), 0.2), rnd("h.inv%d" % N, (N,)).abs() + 0.5
    for a_, b_ in zip(ops.gemm_nt_bnbwd(A, W, refm, bsc, bsh, mu, inv, 0.01), km.gemm_nt_bnbwd(A, W, refm, bsc, bsh, mu, inv, 0.01)):
        close(a_, b_, rtol=2e-3, atol=2e-3, what="bnbwd")


def test_networks_f16_close_to_f32(sp):
    """fp16 MFMA operands against fp32: the stage in front of EdgeConv2's graph and the discriminator logits stay within fp16-
    operand accuracy; most feature-space kNN rows coincide (a flipped near-tie row changes the downstream features discretely,
    so the generated cloud itself is compared statistically, not element-wise); a WGAN-GP train step runs and stays finite."""
    B, N = 4, 512
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
    z = fr.latent(B, N, seed=82).cuda()
    real = fr.synthetic_real(B, N, seed=81).cuda()
    out = {}
    for kind in ("f32", "f16"):
        sp.ops.set_mfma_operands(kind)
        try:
            G = _load(sp.Generator(Opts), fr.init_params(orc.generator_shapes(), salt=8)).train()
            D = _load(sp.Discriminator(Opts), fr.init_params(orc.discriminator_shapes(), salt=8)).train()
            with torch.no_grad():
                fake = G(x, z)
                logit = D(real.transpose(2, 1).contiguous())
            out[kind] = (G.last_x1.clone(), G.EdgeConv2.last_idx.clone(), fake.clone(), logit.clone())
            if kind == "f16":
                tr = sp.TrainStep(G, D, gan="wgan", use_gp=True, lambda_gp=10.0)
                info = tr.step(x, real, z, fr.latent(B, N, seed=83).cuda(), alpha=fr.uniform("f16.alpha", (B, 1, 1), 0.0, 1.0).cuda())
                assert torch.isfinite(info["loss_d"]).item() and torch.isfinite(info["loss_g"]).item()
                assert all(torch.isfinite(p).all().item() for p in list(G.parameters()) + list(D.parameters()))
        finally:
            sp.ops.set_mfma_operands("f32")
    a, b = out["f32"], out["f16"]
    assert ((a[0] - b[0]).norm() / a[0].norm()).item() < 3e-3                       # x1: EdgeConv1 + AdaIN1
    assert (a[1] == b[1]).all(dim=1).float().mean().item() > 0.9                    # EdgeConv2 kNN rows
    assert ((a[3] - b[3]).abs().max() / a[3].abs().max()).item() < 2e-2             # D logits
    assert abs(a[2].std().item() - b[2].std().item()) / a[2].std().item() < 0.1     # generated cloud: same scale
