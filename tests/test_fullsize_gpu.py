"""GPU, BASELINE.json full sizes: size-independent properties of the HIP path where the oracle would take minutes.
  * C2 (N=2048, B=32, WGAN-GP): one D+G step is finite, the sphere graph equals the reference's (golden), the
    whole step is run-to-run bit-deterministic (no float atomics), BN running stats advanced 2x (G) / 5x (D);
  * D is batch-permutation equivariant; G's output is invariant to a permutation of the *latent-tiled* batch rows;
  * C4 size (N=4096): kNN rows are ascending in exact distance and hold the k+1 smallest (checked on sampled rows)."""
import numpy as np
import pytest
import torch

from helpers import golden
from spgan import fixture_rng as fr

pytestmark = pytest.mark.gpu


class Opts:
    np = 2048; nk = 20; nz = 128; softmax = True; off = False; attn = False
    use_head = False; eql = False; z_norm = False; small_d = False


def _models(seed=123):
    import spgan
    torch.manual_seed(seed)
    return spgan.Generator(Opts).cuda(), spgan.Discriminator(Opts).cuda()


def _inputs(B, N):
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
    real = fr.synthetic_real(B, N, seed=1234).cuda()
    z1, z2 = fr.latent(B, N, seed=1).cuda(), fr.latent(B, N, seed=2).cuda()
    alpha = fr.uniform("fs.alpha", (B, 1, 1), 0.0, 1.0).cuda()
    return x, real, z1, z2, alpha


def test_c2_step_properties():
    import spgan
    B, N = 32, 2048
    x, real, z1, z2, alpha = _inputs(B, N)
    runs = []
    for _ in range(2):
        G, D = _models()
        tr = spgan.TrainStep(G, D, gan="wgan", use_gp=True, lambda_gp=10.0)
        info = tr.step(x, real, z1, z2, alpha=alpha)
        torch.cuda.synchronize()
        assert torch.isfinite(info["loss_d"]).item() and torch.isfinite(info["loss_g"]).item()
        runs.append((torch.cat([p.detach().reshape(-1) for p in G.parameters()]).clone(),
                     torch.cat([p.detach().reshape(-1) for p in D.parameters()]).clone(),
                     info["loss_d"].item(), info["loss_g"].item()))
        # the sphere graph of every shape is the reference's, index for index
        ref = torch.from_numpy(golden("g1_edge_features.npz")["sphere2048|idx"].astype(np.int64))
        loc = spgan.ops.idx_to_local64(G.EdgeConv1.last_idx, B, N).view(B, N, 10).cpu()
        assert (loc == ref[None]).all()
        G.flush_bn_counts(); D.flush_bn_counts()
        assert int(G.global_conv[1].num_batches_tracked) == 2 and int(D.fc2[1].num_batches_tracked) == 5   # model.py call order
        assert torch.isfinite(G.EdgeConv2.conv_w[4].running_var).all()
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1]), "train step is not bit-deterministic"
    assert runs[0][2] == runs[1][2] and runs[0][3] == runs[1][3]


def test_discriminator_batch_permutation_equivariance():
    B, N = 32, 2048
    _, D = _models()
    x = fr.synthetic_real(B, N, seed=77).transpose(2, 1).contiguous().cuda()
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(0)).cuda()
    with torch.no_grad():
        a = D(x)
        b = D(x[perm].contiguous())
    assert torch.allclose(a[perm], b, rtol=1e-4, atol=1e-5), (a[perm] - b).abs().max().item()


def test_generator_shape_independence_of_sphere_rows():
    """Every shape of a batch that shares the latent sees the same sphere: identical latents => identical outputs."""
    B, N = 8, 2048
    G, _ = _models()
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
    z = fr.latent(1, N, seed=5).repeat(B, 1, 1).cuda()
    with torch.no_grad():
        out = G(x, z)
    assert out.shape == (B, 3, N) and torch.isfinite(out).all() and out.abs().max().item() <= 1.0
    assert torch.allclose(out[0], out[B - 1], rtol=0, atol=1e-6)


@pytest.mark.parametrize("N,C,mode", [(4096, 3, 1), (4096, 64, 0)])
def test_knn_full_size_invariants(N, C, mode):
    import spgan
    B, k = 4, 10
    x = (fr.sphere_template(N)[None].repeat(B, 1, 1) if C == 3 else fr.normal("fs.knn", (B, N, C), 0.5)).cuda().reshape(B * N, C).contiguous()
    idx = spgan.ops.knn(x, B, N, k, mode).cpu().long()
    rows = torch.arange(0, B * N, 97)
    xd = x.cpu().double()
    for r in rows.tolist():
        b = r // N
        d = ((xd[b * N:(b + 1) * N] - xd[r]) ** 2).sum(1)
        srt = torch.sort(d)[0]
        got = d[idx[r] - b * N]
        tol = 0.0 if mode == 1 else 3e-5 * C
        assert (got - srt[1:k + 1]).abs().max().item() <= tol, (r, (got - srt[1:k + 1]).abs().max().item())
        assert (idx[r] >= b * N).all() and (idx[r] < (b + 1) * N).all()


@pytest.mark.parametrize("B,N,gan,use_gp", [(16, 4096, "wgan", True), (4, 512, "ls", False)])
def test_c4_c1_step_properties(B, N, gan, use_gp):
    """BASELINE configs[3] per-GPU shape (N=4096, b=16) and configs[0] (N=512, bs=4, LS): the step is finite, replayable as a
    hipGraph with bit-identical results, and the sphere graph equals the reference's."""
    import spgan

    class O(Opts):
        np = N
    x, real, z1, z2, alpha = _inputs(B, N)
    res = []
    for graph in (False, True):
        torch.manual_seed(123)
        G, D = spgan.Generator(O).cuda(), spgan.Discriminator(O, num_point=N).cuda()
        tr = spgan.TrainStep(G, D, gan=gan, use_gp=use_gp, lambda_gp=10.0, graph=graph, graph_warmup=1)
        for i in range(3):
            info = tr.step(x, real, z1 if i % 2 == 0 else z2, z2 if i % 2 == 0 else z1, alpha=alpha if use_gp else None)
        torch.cuda.synchronize()
        assert torch.isfinite(info["loss_d"]).item() and torch.isfinite(info["loss_g"]).item()
        res.append(torch.cat([p.detach().reshape(-1) for p in list(G.parameters()) + list(D.parameters())]).clone())
        ref = torch.from_numpy(golden("g1_edge_features.npz")["sphere%d|idx" % N].astype(np.int64))
        loc = spgan.ops.idx_to_local64(G.EdgeConv1.last_idx, B, N).view(B, N, 10).cpu()
        assert (loc == ref[None]).all()
    assert torch.equal(res[0], res[1]), "graph replay differs from eager issue"
