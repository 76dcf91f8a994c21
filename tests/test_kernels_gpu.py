"""GPU: every HIP kernel (through the C ABI, via spgan.ops) against its plain-PyTorch model."""
import numpy as np
import pytest
import torch

import kernel_model as km
from spgan import fixture_rng as fr

pytestmark = pytest.mark.gpu


def dev(t):
    return t.cuda()


def rnd(name, shape, std=1.0):
    return fr.normal("kt." + name, shape, std).cuda()


def close(a, b, rtol=2e-5, atol=1e-6, what=""):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).norm() / max(b.norm().item(), 1e-30)
    mx = (a - b).abs().max().item()
    assert err <= rtol or mx <= atol, "%s: rel-L2 %.3e max-abs %.3e" % (what, err, mx)


@pytest.fixture(scope="module")
def ops():
    from spgan import ops as o
    from spgan import _lib
    _lib.load()
    return o


# ----------------------------------------------------------------------------- graph
def knn_tie_aware(idx, x_pm, B, N, k, tol):
    """Returned neighbours must be the k+1 smallest minus rank 0, ascending, up to distance ties < tol."""
    C = x_pm.shape[1]
    x = x_pm.view(B, N, C).double().cpu()
    d = ((x[:, :, None, :] - x[:, None, :, :]) ** 2).sum(-1)            # exact distances
    loc = (idx.cpu().long().view(B, N, k) - (torch.arange(B) * N).view(B, 1, 1))
    assert loc.min() >= 0 and loc.max() < N
    srt = torch.sort(d, dim=2)[0]
    got = torch.gather(d, 2, loc)
    # rank r of the result must carry the (r+1)-th smallest distance (rank 0 dropped)
    assert (got - srt[:, :, 1:k + 1]).abs().max().item() <= tol, (got - srt[:, :, 1:k + 1]).abs().max().item()
    # no duplicates in a row
    assert (torch.sort(loc, dim=2)[0].diff(dim=2) != 0).all()


@pytest.mark.parametrize("N", [256, 512, 2048, 4096])
def test_knn_sphere_exact(ops, N):
    d = np.load(__import__("os").path.join(__import__("helpers").GOLDEN, "g1_edge_features.npz"))
    x = fr.sphere_template(N).cuda()
    B = 3
    xb = x[None].repeat(B, 1, 1).reshape(B * N, 3).contiguous()
    idx = ops.knn(xb, B, N, 10, mode=1)
    ref = torch.from_numpy(d["sphere%d|idx" % N].astype(np.int64))
    for b in range(B):
        assert torch.equal(idx[b * N:(b + 1) * N].cpu().long() - b * N, ref), "sphere kNN differs from the reference (N=%d, b=%d)" % (N, b)


@pytest.mark.parametrize("B,N,C,k", [(2, 256, 64, 10), (2, 512, 64, 10), (2, 300, 5, 10), (1, 2048, 64, 10), (2, 130, 128, 20), (1, 77, 16, 32)])
def test_knn_features(ops, B, N, C, k):
    if (N, C) in ((256, 64), (512, 64), (300, 5)):
        x_cm = fr.normal("g1.feat.%d.%d" % (N, C), (2, C, N), 0.5).cuda()
    else:
        x_cm = rnd("knn.%d.%d" % (N, C), (B, C, N), 0.5)
    x_pm = km.cm_to_pm(x_cm)
    idx = ops.knn(x_pm, B, N, k, mode=0)
    knn_tie_aware(idx, x_pm, B, N, k, tol=2e-5 * C)
    ref = km.knn(x_pm, B, N, k, 0)
    agree = (idx == ref).all(dim=1).float().mean().item()
    assert agree >= 0.995, "row agreement with the fp32 model only %.4f" % agree
    if (N, C) in ((256, 64), (512, 64), (300, 5)):
        g = np.load(__import__("os").path.join(__import__("helpers").GOLDEN, "g1_edge_features.npz"))
        gold = torch.from_numpy(g["feat%d_%d|idx" % (N, C)].astype(np.int64)).view(2 * N, k)
        loc = idx.cpu().long() - (torch.arange(2).repeat_interleave(N) * N).view(-1, 1)
        rows = (loc == gold).all(dim=1)
        # every disagreeing row must be a near-tie in the reference's own sorted distances
        sd = torch.from_numpy(g["feat%d_%d|sorted_dist" % (N, C)]).view(2 * N, 12)
        gaps = sd.diff(dim=1).abs().min(dim=1)[0]
        assert rows.float().mean().item() >= 0.995
        assert (gaps[~rows] < 1e-4).all(), "a non-tie row disagrees with the reference"


@pytest.mark.parametrize("B,N,k", [(3, 300, 10), (2, 2048, 10), (2, 4096, 10), (1, 4096, 20), (2, 77, 32)])
def test_csr(ops, B, N, k):
    """In-edge lists: the LDS-segment route (N*k <= 65536 local edge ids, 16 bits each) and the in-place one behind it (4096 x 20)."""
    x = rnd("csr.x.%d" % N, (B * N, 3))
    idx = ops.knn(x, B, N, k, mode=1)
    rowptr, src = ops.csr_build(idx, B, N)
    rp, sr = km.csr_build(idx, B, N)
    assert torch.equal(rowptr.cpu(), rp.cpu()) and torch.equal(src.cpu(), sr.cpu())


def test_csr_hub_point(ops):
    """One point that is everybody's neighbour (a segment of N entries) next to empty segments."""
    B, N, k = 2, 512, 4
    idx = torch.zeros((B * N, k), dtype=torch.int32, device="cuda")
    ar = torch.arange(B * N, device="cuda", dtype=torch.int32)
    base = (ar // N) * N
    idx[:, 0] = base                                   # every point -> point 0 of its shape
    for r in range(1, k):
        idx[:, r] = base + (ar - base + r) % N
    rowptr, src = ops.csr_build(idx.contiguous(), B, N)
    rp, sr = km.csr_build(idx, B, N)
    assert torch.equal(rowptr.cpu(), rp.cpu()) and torch.equal(src.cpu(), sr.cpu())


def test_edge_features_and_idx(ops):
    B, C, N, k = 2, 7, 130, 10
    x = rnd("ef.x", (B, C, N))
    idx = ops.knn(km.cm_to_pm(x), B, N, k, mode=0)
    loc = ops.idx_to_local64(idx, B, N)
    assert torch.equal(loc, km.idx_to_local64(idx, B, N))
    assert torch.equal(ops.idx_from_local64(loc, B, N, k), idx)
    assert torch.equal(ops.edge_features_cm(x, loc, k), km.edge_features_cm(x, loc, k))


def test_layout(ops):
    x = rnd("lay.x", (3, 37, 130))
    pm = ops.cm_to_pm(x)
    assert torch.equal(pm, km.cm_to_pm(x))
    assert torch.equal(ops.pm_to_cm(pm, 3, 130), x)
    a, b = rnd("lay.a", (100, 3)), rnd("lay.b", (100, 17))
    assert torch.equal(ops.concat2(a, b), km.concat2(a, b))


# ----------------------------------------------------------------------------- gemm_nt
@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (300, 64, 3), (1000, 256, 131), (4, 1, 64), (129, 3, 64), (640, 1024, 256),
                                   (2048, 128, 1280), (32, 512, 1024), (513, 33, 37)])
def test_gemm_nt_plain(ops, M, N, K):
    A, W, b = rnd("g.A%d" % K, (M, K)), rnd("g.W%d" % N, (N, K), 0.2), rnd("g.b", (N,))
    close(ops.gemm_nt(A, W, b), km.gemm_nt(A, W, b), what="plain")
    close(ops.gemm_nt(A, W, None, act=ops.ACT_LRELU, slope=0.01), km.gemm_nt(A, W, None, act=1, slope=0.01), what="lrelu")
    close(ops.gemm_nt(A, W, b, act=ops.ACT_TANH), km.gemm_nt(A, W, b, act=2), what="tanh")
    y, mean, var = ops.gemm_nt(A, W, b, stats=True)
    y2, mean2, var2 = km.gemm_nt(A, W, b, stats=True)
    close(y, y2, what="stats.y"); close(mean, mean2, atol=1e-5, what="stats.mean"); close(var, var2, what="stats.var")


def test_gemm_nt_asymmetric_layout(ops):
    """A=I with an asymmetric W catches row/col swaps of the MFMA C/D layout."""
    K = 64
    A = torch.eye(K, device="cuda")
    W = (torch.arange(48 * K, device="cuda", dtype=torch.float32).view(48, K) / 7.0)
    close(ops.gemm_nt(A, W), W.t(), what="identity")


def test_gemm_nt_views_and_rowbias(ops):
    M, N, K, G = 512, 256, 128, 128
    big = rnd("v.A", (M, 200))
    A = big[:, 40:40 + K]                       # column slice: lda=200, not 16B aligned start? 40*4=160 -> aligned
    Wfull = rnd("v.W", (N, 640), 0.1)
    W = Wfull[:, 512:]                          # the tail.0 split (Generator.py:189,194)
    rb = rnd("v.rb", (M // G, N))
    close(ops.gemm_nt(A, W, None, rowbias=rb, rows_per_group=G, act=ops.ACT_LRELU, slope=0.01),
          km.gemm_nt(A, W, None, rowbias=rb, rows_per_group=G, act=1, slope=0.01), what="rowbias")
    A2 = big[:, 3:3 + K]                        # unaligned view -> scalar load path
    close(ops.gemm_nt(A2, W), km.gemm_nt(A2, W), what="unaligned")


def test_gemm_nt_prologues(ops):
    M, N, K = 700, 128, 64
    A, W, b = rnd("p.A", (M, K)), rnd("p.W", (N, K), 0.2), rnd("p.b", (N,))
    sc, sh = rnd("p.sc", (K,)).abs() + 0.5, rnd("p.sh", (K,), 0.3)
    y, m, v = ops.gemm_nt(A, W, b, pro=(sc, sh, 0.01), stats=True)
    y2, m2, v2 = km.gemm_nt(A, W, b, pro=(sc, sh, 0.01), stats=True)
    close(y, y2, what="affine"); close(m, m2, atol=1e-5); close(v, v2)
    # edge operand
    B, Np, k, H = 2, 200, 10, 32
    P = rnd("p.P", (B * Np, 96))                 # P occupies columns [0,32) of a wider PQR tensor
    idx = ops.knn(rnd("p.x", (B * Np, 3)), B, Np, k, mode=1)
    eb = rnd("p.eb", (H,), 0.1)
    W2 = rnd("p.W2", (64, H), 0.2)
    sc, sh = rnd("p.sc2", (H,)).abs() + 0.5, rnd("p.sh2", (H,), 0.3)
    y, m, v = ops.gemm_nt(P[:, :H], W2, b[:64], pro=(sc, sh, 0.01), edge=(idx, eb), stats=True)
    y2, m2, v2 = km.gemm_nt(P[:, :H], W2, b[:64], pro=(sc, sh, 0.01), edge=(idx, eb), stats=True)
    close(y, y2, what="edge"); close(m, m2, atol=1e-5); close(v, v2)


def test_gemm_nt_backward_epilogues(ops):
    M, N, K = 900, 64, 128
    A, W = rnd("e.A", (M, K)), rnd("e.W", (N, K), 0.2)
    ref = rnd("e.ref", (M, N))
    close(ops.gemm_nt_maskout(A, W, ref, 0.01), km.gemm_nt_maskout(A, W, ref, 0.01), what="maskout")
    sc, sh = rnd("e.sc", (N,)), rnd("e.sh", (N,), 0.3)
    mean, inv = rnd("e.mu", (N,), 0.2), rnd("e.inv", (N,)).abs() + 0.5
    for a, b in zip(ops.gemm_nt_bnbwd(A, W, ref, sc, sh, mean, inv, 0.01), km.gemm_nt_bnbwd(A, W, ref, sc, sh, mean, inv, 0.01)):
        close(a, b, rtol=5e-5, atol=2e-4, what="bnbwd")
    B, Np, k = 2, 150, 10
    idx = ops.knn(rnd("e.x", (B * Np, 3)), B, Np, k, mode=1)
    P = rnd("e.P", (B * Np, N))
    A3 = rnd("e.A3", (B * Np * k, K))
    eb = rnd("e.eb", (N,), 0.1)
    for a, b in zip(ops.gemm_nt_bnbwd(A3, W, P, sc, sh, mean, inv, 0.01, edge=(idx, eb)),
                    km.gemm_nt_bnbwd(A3, W, P, sc, sh, mean, inv, 0.01, edge=(idx, eb))):
        close(a, b, rtol=5e-5, atol=2e-4, what="edge_bnbwd")


# ----------------------------------------------------------------------------- gemm_tn
@pytest.mark.parametrize("M,Na,Nb", [(1000, 64, 3), (4096, 128, 64), (777, 256, 128), (5000, 1024, 256), (32, 512, 1024), (3000, 128, 1280), (100, 1, 64)])
def test_gemm_tn(ops, M, Na, Nb):
    A, Bm = rnd("t.A%d" % Na, (M, Na)), rnd("t.B%d" % Nb, (M, Nb))
    close(ops.gemm_tn(A, Bm), km.gemm_tn(A, Bm), rtol=3e-5, what="tn")
    sc, sh = rnd("t.sc", (Nb,)).abs() + 0.5, rnd("t.sh", (Nb,), 0.3)
    close(ops.gemm_tn(A, Bm, pro=(sc, sh, 0.01)), km.gemm_tn(A, Bm, pro=(sc, sh, 0.01)), rtol=3e-5, what="tn.affine")
    out = rnd("t.out", (Na, Nb)); out2 = out.clone()
    ops.gemm_tn(A, Bm, out=out, beta=1.0); km.gemm_tn(A, Bm, out=out2, beta=1.0)
    close(out, out2, rtol=3e-5, what="tn.beta")


def test_gemm_tn_edge(ops):
    B, Np, k, H, F_ = 2, 200, 10, 32, 64
    idx = ops.knn(rnd("te.x", (B * Np, 3)), B, Np, k, mode=1)
    P = rnd("te.P", (B * Np, H))
    dY = rnd("te.dY", (B * Np * k, F_))
    eb = rnd("te.eb", (H,), 0.1)
    sc, sh = rnd("te.sc", (H,)).abs() + 0.5, rnd("te.sh", (H,), 0.3)
    close(ops.gemm_tn(dY, P, pro=(sc, sh, 0.01), edge=(idx, eb)), km.gemm_tn(dY, P, pro=(sc, sh, 0.01), edge=(idx, eb)), rtol=3e-5, what="tn.edge")


# ----------------------------------------------------------------------------- reductions / norms
def test_reductions(ops):
    X = rnd("r.X", (6 * 300, 70)) + 3.0
    for G in (300, 1800):
        for a, b in zip(ops.colstats(X, G, 0.2), km.colstats(X, G, 0.2)):
            close(a, b, rtol=1e-5, what="colstats")
        close(ops.colsum(X, G), km.colsum(X, G), rtol=1e-5, what="colsum")
    Xv = rnd("r.Xv", (512, 96))[:, 16:80]
    close(ops.colsum(Xv), km.colsum(Xv), rtol=1e-5, what="colsum.view")


def test_bn_and_pool(ops):
    Cn, M = 70, 640
    y = rnd("b.y", (M, Cn)) * 2 + 0.5
    mean, var = km.colstats(y, M)
    gamma, beta = rnd("b.g", (Cn,)).abs() + 0.5, rnd("b.b", (Cn,), 0.2)
    rm, rv = torch.zeros(Cn, device="cuda"), torch.ones(Cn, device="cuda")
    rm2, rv2 = rm.clone(), rv.clone()
    o = ops.bn_prepare(mean[0].contiguous(), var[0].contiguous(), gamma, beta, M, True, rm, rv)
    o2 = km.bn_prepare(mean[0], var[0], gamma, beta, M, True, rm2, rv2)
    for a, b in zip(o, o2):
        close(a, b, rtol=1e-6)
    close(rm, rm2, rtol=1e-6); close(rv, rv2, rtol=1e-6)
    oe = ops.bn_prepare(None, None, gamma, beta, M, False, rm, rv)
    oe2 = km.bn_prepare(None, None, gamma, beta, M, False, rm2, rv2)
    for a, b in zip(oe, oe2):
        close(a, b, rtol=1e-6)
    g = rnd("b.gr", (M, Cn))
    sums = torch.cat([g.sum(0), (g * ((y - o[3]) * o[2])).sum(0)])
    close(ops.bn_bwd_apply(g, y, o[3], o[2], gamma, sums, M), km.bn_bwd_apply(g, y, o[3], o[2], gamma, sums, M), rtol=1e-5)
    B, N = 5, 128
    out, arg = ops.maxpool(y, B, N, o[0], o[1], 0.01)
    out2, arg2 = km.maxpool(y, B, N, o[0], o[1], 0.01)
    close(out, out2, rtol=1e-6); assert torch.equal(arg, arg2)
    out, arg = ops.maxpool(y, B, N)
    out2, arg2 = km.maxpool(y, B, N)
    assert torch.equal(out, out2) and torch.equal(arg, arg2)


def _col_blocks(ops, A, W, hint):
    """N-tiles spgan_gemm_nt would use for a plain product of these operands under the given tile_hint"""
    from spgan import _lib
    import ctypes as C
    a = _lib.GemmNTArgs()
    a.A = A.data_ptr(); a.lda = A.stride(0); a.W = W.data_ptr(); a.ldw = W.stride(0); a.M, a.N, a.K = A.shape[0], W.shape[0], A.shape[1]
    a.Y = A.data_ptr(); a.ldy = W.shape[0]; a.tile_hint = hint
    return _lib.load().spgan_gemm_nt_col_blocks(C.byref(a))


@pytest.mark.parametrize("hint", [0, 2])
@pytest.mark.parametrize("M,N,K", [(128, 64, 32), (129, 65, 36), (1024, 256, 256), (1000, 192, 132), (4096, 1024, 256), (640, 320, 64), (2048, 64, 640),
                                   (257, 128, 1280), (65, 33, 8), (512, 256, 128), (768, 512, 64), (2304, 256, 32)])
def test_gemm_nt_full_and_partial_tiles(ops, M, N, K, hint):
    """Every epilogue on shapes whose output tiles are all inside, all ragged, or mixed (the straight-line path serves the inside
    tiles, the generic path the rest -- both must agree with the model on the same launch), operands as column slices.
    hint = 2: the 256 x 256-tile kernel (csrc/gemm_wide.hip) wherever the shape is eligible -- the same checks against the same models."""
    wide = M % 256 == 0 and N % 256 == 0 and K % 32 == 0
    if hint == 2 and not wide:
        pytest.skip("not a 256 x 256-tile shape")
    with ops.nt_tile_hint(hint):
        if hint == 2:
            wA, wW = rnd("ft.A%d%d" % (M, K), (M, K + 8)), rnd("ft.W%d%d" % (N, K), (N, K + 4), 0.1)
            assert _col_blocks(ops, wA[:, 4:4 + K], wW[:, :K], 2) == N // 256 and _col_blocks(ops, wA[:, 4:4 + K], wW[:, :K], 1) > N // 256
        _full_and_partial_tiles(ops, M, N, K)


def _full_and_partial_tiles(ops, M, N, K):
    wideA, wideW = rnd("ft.A%d%d" % (M, K), (M, K + 8)), rnd("ft.W%d%d" % (N, K), (N, K + 4), 0.1)
    A, W = wideA[:, 4:4 + K], wideW[:, :K]
    b = rnd("ft.b%d" % N, (N,))
    for act in (0, 1, 2):
        close(ops.gemm_nt(A, W, b, act=act, slope=0.2), km.gemm_nt(A, W, b, act=act, slope=0.2), rtol=5e-5, what="act%d" % act)
    sc, sh = rnd("ft.sc%d" % K, (K,)).abs() + 0.5, rnd("ft.sh%d" % K, (K,), 0.3)
    y, m, v = ops.gemm_nt(A, W, b, pro=(sc, sh, 0.01), stats=True)
    y2, m2, v2 = km.gemm_nt(A, W, b, pro=(sc, sh, 0.01), stats=True)
    close(y, y2, rtol=5e-5, what="affine"); close(m, m2, atol=2e-5, what="mean"); close(v, v2, rtol=1e-4, what="var")
    for G_ in (1, 64, 128, 256):                       # dense addend / two groups per tile / one group per tile / per two tiles
        rb = rnd("ft.rb%d.%d.%d" % (M, N, G_), ((M + G_ - 1) // G_, N + 4))[:, :N]
        close(ops.gemm_nt(A, W, b, rowbias=rb, rows_per_group=G_, act=1, slope=0.1), km.gemm_nt(A, W, b, rowbias=rb, rows_per_group=G_, act=1, slope=0.1),
              rtol=5e-5, what="rowbias G=%d" % G_)
        for name, a_, b_ in zip(("g", "s0", "s1"), ops.gemm_nt_bnbwd(A, W, rnd("ft.ref2%d%d" % (M, N), (M, N)), rnd("ft.q%d" % N, (N,)), rnd("ft.r%d" % N, (N,), 0.3),
                                                                      rnd("ft.s%d" % N, (N,), 0.2), rnd("ft.t%d" % N, (N,)).abs() + 0.5, 0.01, rowadd=rb.contiguous() if G_ == 1 else None),
                                km.gemm_nt_bnbwd(A, W, rnd("ft.ref2%d%d" % (M, N), (M, N)), rnd("ft.q%d" % N, (N,)), rnd("ft.r%d" % N, (N,), 0.3),
                                                 rnd("ft.s%d" % N, (N,), 0.2), rnd("ft.t%d" % N, (N,)).abs() + 0.5, 0.01, rowadd=rb.contiguous() if G_ == 1 else None)):
            close(a_, b_, rtol=1e-4, atol=5e-4, what="bnbwd rowadd " + name)
    out = torch.full((M, N + 12), 9.0, device="cuda")
    ops.gemm_nt(A, W, b, out=out[:, 4:4 + N])
    close(out[:, 4:4 + N], km.gemm_nt(A, W, b), rtol=5e-5, what="strided out")
    assert (out[:, :4] == 9).all() and (out[:, 4 + N:] == 9).all()
    ref = rnd("ft.ref%d%d" % (M, N), (M, N + 4))[:, :N]
    close(ops.gemm_nt_maskout(A, W, ref, 0.01), km.gemm_nt_maskout(A, W, ref, 0.01), rtol=5e-5, what="maskout")
    bsc, bsh = rnd("ft.bsc%d" % N, (N,)), rnd("ft.bsh%d" % N, (N,), 0.3)
    mean, inv = rnd("ft.mu%d" % N, (N,), 0.2), rnd("ft.inv%d" % N, (N,)).abs() + 0.5
    for name, a_, b_ in zip(("g", "s0", "s1"), ops.gemm_nt_bnbwd(A, W, ref, bsc, bsh, mean, inv, 0.01, bias=b),
                            km.gemm_nt_bnbwd(A, W, ref, bsc, bsh, mean, inv, 0.01, bias=b)):
        close(a_, b_, rtol=1e-4, atol=5e-4, what="bnbwd " + name)
    if M % 128 == 0 and M > 64:
        gamma, beta = rnd("ft.ga%d" % N, (N,)).abs() + 0.5, rnd("ft.be%d" % N, (N,), 0.1)
        rows = 128 if M % 256 else 256
        got = ops.gemm_bn_pool(A, W, b, (gamma, beta, None, None), rows, 0.01, pro=(sc, sh, 0.01), keep_y=True)
        want = km.gemm_bn_pool(A, W, b, (gamma, beta, None, None), rows, 0.01, pro=(sc, sh, 0.01), keep_y=True)
        close(got[0], want[0], rtol=5e-5, what="pool y")
        for g_, w_ in zip(got[1], want[1]):
            close(g_, w_, rtol=1e-4, atol=2e-5, what="pool bn")
        close(got[2], want[2], rtol=1e-4, atol=2e-5, what="pooled")
    A2, B2 = rnd("ft.ta%d" % M, (M, N)), rnd("ft.tb%d" % M, (M, K))
    close(ops.gemm_tn(A2, B2), A2.double().t().matmul(B2.double()).float(), rtol=5e-5, atol=1e-4, what="tn")
