"""GPU: EdgeBlock gather-side kernels, AdaIN, pooled-BN backward, double-backward helpers, Adam --
each HIP kernel (through the C ABI) against its plain-PyTorch model."""
import pytest
import torch

import kernel_model as km
from spgan import fixture_rng as fr
from test_kernels_gpu import close, rnd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from spgan import ops as o
    from spgan import _lib
    _lib.load()
    return o


def _graph(ops, B, N, k, tag):
    x = rnd("k2.x." + tag, (B * N, 3))
    idx = ops.knn(x, B, N, k, mode=1)
    return idx, ops.csr_build(idx, B, N)


@pytest.mark.parametrize("H,F_,C", [(32, 64, 3), (64, 128, 64)])
def test_edge_wcat(ops, H, F_, C):
    Ww0, Wx = rnd("wc.a", (H, C)), rnd("wc.b", (F_, 2 * C))
    assert torch.equal(ops.edge_wcat(Ww0, Wx), km.edge_wcat(Ww0, Wx))
    w, wt = ops.edge_wcat(Ww0, Wx, transposed=True)
    assert torch.equal(w, km.edge_wcat(Ww0, Wx)) and torch.equal(wt, w.t().contiguous())
    d = rnd("wc.d", (H + 2 * F_, C))
    for a, b in zip(ops.edge_wcat_bwd(d, H, F_), km.edge_wcat_bwd(d, H, F_)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("F_,k", [(64, 20), (128, 10), (24, 3)])
def test_conv_out_weight_pm(ops, F_, k):
    """conv_out.weight [F,F,1,k] -> Wo [F, k*F] (K index r*F + c) and its transpose, one launch."""
    w = rnd("cow.%d.%d" % (F_, k), (F_, F_, 1, k))
    wo, wot = ops.conv_out_weight_pm(w)
    ref, ref_t = km.conv_out_weight_pm(w)
    assert torch.equal(wo, ref) and torch.equal(wot, ref_t)


def test_multi_add_strided_pairs(ops):
    """spgan_multi_add3: a permuted source (conv_out gradient [F,k,F] into [F,F,1,k]), column blocks of one destination, a plain pair
    with differing shapes -- all in one launch, against torch's own accumulation."""
    F_, k = 64, 10
    dst_w = rnd("ma3.w", (F_, F_, 1, k)); g = rnd("ma3.g", (F_, k * F_))
    dst_cat = rnd("ma3.c", (128, 192, 1)); a, b = rnd("ma3.a", (128, 64)), rnd("ma3.b", (128, 128))
    dst_p = rnd("ma3.p", (37, 5)); c = rnd("ma3.pp", (185,))
    want_w = dst_w + g.view(F_, k, F_).permute(0, 2, 1).unsqueeze(2)
    want_cat = dst_cat + torch.cat([a, b], dim=1).unsqueeze(2)
    want_p = dst_p + c.view(37, 5)
    flat = dst_cat.view(128, 192)
    ops.multi_add([dst_w, flat[:, :64], flat[:, 64:], dst_p], [g.view(F_, k, F_).permute(0, 2, 1).unsqueeze(2), a, b, c])
    assert torch.equal(dst_w, want_w) and torch.equal(dst_cat, want_cat) and torch.equal(dst_p, want_p)
    with pytest.raises(ValueError):
        ops.multi_add([dst_p], [rnd("ma3.bad", (5, 37)).t()[:, :4]])


def test_rowscale_outer_accumulates_in_place(ops):
    X, a, b, d, v = rnd("rso.x", (96, 40)), rnd("rso.a", (96,)), rnd("rso.b", (96,)), rnd("rso.d", (96,)), rnd("rso.v", (40,))
    base = rnd("rso.base", (96, 40))
    want = base + km.rowscale_outer(X, a, b, d, v)
    out = ops.rowscale_outer(X, a, b, d, v, out=base.clone(), accumulate=True)
    close(out, want, rtol=1e-6, atol=1e-6)
    assert torch.equal(ops.rowscale_outer(X, a, out=torch.empty_like(X)), ops.rowscale_outer(X, a))


@pytest.mark.parametrize("B,N,k,H,F_", [(2, 200, 10, 32, 64), (2, 130, 10, 64, 128), (1, 77, 5, 16, 32)])
def test_edge_pipeline_kernels(ops, B, N, k, H, F_):
    M = B * N
    idx, (rowptr, src) = _graph(ops, B, N, k, "%d" % N)
    PQR = rnd("ep.PQR%d" % N, (M, H + 2 * F_))
    b1, bx = rnd("ep.b1", (H,), 0.1), rnd("ep.bx", (F_,), 0.1)
    for a, b in zip(ops.edge_stats(PQR, idx, b1, bx), km.edge_stats(PQR, idx, b1, bx)):
        close(a, b, rtol=2e-5, atol=2e-6, what="edge_stats")
    h2 = rnd("ep.h2%d" % N, (M * k, F_))
    sc2, sh2 = rnd("ep.sc2", (F_,)).abs() + 0.5, rnd("ep.sh2", (F_,), 0.3)
    scx, shx = rnd("ep.scx", (F_,)).abs() + 0.5, rnd("ep.shx", (F_,), 0.3)
    T = ops.edge_attend_fwd(h2, sc2, sh2, PQR, idx, bx, scx, shx, 0.01)
    close(T, km.edge_attend_fwd(h2, sc2, sh2, PQR, idx, bx, scx, shx, 0.01), rtol=1e-5, what="attend_fwd")
    dT = rnd("ep.dT%d" % N, (M, k * F_))
    m2, i2 = rnd("ep.m2", (F_,), 0.2), rnd("ep.i2", (F_,)).abs() + 0.5
    mx, ix = rnd("ep.mx", (F_,), 0.2), rnd("ep.ix", (F_,)).abs() + 0.5
    got = ops.edge_attend_bwd(dT, h2, sc2, sh2, m2, i2, PQR, idx, bx, scx, shx, mx, ix, 0.01)
    ref = km.edge_attend_bwd(dT, h2, sc2, sh2, m2, i2, PQR, idx, bx, scx, shx, mx, ix, 0.01)
    for a, b, w in zip(got, ref, ("g2", "gy", "sums2", "sumsy")):
        close(a, b, rtol=5e-5, atol=5e-5, what="attend_bwd." + w)
    g1 = rnd("ep.g1%d" % N, (M * k, H))
    gam1, gamx = rnd("ep.gam1", (H,)).abs() + 0.5, rnd("ep.gamx", (F_,)).abs() + 0.5
    m1, i1 = rnd("ep.m1", (H,), 0.2), rnd("ep.i1", (H,)).abs() + 0.5
    s1, sx = rnd("ep.s1", (2 * H,), 3.0), rnd("ep.sx", (2 * F_,), 3.0)
    d1 = ops.edge_scatter(g1, got[1], PQR, idx, rowptr, src, b1, m1, i1, gam1, s1, bx, mx, ix, gamx, sx)
    d2 = km.edge_scatter(g1, got[1], PQR, idx, rowptr, src, b1, m1, i1, gam1, s1, bx, mx, ix, gamx, sx)
    close(d1, d2, rtol=2e-5, atol=1e-5, what="edge_scatter")
    # determinism: two runs are bit-identical (no float atomics anywhere)
    assert torch.equal(d1, ops.edge_scatter(g1, got[1], PQR, idx, rowptr, src, b1, m1, i1, gam1, s1, bx, mx, ix, gamx, sx))


@pytest.mark.parametrize("B,N,C", [(3, 200, 64), (2, 128, 128), (2, 77, 20)])
def test_adain(ops, B, N, C):
    M = B * N
    x, gb, dout = rnd("ad.x%d" % C, (M, C)), rnd("ad.gb%d" % C, (M, 2 * C)), rnd("ad.do%d" % C, (M, C))
    for slope in (1.0, 0.2):
        mean, var = km.colstats(x, N, slope)
        mean, var = mean.contiguous(), var.contiguous()
        close(ops.adain_fwd(x, N, slope, mean, var, gb), km.adain_fwd(x, N, slope, mean, var, gb), rtol=1e-6, what="adain_fwd")
        for a, b, w in zip(ops.adain_bwd(dout, x, N, slope, mean, var, gb), km.adain_bwd(dout, x, N, slope, mean, var, gb), ("dx", "dgb")):
            close(a, b, rtol=2e-5, atol=1e-5, what="adain_bwd." + w)


def test_pool_bn_backward(ops):
    B, N, C = 4, 160, 200
    M = B * N
    y = rnd("pb.y", (M, C)) * 2
    mean, var = km.colstats(y, M)
    gamma, beta = rnd("pb.g", (C,)).abs() + 0.5, rnd("pb.b", (C,), 0.2)
    sc, sh, inv, mu = km.bn_prepare(mean[0], var[0], gamma, beta, M)
    pooled, arg = km.maxpool(y, B, N, sc, sh, 0.01)
    gpool = rnd("pb.gp", (B, C))
    gv, sums = ops.pool_bwd_stats(gpool, pooled, arg, y, mu, inv, 0.01)
    gv2, sums2 = km.pool_bwd_stats(gpool, pooled, arg, y, mu, inv, 0.01)
    close(gv, gv2, rtol=1e-6); close(sums, sums2, rtol=1e-5, atol=1e-5)
    close(ops.bn_bwd_apply_sparse(gv, arg, y, N, mu, inv, gamma, sums, M), km.bn_bwd_apply_sparse(gv2, arg, y, N, mu, inv, gamma, sums2, M), rtol=1e-5)
    dst = rnd("pb.dst", (M, C)); dst2 = dst.clone()
    ops.maxpool_bwd_add(gpool, arg, dst); km.maxpool_bwd_add(gpool, arg, dst2)
    assert torch.equal(dst, dst2)
    assert torch.equal(ops.scatter_rows(gv, arg, M), km.scatter_rows(gv, arg, M))
    assert torch.equal(ops.gather_rows(y, arg), km.gather_rows(y, arg))


def test_sparse_affine_operand(ops):
    """The lazily evaluated BatchNorm-backward operand inside gemm_tn / gemm_nt_bnbwd == the materialised one."""
    B, N, C, Cp = 4, 160, 200, 72
    M = B * N
    y = rnd("sa.y", (M, C)) * 2
    mean, var = km.colstats(y, M)
    gamma, beta = rnd("sa.g", (C,)).abs() + 0.5, rnd("sa.b", (C,), 0.2)
    sc, sh, inv, mu = km.bn_prepare(mean[0], var[0], gamma, beta, M)
    pooled, arg = km.maxpool(y, B, N, sc, sh, 0.01)
    gval, sums = km.pool_bwd_stats(rnd("sa.gp", (B, C)), pooled, arg, y, mu, inv, 0.01)
    dense = km.bn_bwd_apply_sparse(gval, arg, y, N, mu, inv, gamma, sums, M)
    sa = ops.sparse_bn_bwd_operand(gval, arg, y, N, mu, inv, gamma, sums, M)
    Bm = rnd("sa.B", (M, Cp))
    psc, psh = rnd("sa.psc", (Cp,)).abs() + 0.5, rnd("sa.psh", (Cp,), 0.3)
    close(ops.gemm_tn(sa, Bm, pro=(psc, psh, 0.01)), km.gemm_tn(dense, Bm, pro=(psc, psh, 0.01)), rtol=5e-5, what="tn.sparse_affine")
    W = rnd("sa.W", (Cp, C), 0.2)
    yref = rnd("sa.yref", (M, Cp))
    m2, i2 = rnd("sa.m2", (Cp,), 0.2), rnd("sa.i2", (Cp,)).abs() + 0.5
    for a, b in zip(ops.gemm_nt_bnbwd(sa, W, yref, psc, psh, m2, i2, 0.01), km.gemm_nt_bnbwd(dense, W, yref, psc, psh, m2, i2, 0.01)):
        close(a, b, rtol=5e-5, atol=2e-4, what="nt_bnbwd.sparse_affine")
    # aligned fast path (C % 4 == 0) and the generic path (odd leading dimension) agree
    y2 = y[:, :197].contiguous()
    sa2 = ops.SparseAffine(y2, sa.alpha[:197].contiguous(), sa.beta[:197].contiguous(), sa.sp_val[:, :197].contiguous(), arg[:, :197].contiguous(), N)
    d2 = dense[:, :197].contiguous()
    close(ops.gemm_tn(sa2, Bm), km.gemm_tn(d2, Bm), rtol=5e-5, what="tn.sparse_affine.unaligned")


def test_sparse_affine_operand_wide_tiles(ops):
    """The same lazily evaluated operand through the 256 x 256-tile kernel (tile_hint 2; shapes = whole tiles, 256-row shapes)."""
    B, N, C, Cp = 3, 256, 256, 256
    M = B * N
    y = rnd("saw.y", (M, C)) * 2
    mean, var = km.colstats(y, M)
    gamma, beta = rnd("saw.g", (C,)).abs() + 0.5, rnd("saw.b", (C,), 0.2)
    sc, sh, inv, mu = km.bn_prepare(mean[0], var[0], gamma, beta, M)
    pooled, arg = km.maxpool(y, B, N, sc, sh, 0.01)
    gval, sums = km.pool_bwd_stats(rnd("saw.gp", (B, C)), pooled, arg, y, mu, inv, 0.01)
    dense = km.bn_bwd_apply_sparse(gval, arg, y, N, mu, inv, gamma, sums, M)
    sa = ops.sparse_bn_bwd_operand(gval, arg, y, N, mu, inv, gamma, sums, M)
    W = rnd("saw.W", (Cp, C), 0.2)
    yref = rnd("saw.yref", (M, Cp))
    psc, psh = rnd("saw.psc", (Cp,)).abs() + 0.5, rnd("saw.psh", (Cp,), 0.3)
    m2, i2 = rnd("saw.m2", (Cp,), 0.2), rnd("saw.i2", (Cp,)).abs() + 0.5
    want = km.gemm_nt_bnbwd(dense, W, yref, psc, psh, m2, i2, 0.01)
    for hint in (1, 2):
        with ops.nt_tile_hint(hint):
            for a, b in zip(ops.gemm_nt_bnbwd(sa, W, yref, psc, psh, m2, i2, 0.01), want):
                close(a, b, rtol=5e-5, atol=2e-4, what="nt_bnbwd.sparse_affine hint %d" % hint)


def test_double_backward_helpers(ops):
    M, C = 700, 96
    u, y, gz = rnd("db.u", (M, C)), rnd("db.y", (M, C)) * 2, rnd("db.gz", (M, C))
    mean, var = km.colstats(y, M)
    gamma, beta = rnd("db.g", (C,)).abs() + 0.5, rnd("db.b", (C,), 0.2)
    sc, sh, inv, mu = km.bn_prepare(mean[0], var[0], gamma, beta, M)
    got = ops.bn_dbl_stats(u, y, gz, mu, inv)
    ref = km.bn_dbl_stats(u, y, gz, mu, inv)
    for a, b in zip(got, ref):
        close(a, b, rtol=2e-5, atol=2e-4, what="bn_dbl_stats")
    S1 = rnd("db.S1", (C,), 3.0)
    for a, b in zip(ops.bn_dbl_apply(u, y, gz, mu, inv, sc, sh, 0.01, gamma, S1, ref[0], ref[1], M),
                    km.bn_dbl_apply(u, y, gz, mu, inv, sc, sh, 0.01, gamma, S1, ref[0], ref[1], M)):
        close(a, b, rtol=1e-5, what="bn_dbl_apply")
    close(ops.col_scale_add(u, gz, gamma), km.col_scale_add(u, gz, gamma), rtol=1e-6)
    for act in (0, 1, 2):
        close(ops.act_bwd(u, torch.tanh(y), act, 0.01), km.act_bwd(u, torch.tanh(y), act, 0.01), rtol=1e-6)
    close(ops.tanh_bwd(u, torch.tanh(y)), km.tanh_bwd(u, torch.tanh(y)), rtol=1e-6)


def test_fused_bn_finalize_and_small_kernels(ops):
    M, N, K = 1000, 96, 64
    A, W, b = rnd("fb.A", (M, K)), rnd("fb.W", (N, K), 0.2), rnd("fb.b", (N,))
    gamma, beta = rnd("fb.g", (N,)).abs() + 0.5, rnd("fb.be", (N,), 0.2)
    rm, rv = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
    rm2, rv2 = rm.clone(), rv.clone()
    y, st = ops.gemm_nt(A, W, b, bn=(gamma, beta, rm, rv))
    y2, st2 = km.gemm_nt(A, W, b, bn=(gamma, beta, rm2, rv2))
    close(y, y2, what="gemm_bn.y")
    for a, c in zip(st, st2):
        close(a, c, rtol=2e-5, atol=1e-6, what="gemm_bn.stats")
    close(rm, rm2, rtol=1e-5, atol=1e-7); close(rv, rv2, rtol=1e-5)
    y, st = ops.gemm_nt(A, W, b, bn=(gamma, beta, None, None))          # no running stats
    close(st[0], st2[0], rtol=2e-5)
    C = 70
    vec = [rnd("fb.v%d" % i, (C,)) for i in range(7)]
    vec[5] = vec[5].abs() + 0.5; vec[6] = vec[6].abs() + 0.5
    co, co2 = ops.bn_dbl_coeffs(*vec, 640), km.bn_dbl_coeffs(*vec, 640)
    close(co, co2, rtol=1e-5, atol=1e-6, what="bn_dbl_coeffs")
    s0, s1 = rnd("fb.s0", (C,)), rnd("fb.s1", (C,))
    for a, c in zip(ops.bn_dbl_phaseb(co2.contiguous(), vec[5], vec[6], s0, s1), km.bn_dbl_phaseb(co2, vec[5], vec[6], s0, s1)):
        close(a, c, rtol=1e-5, atol=1e-6, what="bn_dbl_phaseb")
    for a, c in zip(ops.bn_dbl_phaseb(co2.contiguous(), vec[5], vec[6], None, None), km.bn_dbl_phaseb(co2, vec[5], vec[6], None, None)):
        close(a, c, rtol=1e-5, atol=1e-6, what="bn_dbl_phaseb.none")
    for sa, sb in ((s0, s1), (None, None)):      # both steps as one launch: the same values as the two-launch route
        for a, c in zip(ops.bn_dbl_phaseb(tuple(vec[:5]) + (640,), vec[5], vec[6], sa, sb), ops.bn_dbl_phaseb(co, vec[5], vec[6], sa, sb)):
            close(a, c, rtol=1e-6, atol=1e-7, what="bn_dbl_phaseb from sums")
    dsts = [rnd("fb.d%d" % i, (n,)) for i, n in enumerate((5, 1000, 70001, 3))]
    srcs = [rnd("fb.s%d" % i, (n,)) for i, n in enumerate((5, 1000, 70001, 3))]
    ref = [d + s for d, s in zip(dsts, srcs)]
    ops.multi_add(dsts, srcs)
    for a, c in zip(dsts, ref):
        assert torch.equal(a, c)


def test_adam_and_axpby(ops):
    n = 100003
    p, g = rnd("am.p", (n,)), rnd("am.g", (n,), 0.01)
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    p2, m2, v2 = p.clone(), m.clone(), v.clone()
    for step in (1, 2, 3):
        ops.adam_step(p, g, m, v, step)
        km.adam_step(p2, g, m2, v2, step)
    close(p, p2, rtol=1e-6); close(m, m2, rtol=1e-6); close(v, v2, rtol=1e-6)
    y = rnd("am.y", (n,)); y2 = y.clone()
    ops.axpby(0.5, g, 2.0, y); km.axpby(0.5, g, 2.0, y2)
    close(y, y2, rtol=1e-6)


@pytest.mark.parametrize("rows,cols", [(64, 2048), (33, 4096), (17, 1000), (5, 37), (3, 8192)])
def test_softmax_rows(ops, rows, cols):
    S = rnd("sm.s.%d.%d" % (rows, cols), (rows, cols)) * 4.0
    ref = torch.softmax(S.double(), -1)
    P = ops.softmax_rows(S.clone())
    assert (P.double() - ref).abs().max().item() <= 2e-7
    assert (P.sum(-1) - 1).abs().max().item() <= 1e-5
    dP = rnd("sm.d.%d.%d" % (rows, cols), (rows, cols))
    want = km.softmax_rows_bwd(P.double(), dP.double().clone())
    got = ops.softmax_rows_bwd(P, dP.clone())
    close(got, want.float(), 1e-5)


@pytest.mark.parametrize("n", [4, 640 * 256, 640 * 2048 * 3 + 8])
def test_scale_residual(ops, n):
    o, x, dy = rnd("sr.o%d" % n, (n,)), rnd("sr.x%d" % n, (n,)), rnd("sr.d%d" % n, (n,))
    gamma = torch.tensor(0.37, device="cuda")
    assert (ops.scale_residual(o, x, gamma) - (gamma * o + x)).abs().max().item() <= 1e-6
    d_o, dg = ops.scale_residual_bwd(dy, o, gamma)
    close(d_o, gamma * dy, 1e-6)
    want = (dy.double() * o.double()).sum().item()
    assert abs(dg.item() - want) <= 1e-5 * max(1.0, abs(want)) + 1e-3 * (n ** 0.5) * 1e-3


def test_gemm_nt_into_column_slice(ops):
    A, W = rnd("gs.a", (300, 96)), rnd("gs.w", (80, 96))
    buf = torch.full((300, 480), 7.0, device="cuda")
    ops.gemm_nt(A, W, out=buf[:, 160:240])
    close(buf[:, 160:240], km.gemm_nt(A, W), 1e-5)
    assert (buf[:, :160] == 7).all() and (buf[:, 240:] == 7).all()
    big = rnd("gs.b", (2048, 2048)); th = rnd("gs.t", (2048, 80))
    out = torch.full((2048, 480), 3.0, device="cuda")
    ops.gemm_tn(big, th, out=out[:, 80:160])
    close(out[:, 80:160], (big.double().t() @ th.double()).float(), 2e-5)
    assert (out[:, :80] == 3).all() and (out[:, 160:] == 3).all()


@pytest.mark.parametrize("Z,M,N,K", [(3, 256, 256, 80), (2, 300, 77, 50), (4, 2048, 80, 2048), (2, 40, 320, 64)])
def test_gemm_nt_batched(ops, Z, M, N, K):
    A, W = rnd("gb.a%d" % M, (Z, M, K)), rnd("gb.w%d" % M, (Z, N, K))
    want = torch.bmm(A.double(), W.double().transpose(1, 2)).float()
    close(ops.gemm_nt_batched(A, W), want, 2e-5)
    # strided destination (column slice of a wider buffer) and strided operands
    buf = torch.full((Z, M, N + 24), 5.0, device="cuda")
    ops.gemm_nt_batched(A, W, out=buf[:, :, 8:8 + N])
    close(buf[:, :, 8:8 + N], want, 2e-5)
    assert (buf[:, :, :8] == 5).all() and (buf[:, :, 8 + N:] == 5).all()
    wide = rnd("gb.wide%d" % M, (Z, M, K + 16))
    close(ops.gemm_nt_batched(wide[:, :, 8:8 + K], W), torch.bmm(wide[:, :, 8:8 + K].double(), W.double().transpose(1, 2)).float(), 2e-5)
    # one weight shared by all products (batch stride 0)
    close(ops.gemm_nt_batched(A, W[:1].expand(Z, N, K)), torch.matmul(A.double(), W[0].double().t()).float(), 2e-5)


def test_gemm_tn_deferred_reduce(ops):
    """gemm_tn(defer=True) + flush_tn(): one batched launch finishes several weight gradients bit-identically to the immediate path."""
    shapes = [(65536, 256, 128), (4096, 64, 3), (1000, 100, 36), (32, 512, 1024), (20480, 128, 640)]
    now, later = [], []
    for i, (M, Na, Nb) in enumerate(shapes):
        A, B = rnd("dt.a%d" % i, (M, Na)), rnd("dt.b%d" % i, (M, Nb))
        now.append(ops.gemm_tn(A, B))
        later.append(ops.gemm_tn(A, B, defer=True))
    acc0 = rnd("dt.acc", (256, 128)); acc1 = acc0.clone()
    A, B = rnd("dt.a0", (65536, 256)), rnd("dt.b0", (65536, 128))
    ops.gemm_tn(A, B, out=acc0, beta=1.0)
    ops.gemm_tn(A, B, out=acc1, beta=1.0, defer=True)
    ops.flush_tn()
    ops.flush_tn()                                                   # idempotent
    for a, b in zip(now, later):
        assert torch.equal(a, b)
    assert torch.equal(acc0, acc1)


def test_multi_transpose(ops):
    shapes = [(64, 3), (128, 64), (1024, 256), (257, 33), (1, 7), (512, 1024)]
    srcs = [rnd("mt.%d" % i, s) for i, s in enumerate(shapes)]
    wide = rnd("mt.wide", (256, 640))
    srcs.append(wide[:, 512:])                               # column slice: row stride 640
    outs = ops.multi_transpose(srcs)
    for w, t in zip(srcs, outs):
        assert t.is_contiguous() and torch.equal(t, w.t().contiguous())
    many = [rnd("mt.m%d" % i, (8 + i, 5 + (i % 7))) for i in range(70)]      # more than one launch (64 per launch)
    for w, t in zip(many, ops.multi_transpose(many)):
        assert torch.equal(t, w.t().contiguous())


@pytest.mark.parametrize("M,Na,Nb", [(65536, 64, 3), (65536, 3, 64), (2048, 160, 3), (1000, 1, 37), (777, 129, 4), (300, 4, 4), (65536, 3, 256)])
def test_gemm_tn_skinny(ops, M, Na, Nb):
    """Weight gradients of the 3-channel layers (one operand with <= 4 columns): the streaming kernel instead of MFMA tiles."""
    A, B = rnd("sk.a%d%d" % (M, Na), (M, Na + 4))[:, :Na], rnd("sk.b%d%d" % (M, Nb), (M, Nb + 5))[:, 2:2 + Nb]
    want = A.double().t().matmul(B.double()).float()
    close(ops.gemm_tn(A, B), want, rtol=3e-5, atol=2e-4, what="skinny")
    acc = rnd("sk.c%d%d" % (Na, Nb), (Na, Nb)); acc2 = acc.clone()
    ops.gemm_tn(A, B, out=acc, beta=1.0)
    close(acc, acc2 + want, rtol=3e-5, atol=2e-4, what="skinny beta")
    d = ops.gemm_tn(A, B, defer=True); ops.flush_tn()
    assert torch.equal(d, ops.gemm_tn(A, B))
    sc, sh = rnd("sk.sc%d" % Nb, (Nb,)).abs() + 0.5, rnd("sk.sh%d" % Nb, (Nb,), 0.3)       # a prologue keeps the MFMA kernel on the skinny plan
    close(ops.gemm_tn(A, B, pro=(sc, sh, 0.01)), km.gemm_tn(A, B, pro=(sc, sh, 0.01)), rtol=5e-5, atol=3e-4, what="skinny plan, mfma kernel")


@pytest.mark.parametrize("parts,n", [(2, 8), (4, 1000), (8, 245088), (8, 146289), (3, 7)])
def test_reduce_chunks(ops, parts, n):
    """The local sum of a one-hop all-reduce: rows added in ascending order (bit-identical to that fixed-order sum)."""
    n4 = (n + 3) // 4 * 4
    recv = rnd("rc.%d.%d" % (parts, n), (parts, n4))
    got = ops.reduce_chunks(recv)
    want = recv[0].clone()
    for j in range(1, parts):
        want = want + recv[j]
    assert torch.equal(got, want)
    out = torch.empty(n4, device="cuda")
    assert ops.reduce_chunks(recv, out) is out and torch.equal(out, want)
    if n != n4:                                             # a row length that is not a multiple of 4: the scalar tail path
        r2 = rnd("rc2.%d.%d" % (parts, n), (parts, n))
        if r2.data_ptr() % 16 == 0:
            w2 = r2[0].clone()
            for j in range(1, parts):
                w2 = w2 + r2[j]
            assert torch.equal(ops.reduce_chunks(r2), w2)


@pytest.mark.parametrize("hint", [0, 2])
@pytest.mark.parametrize("groups,Mg,N,K", [(3, 512, 256, 128), (2, 768, 128, 64), (3, 256, 1024, 256), (1, 512, 256, 32), (3, 20480, 128, 64),
                                           (2, 16384, 256, 32), (5, 256, 64, 32),
                                           (3, 24576, 256, 128)])   # 96 tiles of 256 x 256 per pass, 288 grouped: straddles the tile-size rule
def test_gemm_bn_groups(ops, groups, Mg, N, K, hint):
    """Several passes through one layer as ONE product (per-pass BatchNorm of the operand and of the output, running statistics
    updated pass after pass): against `groups` separate calls of the single-pass ops -- the activations bit for bit."""
    M = groups * Mg
    A, W, b = rnd("gg.A%d.%d" % (M, K), (M, K)), rnd("gg.W%d.%d" % (N, K), (N, K), 0.1), rnd("gg.b%d" % N, (N,))
    A = A + torch.arange(groups, device="cuda").repeat_interleave(Mg).view(M, 1) * 0.5          # the passes have different statistics
    sc, sh = rnd("gg.sc%d%d" % (groups, K), (groups, K)).abs() + 0.5, rnd("gg.sh%d%d" % (groups, K), (groups, K), 0.3)
    gamma, beta = rnd("gg.ga%d" % N, (N,)).abs() + 0.5, rnd("gg.be%d" % N, (N,), 0.1)
    with ops.nt_tile_hint(hint):
        for pro in (None, (sc, sh, 0.01)):
            rm, rv = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
            Y, out = ops.gemm_bn_groups(A, W, b, (gamma, beta, rm, rv), groups, pro=pro)
            rm2, rv2 = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
            for g in range(groups):
                pg = None if pro is None else (pro[0][g].contiguous(), pro[1][g].contiguous(), pro[2])
                y1, st1 = ops.gemm_nt(A[g * Mg:(g + 1) * Mg], W, b, pro=pg, bn=(gamma, beta, rm2, rv2))
                assert torch.equal(Y[g * Mg:(g + 1) * Mg], y1), "pass %d: activations differ" % g
                for q in range(4):
                    assert torch.equal(out[q, g], st1[q]), (g, q)
            assert torch.equal(rm, rm2) and torch.equal(rv, rv2)
            Ym, outm = km.gemm_bn_groups(A, W, b, (gamma, beta, torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")), groups, pro=pro)
            close(Y, Ym, rtol=5e-5, what="Y vs model"); close(out, outm, rtol=2e-4, atol=2e-5, what="bn vs model")
        rows = 256
        rm, rv = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
        out, pooled, arg, yarg = ops.gemm_bn_groups(A, W, b, (gamma, beta, rm, rv), groups, pro=(sc, sh, 0.01), rows=rows, slope=0.01)
        rm2, rv2 = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
        Bg = Mg // rows
        for g in range(groups):
            _, st1, p1, a1, ya1 = ops.gemm_bn_pool(A[g * Mg:(g + 1) * Mg], W, b, (gamma, beta, rm2, rv2), rows, 0.01,
                                                   pro=(sc[g].contiguous(), sh[g].contiguous(), 0.01))
            assert torch.equal(pooled[g * Bg:(g + 1) * Bg], p1) and torch.equal(arg[g * Bg:(g + 1) * Bg], a1) and torch.equal(yarg[g * Bg:(g + 1) * Bg], ya1)
            for q in range(4):
                assert torch.equal(out[q, g], st1[q]), (g, q)
        assert torch.equal(rm, rm2) and torch.equal(rv, rv2)


@pytest.mark.parametrize("M,C,Nout", [(2048, 256, 128), (1024, 128, 64), (700, 132, 40), (4096, 64, 256)])
def test_bn_bwd_lazy_operand(ops, M, C, Nout):
    """ops.bn_bwd_lazy: the BatchNorm-backward tensor dy = p*g + q*y + r evaluated on the operand loads of its consumers
    (spgan_gemm_tn_args.A2: weight gradient, plain / affine / per-edge B side; spgan_gemm_nt_args.A2: input gradient with the
    BNBWD and EDGE_BNBWD epilogues, and plain) against the materialised bn_bwd_apply."""
    g, y = rnd("lz.g%d.%d" % (M, C), (M, C)), rnd("lz.y%d.%d" % (M, C), (M, C), 2.0) + 0.3
    mean, var = y.mean(0), y.var(0, unbiased=False)
    inv = 1.0 / torch.sqrt(var + 1e-5)
    gamma = rnd("lz.ga%d" % C, (C,)).abs() + 0.5
    xh = (y - mean) * inv
    sums = torch.cat([g.sum(0), (g * xh).sum(0)])
    dense = ops.bn_bwd_apply(g, y, mean, inv, gamma, sums, M)
    lazy = ops.bn_bwd_lazy(g, y, mean, inv, gamma, sums, M)
    close(lazy.dense(), dense, rtol=2e-6, atol=1e-6, what="coefficients")
    close(lazy.dense(), km.bn_bwd_lazy(g, y, mean, inv, gamma, sums, M).dense(), rtol=1e-6, atol=1e-6, what="vs model")
    # weight gradient: A side lazy
    prev = rnd("lz.prev%d.%d" % (M, Nout), (M, Nout))
    sc, sh = rnd("lz.sc%d" % Nout, (Nout,)).abs() + 0.5, rnd("lz.sh%d" % Nout, (Nout,), 0.3)
    for pro in (None, (sc, sh, 0.01)):
        close(ops.gemm_tn(lazy, prev, pro=pro), ops.gemm_tn(dense, prev, pro=pro), rtol=5e-6, atol=1e-5, what="gemm_tn A2")
    out = rnd("lz.out%d.%d" % (C, Nout), (C, Nout))
    close(ops.gemm_tn(lazy, prev, out=out.clone(), beta=1.0), ops.gemm_tn(dense, prev, out=out.clone(), beta=1.0), rtol=5e-6, atol=1e-5, what="gemm_tn A2 beta")
    # input gradient through the previous layer's BatchNorm-backward epilogue
    W = rnd("lz.W%d.%d" % (Nout, C), (Nout, C), 0.1)
    bsc, bsh, mu2, inv2 = rnd("lz.bsc%d" % Nout, (Nout,)), rnd("lz.bsh%d" % Nout, (Nout,), 0.3), rnd("lz.mu%d" % Nout, (Nout,), 0.2), rnd("lz.inv%d" % Nout, (Nout,)).abs() + 0.5
    for a_, b_ in zip(ops.gemm_nt_bnbwd(lazy, W, prev, bsc, bsh, mu2, inv2, 0.01), ops.gemm_nt_bnbwd(dense, W, prev, bsc, bsh, mu2, inv2, 0.01)):
        close(a_, b_, rtol=5e-6, atol=2e-5, what="gemm_nt_bnbwd A2")
    # 3-column consumers (D's first layer): the streaming weight-gradient kernel and the plain input-gradient product
    x3 = rnd("lz.x3%d" % M, (M, 3))
    close(ops.gemm_tn(lazy, x3), ops.gemm_tn(dense, x3), rtol=5e-6, atol=1e-5, what="streaming gemm_tn A2")
    W3 = rnd("lz.W3%d" % C, (3, C), 0.2)
    close(ops.gemm_nt(lazy, W3), ops.gemm_nt(dense, W3), rtol=5e-6, atol=1e-5, what="gemm_nt A2 (3 output columns)")
    close(ops.gemm_nt(lazy, W), ops.gemm_nt(dense, W), rtol=5e-6, atol=1e-5, what="gemm_nt A2")
    # the finalize launch of a BNBWD product emits the NEXT lazy operand's coefficients (coef_bn)
    gam2 = rnd("lz.gam2%d" % Nout, (Nout,)).abs() + 0.5
    g2, s0, s1, coef = ops.gemm_nt_bnbwd(dense, W, prev, bsc, bsh, mu2, inv2, 0.01, coef_bn=(gam2, M))
    ref = ops.bn_bwd_lazy(g2, prev, mu2, inv2, gam2, torch.cat([s0, s1]), M)
    close(coef, ref.coef, rtol=1e-6, atol=1e-7, what="coef from the finalize launch")


def test_bn_bwd_lazy_operand_edge_consumers(ops):
    """The EdgeBlock's two consumers of the per-edge BatchNorm backward (conv_w.4 -> conv_w.3): weight gradient with the per-edge B
    operand, input gradient with the EDGE_BNBWD epilogue."""
    B, N, k, H, F_ = 2, 256, 10, 64, 128
    M = B * N
    E = M * k
    g, y = rnd("lze.g", (E, F_)), rnd("lze.y", (E, F_), 1.5)
    mean, var = y.mean(0), y.var(0, unbiased=False)
    inv = 1.0 / torch.sqrt(var + 1e-5)
    gamma = rnd("lze.ga", (F_,)).abs() + 0.5
    sums = torch.cat([g.sum(0), (g * ((y - mean) * inv)).sum(0)])
    dense = ops.bn_bwd_apply(g, y, mean, inv, gamma, sums, E)
    lazy = ops.bn_bwd_lazy(g, y, mean, inv, gamma, sums, E)
    Pm = rnd("lze.P", (M, H + 2 * F_))
    x = rnd("lze.x", (M, 16), 0.5)
    idx = ops.knn(x, B, N, k, 0)
    b1 = rnd("lze.b1", (H,), 0.1)
    sc, sh = rnd("lze.sc", (H,)).abs() + 0.5, rnd("lze.sh", (H,), 0.3)
    a = ops.gemm_tn(lazy, Pm[:, :H], pro=(sc, sh, 0.01), edge=(idx, b1))
    b = ops.gemm_tn(dense, Pm[:, :H], pro=(sc, sh, 0.01), edge=(idx, b1))
    close(a, b, rtol=5e-6, atol=2e-5, what="edge gemm_tn A2")
    W2t = rnd("lze.W", (H, F_), 0.1)
    mu1, inv1 = rnd("lze.mu1", (H,), 0.2), rnd("lze.inv1", (H,)).abs() + 0.5
    for a_, b_ in zip(ops.gemm_nt_bnbwd(lazy, W2t, Pm[:, :H], sc, sh, mu1, inv1, 0.01, edge=(idx, b1)),
                      ops.gemm_nt_bnbwd(dense, W2t, Pm[:, :H], sc, sh, mu1, inv1, 0.01, edge=(idx, b1))):
        close(a_, b_, rtol=5e-6, atol=2e-5, what="edge gemm_nt_bnbwd A2")


@pytest.mark.parametrize("M,Na,Nb,defer", [(4096, 256, 128, True), (65536, 64, 256, True), (3000, 132, 36, False), (20480, 128, 1280, False), (700, 3, 64, False)])
def test_gemm_tn_colsum_by_product(ops, M, Na, Nb, defer):
    """gemm_tn(with_colsum=True): the column sums of the A operand (the bias gradient next to a weight gradient) come out of the
    product's own launch and its split reduction -- against the separate colsum pass; with a lazy BatchNorm-backward operand too."""
    A, Bm = rnd("cs.A%d.%d" % (M, Na), (M, Na)), rnd("cs.B%d.%d" % (M, Nb), (M, Nb))
    out, cs = ops.gemm_tn(A, Bm, defer=defer, with_colsum=True)
    if defer:
        ops.flush_tn()
    close(out, km.gemm_tn(A, Bm), rtol=2e-5, what="product")
    close(cs, A.double().sum(0).float(), rtol=2e-6, atol=2e-5, what="column sums")
    out2, cs2 = ops.gemm_tn(A, Bm, with_colsum=True)
    assert torch.equal(cs, cs2), "not deterministic"
    if Na % 4 == 0 and M > 64:
        y = rnd("cs.y%d.%d" % (M, Na), (M, Na), 2.0)
        mean, inv = y.mean(0), 1.0 / torch.sqrt(y.var(0, unbiased=False) + 1e-5)
        sums = torch.cat([A.sum(0), (A * ((y - mean) * inv)).sum(0)])
        lazy = ops.bn_bwd_lazy(A, y, mean, inv, None, sums, M)
        o3, cs3 = ops.gemm_tn(lazy, Bm, with_colsum=True)
        close(cs3, lazy.dense().double().sum(0).float(), rtol=1e-5, atol=5e-4, what="column sums of the lazy operand")
        close(o3, km.gemm_tn(lazy.dense(), Bm), rtol=2e-5, atol=1e-4, what="product of the lazy operand")


@pytest.mark.parametrize("M,C", [(4096, 256), (20000, 128), (1500, 68)])
def test_gemm_tn_gram_of_an_activation_without_materialising_it(ops, M, C):
    """a^T a and colsum(a) of a = lrelu(y*scale + shift) from ONE launch over y (A-side and B-side prologues, column-sum by-product):
    what the collapsed 256 -> 1024 backward needs of a3 = lrelu(bn3(y3))."""
    y = rnd("gram.y%d.%d" % (M, C), (M, C), 1.5)
    sc, sh = rnd("gram.sc%d" % C, (C,)).abs() + 0.5, rnd("gram.sh%d" % C, (C,), 0.3)
    a = ops.affine_act(y, sc, sh, 0.01)
    gram, cs = ops.gemm_tn(y, y, a_pro=(sc, sh, 0.01), pro=(sc, sh, 0.01), with_colsum=True)
    close(gram, (a.double().t() @ a.double()).float(), rtol=2e-5, what="gram")
    close(cs, a.double().sum(0).float(), rtol=2e-6, atol=1e-4, what="colsum")
    close(gram, ops.gemm_tn(a, a), rtol=1e-6, what="vs the materialised operand")


# ------------------------------------------------------------------ streaming gemm_nt kernels (3 coordinate columns in or out)
@pytest.mark.parametrize("M,N,K", [(65536, 64, 3), (4096, 128, 3), (1000, 64, 3), (333, 256, 4), (2048, 8, 2), (700, 512, 3)])
def test_gemm_nt_k4_streaming(ops, M, N, K):
    """K <= 4 products run on the streaming kernel (csrc/gemm.hip gemm_nt_k4_kernel): against the model and -- bit for bit -- against
    the MFMA kernel (tile hint 1 keeps it), linear / activation / per-group bias rows / masked epilogues, ragged row counts; with column
    statistics the product stays on the MFMA kernel (same contract)."""
    A, W, b = rnd("k4.A%d.%d" % (M, K), (M, K)), rnd("k4.W%d.%d" % (N, K), (N, K), 0.3), rnd("k4.b%d" % N, (N,))
    close(ops.gemm_nt(A, W, b), km.gemm_nt(A, W, b), rtol=2e-6, what="linear")
    with ops.nt_tile_hint(1):
        mf = ops.gemm_nt(A, W, b, stats=True)
        mf_act = ops.gemm_nt(A, W, b, act=ops.ACT_LRELU, slope=0.2)
    assert torch.equal(ops.gemm_nt(A, W, b), mf[0]) and torch.equal(ops.gemm_nt(A, W, b, act=ops.ACT_LRELU, slope=0.2), mf_act)
    y, mean, var = ops.gemm_nt(A, W, b, stats=True)
    y2, mean2, var2 = km.gemm_nt(A, W, b, stats=True)
    close(y, y2, rtol=2e-6, what="stats.y"); close(mean, mean2, rtol=1e-5, atol=1e-6, what="mean"); close(var, var2, rtol=2e-5, what="var")
    assert torch.equal(y, mf[0]), "the streaming kernel sums in the MFMA kernel's k order: same bits"
    close(var, mf[2], rtol=2e-5, what="var vs MFMA kernel")
    close(ops.gemm_nt(A, W, b, act=ops.ACT_LRELU, slope=0.2), km.gemm_nt(A, W, b, act=km.ACT_LRELU, slope=0.2), rtol=2e-6, what="lrelu")
    close(ops.gemm_nt(A, W, None, act=ops.ACT_TANH), km.gemm_nt(A, W, None, act=km.ACT_TANH), rtol=2e-6, what="tanh")
    ref = rnd("k4.ref%d.%d" % (M, N), (M, N))
    close(ops.gemm_nt_maskout(A, W, ref, 0.2), km.gemm_nt_maskout(A, W, ref, 0.2), rtol=2e-6, what="maskout")
    if M % 128 == 0:                                      # per-shape rows of a [groups, N] addend (the generator's latent part of its first conv)
        rb = rnd("k4.rb%d" % N, (M // 128, N))
        got_rb = ops.gemm_nt(A, W, b, rowbias=rb, rows_per_group=128, act=ops.ACT_LRELU, slope=0.2)
        close(got_rb, km.gemm_nt(A, W, b, rowbias=rb, rows_per_group=128, act=km.ACT_LRELU, slope=0.2), rtol=2e-6, what="rowbias")
        with ops.nt_tile_hint(1):
            assert torch.equal(got_rb, ops.gemm_nt(A, W, b, rowbias=rb, rows_per_group=128, act=ops.ACT_LRELU, slope=0.2))
        ya, ma, va = ops.gemm_nt(A, W, b, rowbias=rb, rows_per_group=128, stats=True)
        yb, mb, vb = km.gemm_nt(A, W, b, rowbias=rb, rows_per_group=128, stats=True)
        close(ya, yb, rtol=2e-6); close(ma, mb, rtol=1e-5, atol=1e-6); close(va, vb, rtol=2e-5, what="rowbias var")
    wide = torch.zeros((M, N + 8), device=A.device)       # a column slice of a wider buffer as destination
    ops.gemm_nt(A, W, b, out=wide[:, 4:4 + N])
    close(wide[:, 4:4 + N], km.gemm_nt(A, W, b), rtol=2e-6, what="strided out"); assert float(wide[:, :4].abs().max()) == 0 and float(wide[:, 4 + N:].abs().max()) == 0
    bn = (rnd("k4.g%d" % N, (N,)).abs() + 0.5, rnd("k4.be%d" % N, (N,), 0.2), torch.zeros(N, device=A.device), torch.ones(N, device=A.device))
    bn2 = tuple(t.clone() for t in bn)
    (ya, pa), (yb, pb) = ops.gemm_nt(A, W, b, bn=bn), km.gemm_nt(A, W, b, bn=bn2)
    for g, w_, what in zip(pa, pb, ("scale", "shift", "invstd", "mean")):
        close(g, w_, rtol=2e-5, atol=1e-6, what="bn." + what)
    close(bn[2], bn2[2], rtol=2e-5, atol=1e-7, what="running mean"); close(bn[3], bn2[3], rtol=2e-5, what="running var")


# ---------------------------------------------------------------- the fused layer backward (csrc/gemm_dual.hip)
@pytest.mark.parametrize("M,lazy", [(8192, True), (65536, True), (20000, False)])
def test_gemm_dual_plain(ops, M, lazy):
    """ops.gemm_dual -- weight gradient AND masked input gradient of a conv layer behind BatchNorm + LeakyReLU from ONE staging of the dy
    tile -- against the two separate launches it replaces (gemm_tn + gemm_nt_bnbwd; other summation orders) and against a float64 model;
    lazy (two-tensor) and dense dy; the coefficients of the next lazy operand from its finalize launch; accumulation into `out`."""
    Na, Nb = 128, 64
    g, y = rnd("gd.g%d" % M, (M, Na)), rnd("gd.y%d" % M, (M, Na), 2.0) + 0.3
    mean, inv = y.mean(0), 1.0 / torch.sqrt(y.var(0, unbiased=False) + 1e-5)
    gamma = rnd("gd.ga", (Na,)).abs() + 0.5
    sums = torch.cat([g.sum(0), (g * ((y - mean) * inv)).sum(0)])
    dy = ops.bn_bwd_lazy(g, y, mean, inv, gamma, sums, M) if lazy else ops.bn_bwd_apply(g, y, mean, inv, gamma, sums, M)
    dense = dy.dense() if lazy else dy
    W = rnd("gd.W", (Na, Nb), 0.1)
    prev = rnd("gd.prev%d" % M, (M, Nb), 1.5)
    sc, sh = rnd("gd.sc", (Nb,)), rnd("gd.sh", (Nb,), 0.3)
    mu, iv = rnd("gd.mu", (Nb,), 0.2), rnd("gd.iv", (Nb,)).abs() + 0.5
    assert ops.gemm_dual_ok(dy, W, prev)
    gam2 = rnd("gd.gam2", (Nb,)).abs() + 0.5
    dW, gz, s0, s1, coef = ops.gemm_dual(dy, W, prev, sc, sh, mu, iv, 0.01, coef_bn=(gam2, M), defer=False)
    # float64 model
    d64, p64 = dense.double(), prev.double()
    z = p64 * sc.double() + sh.double()
    a64 = torch.where(z > 0, z, z * 0.01)
    close(dW, (d64.t() @ a64).float(), rtol=2e-5, atol=2e-5 * float((d64.t() @ a64).abs().max()), what="weight gradient vs float64")
    g64 = (d64 @ W.double()) * torch.where(z > 0, 1.0, 0.01)
    xh = (p64 - mu.double()) * iv.double()
    close(gz, g64.float(), rtol=1e-5, atol=1e-5 * float(g64.abs().max()), what="input gradient vs float64")
    close(s0, g64.sum(0).float(), rtol=1e-5, atol=3e-5 * float(g64.abs().sum(0).max()), what="sum g")
    close(s1, (g64 * xh).sum(0).float(), rtol=1e-5, atol=3e-5 * float((g64 * xh).abs().sum(0).max()), what="sum g*xhat")
    # the two launches it replaces
    close(dW, ops.gemm_tn(dy, prev, pro=(sc, sh, 0.01)), rtol=1e-5, atol=1e-5 * float(dW.abs().max()), what="vs gemm_tn")
    g2, t0, t1, coef2 = ops.gemm_nt_bnbwd(dy, W.t().contiguous(), prev, sc, sh, mu, iv, 0.01, coef_bn=(gam2, M))
    close(gz, g2, rtol=2e-6, atol=2e-6 * float(g2.abs().max()), what="vs gemm_nt_bnbwd")
    close(s0, t0, rtol=1e-5, atol=1e-5 * float(g2.abs().sum(0).max())); close(s1, t1, rtol=1e-5, atol=3e-5 * float((g64 * xh).abs().sum(0).max()))
    close(coef, ops.bn_bwd_lazy(gz, prev, mu, iv, gam2, torch.cat([s0, s1]), M).coef, rtol=1e-6, atol=1e-7, what="coef from the finalize launch")
    # deterministic; accumulation into an existing gradient by the (deferred) split reduction
    dWb, gzb, s0b, s1b = ops.gemm_dual(dy, W, prev, sc, sh, mu, iv, 0.01, defer=False)
    assert torch.equal(dW, dWb) and torch.equal(gz, gzb) and torch.equal(s0, s0b) and torch.equal(s1, s1b)
    # phase B of the double backward from the finalize launch: the same values as the per-channel launch on the returned sums
    pbv = [rnd("gd.pb%d" % i, (Nb,)) for i in range(5)]
    dWp, gzp, s0p, s1p, psums, pdg = ops.gemm_dual(dy, W, prev, sc, sh, mu, iv, 0.01, defer=False, phaseb=(tuple(pbv) + (M,), gam2, iv))
    assert torch.equal(dWp, dW) and torch.equal(gzp, gz) and torch.equal(s0p, s0) and torch.equal(s1p, s1)
    rs, rd = ops.bn_dbl_phaseb(tuple(pbv) + (M,), gam2, iv, s0, s1)
    close(psums, rs, rtol=1e-6, atol=1e-7, what="phase-B sums from the finalize launch"); close(pdg, rd, rtol=1e-6, atol=1e-7, what="phase-B dgamma")
    acc = rnd("gd.acc", (Na, Nb))
    out = acc.clone()
    ops.gemm_dual(dy, W, prev, sc, sh, mu, iv, 0.01, out=out, beta=1.0)
    ops.flush_tn()
    close(out, acc + dW, rtol=1e-6, atol=1e-6 * float(dW.abs().max()), what="beta = 1 accumulation")


def test_gemm_dual_edge(ops):
    """The EdgeBlock's conv_w.3 backward (per-edge operand: pre[e] = P[idx[e]] - P[e // k] + b1) in one launch against the two it replaces
    and a float64 model, at a size with several chunks per workgroup."""
    B, N, k, H, F_ = 4, 1024, 10, 64, 128
    M = B * N
    E = M * k
    g, y = rnd("gde.g", (E, F_)), rnd("gde.y", (E, F_), 1.5)
    mean, inv = y.mean(0), 1.0 / torch.sqrt(y.var(0, unbiased=False) + 1e-5)
    gamma = rnd("gde.ga", (F_,)).abs() + 0.5
    sums = torch.cat([g.sum(0), (g * ((y - mean) * inv)).sum(0)])
    lazy = ops.bn_bwd_lazy(g, y, mean, inv, gamma, sums, E)
    Pm = rnd("gde.P", (M, H + 2 * F_))
    idx = ops.knn(rnd("gde.x", (M, 16), 0.5), B, N, k, 0)
    b1 = rnd("gde.b1", (H,), 0.1)
    sc, sh = rnd("gde.sc", (H,)), rnd("gde.sh", (H,), 0.3)
    mu1, inv1 = rnd("gde.mu1", (H,), 0.2), rnd("gde.inv1", (H,)).abs() + 0.5
    W2 = rnd("gde.W", (F_, H), 0.1)
    assert ops.gemm_dual_ok(lazy, W2, Pm[:, :H], edge=(idx, b1))
    dW, g1, s0, s1 = ops.gemm_dual(lazy, W2, Pm[:, :H], sc, sh, mu1, inv1, 0.01, edge=(idx, b1), defer=False)
    a = ops.gemm_tn(lazy, Pm[:, :H], pro=(sc, sh, 0.01), edge=(idx, b1))
    close(dW, a, rtol=1e-5, atol=1e-5 * float(a.abs().max()), what="edge weight gradient vs gemm_tn")
    g2, t0, t1 = ops.gemm_nt_bnbwd(lazy, W2.t().contiguous(), Pm[:, :H], sc, sh, mu1, inv1, 0.01, edge=(idx, b1))
    close(g1, g2, rtol=2e-6, atol=2e-6 * float(g2.abs().max()), what="edge input gradient vs gemm_nt_bnbwd")
    close(s0, t0, rtol=1e-5, atol=1e-5 * float(g2.abs().sum(0).max())); close(s1, t1, rtol=1e-5, atol=1e-4 * float(g2.abs().sum(0).max()))
    # float64 model of the per-edge operand
    i = torch.arange(M, device=g.device).repeat_interleave(k)
    pre = (Pm[idx.reshape(-1).long(), :H].double() - Pm[i, :H].double()) + b1.double()
    z = pre * sc.double() + sh.double()
    d64 = lazy.dense().double()
    ref = d64.t() @ torch.where(z > 0, z, z * 0.01)
    close(dW, ref.float(), rtol=2e-5, atol=2e-5 * float(ref.abs().max()), what="edge weight gradient vs float64")
    r64 = (d64 @ W2.double()) * torch.where(z > 0, 1.0, 0.01)
    close(g1, r64.float(), rtol=1e-5, atol=1e-5 * float(r64.abs().max()), what="edge input gradient vs float64")


@pytest.mark.parametrize("M,Nb,mode", [(16384, 128, "lazy"), (65536, 128, "dense"), (16384, 256, "act"), (65536, 256, "act")])
def test_gemm_dual_wide(ops, M, Nb, mode):
    """The 256-column geometry of ops.gemm_dual (512 threads, 2 / 4 column parts per row run): the Discriminator's mlps.6 backward
    (lazy / dense dy, Nb = 128) and the collapsed fc2.0 (dy := a3 = lrelu(bn3(y3)) formed on load, W := a symmetric K x K matrix, pre := y3
    itself, bias + dense row addend in front of the mask, colsum(a3) as a by-product; Nb = 256) -- against float64 and the launches it replaces."""
    Na = 256
    sc, sh = rnd("gdw.sc%d" % Nb, (Nb,)), rnd("gdw.sh%d" % Nb, (Nb,), 0.3)
    mu, iv = rnd("gdw.mu%d" % Nb, (Nb,), 0.2), rnd("gdw.iv%d" % Nb, (Nb,)).abs() + 0.5
    if mode == "act":
        y3 = rnd("gdw.y3.%d" % M, (M, Na), 1.5)
        dy = ops.ActOperand(y3, sc, sh, 0.01)
        dense, prev = dy.dense(), y3
        Wm = rnd("gdw.G", (Na, Na), 0.05); Wm = (Wm + Wm.t()).contiguous()
        bias, radd = rnd("gdw.cv", (Nb,), 0.2), rnd("gdw.E%d" % M, (M, Nb), 0.3)
        kw = dict(bias=bias, rowadd=radd, with_colsum=True)
    else:
        g, y = rnd("gdw.g%d" % M, (M, Na)), rnd("gdw.y%d" % M, (M, Na), 2.0) + 0.3
        mean, inv = y.mean(0), 1.0 / torch.sqrt(y.var(0, unbiased=False) + 1e-5)
        gamma = rnd("gdw.ga", (Na,)).abs() + 0.5
        sums = torch.cat([g.sum(0), (g * ((y - mean) * inv)).sum(0)])
        dy = ops.bn_bwd_lazy(g, y, mean, inv, gamma, sums, M) if mode == "lazy" else ops.bn_bwd_apply(g, y, mean, inv, gamma, sums, M)
        dense = dy.dense() if mode == "lazy" else dy
        prev = rnd("gdw.prev%d" % M, (M, Nb), 1.5)
        Wm = rnd("gdw.W", (Na, Nb), 0.1)
        bias = radd = None
        kw = {}
    assert ops.gemm_dual_ok(dy, Wm, prev)
    res = ops.gemm_dual(dy, Wm, prev, sc, sh, mu, iv, 0.01, defer=False, **kw)
    dW, gz, s0, s1 = res[:4]
    d64, p64 = dense.double(), prev.double()
    z = p64 * sc.double() + sh.double()
    a64 = torch.where(z > 0, z, z * 0.01)
    ref = d64.t() @ a64
    close(dW, ref.float(), rtol=2e-5, atol=2e-5 * float(ref.abs().max()), what="weight gradient vs float64")
    acc = d64 @ Wm.double()
    if bias is not None:
        acc = acc + bias.double() + radd.double()
    g64 = acc * torch.where(z > 0, 1.0, 0.01)
    xh = (p64 - mu.double()) * iv.double()
    close(gz, g64.float(), rtol=1e-5, atol=1e-5 * float(g64.abs().max()), what="input gradient vs float64")
    close(s0, g64.sum(0).float(), rtol=1e-5, atol=3e-5 * float(g64.abs().sum(0).max()), what="sum g")
    close(s1, (g64 * xh).sum(0).float(), rtol=1e-5, atol=3e-5 * float((g64 * xh).abs().sum(0).max()), what="sum g*xhat")
    if mode == "act":
        close(res[4], d64.sum(0).float(), rtol=1e-5, atol=3e-5 * float(d64.abs().sum(0).max()), what="colsum(a3)")
        g2, t0, t1 = ops.gemm_nt_bnbwd(prev, Wm, prev, sc, sh, mu, iv, 0.01, pro=(sc, sh, 0.01), bias=bias, rowadd=radd)
        gram, cs = ops.gemm_tn(prev, prev, a_pro=(sc, sh, 0.01), pro=(sc, sh, 0.01), with_colsum=True)
        close(dW, gram, rtol=1e-5, atol=1e-5 * float(gram.abs().max()), what="vs gemm_tn (Gram)")
        close(res[4], cs, rtol=1e-5, atol=1e-5 * float(cs.abs().max()), what="colsum vs gemm_tn by-product")
    else:
        g2, t0, t1 = ops.gemm_nt_bnbwd(dy, Wm.t().contiguous(), prev, sc, sh, mu, iv, 0.01)
        close(dW, ops.gemm_tn(dy, prev, pro=(sc, sh, 0.01)), rtol=1e-5, atol=1e-5 * float(dW.abs().max()), what="vs gemm_tn")
    close(gz, g2, rtol=3e-6, atol=3e-6 * float(g2.abs().max()), what="vs gemm_nt_bnbwd")
    res2 = ops.gemm_dual(dy, Wm, prev, sc, sh, mu, iv, 0.01, defer=False, **kw)
    assert all(torch.equal(a_, b_) for a_, b_ in zip(res, res2)), "not deterministic"


@pytest.mark.parametrize("M,N,K", [(1024, 256, 256), (512, 96, 64), (96, 32, 32), (2048, 64, 128), (1024, 256, 224)])
def test_gemm_nt_small_row_products(ops, M, N, K):
    """Weight-by-weight products (few rows, K <= 256): bias, per-row addend (rows_per_group = 1: a full [M,N] tensor, also in place --
    how fc2.0's double-backward weight gradient sums its terms), activation; against float64."""
    A, W, b = rnd("mid.a.%d.%d" % (M, K), (M, K)), rnd("mid.w.%d.%d" % (N, K), (N, K), 0.2), rnd("mid.b.%d" % N, (N,))
    ref = A.double() @ W.double().t()
    close(ops.gemm_nt(A, W, exact=True), ref, rtol=2e-6, atol=2e-5)
    close(ops.gemm_nt(A, W, b, act=ops.ACT_LRELU, slope=0.2, exact=True), torch.nn.functional.leaky_relu(ref + b.double(), 0.2), rtol=2e-6, atol=2e-5)
    add = rnd("mid.add.%d.%d" % (M, N), (M, N))
    want = ref + add.double()
    out = add.clone()
    got = ops.gemm_nt(A, W, rowbias=out, rows_per_group=1, out=out, exact=True)      # accumulate in place
    assert got.data_ptr() == out.data_ptr()
    close(out, want, rtol=2e-6, atol=2e-5)
    grp = rnd("mid.grp.%d" % N, (M // 32, N))
    close(ops.gemm_nt(A, W, rowbias=grp, rows_per_group=32, exact=True), ref + grp.double().repeat_interleave(32, dim=0), rtol=2e-6, atol=2e-5)
