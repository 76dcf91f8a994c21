"""GPU: the pieces of the collapsed backward of the layer in front of D's max-pool (row-sparse products, materialised
BN+LeakyReLU, row scaling, BNBWD epilogue with bias / dense addend) and the collapsed form against the direct one."""
import pytest
import torch

import kernel_model as km
from test_kernels_gpu import close, ops, rnd  # noqa: F401  (ops is a fixture)

pytestmark = pytest.mark.gpu


def _sparse(name, B, rows, Cs):
    val = rnd(name + ".val", (B, Cs))
    g = torch.Generator().manual_seed(hash(name) % 1000)
    arg = (torch.randint(0, rows, (B, Cs), generator=g) + torch.arange(B)[:, None] * rows).int().cuda()
    return val, arg


@pytest.mark.parametrize("B,rows,Cs,N", [(3, 300, 1024, 256), (2, 128, 70, 37), (1, 2048, 1024, 256), (2, 64, 1000, 600)])
def test_sparse_rows(ops, B, rows, Cs, N):
    val, arg = _sparse("sr%d" % Cs, B, rows, Cs)
    if Cs == 70:
        arg[:, :40] = arg[:, :1]                      # many channels share one row
    if Cs == 1024 and rows == 300:                    # ... dozens of channels on each of a few rows (what a shape's extreme points look like): the 8-rows-in-flight path
        arg[:, :600] = (torch.arange(B)[:, None] * rows + (torch.arange(600)[None, :] % 7) * 13).int().cuda()
    W = rnd("sr.W%d" % N, (Cs, N), 0.3)
    E = ops.sparse_rows_nt(val, arg, rows, W)
    close(E, km.sparse_rows_nt(val, arg, rows, W), what="nt")
    assert torch.equal(E, ops.sparse_rows_nt(val, arg, rows, W)), "not deterministic"
    Bm = rnd("sr.B%d" % N, (B * rows, N))
    out = rnd("sr.out", (Cs, N)); out2 = out.clone()
    ops.sparse_rows_tn(val, arg, rows, Bm, out); km.sparse_rows_tn(val, arg, rows, Bm, out2)
    close(out, out2, what="tn")
    sc, sh = rnd("sr.sc", (N,)).abs() + 0.5, rnd("sr.sh", (N,), 0.3)
    out = torch.zeros(Cs, N, device="cuda"); out2 = out.clone()
    ops.sparse_rows_tn(val, arg, rows, Bm, out, pro=(sc, sh, 0.01)); km.sparse_rows_tn(val, arg, rows, Bm, out2, pro=(sc, sh, 0.01))
    close(out, out2, what="tn.pro")


def test_affine_act_rowscale(ops):
    X = rnd("aa.X", (777, 256)); sc, sh = rnd("aa.sc", (256,)), rnd("aa.sh", (256,), 0.3)
    close(ops.affine_act(X, sc, sh, 0.01), km.affine_act(X, sc, sh, 0.01), what="affine_act")
    Xv = rnd("aa.Xv", (100, 90))[:, 7:70]             # unaligned view -> scalar kernel
    close(ops.affine_act(Xv, sc[:63].contiguous(), sh[:63].contiguous(), 0.2), km.affine_act(Xv, sc[:63], sh[:63], 0.2), what="affine_act.view")
    W = rnd("aa.W", (1024, 256), 0.2); a, b, d, v = rnd("aa.a", (1024,)), rnd("aa.b", (1024,)), rnd("aa.d", (1024,)), rnd("aa.v", (256,))
    close(ops.rowscale_outer(W, a), km.rowscale_outer(W, a), what="rowscale")
    close(ops.rowscale_outer(W, a, b, d, v), km.rowscale_outer(W, a, b, d, v), what="rowscale_outer")


def test_bnbwd_with_prologue_bias_rowadd(ops):
    M, N, K = 900, 256, 256
    A, W, ref = rnd("bb.A", (M, K)), rnd("bb.W", (N, K), 0.2), rnd("bb.ref", (M, N))
    psc, psh = rnd("bb.psc", (K,)).abs() + 0.5, rnd("bb.psh", (K,), 0.3)
    sc, sh, mean, inv = rnd("bb.sc", (N,)), rnd("bb.sh", (N,), 0.3), rnd("bb.mu", (N,), 0.2), rnd("bb.inv", (N,)).abs() + 0.5
    bias, E = rnd("bb.bias", (N,)), rnd("bb.E", (M, N))
    for a, b in zip(ops.gemm_nt_bnbwd(A, W, ref, sc, sh, mean, inv, 0.01, pro=(psc, psh, 0.01), bias=bias, rowadd=E),
                    km.gemm_nt_bnbwd(A, W, ref, sc, sh, mean, inv, 0.01, pro=(psc, psh, 0.01), bias=bias, rowadd=E)):
        close(a, b, rtol=5e-5, atol=2e-4, what="bnbwd+")


def test_collapsed_equals_direct(ops):
    """dz.W and dz^T.a through the collapsed identities == the direct GEMMs on the lazily evaluated operand."""
    B, N, Cin, Cout = 4, 512, 256, 1024
    M = B * N
    y3 = rnd("cd.y3", (M, Cin)); sc3, sh3 = rnd("cd.sc3", (Cin,)).abs() + 0.5, rnd("cd.sh3", (Cin,), 0.3)
    W, b4 = rnd("cd.W", (Cout, Cin), 0.06), rnd("cd.b4", (Cout,), 0.1)
    a3 = ops.affine_act(y3, sc3, sh3, 0.01)
    y4 = ops.gemm_nt(a3, W, b4)
    alpha, beta = rnd("cd.al", (Cout,), 0.05), rnd("cd.be", (Cout,), 0.05)
    val, arg = _sparse("cd", B, N, Cout)
    dz = ops.SparseAffine(y4, alpha, beta, val, arg, N)
    # direct
    dW_ref = ops.gemm_tn(dz, y3, pro=(sc3, sh3, 0.01))
    dz_dense = (y4 * alpha + beta + km._sparse_dense(val, arg, N)).contiguous()
    dx_ref = ops.gemm_nt(dz_dense, W.t().contiguous())
    # collapsed
    gram = ops.gemm_tn(a3, a3)
    dW = ops.rowscale_outer(ops.gemm_nt(W, gram), alpha, b4, beta, ops.colsum(a3)[0])
    ops.sparse_rows_tn(val, arg, N, a3, dW)
    G4 = ops.gemm_tn(W, ops.rowscale_outer(W, alpha))
    cvec = ops.gemm_nt(b4.view(1, -1), W.t().contiguous(), pro=(alpha, beta, 1.0))[0]
    dx = ops.gemm_nt(a3, G4, cvec) + ops.sparse_rows_nt(val, arg, N, W)
    close(dW, dW_ref, rtol=2e-4, what="collapsed dW")
    close(dx, dx_ref, rtol=2e-4, what="collapsed dx")


def test_double_backward_collapse_pieces(ops):
    B, N, K, C = 4, 256, 256, 1024
    q = rnd("dc.q", (B * N, K)); W = rnd("dc.W", (C, K), 0.06)
    val, arg = _sparse("dc", B, N, C)
    close(ops.gather_rowdot(q, arg, W), km.gather_rowdot(q, arg, W), rtol=3e-5, what="gather_rowdot")
    T = rnd("dc.T", (C, K))
    close(ops.rowdot(W, T), km.rowdot(W, T), rtol=3e-5, what="rowdot")
    v = lambda n, s=1.0: rnd("dc." + n, (C,), s)
    uarg, gval, yarg, pooled = rnd("dc.ua", (B, C)), rnd("dc.gv", (B, C)), rnd("dc.ya", (B, C)), rnd("dc.po", (B, C))
    args = (uarg, gval, yarg, pooled, v("U0"), v("quad"), v("b", 0.1), v("mu", 0.2), v("inv").abs() + 0.5, v("ga").abs() + 0.5, v("S0"), v("S1"), B * N, 0.01)
    for a, b, n in zip(ops.bn_dbl_pool(*args), km.bn_dbl_pool(*args), ("t", "spB", "coeffs")):
        close(a, b, rtol=3e-5, what="bn_dbl_pool." + n)


@pytest.mark.parametrize("B,N,K,C", [(3, 256, 64, 128), (2, 1024, 256, 1024), (4, 128, 16, 70)])
def test_gemm_bn_pool(ops, B, N, K, C):
    """GEMM + train-mode BN + LeakyReLU + max over N with the pooling partials in the GEMM epilogue == GEMM, BN, maxpool."""
    M = B * N
    A, W, b = rnd("gp.A%d" % K, (M, K)), rnd("gp.W%d" % C, (C, K), 0.2), rnd("gp.b%d" % C, (C,), 0.1)
    gamma, beta = rnd("gp.g%d" % C, (C,)), rnd("gp.be%d" % C, (C,), 0.2)      # gamma of both signs: max and min branches
    gamma[3] = 0.0
    psc, psh = rnd("gp.psc%d" % K, (K,)).abs() + 0.5, rnd("gp.psh%d" % K, (K,), 0.3)
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    rm2, rv2 = rm.clone(), rv.clone()
    for keep in (False, True):
        y, st, pooled, arg, yarg = ops.gemm_bn_pool(A, W, b, (gamma, beta, rm, rv), N, 0.01, pro=(psc, psh, 0.01), keep_y=keep)
        y2, st2, pooled2, arg2, yarg2 = km.gemm_bn_pool(A, W, b, (gamma, beta, rm2, rv2), N, 0.01, pro=(psc, psh, 0.01), keep_y=True)
        assert (y is None) == (not keep)
        if keep:
            close(y, y2, what="y")
        for a_, b_ in zip(st, st2):
            close(a_, b_, rtol=2e-5, atol=2e-6, what="bn state")
        close(pooled, pooled2, rtol=2e-5, atol=2e-6, what="pooled")
        live = gamma != 0          # gamma == 0: every row ties; the fused path reports the first row but cannot know y there
        close(yarg[:, live], yarg2[:, live], rtol=2e-5, atol=2e-6, what="yarg")
        # the arg-max rows carry the pooled value (exact index equality is only guaranteed up to rounding ties)
        cols = torch.arange(C, device="cuda").view(1, C).expand(B, C)
        assert ((arg.long() // N) == torch.arange(B, device="cuda").view(B, 1)).all()
        close(y2[arg.long(), cols][:, live], yarg2[:, live], rtol=2e-5, atol=2e-6, what="y at argmax")
        assert (arg[:, ~live] == arg2[:, ~live]).all()
        agree = (arg == arg2).float().mean().item()
        assert agree > 0.99 or C == 70, agree
    close(rm, rm2, rtol=1e-5, atol=1e-6, what="running_mean"); close(rv, rv2, rtol=1e-5, what="running_var")


@pytest.mark.parametrize("C,K", [(1024, 256), (512, 256), (256, 32)])
def test_wt_diag_w(C, K):
    """ops.wt_diag_w: W^T diag(alpha) W and (alpha*b + beta).W in one launch against float64 and against the launches it replaces."""
    from spgan import ops
    from test_kernels_gpu import close, rnd
    W = rnd("wdw.W%d" % C, (C, K), 0.1)
    alpha, beta, b = rnd("wdw.a%d" % C, (C,)), rnd("wdw.b%d" % C, (C,), 0.3), rnd("wdw.c%d" % C, (C,), 0.2)
    G, cvec = ops.wt_diag_w(W, alpha, beta, b)
    ref = (W.double() * alpha.double()[:, None]).t() @ W.double()
    close(G, ref.float(), rtol=3e-6, atol=3e-6 * float(ref.abs().max()), what="G vs float64")
    cref = (alpha.double() * b.double() + beta.double()) @ W.double()
    close(cvec, cref.float(), rtol=3e-6, atol=3e-6 * float(cref.abs().max()), what="cvec vs float64")
    close(G, ops.gemm_tn(W, ops.rowscale_outer(W, alpha)), rtol=3e-6, atol=3e-6 * float(ref.abs().max()), what="G vs gemm_tn")
    G2 = ops.wt_diag_w(W, alpha)
    assert torch.equal(G, G2) and torch.equal(ops.wt_diag_w(W, alpha, beta, b)[1], cvec)


@pytest.mark.parametrize("C,K,B,rows", [(1024, 256, 4, 256), (512, 64, 3, 96), (1024, 256, 32, 2048)])
def test_collapse_prep_equals_its_parts(C, K, B, rows):
    """ops.collapse_prep: one or two wt_diag_w problems and the sparse-row product S.W from ONE launch -- bit-identical to the separate launches
    (the same device functions), for one problem with cvec, and for two (the double backward: no cvec / cvec)."""
    from spgan import ops
    from test_kernels_gpu import rnd
    W = rnd("cp.W%d" % C, (C, K), 0.1)
    a1, a2, beta, b = rnd("cp.a1%d" % C, (C,)), rnd("cp.a2%d" % C, (C,)), rnd("cp.b%d" % C, (C,), 0.3), rnd("cp.c%d" % C, (C,), 0.2)
    val = rnd("cp.val%d.%d" % (C, B), (B, C))
    g = torch.Generator().manual_seed(C + rows)
    local = torch.randint(0, rows, (B, C), generator=g)
    local[:, : C // 4] = 7 % rows                                   # a hub row: a quarter of the channels share one arg-max
    arg = (local + torch.arange(B)[:, None] * rows).to(torch.int32).cuda()
    E_ref = ops.sparse_rows_nt(val, arg, rows, W)
    G_ref, c_ref = ops.wt_diag_w(W, a2, beta, b)
    ((G, cv),), E = ops.collapse_prep(W, [(a2, beta, b)], val, arg, rows)
    assert torch.equal(G, G_ref) and torch.equal(cv, c_ref) and torch.equal(E, E_ref)
    (G1, (G2, cv2)), E2 = ops.collapse_prep(W, [(a1, None, None), (a2, beta, b)], val, arg, rows)
    assert torch.equal(G1, ops.wt_diag_w(W, a1)) and torch.equal(G2, G_ref) and torch.equal(cv2, c_ref) and torch.equal(E2, E_ref)


@pytest.mark.parametrize("C,N,K,B,rows", [(1024, 256, 256, 32, 2048), (512, 64, 96, 3, 64), (256, 32, 32, 1, 40)])
def test_wgrad_collapse_all_terms(C, N, K, B, rows):
    """ops.wgrad_collapse: scaled product + rank-1 term + second (transposed-operand) product + sparse gather term, plain / accumulating, with the
    raw product as a by-product -- against float64 and against the launches it replaces."""
    import kernel_model as km
    from spgan import ops
    from test_kernels_gpu import close, rnd
    W, X1, X2 = rnd("wg.W%d" % C, (C, K), 0.1), rnd("wg.X1%d" % N, (N, K)), rnd("wg.X2%d" % N, (K, N))
    a1, b1, d1, a2 = rnd("wg.a%d" % C, (C,)), rnd("wg.b%d" % C, (C,)), rnd("wg.d%d" % C, (C,)), rnd("wg.e%d" % C, (C,))
    v1 = rnd("wg.v%d" % N, (N,))
    val = rnd("wg.val%d" % C, (B, C))
    g = torch.Generator().manual_seed(C + N)
    arg = (torch.randint(0, rows, (B, C), generator=g) + torch.arange(B)[:, None] * rows).to(torch.int32).cuda()
    Bm = rnd("wg.Bm%d" % N, (B * rows, N))
    pro = (rnd("wg.ps%d" % N, (N,)).abs() + 0.5, rnd("wg.ph%d" % N, (N,), 0.2), 0.01)
    def ref64(x2, sparse, base=None):
        o = a1.double()[:, None] * (W.double() @ X1.double().t()) + (a1.double() * b1.double() + d1.double())[:, None] * v1.double()[None, :]
        if x2:
            o = o + a2.double()[:, None] * (W.double() @ X2.double())
        if sparse:
            add = torch.zeros((C, N), dtype=torch.float64, device=W.device)
            km.sparse_rows_tn(val.double(), arg, rows, Bm.double(), add, pro=(pro[0].double(), pro[1].double(), pro[2]))
            o = o + add
        return o if base is None else o + base.double()
    tol = dict(rtol=3e-6, atol=3e-5)
    close(ops.wgrad_collapse(W, X1, a1, b1, d1, v1), ref64(False, False), **tol)
    out, T = ops.wgrad_collapse(W, X1, a1, b1, d1, v1, sparse=(val, arg, rows, Bm, pro), want_T=True)
    close(out, ref64(False, True), **tol); close(T, W.double() @ X1.double().t(), **tol)
    base = rnd("wg.base%d" % C, (C, N))
    acc = base.clone()
    ops.wgrad_collapse(W, X1, a1, b1, d1, v1, X2=X2, x2_t=True, a2=a2, sparse=(val, arg, rows, Bm, pro), out=acc, accumulate=True)
    close(acc, ref64(True, True, base), **tol)
    close(ops.wgrad_collapse(W, X1, a1, X2=X2.t().contiguous(), a2=a2), a1.double()[:, None] * (W.double() @ X1.double().t()) + a2.double()[:, None] * (W.double() @ X2.double()), **tol)
    # against the launches it replaces (same fp32 data, other summation order)
    old = ops.rowscale_outer(ops.gemm_nt(W, X1, exact=True), a1, b1, d1, v1)
    ops.sparse_rows_tn(val, arg, rows, Bm, old, pro=pro)
    close(ops.wgrad_collapse(W, X1, a1, b1, d1, v1, sparse=(val, arg, rows, Bm, pro)), old, rtol=3e-6, atol=3e-5)
    assert torch.equal(ops.wgrad_collapse(W, X1, a1, b1, d1, v1, sparse=(val, arg, rows, Bm, pro)), out), "not deterministic"


def test_dbl_top_dots_equals_its_three_launches():
    """ops.dbl_top_dots: gather_rowdot, rowdot and W.cq from one grid -- the first two bit-identical to their own kernels (same arithmetic), the
    matrix-vector product against float64."""
    from spgan import ops
    from test_kernels_gpu import close, rnd
    B, rows, C, K = 5, 96, 1024, 256
    Q, W, T, cq = rnd("dtd.Q", (B * rows, K)), rnd("dtd.W", (C, K), 0.1), rnd("dtd.T", (C, K)), rnd("dtd.cq", (K,))
    g = torch.Generator().manual_seed(11)
    arg = (torch.randint(0, rows, (B, C), generator=g) + torch.arange(B)[:, None] * rows).to(torch.int32).cuda()
    uarg, quad, U0 = ops.dbl_top_dots(Q, arg, W, T, cq)
    assert torch.equal(uarg, ops.gather_rowdot(Q, arg, W)) and torch.equal(quad, ops.rowdot(W, T))
    close(U0, W.double() @ cq.double(), rtol=3e-6, atol=3e-6)
