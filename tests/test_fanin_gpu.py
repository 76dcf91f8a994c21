"""GPU: column records merged inside the producing launch by the last-arriving workgroup (csrc/fanin.hpp) against the separate
finalize launches of round 1 (ops.FANIN[0] = False) and the kernel models: one and two merge levels, ragged row/column tiles,
bit-reproducibility from run to run (the merge order is fixed, not the arrival order), counters left at zero."""
import pytest
import torch

import kernel_model as km
from test_kernels_gpu import close, ops, rnd  # noqa: F401  (ops is a fixture)

pytestmark = pytest.mark.gpu


@pytest.fixture()
def both(ops):
    def run(fn):
        out = []
        keep = ops.FANIN[0]
        for flag in (True, False):
            ops.FANIN[0] = flag
            try:
                out.append(fn())
            finally:
                ops.FANIN[0] = keep
        return out
    return run


def _flat(res):
    out = []
    for r in (res if isinstance(res, (tuple, list)) else [res]):
        if isinstance(r, (tuple, list)):
            out.extend(_flat(r))
        elif r is not None:
            out.append(r)
    return out


@pytest.mark.parametrize("M,N,K", [(65536, 256, 128), (8192, 64, 64), (700, 40, 36), (128 * 49, 1024, 256), (655360, 128, 64), (32, 512, 128), (48, 100, 64)])
def test_gemm_bn_statistics_in_launch(ops, both, M, N, K):
    A, W, b = rnd("fi.A%d.%d" % (M, K), (M, K)), rnd("fi.W%d.%d" % (N, K), (N, K), 0.1), rnd("fi.b%d" % N, (N,))
    gamma, beta = rnd("fi.g%d" % N, (N,)).abs() + 0.5, rnd("fi.be%d" % N, (N,), 0.2)

    def fwd():
        rm, rv = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
        y, st = ops.gemm_nt(A, W, b, bn=(gamma, beta, rm, rv))
        return [y, *st, rm, rv]
    fused, split = both(fwd)
    rm, rv = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
    ym, stm = km.gemm_nt(A, W, b, bn=(gamma, beta, rm, rv))
    for a_, b_, m_ in zip(fused, split, [ym, *stm, rm, rv]):
        close(a_, b_, rtol=2e-6, atol=1e-6, what="fused vs separate finalize")
        close(a_, m_, rtol=2e-4, atol=2e-5, what="fused vs model")
    ops.FANIN[0] = True
    again = fwd()
    assert all(torch.equal(x, y) for x, y in zip(fused, again)), "in-launch merge is not reproducible"
    # count_rep only changes the unbiased-variance factor of the running statistics
    rm2, rv2 = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
    ops.gemm_nt(A, W, b, bn=(gamma, beta, rm2, rv2), count_rep=4)
    rm3, rv3 = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
    km.gemm_nt(A, W, b, bn=(gamma, beta, rm3, rv3), count_rep=4)
    close(rv2, rv3, rtol=2e-4, atol=2e-5); close(rm2, rm3, rtol=2e-4, atol=2e-5)
    ymv = ops.gemm_nt(A, W, b, stats=True)
    ops.FANIN[0] = False
    close(ymv[1], fused[4], rtol=1e-6, atol=1e-6, what="stats=True mean")
    ring = ops._FANIN_RING[A.device][0]
    assert int(ring.abs().sum()) == 0, "fan-in counters must be left at zero"


@pytest.mark.parametrize("M,N,K", [(65536, 128, 256), (4096, 64, 128), (1000, 70, 52), (32, 256, 64)])
def test_gemm_bnbwd_sums_in_launch(ops, both, M, N, K):
    A, W = rnd("fb.A%d.%d" % (M, K), (M, K)), rnd("fb.W%d.%d" % (N, K), (N, K), 0.1)
    ref = rnd("fb.ref%d.%d" % (M, N), (M, N))
    sc, sh, mu, inv = rnd("fb.sc%d" % N, (N,)), rnd("fb.sh%d" % N, (N,), 0.3), rnd("fb.mu%d" % N, (N,), 0.2), rnd("fb.inv%d" % N, (N,)).abs() + 0.5
    fused, split = both(lambda: list(ops.gemm_nt_bnbwd(A, W, ref, sc, sh, mu, inv, 0.01)))
    model = km.gemm_nt_bnbwd(A, W, ref, sc, sh, mu, inv, 0.01)
    for a_, b_, m_ in zip(fused, split, model):
        close(a_, b_, rtol=2e-6, atol=2e-5, what="fused vs separate")
        close(a_, m_, rtol=3e-4, atol=3e-3, what="fused vs model")
    assert fused[1].data_ptr() + 4 * N == fused[2].data_ptr(), "sums must be contiguous [s0 | s1]"
    ops.FANIN[0] = True
    try:
        again = ops.gemm_nt_bnbwd(A, W, ref, sc, sh, mu, inv, 0.01)
    finally:
        ops.FANIN[0] = False
    assert all(torch.equal(x, y) for x, y in zip(fused, again))


def test_fanin_many_launches_back_to_back(ops):
    """300 launches issued without a host sync in between (the ring of counters wraps): every launch must see zeroed counters."""
    M, N, K = 8192, 256, 64
    A, W = rnd("fs.A", (M, K)), rnd("fs.W", (N, K), 0.1)
    gamma, beta = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
    first = None
    ops.FANIN[0] = True
    try:
        for i in range(300):
            y, st = ops.gemm_nt(A, W, None, bn=(gamma, beta, None, None))
            if first is None:
                first = [t.clone() for t in st]
    finally:
        ops.FANIN[0] = False
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(first, st))
    assert int(ops._FANIN_RING[A.device][0].abs().sum()) == 0


def test_edge_stats_bn_two_layers_one_launch(ops):
    """spgan_colstats_finalize_bn2: both per-edge BatchNorm layers of an EdgeBlock from one record set."""
    M, k, H, F_, C = 2048, 10, 64, 128, 64
    PQR = rnd("e2.PQR", (M, H + 2 * F_))
    g = torch.Generator().manual_seed(3)
    idx = torch.randint(0, M, (M, k), generator=g).int().cuda()
    b1, bx = rnd("e2.b1", (H,), 0.1), rnd("e2.bx", (F_,), 0.1)
    mk = lambda n, c: (rnd(n + ".g", (c,)).abs() + 0.5, rnd(n + ".b", (c,), 0.2), torch.zeros(c, device="cuda"), torch.ones(c, device="cuda"))
    for rep in (1, 4):
        bw, bxp = mk("e2.w", H), mk("e2.x", F_)
        bw2 = tuple(t.clone() for t in bw); bx2 = tuple(t.clone() for t in bxp)
        got = ops.edge_stats_bn(PQR, idx, b1, bx, bw, bxp, rep)
        ref = km.edge_stats_bn(PQR, idx, b1, bx, bw2, bx2, rep)
        for a, b in zip(got[0] + got[1], ref[0] + ref[1]):
            close(a, b, rtol=2e-5, atol=2e-6, what="edge_stats_bn")
        for a, b in zip(bw[2:] + bxp[2:], bw2[2:] + bx2[2:]):
            close(a, b, rtol=2e-5, atol=2e-6, what="running statistics")


@pytest.mark.parametrize("M,C,G", [(32, 1024, 32), (64, 64, 64), (100, 37, 100), (256, 48, 128), (4096, 96, 64)])
def test_colsum_of_short_groups_is_one_launch(ops, M, C, G):
    X = rnd("cs.X%d.%d" % (M, C), (M, C))
    close(ops.colsum(X, G), km.colsum(X, G), rtol=2e-6, atol=1e-5, what="colsum")
    m, v = ops.colstats(X, G, 0.2)
    m2, v2 = km.colstats(X, G, 0.2)
    close(m, m2, rtol=2e-6, atol=1e-6); close(v, v2, rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("M,N,K", [(32, 256, 64), (64, 512, 256), (32, 64, 1), (100, 128, 64)])
def test_maskout_with_column_sums(ops, M, N, K):
    """gemm_nt_maskout(with_colsum=True): the bias gradient of the layer below from the same launch (M <= 64, aligned) or a follow-up."""
    A, W, ref = rnd("mo.A%d.%d" % (M, K), (M, K)), rnd("mo.W%d.%d" % (N, K), (N, K), 0.2), rnd("mo.r%d.%d" % (M, N), (M, N))
    y, cs = ops.gemm_nt_maskout(A, W, ref, 0.01, with_colsum=True)
    y2, cs2 = km.gemm_nt_maskout(A, W, ref, 0.01, with_colsum=True)
    close(y, y2, rtol=2e-5, atol=2e-5, what="maskout"); close(cs, cs2, rtol=2e-5, atol=1e-4, what="column sums")
    assert torch.equal(y, ops.gemm_nt_maskout(A, W, ref, 0.01))


def test_pool_bwd_stats_with_prep(ops):
    B, C, N = 8, 1024, 256
    M = B * N
    gpool, pooled, y = rnd("pp.g", (B, C)), rnd("pp.p", (B, C)), rnd("pp.y", (M, C))
    g = torch.Generator().manual_seed(2)
    arg = (torch.randint(0, N, (B, C), generator=g) + torch.arange(B)[:, None] * N).int().cuda()
    mu, inv, gamma = rnd("pp.mu", (C,), 0.2), rnd("pp.inv", (C,)).abs() + 0.5, rnd("pp.ga", (C,)).abs() + 0.5
    gval, sums, sa = ops.pool_bwd_stats(gpool, pooled, arg, y, mu, inv, 0.01, prep=(gamma, M, y, N))
    gval2, sums2 = ops.pool_bwd_stats(gpool, pooled, arg, y, mu, inv, 0.01)
    sb = ops.sparse_bn_bwd_operand(gval2, arg, y, N, mu, inv, gamma, sums2, M)
    assert torch.equal(gval, gval2) and torch.equal(sums, sums2)
    for a, b in ((sa.alpha, sb.alpha), (sa.beta, sb.beta), (sa.sp_val, sb.sp_val)):
        assert torch.equal(a, b)
    assert sa.rows == sb.rows and torch.equal(sa.sp_arg, sb.sp_arg)
