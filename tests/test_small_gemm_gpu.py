"""GPU: the M <= 64 gemm_nt path (per-shape linears: D's fc head, G's global_conv) against the torch model."""
import pytest
import torch

import kernel_model as km
from test_kernels_gpu import close, ops, rnd  # noqa: F401  (ops is a fixture)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K", [(32, 512, 1024), (32, 256, 512), (37, 6, 1280), (4, 1, 64), (64, 130, 2052), (1, 3, 4), (16, 64, 256)])
def test_small_m_linear(ops, M, N, K):
    A, W, b = rnd("s.A%d.%d" % (M, K), (M, K)), rnd("s.W%d.%d" % (N, K), (N, K), 0.1), rnd("s.b%d" % N, (N,))
    close(ops.gemm_nt(A, W, b), km.gemm_nt(A, W, b), what="plain")
    close(ops.gemm_nt(A, W, None), A @ W.t(), what="matmul")
    close(ops.gemm_nt(A, W, b, act=ops.ACT_LRELU, slope=0.01), km.gemm_nt(A, W, b, act=1, slope=0.01), what="lrelu")
    close(ops.gemm_nt(A, W, b, act=ops.ACT_TANH), km.gemm_nt(A, W, b, act=2), what="tanh")
    sc, sh = rnd("s.sc%d" % K, (K,)).abs() + 0.5, rnd("s.sh%d" % K, (K,), 0.3)
    close(ops.gemm_nt(A, W, b, pro=(sc, sh, 0.01)), km.gemm_nt(A, W, b, pro=(sc, sh, 0.01)), what="affine")
    ref = rnd("s.ref%d.%d" % (M, N), (M, N))
    close(ops.gemm_nt_maskout(A, W, ref, 0.01), km.gemm_nt_maskout(A, W, ref, 0.01), what="maskout")
    # statistics requested -> generic kernel; same numbers up to summation order
    y, mean, var = ops.gemm_nt(A, W, b, stats=True)
    close(y, ops.gemm_nt(A, W, b), what="generic==small")


def test_small_m_rowbias_and_views(ops):
    M, N, K = 24, 40, 128
    big = rnd("sv.A", (M, 200))
    A = big[:, 40:40 + K]
    W = rnd("sv.W", (N, 640), 0.1)[:, 512:]
    rb = rnd("sv.rb", (M // 8, N))
    close(ops.gemm_nt(A, W, None, rowbias=rb, rows_per_group=8, act=ops.ACT_LRELU, slope=0.2),
          km.gemm_nt(A, W, None, rowbias=rb, rows_per_group=8, act=1, slope=0.2), what="rowbias")
    A2 = big[:, 3:3 + K]                        # unaligned view -> generic scalar-load kernel
    close(ops.gemm_nt(A2, W), km.gemm_nt(A2, W), what="unaligned")


def test_small_m_deterministic(ops):
    A, W = rnd("sd.A", (32, 1024)), rnd("sd.W", (512, 1024), 0.1)
    y0 = ops.gemm_nt(A, W)
    for _ in range(3):
        assert torch.equal(ops.gemm_nt(A, W), y0)


@pytest.mark.parametrize("M,N,K", [(32, 128, 128), (32, 512, 128), (32, 128, 512), (4, 64, 256), (64, 20, 64), (17, 7, 36)])
def test_small_m_statistics_epilogues(M, N, K):
    """M <= 64 with column statistics (LINEAR + stats, BatchNorm-backward epilogue): served by the small-M kernel (one workgroup walks
    all rows of its columns) instead of a single 128-row MFMA tile."""
    import kernel_model as km
    from spgan import ops
    from test_kernels_gpu import close, rnd
    A, W, b = rnd("sm.A%d%d" % (M, K), (M, K)), rnd("sm.W%d%d" % (N, K), (N, K), 0.2), rnd("sm.b%d" % N, (N,))
    sc, sh = rnd("sm.sc%d" % K, (K,)).abs() + 0.5, rnd("sm.sh%d" % K, (K,), 0.3)
    for pro in (None, (sc, sh, 0.01)):
        y, m, v = ops.gemm_nt(A, W, b, pro=pro, stats=True, act=1, slope=0.2)
        y2, m2, v2 = km.gemm_nt(A, W, b, pro=pro, stats=True, act=1, slope=0.2)
        close(y, y2, rtol=5e-5, what="y"); close(m, m2, atol=2e-5, what="mean"); close(v, v2, rtol=1e-4, atol=1e-6, what="var")
        ref = rnd("sm.ref%d%d" % (M, N), (M, N))
        bsc, bsh = rnd("sm.bsc%d" % N, (N,)), rnd("sm.bsh%d" % N, (N,), 0.3)
        mean, inv = rnd("sm.mu%d" % N, (N,), 0.2), rnd("sm.inv%d" % N, (N,)).abs() + 0.5
        for name, a_, b_ in zip(("g", "s0", "s1"), ops.gemm_nt_bnbwd(A, W, ref, bsc, bsh, mean, inv, 0.01, pro=pro, bias=b),
                                km.gemm_nt_bnbwd(A, W, ref, bsc, bsh, mean, inv, 0.01, pro=pro, bias=b)):
            close(a_, b_, rtol=1e-4, atol=5e-4, what="bnbwd " + name)
