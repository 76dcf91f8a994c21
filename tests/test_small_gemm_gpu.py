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
