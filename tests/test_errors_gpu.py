"""GPU: error behaviour of the boundary (status codes become exceptions, nothing exits or falls back)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_argument_errors():
    from spgan import metrics, ops
    x = torch.randn(2 * 16, 3, device="cuda")
    with pytest.raises(RuntimeError, match="status -22"):
        ops.knn(x, 2, 16, 16, mode=1)                               # k + 1 > N
    with pytest.raises(RuntimeError, match="status -22"):
        ops.knn(torch.randn(32, 200, device="cuda"), 2, 16, 4, mode=0)   # C > 128
    with pytest.raises(ValueError):
        ops.gemm_nt(torch.randn(8, 5, device="cuda"), torch.randn(4, 6, device="cuda"))
    with pytest.raises(RuntimeError, match="GPU"):
        ops.gemm_nt(torch.randn(8, 4), torch.randn(4, 4))           # CPU tensors: there is no CPU path
    with pytest.raises(TypeError):
        ops.gemm_nt(torch.randn(8, 4, device="cuda").double(), torch.randn(4, 4, device="cuda"))
    with pytest.raises(ValueError):
        ops.gemm_nt(torch.randn(8, 4, device="cuda").t(), torch.randn(4, 8, device="cuda"))   # column stride != 1
    g = (torch.ones(4, device="cuda"), torch.zeros(4, device="cuda"), None, None)
    with pytest.raises(ValueError):
        ops.gemm_bn_pool(torch.randn(200, 4, device="cuda"), torch.randn(4, 4, device="cuda"), None, g, 100, 0.01)   # rows % 128
    with pytest.raises(RuntimeError, match="status -22"):
        metrics.pairwise_cd(torch.randn(1, 5000, 3, device="cuda"), torch.randn(1, 8, 3, device="cuda"))       # N > 4096
    with pytest.raises(ValueError):
        metrics.ChamferDistance()(torch.randn(2, 8, 2, device="cuda"), torch.randn(2, 8, 3, device="cuda"))


def test_smallest_and_ragged_sizes():
    """k+1 == N, single shape, sizes that are not multiples of any tile."""
    import kernel_model as km
    from spgan import ops
    x = torch.randn(11, 3, device="cuda")
    idx = ops.knn(x, 1, 11, 10, mode=1)
    assert torch.equal(idx, km.knn(x, 1, 11, 10, 1))
    assert (torch.sort(idx, dim=1)[0] != torch.arange(11, device="cuda").view(-1, 1)).all()      # rank 0 (itself) is dropped
    A, W = torch.randn(1, 4, device="cuda"), torch.randn(1, 4, device="cuda")
    assert torch.allclose(ops.gemm_nt(A, W), A @ W.t(), atol=1e-6)
    A, W = torch.randn(129, 7, device="cuda"), torch.randn(5, 7, device="cuda")
    assert torch.allclose(ops.gemm_nt(A, W), A @ W.t(), atol=1e-5)
    Bm = torch.randn(129, 3, device="cuda")
    assert torch.allclose(ops.gemm_tn(A, Bm), A.t() @ Bm, atol=1e-4)
