"""GPU: feature-space kNN on the matrix cores (k <= 10, 16 < C <= 128) -- exact-distance and model checks on ragged sizes."""
import pytest
import torch

import kernel_model as km
from test_kernels_gpu import knn_tie_aware, ops, rnd  # noqa: F401  (ops is a fixture)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,N,C,k", [(3, 333, 40, 7), (2, 130, 128, 10), (2, 97, 20, 10), (1, 4096, 64, 10), (4, 512, 64, 10), (2, 1000, 100, 3)])
def test_knn_mfma(ops, B, N, C, k):
    x = rnd("knnm.%d.%d.%d" % (B, N, C), (B * N, C), 0.5)
    idx = ops.knn(x, B, N, k, mode=0)
    knn_tie_aware(idx, x, B, N, k, tol=2e-5 * C)
    ref = km.knn(x, B, N, k, 0)
    agree = (idx == ref).all(dim=1).float().mean().item()
    assert agree >= 0.995, "row agreement with the fp32 model only %.4f" % agree
    assert torch.equal(ops.knn(x, B, N, k, mode=0), idx), "not deterministic"


def test_knn_mfma_duplicates_and_ties(ops):
    """Exact duplicates: rank 0 is dropped positionally (the lower index of equal distances comes first)."""
    B, N, C, k = 1, 256, 64, 10
    x = rnd("knnm.dup", (N, C), 0.5)
    x[1::2] = x[0::2]                               # every point has an exact twin
    idx = ops.knn(x.contiguous(), B, N, k, mode=0).cpu().long()
    ar = torch.arange(N)
    twin = ar ^ 1
    # the twin (distance == self distance up to rounding) must be among rank 0/1: either dropped as rank 0 or returned first
    first = idx[:, 0]
    assert ((first == twin) | (first == ar)).all()


@pytest.mark.parametrize("B,N,C,k", [(3, 333, 40, 7), (2, 97, 20, 10), (1, 31, 64, 10), (2, 12, 17, 10), (5, 129, 64, 1), (1, 4096, 64, 10), (32, 2048, 64, 10)])
def test_knn_tile_images_route_returns_the_same_indices(ops, B, N, C, k):
    """csrc/knn_pipe.hip (pre-split tile images, software-pipelined scan) keeps the arithmetic of the single-launch kernel:
    identical indices on ragged N (tiles past N carry +inf norms), C < 64 (zero-padded channels), one-tile shapes and k < 10."""
    import os
    if os.environ.get("SPGAN_KNN_BF16X3") == "0":
        pytest.skip("the A/B switch puts the single-launch route on the fp32-MFMA kernel: other rounding, near-ties may resolve differently")
    x = rnd("knnp.%d.%d.%d" % (B, N, C), (B * N, C), 0.5)
    from spgan import _lib
    assert _lib.load().spgan_knn_ws_bytes(B, N, C, k, 0) > 0
    try:
        ops.KNN_PIPELINED[0] = False
        ref = ops.knn(x, B, N, k, mode=0)
        ops.KNN_PIPELINED[0] = True
        idx = ops.knn(x, B, N, k, mode=0)
    finally:
        ops.KNN_PIPELINED[0] = True
    assert torch.equal(idx, ref)


def test_knn_tile_images_route_falls_through_outside_its_shapes(ops):
    from spgan import _lib
    lib = _lib.load()
    assert lib.spgan_knn_ws_bytes(2, 100, 3, 10, 1) == 0      # coordinate mode
    assert lib.spgan_knn_ws_bytes(2, 100, 128, 10, 0) == 0    # C > 64
    assert lib.spgan_knn_ws_bytes(2, 100, 64, 20, 0) == 0     # k > 10
    x = rnd("knnp.fall", (200, 128), 0.5)
    idx = torch.empty((200, 10), dtype=torch.int32, device=x.device)
    st = lib.spgan_knn_ws(x.data_ptr(), 2, 100, 128, 10, 0, idx.data_ptr(), None, 0, torch.cuda.current_stream().cuda_stream)
    assert st == 0 and torch.equal(idx, ops.knn(x, 2, 100, 10, mode=0))
