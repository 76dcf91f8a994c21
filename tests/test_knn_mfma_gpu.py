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
