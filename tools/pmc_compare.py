#!/usr/bin/env python3
"""spgan.gemm_nt and the vendor SGEMM (torch.mm) on the same two shapes, for rocprofv3 --pmc passes (diagnostics)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd")]
import torch
from spgan import ops
for (M, N, K) in [(65536, 1024, 256), (65536, 256, 1024)]:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05
    Wt = W.t().contiguous()
    for _ in range(4):
        ops.gemm_nt(A, W)
    for _ in range(4):
        torch.mm(A, Wt)
torch.cuda.synchronize()
