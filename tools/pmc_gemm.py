#!/usr/bin/env python3
"""Launch the dominant kernel (gemm_nt at D.fc2.0: BN+LeakyReLU prologue, column-statistics + pooling epilogue, output not
stored) and the EdgeConv2 conv_out GEMM a few times in isolation, for rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, separate runs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd")]
import torch
from spgan import ops
B, Npts = 32, 2048
M, N, K = B * Npts, 1024, 256
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
sc, sh = torch.rand(K, device="cuda") + 0.5, torch.randn(K, device="cuda") * 0.1
gamma, beta = torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda") * 0.1
for _ in range(6):
    ops.gemm_bn_pool(A, W, b, (gamma, beta, None, None), Npts, 0.01, pro=(sc, sh, 0.01))
M, N, K = 32 * 2048, 128, 1280
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
for _ in range(6):
    ops.gemm_nt(A, W, b)
torch.cuda.synchronize()
