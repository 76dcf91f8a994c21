#!/usr/bin/env python3
"""Calibration: vendor SGEMM (torch.mm -> rocBLAS/hipBLASLt, fp32) vs spgan.gemm_nt on the step's main shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd"), ROOT]
import torch
from spgan import ops
from bench_kernels import timeit
torch.backends.cuda.matmul.allow_tf32 = False
for (M, N, K) in [(65536, 1024, 256), (65536, 256, 1024), (65536, 128, 1280), (65536, 256, 128), (65536, 256, 256), (655360, 128, 64)]:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05
    Wt = W.t().contiguous()
    tv = timeit(lambda: torch.mm(A, Wt), n=20)
    tv2 = timeit(lambda: torch.mm(A, W.t()), n=20)
    to = timeit(lambda: ops.gemm_nt(A, W), n=20)
    f = 2.0 * M * N * K / 1e9
    print("M=%7d N=%5d K=%5d  vendor(NN) %7.1f us %6.1f TF | vendor(NT) %7.1f us %6.1f TF | spgan %7.1f us %6.1f TF" % (M, N, K, tv * 1e3, f / tv, tv2 * 1e3, f / tv2, to * 1e3, f / to))
