#!/usr/bin/env python3
"""Micro-timings of the individual HIP kernels at the C2 shapes (B=32, N=2048).  Development aid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd"), ROOT]
import torch
from spgan import ops

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n   # ms

def main():
    B, N, k = 32, 2048, 10
    M = B * N
    dev = "cuda"
    print(torch.cuda.get_device_name(0))
    rows = []
    for (m, n, kk, tag) in [(M, 64, 3, "D.L1"), (M, 128, 64, "D.L2"), (M, 256, 128, "D.L3"), (M, 1024, 256, "D.L4"),
                            (M, 128, 131, "G.head0"), (M, 128, 128, "G.head2"), (M, 256, 128, "G.tail0"), (M, 64, 256, "G.tail2"),
                            (M * k, 128, 64, "EC2.conv_w3"), (M, 128, 1280, "EC2.conv_out"), (M, 1280, 128, "EC2.dT"), (M, 256, 1024, "D.dgrad4")]:
        A = torch.randn(m, kk, device=dev); W = torch.randn(n, kk, device=dev) * 0.1; b = torch.randn(n, device=dev)
        t = timeit(lambda: ops.gemm_nt(A, W, b))
        ts = timeit(lambda: ops.gemm_nt(A, W, b, stats=True))
        rows.append("gemm_nt %-14s M=%7d N=%5d K=%5d  %8.3f ms  %7.1f TF   (+stats %8.3f ms)" % (tag, m, n, kk, t, 2.0 * m * n * kk / t / 1e9, ts))
    for (m, na, nb, tag) in [(M, 1024, 256, "D.dW4"), (M, 256, 128, "D.dW3"), (M, 64, 3, "D.dW1"), (M * k, 128, 64, "EC2.dW2"), (M, 128, 1280, "EC2.dWo")]:
        A = torch.randn(m, na, device=dev); Bm = torch.randn(m, nb, device=dev)
        t = timeit(lambda: ops.gemm_tn(A, Bm))
        rows.append("gemm_tn %-14s M=%7d Na=%5d Nb=%5d %8.3f ms  %7.1f TF" % (tag, m, na, nb, t, 2.0 * m * na * nb / t / 1e9))
    for C, mode in ((3, 1), (3, 0), (64, 0), (128, 0)):
        x = torch.randn(M, C, device=dev)
        t = timeit(lambda: ops.knn(x, B, N, k, mode), n=5, warm=1)
        rows.append("knn C=%3d mode=%d  %8.3f ms" % (C, mode, t))
    idx = ops.knn(torch.randn(M, 3, device=dev), B, N, k, 1)
    rows.append("csr_build        %8.3f ms" % timeit(lambda: ops.csr_build(idx, B, N)))
    y = torch.randn(M, 1024, device=dev)
    rows.append("maxpool [M,1024] %8.3f ms  %6.1f GB/s" % ((t := timeit(lambda: ops.maxpool(y, B, N))), y.numel() * 4 / t / 1e6))
    rows.append("colstats[M,1024] %8.3f ms" % timeit(lambda: ops.colstats(y, M)))
    print("\n".join(rows))

if __name__ == "__main__":
    main()
