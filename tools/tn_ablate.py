#!/usr/bin/env python3
"""Ablation of gemm_tn prologue combinations at the D.L4 weight-gradient shape (development aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd"), ROOT]
import torch
from spgan import ops
from bench_kernels import timeit

def main():
    B, N = 32, 2048
    M, Na, Nb = B * N, 1024, 256
    dev = "cuda"
    for data in ("randn", "zeros"):
        mk = (lambda *s: torch.randn(*s, device=dev)) if data == "randn" else (lambda *s: torch.zeros(*s, device=dev))
        A, Bm = mk(M, Na), mk(M, Nb)
        sc, sh = torch.rand(Nb, device=dev) + 0.5, torch.randn(Nb, device=dev) * 0.1
        al, be = torch.rand(Na, device=dev) + 0.5, torch.randn(Na, device=dev) * 0.1
        spv = torch.randn(B, Na, device=dev)
        spa = (torch.randint(0, N, (B, Na), device=dev) + torch.arange(B, device=dev)[:, None] * N).int()
        sa = ops.SparseAffine(A, al, be, spv, spa, N)
        sa0 = ops.SparseAffine(A, al, be, torch.zeros_like(spv), torch.full_like(spa, -1), N)
        for tag, fn in (("plain", lambda: ops.gemm_tn(A, Bm)), ("b-affine", lambda: ops.gemm_tn(A, Bm, pro=(sc, sh, 0.01))),
                        ("a-affine+sparse", lambda: ops.gemm_tn(sa, Bm)), ("a-affine+sparse(-1)", lambda: ops.gemm_tn(sa0, Bm)),
                        ("both", lambda: ops.gemm_tn(sa, Bm, pro=(sc, sh, 0.01)))):
            t = timeit(fn, n=10)
            print("%-6s %-20s %8.1f us  %6.1f TF" % (data, tag, t * 1e3, 2.0 * M * Na * Nb / t / 1e9))

if __name__ == "__main__":
    main()
