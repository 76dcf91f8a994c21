#!/usr/bin/env python3
"""Times ops.gemm_dual (csrc/gemm_dual.hip) against the two launches it replaces (gemm_tn + gemm_nt_bnbwd) at the step's shapes:
the EdgeBlock's conv_w.3 backward (E = 655,360 edges, per-edge operand) and the Discriminator's mlps.3 backward (M = 65,536).
    python tools/dual_bench.py > gpurun_out/dual_bench.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sp-gan_amd")]
import torch   # noqa: E402

from spgan import ops   # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    Na, Nb = 128, 64
    for name, M, edge in (("EdgeConv2 conv_w.3 (per-edge operand)", 655360, True), ("D mlps.3", 65536, False), ("D mlps.3 x 2 passes", 131072, False)):
        g, y = torch.randn(M, Na, device=dev), torch.randn(M, Na, device=dev)
        coef = torch.randn(3, Na, device=dev)
        dy = ops.Affine2(g, y, coef)
        W = torch.randn(Na, Nb, device=dev) * 0.1
        Wt = W.t().contiguous()
        sc, sh, mu, iv = (torch.randn(Nb, device=dev) for _ in range(4))
        if edge:
            k = 10
            P = torch.randn(M // k, Nb + 256, device=dev)
            idx = torch.randint(0, M // k, (M // k, k), device=dev, dtype=torch.int32)
            e = (idx, torch.randn(Nb, device=dev))
            yref = P[:, :Nb]
        else:
            e, yref = None, torch.randn(M, Nb, device=dev)
        t_dual = timeit(lambda: (ops.gemm_dual(dy, W, yref, sc, sh, mu, iv, 0.01, edge=e), ops.flush_tn()))
        t_two = timeit(lambda: (ops.gemm_tn(dy, yref, pro=(sc, sh, 0.01), edge=e, defer=True), ops.gemm_nt_bnbwd(dy, Wt, yref, sc, sh, mu, iv, 0.01, edge=e), ops.flush_tn()))
        flops = 4.0 * M * Na * Nb
        byt = M * (2 * Na + (2 if edge else 1) * Nb + Nb) * 4.0
        print("%-40s M=%7d  dual %7.1f us (%5.1f TF, %4.2f TB/s incl. gathers)   two launches %7.1f us   x%.2f" % (
            name, M, t_dual, flops / t_dual / 1e6, byt / t_dual / 1e6, t_two, t_two / t_dual))


if __name__ == "__main__":
    main()
