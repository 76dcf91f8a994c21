#!/usr/bin/env python3
"""A/B of the small-row gemm_nt products (W [1024,256] x [256,256]): the 32 x 32-tile one-shot-K kernel (csrc/gemm_mid.hip, tile_hint 0)
against the 128 x 64-tile kernel (tile_hint 1).  Launches back to back inside a captured graph of 20 (what the train step sees), cold-ish
operands (a 64 MB fill between launches evicts nothing from the 256 MB MALL but keeps the L2 honest)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd")]
import torch
from spgan import ops
for M, N, K in ((1024, 256, 256), (2048, 64, 128), (512, 256, 256)):
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda")
    for hint in (1, 0):
        with ops.nt_tile_hint(hint):
            for _ in range(3):
                ops.gemm_nt(A, W, exact=True)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                with torch.cuda.graph(g, stream=s):
                    for _ in range(20):
                        ops.gemm_nt(A, W, exact=True)
            ts = []
            for _ in range(10):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / 20)
            ts.sort()
            print("M=%d N=%d K=%d  tile_hint %d: %.2f us per launch (median of 10 replays of 20)" % (M, N, K, hint, ts[len(ts) // 2]))
