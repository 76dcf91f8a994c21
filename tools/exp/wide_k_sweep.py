"""Fixed per-tile cost of the 256 x 256-tile gemm_nt: time the D.fc2.0-shaped launch (BatchNorm + LeakyReLU prologue, statistics + pooling
epilogue, nothing stored) at several K and fit t = a + b*K (development aid for a persistent-workgroup variant)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "sp-gan_amd"))
import torch
from spgan import ops, _lib
_lib.load()
torch.manual_seed(0)
dev = "cuda"
def t_launch(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in (65536, 196608):
    rows = []
    for K in (64, 128, 256, 512, 1024):
        A = torch.randn(M, K, device=dev); W = torch.randn(1024, K, device=dev) * 0.05; b = torch.randn(1024, device=dev)
        sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
        bn = (torch.ones(1024, device=dev), torch.zeros(1024, device=dev), None, None)
        us = t_launch(lambda: ops.gemm_bn_pool(A, W, b, bn, 2048, 0.01, pro=(sc, sh, 0.01)))
        us_fin = t_launch(lambda: ops.gemm_bn_pool(A[:256], W, b, bn, 256, 0.01, pro=(sc, sh, 0.01)))   # the finalize launches + host cost, roughly
        rows.append((K, us, us_fin))
        print("M %6d K %4d: %8.1f us  (tiny-M call %6.1f us)  %6.1f TF" % (M, K, us, us_fin, 2.0 * M * 1024 * K / us / 1e6))
    (k0, t0, _), (k1, t1, _) = rows[2], rows[3]
    b_ = (t1 - t0) / (k1 - k0); a_ = t0 - b_ * k0
    tiles = M // 256 * 4
    print("  fit over K=256..512: t = %.1f + %.3f*K us; per tile round (%d tiles / 256 CUs = %.0f rounds): fixed %.2f us, per k-tile(32) %.2f us" %
          (a_, b_, tiles, tiles / 256, a_ / (tiles / 256), b_ * 32 / (tiles / 256)))
