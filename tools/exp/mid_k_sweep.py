"""t = a + b*K fits of the mid-size gemm_nt products (M = 65536, N = 128 / 256) on the 128-row kernels (tile hint 1) and the 256 x 256-tile
kernel (hint 2): is the distance from the MFMA time a fixed per-launch cost (overlap would help) or a slope (the inner loop)?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "sp-gan_amd"))
import torch
from spgan import ops, _lib
_lib.load()
torch.manual_seed(0)
dev = "cuda"
def t_launch(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 65536
empty = t_launch(lambda: ops.gemm_nt(torch.empty(256, 32, device=dev), torch.empty(64, 32, device=dev)))
print("host + launch floor of one eager gemm_nt call: %.1f us" % empty)
for N in (128, 256):
    for hint in (1, 2):
        rows = []
        for K in (128, 256, 512, 1024):
            A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05
            with ops.nt_tile_hint(hint):
                us = t_launch(lambda: ops.gemm_nt(A, W))
            rows.append((K, us))
        (k0, t0), (k1, t1) = rows[1], rows[3]
        b_ = (t1 - t0) / (k1 - k0); a_ = t0 - b_ * k0
        mfma = 2.0 * M * N / 157.3e6          # us per unit K at the 2.4 GHz peak
        print("N %4d hint %d: " % (N, hint) + "  ".join("K%d %.1f us (%.0f TF)" % (k, t, 2.0 * M * N * k / t / 1e6) for k, t in rows) +
              "   fit: a = %.1f us, b = %.4f us/K = %.0f %% of peak slope" % (a_, b_, 100 * mfma / b_))
