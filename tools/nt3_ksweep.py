#!/usr/bin/env python3
"""K sweep of the split-bf16 256-row-tile gemm_nt at M = 65536, N = 1024 (1024 tiles of 256 x 256: four rounds over 256 CUs): slope = time per
k-tile of 16, intercept = ramp + epilogue; with the BatchNorm prologue + statistics/pooling epilogue (D.fc2.0's flavour) and plain."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd"), ROOT]
import torch
from spgan import ops
ops.set_mfma_operands("bf16x3")
if "--img" in sys.argv:
    _img = {}
    def _prov(Wt):
        k = (Wt.data_ptr(), tuple(Wt.shape))
        if k not in _img: _img[k] = ops.split_image(Wt)
        return _img[k]
    ops.w_image_provider = _prov
M, N = 65536, 1024
def timeit(f, reps=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
b = torch.randn(N, device="cuda"); gamma, beta = torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda")
rows = []
for K in (64, 128, 256, 512, 1024):
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.1
    sc = torch.rand(K, device="cuda") + 0.5; sh = torch.randn(K, device="cuda") * 0.3
    with ops.nt_tile_hint(2):
        tp = timeit(lambda: ops.gemm_bn_pool(A, W, b, (gamma, beta, None, None), 2048, 0.2, pro=(sc, sh, 0.2)))
        tq = timeit(lambda: ops.gemm_bn_pool(A, W, b, (gamma, beta, None, None), 2048, 0.2))
        Y = torch.empty(M, N, device="cuda")
        tl = timeit(lambda: ops.gemm_nt(A, W, b, out=Y))
    rows.append((K, tp, tq, tl))
    print("K=%4d  pro+pool %7.1f us   pool %7.1f us   plain+store %7.1f us   (TF %5.1f / %5.1f / %5.1f)" % (K, tp, tq, tl, *(2.0 * M * N * K / 1e6 / t for t in (tp, tq, tl))), flush=True)
(k0, a0, b0, c0), (k1, a1, b1, c1) = rows[2], rows[4]
for name, t0, t1 in (("pro+pool", a0, a1), ("pool", b0, b1), ("plain+store", c0, c1)):
    slope = (t1 - t0) / ((k1 - k0) / 16)
    print("%-12s per k-tile %.2f us per launch (4 rounds) = %.0f ns per tile-k-step; intercept %.1f us; MFMA-bound per k-tile at 2.4 GHz: %.2f us" % (name, slope, slope / 4 * 1e3, t0 - slope * k0 / 16, 4 * 96 * 32 / 2.4e3))
