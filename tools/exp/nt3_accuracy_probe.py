"""EXPERIMENT: error statistics of gemm_nt against the float64 product: exact-fp32 MFMA vs split-bf16 (128-row and 256-row-tile kernels); signed
mean (bias) and RMS relative error, on zero-mean and on all-positive operands (where a truncating accumulation shows as a bias)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd")]
from spgan import ops
torch.manual_seed(1)
for (M, N, K) in ((4096, 256, 256), (4096, 256, 1280), (4096, 256, 64)):
    for kind in ("randn", "positive"):
        A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.1
        if kind == "positive":
            A, W = A.abs() + 0.1, W.abs() + 0.01
        ref = A.double() @ W.double().t()
        scale = (A.double().abs() @ W.double().abs().t())          # sum |a||b|: the natural error scale of a length-K dot product
        row = []
        for name, mode, hint in (("f32", "f32", 0), ("x3-128", "bf16x3", 1), ("x3-wide", "bf16x3", 2)):
            ops.set_mfma_operands(mode)
            with ops.nt_tile_hint(hint):
                Y = ops.gemm_nt(A, W)
            e = (Y.double() - ref) / scale
            cs = ((Y.double().sum(0) - ref.sum(0)) / scale.sum(0))          # error of the column sums (what BatchNorm statistics see)
            row.append("%s bias %+.2e rms %.2e | colsum err mean %+.2e rms %.2e" % (name, e.mean().item(), e.pow(2).mean().sqrt().item(), cs.mean().item(), cs.pow(2).mean().sqrt().item()))
        ops.set_mfma_operands("f32")
        print("M=%d N=%d K=%4d %-8s | %s   (2^-24 = %.2e)" % (M, N, K, kind, " | ".join(row), 2.0 ** -24), flush=True)
