#!/usr/bin/env python3
"""gemm_tn at the step's weight-gradient shapes: exact-fp32 MFMA kernel against the split-bf16 kernels (ops.TN_SPLIT_BF16), incl. the split-sum
reduction; us per launch in a hot loop, TF fp32-equivalent, error against the float64 product."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd"), ROOT]
import torch
from spgan import ops
def timeit(f, reps=16):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
torch.manual_seed(0)
for (M, Na, Nb, pro) in ((65536, 256, 256, 1), (65536, 256, 256, 2), (65536, 256, 128, 2), (65536, 128, 1280, 0), (65536, 256, 128, 0), (65536, 128, 128, 0), (196608, 256, 256, 1), (196608, 256, 128, 0), (65536, 320, 64, 0), (65536, 64, 256, 0)):
    A = torch.randn(M, Na, device="cuda") * 1e-3; B = torch.randn(M, Nb, device="cuda")
    sc, sh = torch.rand(Nb, device="cuda") + 0.5, torch.randn(Nb, device="cuda") * 0.3
    p = (sc, sh, 0.2) if pro == 1 else None
    b = B if pro != 1 else torch.nn.functional.leaky_relu(B * sc + sh, 0.2)
    Aop = A
    if pro == 2:      # the lazy two-tensor A operand (BatchNorm backward: p*g + q*y + r)
        y = torch.randn(M, Na, device="cuda"); coef = torch.stack([torch.rand(Na, device="cuda") + 0.5, torch.randn(Na, device="cuda") * 1e-3, torch.randn(Na, device="cuda") * 1e-4])
        Aop = ops.Affine2(A, y, coef)
    ref = (Aop.dense() if pro == 2 else A).double().t() @ b.double()
    row = []
    for name, mode, split in (("f32", "f32", False), ("x3", "bf16x3", True)):
        ops.set_mfma_operands(mode); ops.TN_SPLIT_BF16[0] = split
        out = ops.gemm_tn(Aop, B, pro=p)
        err = ((out.double() - ref).norm() / ref.norm()).item()
        t = timeit(lambda: ops.gemm_tn(Aop, B, pro=p))
        row.append("%s %6.1f us (%5.1f TF) err %.1e" % (name, t, 2.0 * M * Na * Nb / 1e6 / t, err))
    ops.set_mfma_operands("f32"); ops.TN_SPLIT_BF16[0] = False
    print("M=%6d Na=%4d Nb=%4d %s | %s" % (M, Na, Nb, ("plain   ", "affine B", "lazy A  ")[pro], " | ".join(row)), flush=True)
