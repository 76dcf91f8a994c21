#!/usr/bin/env python3
"""A few eager train steps (C2 shape) for rocprofv3 --pmc passes over the whole kernel population (HBM bytes of the
HBM-bound kernels, MfmaUtil of the GEMMs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd"), ROOT]
import torch
import bench
import spgan
dev = torch.device("cuda", 0)
MFMA = next((sys.argv[i + 1] for i, a in enumerate(sys.argv[:-1]) if a == "--mfma"), "f32")      # f32 | f16 | bf16x3 (bench.py --mfma)
spgan.ops.set_mfma_operands(MFMA)
G, D = bench.build_models(dev)
tr = spgan.TrainStep(G, D, gan="wgan", use_gp=True, lambda_gp=10.0, lr_g=1e-4, lr_d=1e-4)
x, real, zs, alpha = bench.make_inputs(dev, 0, bench.PER_GPU_BATCH)
for i in range(3):
    tr.step(x, real, zs[0], zs[1], alpha=alpha)
torch.cuda.synchronize()
