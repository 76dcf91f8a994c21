#!/usr/bin/env python3
"""A/B of the feature-space kNN kernels at the train step's shape (B=32, N=2048, C=64, k=10): fp32-MFMA inner products
(SPGAN_KNN_BF16X3=0) against the split-bf16 ones, and the pre-split tile images + software-pipelined scan of csrc/knn_pipe.hip (default; its time includes the pre-pass launch).  Each variant runs in its own process (the switch is read once);
interleaved launches of one kernel, GPU kept busy, median of 30.  `python tools/knn_ab.py [N [B]]` (C4: 4096 16).  Output -> profiles/r0X_knn_ab.txt."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch
sys.path.insert(0, "%s/sp-gan_amd")
from spgan import ops, fixture_rng as fr
import os
ops.KNN_PIPELINED[0] = os.environ.get("KNN_AB_PIPE", "0") == "1"
B, N, C, k = %d, %d, 64, 10
x = fr.normal("knn.ab", (B * N, C), 0.5).cuda()
for _ in range(5):
    idx = ops.knn(x, B, N, k, 0)
torch.cuda.synchronize()
ts = []
for _ in range(30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); idx = ops.knn(x, B, N, k, 0); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print("median %%.1f us  min %%.1f us  (idx checksum %%d)" %% (ts[len(ts) // 2], ts[0], int(idx.long().sum().item())))
''' % (ROOT, int(sys.argv[2]) if len(sys.argv) > 2 else 32, int(sys.argv[1]) if len(sys.argv) > 1 else 2048)

print("# B=%d N=%d C=64 k=10" % (int(sys.argv[2]) if len(sys.argv) > 2 else 32, int(sys.argv[1]) if len(sys.argv) > 1 else 2048))
for tag, env, pipe in (("fp32 MFMA (32x32x2_f32)", "0", "0"), ("split bf16 (6 x 32x32x16_bf16)", "1", "0"),
                       ("split bf16, tile images + pipelined", "1", "1")):
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, SPGAN_KNN_BF16X3=env, KNN_AB_PIPE=pipe), capture_output=True, text=True)
    print("%-38s %s" % (tag, (r.stdout.strip() or r.stderr.strip()[-300:])))
