"""EXPERIMENT: the split plan of gemm_tn (workgroups per launch = tiles x splits, SPGAN_TN_WGS; default 512 = two per CU) at the step's weight-gradient shapes."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd")]
if not os.environ.get("TN_CHILD"):
    for w in (256, 384, 512, 768, 1024, 1536):
        env = dict(os.environ, TN_CHILD="1", SPGAN_TN_WGS=str(w))
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=env)
    sys.exit(0)
import torch
from spgan import ops
def timeit(f, reps=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
row = []
for (M, Na, Nb) in ((65536, 256, 256), (65536, 128, 1280), (65536, 256, 128), (65536, 128, 128), (65536, 320, 64), (196608, 256, 256), (655360, 128, 64)):
    A = torch.randn(M, Na, device="cuda"); B = torch.randn(M, Nb, device="cuda")
    t = timeit(lambda: ops.gemm_tn(A, B))
    row.append("%dx%dx%d %6.1f us (%5.1f TF)" % (M, Na, Nb, t, 2.0 * M * Na * Nb / 1e6 / t))
print("WGS=%5s | " % os.environ["SPGAN_TN_WGS"] + " | ".join(row), flush=True)
