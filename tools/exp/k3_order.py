"""Which summation order does the fp32 MFMA kernel realise for a K = 3 product?  (development aid for gemm_nt_k4_kernel)"""
import itertools, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "sp-gan_amd"))
import torch
from spgan import ops, _lib
_lib.load()
torch.manual_seed(0)
M, N = 4096, 64
A = torch.randn(M, 3, device="cuda"); W = torch.randn(N, 3, device="cuda") * 0.3
with ops.nt_tile_hint(1):
    ref = ops.gemm_nt(A, W)
sk = ops.gemm_nt(A, W)
print("streaming == mfma:", float((sk == ref).float().mean()))
Ad, Wd = A.double(), W.double()
def fma(a, b, c):
    return (a * b + c).float().double()
for perm in itertools.permutations(range(3)):
    acc = torch.zeros(M, N, dtype=torch.float64, device="cuda")
    for k in perm:
        acc = fma(Ad[:, k:k + 1], Wd[:, k].unsqueeze(0), acc)
    print("fma chain", perm, float((acc.float() == ref).float().mean()))
for pair in ((0, 2, 1), (0, 1, 2), (1, 2, 0)):
    i, j, l = pair
    t = (Ad[:, i:i + 1] * Wd[:, i].unsqueeze(0) + Ad[:, j:j + 1] * Wd[:, j].unsqueeze(0)).float().double()
    acc = fma(Ad[:, l:l + 1], Wd[:, l].unsqueeze(0), t)
    print("exact pair", (i, j), "then", l, float((acc.float() == ref).float().mean()))
