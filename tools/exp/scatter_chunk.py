"""edge_scatter at the benchmark geometry (B=32, N=2048, k=10, H=64, F=128) with the library variants built with -DSPGAN_SCATTER_CHUNK=4/8/16
(development aid).  usage: scatter_chunk.py lib4.so lib8.so lib16.so"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "sp-gan_amd"))
import torch
from spgan import _lib
res = {}
ref = None
for path in sys.argv[1:]:
    _lib.LIB_PATH = os.path.abspath(path); _lib._LIB = None if hasattr(_lib, "_LIB") else None
    for attr in ("_lib", "_handle", "_LIB", "_cached"):
        if hasattr(_lib, attr):
            setattr(_lib, attr, None)
    from spgan import ops
    torch.manual_seed(0)
    B, N, k, H, F_ = 32, 2048, 10, 64, 128
    M = B * N
    feat = torch.randn(M, 64, device="cuda")
    idx = ops.knn(feat, B, N, k, mode=0)
    rowptr, src = ops.csr_build(idx, B, N)
    PQR = torch.randn(M, H + 2 * F_, device="cuda")
    g1 = torch.randn(M * k, H, device="cuda"); gy = torch.randn(M * k, F_, device="cuda")
    v = lambda n: torch.rand(n, device="cuda") + 0.5
    args = (g1, gy, PQR, idx, rowptr, src, v(H), v(H), v(H), v(H), torch.randn(2 * H, device="cuda"), v(F_), v(F_), v(F_), v(F_), torch.randn(2 * F_, device="cuda"))
    out = ops.edge_scatter(*args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.edge_scatter(*args)
    e1.record(); torch.cuda.synchronize()
    print(path, "%.1f us" % (e0.elapsed_time(e1) / 20 * 1e3), "max in-degree", int((rowptr[1:] - rowptr[:-1]).max()))
