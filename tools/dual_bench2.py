#!/usr/bin/env python3
"""gemm_dual at the Discriminator's and the EdgeBlock's shapes through the product entry point (spgan_gemm_dual): us per launch, fraction of
the fp32 matrix peak; beside it the round-4 kernel when tools/exp/libdual_abl.so (tools/exp/dual_abl.hip) is present."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sp-gan_amd")]
from spgan import _lib
from spgan._lib import GemmDualArgs
lib = _lib.load()
old = None
if os.path.exists(os.path.join(ROOT, "tools/exp/libdual_abl.so")):
    old = C.CDLL(os.path.join(ROOT, "tools/exp/libdual_abl.so"))
    old.abl_gemm_dual.argtypes = [C.POINTER(GemmDualArgs), C.c_int, C.c_void_p]
    old.abl_gemm_dual_wgs.argtypes = [C.c_int] * 4
dev = "cuda"


def run(M, Na, Nb, mode, ek=0):
    A, A2 = torch.randn(M, Na, device=dev), torch.randn(M, Na, device=dev)
    W = torch.randn(Na, Nb, device=dev) * 0.1
    vec = [torch.randn(max(Na, Nb), device=dev) for _ in range(9)]
    radd = torch.randn(M, Nb, device=dev)
    if ek:
        Bm = torch.randn(M // ek, Nb + 256, device=dev)
        idx = torch.randint(0, M // ek, (M // ek, ek), device=dev, dtype=torch.int32)
    else:
        Bm = torch.randn(M, Nb, device=dev)
    res = []
    for which, L, wgs in (("new", lib, lib.spgan_gemm_dual_wgs), ("r04", old, old.abl_gemm_dual_wgs if old else None)):
        if L is None:
            continue
        runs = wgs(M, Na, Nb, ek)
        G = torch.empty(M, Nb, device=dev); stats = torch.empty(runs, Nb, 2, device=dev); ws = torch.empty(runs, Na, Nb, device=dev); cs = torch.empty(runs, Na, device=dev)
        a = GemmDualArgs()
        p = lambda t: t.data_ptr()
        a.A = p(A); a.lda = Na; a.A2 = p(A2); a.lda2 = Na; a.p = p(vec[0]); a.q = p(vec[1]); a.r = p(vec[2]); a.W = p(W); a.ldw = Nb; a.B = p(Bm); a.ldb = Bm.shape[1]
        a.b_scale = p(vec[3]); a.b_shift = p(vec[4]); a.b_mean = p(vec[5]); a.b_invstd = p(vec[6]); a.slope = 0.01
        a.G = p(G); a.ldg = Nb; a.stats = p(stats); a.ws = p(ws); a.M, a.Na, a.Nb = M, Na, Nb; a.a_mode = mode; a.a_slope = 0.01
        if ek:
            a.e_idx = p(idx); a.e_k = ek; a.e_bias = p(vec[8])
        if mode == 2:
            a.bias = p(vec[7]); a.rowadd = p(radd); a.ld_rowadd = Nb; a.colsum_ws = p(cs)
        s = torch.cuda.current_stream().cuda_stream
        call = (lambda: L.spgan_gemm_dual(C.byref(a), s)) if which == "new" else (lambda: L.abl_gemm_dual(C.byref(a), 0, s))
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record(); torch.cuda.synchronize()
        res.append((which, e0.elapsed_time(e1) / 20 * 1e3, (G.double().sum().item(), ws.double().sum().item())))
    ideal = 4.0 * M * Na * Nb / 157.3e6
    print("M=%7d Na=%3d Nb=%3d mode=%d k=%2d  ideal %6.1f us | " % (M, Na, Nb, mode, ek, ideal) +
          "   ".join("%s %7.1f us (%.3f of peak)" % (w, t, ideal / t) for w, t, _ in res) +
          ("   | checksums agree to %.1e / %.1e" % tuple(abs(res[0][2][i] - res[1][2][i]) / max(abs(res[1][2][i]), 1e-30) for i in (0, 1)) if len(res) == 2 else ""))


run(65536, 256, 256, 2)
run(65536, 256, 128, 1)
run(65536, 256, 128, 0)
run(65536, 128, 64, 1)
run(65536, 128, 64, 0)
run(655360, 128, 64, 1, ek=10)
run(196608, 256, 256, 2)
