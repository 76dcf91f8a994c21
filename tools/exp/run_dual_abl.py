"""Ablations of gemm_dual at the Discriminator's shapes (tools/exp/dual_abl.hip: the product kernel with runtime switches)."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sp-gan_amd")]
from spgan import _lib
from spgan._lib import GemmDualArgs
lib = C.CDLL(os.path.join(ROOT, "tools/exp", sys.argv[1] if len(sys.argv) > 1 else "libdual_abl.so"))
lib.abl_gemm_dual.argtypes = [C.POINTER(GemmDualArgs), C.c_int, C.c_void_p]
lib.abl_gemm_dual_wgs.argtypes = [C.c_int] * 4
dev = "cuda"
NAMES = {0: "full", 1: "no G store", 2: "no global loads (steady state)", 4: "no dgrad MFMA", 8: "no wgrad MFMA + epilogue", 16: "no LDS staging stores",
         12: "no MFMA at all", 32: "no K-slice exchange (no B1)", 44: "no MFMA, no exchange", 18: "no loads, no staging", 30: "barriers + G store only", 3: "no loads, no G store", 14: "no MFMA, no loads"}
def run(M, Na, Nb, mode):
    A, A2 = torch.randn(M, Na, device=dev), torch.randn(M, Na, device=dev)
    W = torch.randn(Na, Nb, device=dev) * 0.1
    Bm = torch.randn(M, Nb, device=dev)
    vec = [torch.randn(max(Na, Nb), device=dev) for _ in range(8)]
    radd = torch.randn(M, Nb, device=dev)
    runs = lib.abl_gemm_dual_wgs(M, Na, Nb, 0)
    G = torch.empty(M, Nb, device=dev); stats = torch.empty(runs, Nb, 2, device=dev); ws = torch.empty(runs, Na, Nb, device=dev); cs = torch.empty(runs, Na, device=dev)
    a = GemmDualArgs()
    p = lambda t: t.data_ptr()
    a.A = p(A); a.lda = Na; a.A2 = p(A2); a.lda2 = Na; a.p = p(vec[0]); a.q = p(vec[1]); a.r = p(vec[2]); a.W = p(W); a.ldw = Nb; a.B = p(Bm); a.ldb = Nb
    a.b_scale = p(vec[3]); a.b_shift = p(vec[4]); a.b_mean = p(vec[5]); a.b_invstd = p(vec[6]); a.slope = 0.01
    a.G = p(G); a.ldg = Nb; a.stats = p(stats); a.ws = p(ws); a.M, a.Na, a.Nb = M, Na, Nb; a.a_mode = mode; a.a_slope = 0.01
    if mode == 2:
        a.bias = p(vec[7]); a.rowadd = p(radd); a.ld_rowadd = Nb; a.colsum_ws = p(cs)
    s = torch.cuda.current_stream().cuda_stream
    print("M=%d Na=%d Nb=%d a_mode=%d (%s)  ideal MFMA time %.1f us" % (M, Na, Nb, mode, ["dense", "lazy two-tensor", "activation on load + bias + rowadd + colsum"][mode], 4.0 * M * Na * Nb / 157.3e6))
    for abl in (0, 1, 2, 16, 18, 4, 8, 12, 14, 30, 32, 44):
        for _ in range(3):
            lib.abl_gemm_dual(C.byref(a), abl, s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lib.abl_gemm_dual(C.byref(a), abl, s)
        e1.record(); torch.cuda.synchronize()
        print("   abl %2d %-36s %7.1f us" % (abl, NAMES.get(abl, ""), e0.elapsed_time(e1) / 20 * 1e3))
run(65536, 256, 256, 2)
run(65536, 256, 128, 1)
run(65536, 128, 64, 1)
run(196608, 256, 256, 2)
