#!/usr/bin/env python3
"""Where does a gemm_nt workgroup spend its life?  s_memtime stamps at kernel entry / after the first loads were issued / first tile
in LDS / end of the k-loop / end of the epilogue, for every workgroup of a launch (csrc/gemm.hip, -DSPGAN_TRACE; the trace buffer is
passed in the otherwise unused e_bias2 field of a plain LINEAR launch).  Build the instrumented library next to this file first:

  cd sp-gan_amd && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../include -Icsrc -DSPGAN_TRACE -c csrc/gemm.hip -o /tmp/gemm_trace.o \
    && hipcc --offload-arch=gfx950 -shared -fPIC /tmp/gemm_trace.o $(ls csrc/*.o | grep -v gemm.o) -o ../tools/_trace/libspgan_hip.so

This is how the per-element epilogues were found to take 25-33 % of a workgroup's lifetime (profiles/r01_wg_timeline.txt)."""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sp-gan_amd"))
from spgan import _lib
from spgan._lib import GemmNTArgs
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_trace", "libspgan_hip.so"))
lib.spgan_gemm_nt.restype = C.c_int; lib.spgan_gemm_nt.argtypes = [C.POINTER(GemmNTArgs), C.c_void_p]
def run(M, N, K, fc2=False):
    """fc2=True: the D.fc2.0 configuration (affine + LeakyReLU prologue, column statistics and pooling partials, output not stored)"""
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; Y = torch.empty(M, N, device="cuda")
    nwg = ((M + 127) // 128 + 7) // 8 * 8 * ((N + 63) // 64) + 64
    trc = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
    a = GemmNTArgs()
    a.A = A.data_ptr(); a.lda = K; a.W = W.data_ptr(); a.ldw = K; a.Y = Y.data_ptr(); a.ldy = N; a.M, a.N, a.K = M, N, K
    a.e_bias2 = trc.data_ptr()
    if fc2:
        tiles = (M + 127) // 128
        sc, sh = torch.rand(K, device="cuda") + 0.5, torch.randn(K, device="cuda") * 0.1
        st = torch.empty(tiles * N * 2, device="cuda"); pv = torch.empty(tiles * N * 2, device="cuda"); pa = torch.empty(tiles * N * 2, dtype=torch.int32, device="cuda")
        a.a_mode = 1; a.p_scale = sc.data_ptr(); a.p_shift = sh.data_ptr(); a.p_slope = 0.01
        a.stats = st.data_ptr(); a.pool_val = pv.data_ptr(); a.pool_arg = pa.data_ptr(); a.Y = None
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        assert lib.spgan_gemm_nt(C.byref(a), s) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); lib.spgan_gemm_nt(C.byref(a), s); e1.record(); torch.cuda.synchronize()
    t = trc.cpu().numpy().reshape(-1, 8)
    t = t[t[:, 0] > 0]
    d = np.diff(t[:, :5], axis=1).astype(np.float64)
    start = t[:, 0] - t[:, 0].min(); end = t[:, 4] - t[:, 0].min()
    print(("D.fc2.0 config " if fc2 else "") + "M %d N %d K %d: %.1f us (%.1f TF)  WGs %d  kernel span %.0f clk" % (M, N, K, e0.elapsed_time(e1) * 1e3, 2.0 * M * N * K / e0.elapsed_time(e1) / 1e9, len(t), end.max()))
    for name, col in (("gload issue", 0), ("first tile -> LDS", 1), ("main loop", 2), ("epilogue", 3)):
        print("   %-18s median %8.0f  p10 %8.0f  p90 %8.0f clk" % (name, np.median(d[:, col]), np.percentile(d[:, col], 10), np.percentile(d[:, col], 90)))
    life = (t[:, 4] - t[:, 0]).astype(np.float64)
    print("   WG lifetime median %.0f clk; start-time quartiles %s" % (np.median(life), np.percentile(start, [25, 50, 75, 100]).round(0)))
for shp in ((65536, 256, 128), (65536, 256, 256), (65536, 1024, 256), (65536, 128, 128)):
    run(*shp)
run(65536, 1024, 256, fc2=True)
