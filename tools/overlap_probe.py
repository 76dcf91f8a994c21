#!/usr/bin/env python3
"""Do an input-gradient GEMM (gemm_nt) and the weight-gradient GEMM of the same layer (gemm_tn + split-K reduce) overlap when they
are issued on two HIP streams?  Times n repetitions of the pair serial on one stream vs forked on two (inside hipGraphs)."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "sp-gan_amd"))
from spgan import ops

def bench(fn, reps=5):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best

M = 65536
for (Cout, Cin) in ((256, 256), (256, 128), (128, 64), (1024, 256)):
    dy = torch.randn(M, Cout, device="cuda"); x = torch.randn(M, Cin, device="cuda"); Wt = torch.randn(Cin, Cout, device="cuda") * 0.05
    n = 20
    side = torch.cuda.Stream()
    def serial():
        for _ in range(n):
            ops.gemm_nt(dy, Wt); ops.gemm_tn(dy, x)
    def forked():
        cur = torch.cuda.current_stream()
        for _ in range(n):
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                ops.gemm_tn(dy, x)
            ops.gemm_nt(dy, Wt)
        cur.wait_stream(side)
    def only_nt():
        for _ in range(n):
            ops.gemm_nt(dy, Wt)
    def only_tn():
        for _ in range(n):
            ops.gemm_tn(dy, x)
    def eager(fn, reps=5):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best
    print("   eager (no graph): serial %.1f us  forked %.1f us" % (eager(serial) / n * 1e3, eager(forked) / n * 1e3))
    a, b, c, d = bench(serial), bench(forked), bench(only_nt), bench(only_tn)
    print("Cout %4d Cin %4d: nt %.1f us  tn %.1f us  serial %.1f us  forked %.1f us  (saves %.0f%%)" % (Cout, Cin, c / n * 1e3, d / n * 1e3, a / n * 1e3, b / n * 1e3, 100 * (1 - b / a)))
