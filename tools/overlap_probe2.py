#!/usr/bin/env python3
"""Does a matrix-core-bound kernel overlap with an HBM-bound one when the two are issued on forked streams -- eagerly and inside a
captured graph?  Chain A: n x the dominant fused GEMM (D.fc2.0, ~0.3 ms, 88 MB of HBM traffic); chain B: m x a streaming kernel
(torch copy of 256 MB, ~0.1 ms).  (tools/overlap_probe.py asked the same of two MFMA-bound kernels: no gain, as expected.)"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "sp-gan_amd"))
from spgan import ops

M, N, K = 65536, 1024, 256
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
sc, sh = torch.rand(K, device="cuda") + 0.5, torch.randn(K, device="cuda") * 0.1
g, be = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
src = torch.empty(64 * 1024 * 1024, device="cuda"); dst = torch.empty_like(src)
side = torch.cuda.Stream()
nA, nB = 6, 18

def chain_a():
    for _ in range(nA):
        ops.gemm_bn_pool(A, W, b, (g, be, None, None), 2048, 0.01, pro=(sc, sh, 0.01))
def chain_b():
    for _ in range(nB):
        dst.copy_(src)
def serial():
    chain_a(); chain_b()
def forked():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        chain_b()
    chain_a()
    cur.wait_stream(side)

def eager(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
def graphed(fn, reps=5):
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=s):
            fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
print("eager : A alone %.2f ms  B alone %.2f ms  serial %.2f ms  forked %.2f ms" % (eager(chain_a), eager(chain_b), eager(serial), eager(forked)))
print("graph : A alone %.2f ms  B alone %.2f ms  serial %.2f ms  forked %.2f ms" % (graphed(chain_a), graphed(chain_b), graphed(serial), graphed(forked)))
