#!/usr/bin/env python3
"""Does a captured hipGraph run independent branches concurrently?  Two independent chains of small kernels (each ~5 us, few workgroups),
(a) captured on one stream, (b) forked onto two streams inside the capture; replay time per graph.  -> gpurun_out/graph_branches.txt"""
import sys, os
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sp-gan_amd")]
from spgan import ops  # noqa: E402

dev = torch.device("cuda", 0)
CH = 40
xs = [torch.randn(64, 256, device=dev) for _ in range(2)]
Ws = [torch.randn(256, 256, device=dev) * 0.05 for _ in range(2)]


def chain(i):
    h = xs[i]
    for _ in range(CH):
        h = ops.gemm_nt(h, Ws[i])           # M = 64: the small-M kernel, ~6 us, latency-bound
    return h


def bench(g, n=50):
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


chain(0); chain(1); torch.cuda.synchronize()
side = torch.cuda.Stream()
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1, stream=side):
    a = chain(0); b = chain(1)
s2 = torch.cuda.Stream()
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2, stream=side):
    s2.wait_stream(side)
    a = chain(0)
    with torch.cuda.stream(s2):
        b = chain(1)
    side.wait_stream(s2)
print("two chains of %d small launches: one stream %.1f us/replay, forked on two streams %.1f us/replay" % (CH, bench(g1), bench(g2)))
