"""EXPERIMENT: what would batching the step's two generator forwards (D step: G(z_d) without grad; G step: G(z_g)) buy?  Upper bound: one no-grad forward
of 2B shapes against two of B shapes, each replayed as a hipGraph (same weights; BatchNorm over 2B instead of per pass -- timing only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd"), ROOT]
import torch
import bench, spgan
dev = torch.device("cuda", 0)
G, D = bench.build_models(dev)
G.train()
B, N = 32, 2048
def inputs(b):
    x, real, zs, alpha = bench.make_inputs(dev, 0, b)
    return x, zs[0]
def graph_of(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g
def t(g, reps=50):
    for _ in range(5): g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
x1, z1 = inputs(B)
x2, z2 = inputs(2 * B)
def one():
    with torch.no_grad():
        G(x1, z1, pm_out=True); G(x1, z1, pm_out=True)
def two():
    with torch.no_grad():
        G(x2, z2, pm_out=True)
g1, g2 = graph_of(one), graph_of(two)
for _ in range(2):
    print("two forwards of B=32: %.3f ms   one forward of B=64: %.3f ms" % (t(g1), t(g2)))
