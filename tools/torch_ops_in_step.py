#!/usr/bin/env python3
"""Which torch (ATen) kernels still run inside one eager train step, and from where: every one of them is a ~5 us launch in the
replayed graph.  Prints op counts and, for the copy/fill/cat class, the Python call sites."""
import os, sys, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd"), ROOT]
import torch
import bench, spgan
dev = torch.device("cuda", 0)
G, D = bench.build_models(dev)
tr = spgan.TrainStep(G, D, gan="wgan", use_gp=True, lambda_gp=10.0, graph=False)
x, real, zs, alpha = bench.make_inputs(dev, 0, bench.PER_GPU_BATCH)
for i in range(2):
    tr.step(x, real, zs[0], zs[1], alpha=alpha)
torch.cuda.synchronize()
sites = collections.Counter()
from torch.utils._python_dispatch import TorchDispatchMode
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("copy_", "clone", "contiguous", "cat", "fill_", "zero_", "zeros", "ones", "full", "repeat", "add", "mul", "neg", "expand_copy", "sum", "mean")):
            st = [f for f in traceback.extract_stack() if "/spgan/" in f.filename or "bench.py" in f.filename]
            where = "%s:%d" % (os.path.basename(st[-1].filename), st[-1].lineno) if st else "?"
            sites[(name, where)] += 1
        return func(*args, **(kwargs or {}))
with Spy():
    tr.step(x, real, zs[2], zs[3], alpha=alpha)
torch.cuda.synchronize()
for (name, where), n in sorted(sites.items(), key=lambda kv: -kv[1]):
    print("%3d  %-40s %s" % (n, name, where))
