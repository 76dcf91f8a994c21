"""EXPERIMENT: what W's pre-split image would buy the split-bf16 step: a naive provider (image per (address, shape) and weights epoch, one split
launch per miss -- every weight re-split after each optimiser step) installed around the replayed TrainStep; ms/step with and without."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd"), ROOT]
import torch
import bench, spgan
from spgan import ops, nets
dev = torch.device("cuda", 0)
ops.set_mfma_operands("bf16x3")
x, real, zs, alpha = bench.make_inputs(dev, 0, 32)
def run(provider, tag):
    ops.w_image_provider = provider
    G, D = bench.build_models(dev)
    tr = spgan.TrainStep(G, D, gan="wgan", use_gp=True, lambda_gp=10.0, graph=True, graph_warmup=3)
    step = lambda i: tr.step(x, real, zs[(2 * i) % 4], zs[(2 * i + 1) % 4], alpha=alpha)
    for i in range(8): step(i)
    dt, _ = bench.time_steps(tr, step, 30, False, dev)
    print("%-28s %.3f ms/step" % (tag, dt / 30 * 1e3), flush=True)
    ops.w_image_provider = None
cache = {}
misses = [0]
def provider(W):
    if nets._owner(W) is None and not getattr(W, "_is_t", False):
        pass          # computed operands too: keyed by epoch only, so a recycled address within one epoch would be WRONG -- timing probe only
    key = (W.data_ptr(), tuple(W.shape), tuple(W.stride()), ops.WEIGHTS_EPOCH[0])
    img = cache.get(key)
    if img is None:
        misses[0] += 1
        img = cache[key] = ops.split_image(W)
        if len(cache) > 400:
            for k in list(cache)[:200]: del cache[k]
    return img
run(None, "no image")
run(provider, "image per weight and epoch")
print("misses:", misses[0])
run(None, "no image (again)")
