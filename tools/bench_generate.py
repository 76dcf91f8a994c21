#!/usr/bin/env python3
"""Eval-mode generation throughput (G.eval() + G(x, z), Generation/model_test.py:54-64) on one MI355X:
batch of 32 shapes x 2048 points, latent drawn on the device per batch, forward replayed as a hipGraph.  One JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd"), ROOT]
import torch
import bench
import spgan
from spgan.sampling import InputSampler


class SOpts(bench.Opts):
    nv = 0.2; n_rand = False; n_mix = False


def main():
    dev = torch.device("cuda", 0)
    B = bench.PER_GPU_BATCH
    G, _ = bench.build_models(dev)
    G.eval()
    smp = InputSampler(SOpts, device=dev, seed=1)
    x = smp.sphere_generator(B)
    z = smp.noise_generator(B)
    with torch.no_grad():
        for _ in range(3):
            out = G(x, z)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            out = G(x, z)
        for mode in ("eager", "graph"):
            iters = 50
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(iters):
                z.copy_(smp.noise_generator(B))
                if mode == "eager":
                    out = G(x, z)
                else:
                    g.replay()
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(json.dumps({"metric": "eval-mode generation shapes/sec @2048 pts, bs=32", "mode": mode, "value": round(B * iters / dt, 1),
                              "unit": "shapes/s", "ms_per_batch": round(dt / iters * 1e3, 3), "finite": bool(torch.isfinite(out).all().item())}))


if __name__ == "__main__":
    main()
