#!/usr/bin/env python3
"""All-pairs Chamfer matrix throughput: S x R clouds of 2048 points.  One JSON line."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd"), ROOT]
import torch
from spgan import metrics
from bench_kernels import timeit
S = R = 256; N = 2048
A = torch.randn(S, N, 3, device="cuda"); B = torch.randn(R, N, 3, device="cuda")
t = timeit(lambda: metrics.pairwise_cd(A, B), n=3, warm=1)
evals = 2.0 * S * R * N * N
print(json.dumps({"metric": "all-pairs Chamfer matrix", "S": S, "R": R, "points": N, "ms": round(t, 2), "pairs_per_s": round(S * R / t * 1e3, 1),
                  "distance_evals_per_s": round(evals / t * 1e3, 0), "valu_tflops_at_8_flop_per_eval": round(evals * 8 / t * 1e3 / 1e12, 2)}))
a = torch.randn(32, N, 3, device="cuda"); b = torch.randn(32, N, 3, device="cuda")
t2 = timeit(lambda: metrics.nn_distance(a, b), n=10)
print(json.dumps({"metric": "ChamferDistance forward, 32 x 2048 x 2048", "ms": round(t2, 3)}))
u = torch.rand(64, N, 3, device="cuda"); v = torch.rand(64, N, 3, device="cuda") * 0.9
for iters in (50, 300):
    t3 = timeit(lambda: metrics.emdModule()(u, v, 0.005, iters), n=3, warm=1)
    d, asg = metrics.emdModule()(u, v, 0.005, iters)
    uniq = sum(int(asg[i].unique().numel()) for i in range(64)) / (64.0 * N)
    print(json.dumps({"metric": "auction EMD forward, 64 x 2048 points", "iters": iters, "ms": round(t3, 2), "pairs_per_s": round(64 / t3 * 1e3, 1),
                      "distinct_targets_fraction": round(uniq, 4), "mean_emd": round(d.sqrt().mean().item(), 5)}))
pcs = (torch.rand(256, N, 3, device="cuda") - 0.5) * 0.55
t4 = timeit(lambda: metrics.jsd_between_point_cloud_sets(pcs, pcs * 0.9), n=3, warm=1)
print(json.dumps({"metric": "JSD between 2 x 256 clouds of 2048 points, 28^3 grid", "ms": round(t4, 2)}))
