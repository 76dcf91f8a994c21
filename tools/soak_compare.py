#!/usr/bin/env python3
"""Full-size (B=32, N=2048, WGAN-GP) training trajectory: losses of the first steps, to compare builds / kernel selections
(e.g. SPGAN_NT_WIDE=0 vs 1) and to soak-test a few hundred graph-replayed steps for non-finite values.
usage: soak_compare.py <steps> <out.json>"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd"), ROOT]
import torch
import bench, spgan
steps = int(sys.argv[1]); out = sys.argv[2]
dev = torch.device("cuda", 0)
spgan.ops.set_mfma_operands(os.environ.get("SPGAN_MFMA", "f32"))        # "bf16x3": fp32-equivalent products in another summation order
G, D = bench.build_models(dev)
tr = spgan.TrainStep(G, D, gan="wgan", use_gp=True, lambda_gp=10.0, graph=True)
from spgan import fixture_rng as fr
x = fr.sphere_template(2048)[None].repeat(32, 1, 1).to(dev)
rec = []
for i in range(steps):
    real = fr.synthetic_real(32, 2048, seed=100 + i).to(dev)
    zd = fr.latent(32, 2048, 128, seed=5000 + 2 * i)[:, :1, :].contiguous().to(dev)
    zg = fr.latent(32, 2048, 128, seed=5001 + 2 * i)[:, :1, :].contiguous().to(dev)
    alpha = fr.uniform("soak.alpha.%d" % i, (32, 1, 1), 0.0, 1.0).to(dev)
    info = tr.step(x, real, zd, zg, alpha=alpha)
    rec.append((float(info["loss_d"]), float(info["loss_g"])))
ok = all(all(abs(v) < 1e6 and v == v for v in r) for r in rec)
w = torch.cat([p.detach().flatten() for p in list(G.parameters()) + list(D.parameters())])
json.dump({"losses": rec, "finite": ok, "param_norm": float(w.norm()), "param_finite": bool(torch.isfinite(w).all())}, open(out, "w"))
print("steps", steps, "finite", ok, "param norm %.6f" % float(w.norm()), "last", rec[-1])
