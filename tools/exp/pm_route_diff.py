#!/usr/bin/env python3
"""How far apart TrainStep's point-major and channel-major routes end after two steps (tests/test_graph_gpu.py::
test_point_major_route_equals_the_channel_major_one), per state entry and for both kNN routes: max |a-b| against the test's bound."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd"), os.path.join(ROOT, "tests"), ROOT]
import torch
import spgan
from spgan import ops, fixture_rng as fr
from oracle import spgan_oracle as orc
from test_parity_gpu import Opts, _load

B, N = 4, 256
x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
for pipe in (False, True, True):
    ops.KNN_PIPELINED[0] = pipe
    out = []
    for pm in (True, False):
        o = Opts()
        G = _load(spgan.Generator(o), fr.init_params(orc.generator_shapes(), salt=31))
        D = _load(spgan.Discriminator(o), fr.init_params(orc.discriminator_shapes(), salt=31))
        tr = spgan.TrainStep(G, D, gan="wgan", use_gp=True)
        tr.point_major = pm
        torch.manual_seed(1234)
        for s in range(2):
            tr.step(x, fr.synthetic_real(B, N, seed=40 + s).cuda(), fr.latent(B, N, seed=50 + s).cuda(), fr.latent(B, N, seed=60 + s).cuda())
        torch.cuda.synchronize()
        G.flush_bn_counts(); D.flush_bn_counts()
        out.append({k: v.clone() for k, v in list(G.state_dict().items()) + [("D." + k, v) for k, v in D.state_dict().items()]})
    worst = sorted(((((out[0][k].double() - out[1][k].double()).abs().max().item()) / (1e-6 + 1e-5 * out[1][k].double().abs().max().item()), k) for k in out[0]), reverse=True)[:4]
    print("knn pipelined=%s: worst (max|a-b| / bound, entry): %s" % (pipe, ", ".join("%.3f %s" % w for w in worst)))
    chk = sum(float(v.double().sum()) for v in out[0].values())
    print("   state checksum (pm route) %.12e" % chk)
