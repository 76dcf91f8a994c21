#!/usr/bin/env python3
"""Checksums of every spgan.ops call's tensor results during the golden train step (test_train_step_golden, wgangp) -> a text file;
diff two of them (e.g. SPGAN_NT_MID=0 against the default) to find the first op whose result changes.  usage: op_trace.py out.txt"""
import os, sys, inspect
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd"), os.path.join(ROOT, "tests"), ROOT]
import torch
import spgan
from spgan import ops, fixture_rng as fr
from oracle import spgan_oracle as orc
from test_parity_gpu import Opts, _load

out = open(sys.argv[1], "w")
def cs(t):
    if isinstance(t, torch.Tensor) and t.is_floating_point() and t.numel():
        return "%s:%.9e" % (tuple(t.shape), t.double().abs().sum().item())
    if isinstance(t, torch.Tensor):
        return "%s:i%d" % (tuple(t.shape), int(t.long().sum().item()))
    if isinstance(t, (tuple, list)):
        return "[" + ",".join(cs(u) for u in t) + "]"
    g = getattr(t, "g", None)
    return cs(g) if g is not None else type(t).__name__
def wrap(name, fn):
    def w(*a, **k):
        r = fn(*a, **k)
        out.write("%s %s <- %s\n" % (name, cs(r), ",".join(cs(x) for x in a if isinstance(x, (torch.Tensor, tuple, list)) or hasattr(x, "g"))))
        return r
    return w
for name, fn in inspect.getmembers(ops, inspect.isfunction):
    if not name.startswith("_") and fn.__module__ == ops.__name__ and name not in ("check", "capturing", "launch_timer", "nt_tile_hint", "storage16", "bump_weights_epoch", "weights_epoch_of", "gemm_dual_ok"):
        setattr(ops, name, wrap(name, fn))
B, N = 4, 256
o = Opts()
G = _load(spgan.Generator(o), fr.init_params(orc.generator_shapes(), salt=8))
D = _load(spgan.Discriminator(o), fr.init_params(orc.discriminator_shapes(), salt=8))
tr = spgan.TrainStep(G, D, gan="wgan", use_gp=True, lambda_gp=10.0, lr_g=1e-4, lr_d=1e-4)
x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
real = fr.synthetic_real(B, N, seed=81).cuda()
z_d, z_g = fr.latent(B, N, seed=82).cuda(), fr.latent(B, N, seed=83).cuda()
alpha = fr.uniform("graph.alpha", (B, 1, 1), 0.0, 1.0).cuda()
info = tr.step(x, real, z_d, z_g, alpha=alpha, keep_grads=True)
torch.cuda.synchronize()
out.close()
