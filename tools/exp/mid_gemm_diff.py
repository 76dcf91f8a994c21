#!/usr/bin/env python3
"""Every ops.gemm_nt call of one small train step evaluated on both tile routes (tile_hint 0 / 1): report calls whose results differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "sp-gan_amd"), os.path.join(ROOT, "tests"), ROOT]
import torch
import spgan
from spgan import ops, nets, functions, fixture_rng as fr
from oracle import spgan_oracle as orc
from test_parity_gpu import Opts, _load

orig = ops.gemm_nt
seen = []
def both(A, W, bias=None, **kw):
    out_t = kw.get("out")
    rb = kw.get("rowbias")
    alias = out_t is not None and rb is not None and out_t.data_ptr() == rb.data_ptr()
    kw1 = dict(kw)
    if out_t is not None:
        kw1["out"] = out_t.clone()
        if alias:
            kw1["rowbias"] = kw1["out"]
    with ops.nt_tile_hint(1):
        r1 = orig(A, W, bias, **kw1)
    r0 = orig(A, W, bias, **kw)
    a0 = r0[0] if isinstance(r0, tuple) else r0
    a1 = r1[0] if isinstance(r1, tuple) else r1
    if isinstance(a0, torch.Tensor) and a0.dtype == torch.float32:
        d = (a0 - a1).abs().max().item(); sc = a1.abs().max().item()
        if d > 1e-4 * max(sc, 1e-6):
            Ash = tuple(A.shape) if isinstance(A, torch.Tensor) else type(A).__name__
            seen.append((d, sc, Ash, tuple(W.shape), {k: (tuple(v.shape) if isinstance(v, torch.Tensor) else v) for k, v in kw.items()},
                         (A.stride() if isinstance(A, torch.Tensor) else None), W.stride(), alias))
    return r0
ops.gemm_nt = both
B, N = 4, 256
o = Opts()
G = _load(spgan.Generator(o), fr.init_params(orc.generator_shapes(), salt=31))
D = _load(spgan.Discriminator(o), fr.init_params(orc.discriminator_shapes(), salt=31))
tr = spgan.TrainStep(G, D, gan="wgan", use_gp=True, graph=False)
x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
tr.step(x, fr.synthetic_real(B, N, seed=40).cuda(), fr.latent(B, N, seed=50).cuda(), fr.latent(B, N, seed=60).cuda())
torch.cuda.synchronize()
print(len(seen), "differing calls")
for s in seen[:12]:
    print(s)
