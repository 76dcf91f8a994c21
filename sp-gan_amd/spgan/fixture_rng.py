"""Counter-based, name-keyed deterministic tensors.

Weights and inputs for fixtures/benchmarks must be reproducible on any machine
without relying on torch's RNG stream (the GPU box has no reference and no
shared RNG state with the build container).  Every tensor is drawn from a numpy
Philox generator keyed by a hash of its *name*, so the same name always gives
the same values regardless of draw order.
"""
from __future__ import annotations

import hashlib
import math
import os
from typing import Dict, Tuple

import numpy as np
import torch


def _gen(name: str, salt: int = 0) -> np.random.Generator:
    h = hashlib.blake2b(("%s|%d" % (name, salt)).encode(), digest_size=16).digest()
    key = np.frombuffer(h, dtype=np.uint64)
    return np.random.Generator(np.random.Philox(key=key))


def uniform(name: str, shape, lo: float = -1.0, hi: float = 1.0, salt: int = 0) -> torch.Tensor:
    a = _gen(name, salt).uniform(lo, hi, size=tuple(shape)).astype(np.float32)
    return torch.from_numpy(a)


def normal(name: str, shape, std: float = 1.0, mean: float = 0.0, salt: int = 0) -> torch.Tensor:
    a = (_gen(name, salt).standard_normal(size=tuple(shape)) * std + mean).astype(np.float32)
    return torch.from_numpy(a)


def init_params(shapes: Dict[str, Tuple[int, ...]], salt: int = 0, perturb_bn: bool = True) -> Dict[str, torch.Tensor]:
    """Reference-like initial values for a state_dict described by `shapes`:
    conv/linear weight and bias ~ U(+-1/sqrt(fan_in)) (torch default, kaiming a=sqrt(5));
    AdaIN style conv: weight ~ N(0,1), bias = [1]*C ++ [0]*C (Generator.py:32-36);
    BN gamma/beta: 1/0, or (perturb_bn) U(0.5,1.5)/U(-0.2,0.2) so tests exercise them."""
    out = {}
    for name, shp in shapes.items():
        base, leaf = name.rsplit(".", 1)
        wshape = shapes.get(base + ".weight", shp)
        if leaf == "gamma":                                      # Attention's gate (0 at init in the reference: nothing to test)
            t = torch.full(shp, 0.7)
        elif leaf == "weight_orig":                              # equalised-LR layers: N(0,1), scaled at run time
            t = normal(name, shp, 1.0, 0.0, salt)
        elif len(wshape) == 1:                                   # batch-norm affine (and the biases of equalised-LR layers)
            if leaf == "weight":
                t = uniform(name, shp, 0.5, 1.5, salt) if perturb_bn else torch.ones(shp)
            else:
                t = uniform(name, shp, -0.2, 0.2, salt) if perturb_bn else torch.zeros(shp)
        elif ".style" in name:
            if leaf == "weight":
                t = normal(name, shp, 1.0, 0.0, salt)
            else:
                c = shp[0] // 2
                t = torch.cat([torch.ones(c), torch.zeros(c)])
        else:
            fan_in = int(np.prod(wshape[1:]))
            bound = 1.0 / math.sqrt(fan_in)
            t = uniform(name, shp, -bound, bound, salt)
        out[name] = t.contiguous()
    return out


def mid_training_state(shapes: Dict[str, Tuple[int, ...]], buffer_names, salt: int = 0, scale: float = 1e-2, step: int = 7,
                       batches: int = 21) -> Dict[str, object]:
    """A reproducible state "in the middle of training" for the state-carry fixtures (golden G20): Adam moments of the magnitude the
    networks' gradients have (exp_avg ~ N(0, scale), exp_avg_sq = (U(0.5, 1.5)*scale)^2), an optimiser step count, and non-trivial
    BatchNorm running statistics (mean ~ N(0, 0.1), var ~ U(0.5, 1.5), `batches` calls tracked) -- loaded into the reference AND into the
    build before ONE step, so that Adam at step > 1 and a running-statistics update from non-initial buffers are compared at one-step
    tolerances instead of across a chaotic multi-step trajectory.  -> {"m": {name: t}, "v": {name: t}, "step": int, "buffers": {name: t}}."""
    m = {n: normal("mid.m." + n, shp, scale, 0.0, salt) for n, shp in shapes.items()}
    v = {n: (uniform("mid.v." + n, shp, 0.5, 1.5, salt) * scale) ** 2 for n, shp in shapes.items()}
    bufs = {}
    for n in buffer_names:
        base = n.rsplit(".", 1)[0]
        c = shapes[base + ".weight"]
        if n.endswith("running_mean"):
            bufs[n] = normal("mid.rm." + n, c, 0.1, 0.0, salt)
        elif n.endswith("running_var"):
            bufs[n] = uniform("mid.rv." + n, c, 0.5, 1.5, salt)
        elif n.endswith("num_batches_tracked"):
            bufs[n] = torch.tensor(batches, dtype=torch.long)
    return {"m": m, "v": v, "step": step, "buffers": bufs}


_BALLS = None


def sphere_template64(n: int) -> np.ndarray:
    """The normalised template in float64, as the reference holds `self.ball` (Generation/model.py:46-52,159-160)."""
    global _BALLS
    if _BALLS is None:
        _BALLS = np.load(os.path.join(os.path.dirname(__file__), "data", "balls.npz"))
    key = "ball_%d" % n
    if key not in _BALLS:
        raise ValueError("no sphere template with %d points (have %s)" % (n, sorted(_BALLS.keys())))
    pc = _BALLS[key].astype(np.float64)
    pc = pc - np.mean(pc, axis=0)
    pc = pc / np.max(np.sqrt(np.sum(pc ** 2, axis=1)))
    return pc


def sphere_template(n: int) -> torch.Tensor:
    """Unit-sphere template with N points, centred and scaled to unit max radius the way
    the reference does (Generation/model.py:46-52,159-160): float64 maths, then fp32."""
    return torch.Tensor(sphere_template64(n))                 # float64 -> float32 like torch.Tensor(ndarray)


def synthetic_real(b: int, n: int, seed: int = 1234) -> torch.Tensor:
    """Synthetic 'real' clouds [b,n,3]: N(0,I) projected to the unit sphere,
    anisotropic per-cloud scale U[0.3,1], then centred and scaled to unit max radius
    (Common/point_operation.py:21-40 semantics)."""
    g = _gen("synthetic_real", seed)
    p = g.standard_normal(size=(b, n, 3))
    p /= np.linalg.norm(p, axis=-1, keepdims=True)
    p *= g.uniform(0.3, 1.0, size=(b, 1, 3))
    p -= p.mean(axis=1, keepdims=True)
    p /= np.max(np.linalg.norm(p, axis=-1), axis=1)[:, None, None]
    return torch.from_numpy(p.astype(np.float32))


def latent(b: int, n: int, nz: int = 128, nv: float = 0.2, seed: int = 4321) -> torch.Tensor:
    """z ~ N(0, nv^2) [b,1,nz] tiled over the N points (Generation/model.py:128-131)."""
    z = (_gen("latent", seed).standard_normal(size=(b, 1, nz)) * nv).astype(np.float32)
    return torch.from_numpy(np.tile(z, (1, n, 1)))
