"""Input sampling of the reference's train/test loops, generated on the device (Generation/model.py:46-52,122-180):
the per-shape (or per-point / region-mixed) latent noise, the sphere prior and the `.xyz` sample dump.

The reference draws from numpy's global, never-seeded RNG on the host and uploads 33 MB of tiled noise per call; here the
draws come from a seedable torch.Generator on the GPU (reproducible, no PCIe traffic).  Distributions and shapes are the
reference's; the individual random numbers necessarily differ (the reference's are not reproducible either, model.py:40).
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch

from . import fixture_rng


def pc_normalize(pc: torch.Tensor) -> torch.Tensor:
    """model.py:46-52: centre on the centroid, scale the farthest point to radius 1.  pc [N,3]; computed in the input's dtype
    (the reference's numpy version runs in float64: pass a float64 tensor for its exact arithmetic)."""
    pc = pc - pc.mean(dim=0, keepdim=True)
    return pc / pc.pow(2).sum(dim=1).sqrt().max()


class InputSampler:
    """`noise_generator` / `sphere_generator` of Generation/model.py:122-180 as device-side samplers.
    opts fields read: np (points), nz (latent size), nv (noise std), n_rand, n_mix."""

    def __init__(self, opts, device="cuda", seed: Optional[int] = None):
        self.opts = opts
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device)
        if seed is not None:
            self.gen.manual_seed(seed)
        self.ball: Optional[torch.Tensor] = None          # [np,3] normalised template
        self.ball_order: Optional[torch.Tensor] = None    # lazily: argsort of the reference's ball_dist rows

    # ------------------------------------------------------------------ sphere prior
    def _load_ball(self):
        if self.ball is None:
            self.ball = fixture_rng.sphere_template(self.opts.np).to(self.device)      # template/balls/<np>.xyz, pc_normalize'd

    def sphere_generator(self, bs: int = 2, static: bool = True) -> torch.Tensor:
        """model.py:156-180.  static: the template tiled to [bs,np,3]; otherwise np points drawn with replacement per shape."""
        self._load_ball()
        if static:
            return self.ball[None].repeat(bs, 1, 1)
        idx = torch.randint(0, self.ball.shape[0], (bs, self.opts.np), generator=self.gen, device=self.device)
        return self.ball[idx]

    def _region_order(self, centre: torch.Tensor) -> torch.Tensor:
        """Rows of argsort(ball_dist) for the given centre points.  ball_dist is the reference's expression
        -2*|x|^2|y|^2 + |x|^2 + |y|^2 (model.py:164-167: it multiplies the squared norms instead of taking the inner
        product) -- kept, since it defines which points a 'region' contains.  Evaluated the way the reference does -- float64
        numpy on the host, np.argsort's default sort -- so that the ordering (ties included) is the reference's own; the rows are
        computed once per centre and cached on the device (pinned by tests/golden/g15_samplers.npz)."""
        self._load_ball()
        n = self.opts.np
        if self.ball_order is None:
            ball = fixture_rng.sphere_template64(n)
            xx = np.sum(ball ** 2, axis=1).reshape(n, 1)
            self._ball_dist = -2 * xx @ xx.T + xx + xx.T                              # model.py:164-167
            self.ball_order = torch.full((n, n), -1, dtype=torch.int64, device=self.device)
            self._order_known = np.zeros(n, dtype=bool)
        ids = centre.detach().cpu().numpy().astype(np.int64).reshape(-1)
        for i in np.unique(ids[~self._order_known[ids]]):
            self.ball_order[i] = torch.from_numpy(np.argsort(self._ball_dist[i])[::1].copy()).to(self.device)
            self._order_known[i] = True
        return self.ball_order[centre.to(self.device).long()]

    def region_mask(self, centre: torch.Tensor, num: torch.Tensor) -> torch.Tensor:
        """[bs, np] bool: the `num[b]` points closest to `centre[b]` in ball_dist order (model.py:138-143: idx[:num])."""
        order = self._region_order(centre)
        n = self.opts.np
        inside = torch.arange(n, device=self.device)[None, :] < num.to(self.device)[:, None]
        return torch.zeros((order.shape[0], n), dtype=torch.bool, device=self.device).scatter_(1, order, inside)

    # ------------------------------------------------------------------ latent noise
    def noise_generator(self, bs: int = 1, masks: Optional[torch.Tensor] = None, compact: bool = False) -> torch.Tensor:
        """model.py:122-154 -> [bs, np, nz]; compact=True (default mode only: one latent per shape, no n_rand / n_mix / masks)
        returns the un-tiled [bs, 1, nz], which spgan.Generator accepts as is and evaluates per shape instead of per point.
        default: one N(0, nv^2) vector per shape, tiled over the points; n_rand: independent per point; n_mix: with
        probability 1/2 a random region (the `num` points closest to a random centre in ball_dist order, num =
        max(U,0.1)*np) of every shape gets a second vector.  masks [bs,np] (part labels): one N(0, 0.2^2) vector per part
        (the reference's masks branch assigns the index array instead of the drawn vector, model.py:151 -- the evident
        intent is implemented)."""
        o, dev, g = self.opts, self.device, self.gen
        if masks is not None:
            masks = torch.as_tensor(masks, device=dev)
            noise = torch.zeros((masks.shape[0], o.np, o.nz), device=dev)
            for i in range(masks.shape[0]):
                for j in torch.unique(masks[i]).tolist():
                    noise[i, masks[i] == j] = torch.randn((o.nz,), generator=g, device=dev) * 0.2
            return noise
        if o.n_rand:
            noise = torch.randn((bs, o.np, o.nz), generator=g, device=dev) * o.nv
        else:
            noise = torch.randn((bs, 1, o.nz), generator=g, device=dev) * o.nv
            if compact and not getattr(o, "n_mix", False):
                return noise
            noise = noise.repeat(1, o.np, 1)
        if getattr(o, "n_mix", False) and torch.rand((), generator=g, device=dev).item() < 0.5:
            noise2 = torch.randn((bs, o.nz), generator=g, device=dev) * o.nv
            centre = torch.randint(0, o.np, (bs,), generator=g, device=dev)
            num = (torch.rand((bs,), generator=g, device=dev).clamp_min(0.1) * o.np).long()
            sel = self.region_mask(centre, num)                                       # first `num` entries of each order row
            noise = torch.where(sel[:, :, None], noise2[:, None, :], noise)
        return noise


def save_xyz(path: str, points: torch.Tensor) -> None:
    """One shape per file, `x y z` per line with 6 decimals (the np.savetxt(fmt='%.6f') dump of model.py:371-410).
    points [N,3] or [3,N]."""
    p = points.detach().float().cpu()
    if p.dim() != 2 or 3 not in p.shape:
        raise ValueError("save_xyz expects [N,3] or [3,N], got %s" % (tuple(p.shape),))
    if p.shape[1] != 3:
        p = p.t()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.savetxt(path, p.numpy(), fmt="%.6f")
