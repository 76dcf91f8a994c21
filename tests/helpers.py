"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))


def check(d, name, t, rtol=1e-4, atol=1e-6, what=""):
    """Compare tensor `t` against golden entry `name` (full or summarised form).
    rtol is a relative-L2 bound (and an element-wise bound scaled by the max magnitude)."""
    a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    if name + "|full" in d:
        ref = d[name + "|full"]
        assert a.shape == ref.shape, (name, a.shape, ref.shape)
        err = rel_l2(a, ref)
        scale = max(float(np.abs(ref).max()), 1e-30)
        maxerr = float(np.abs(a - ref).max())
        assert err <= rtol or maxerr <= atol, "%s %s: rel-L2 %.3e (max-abs %.3e, scale %.3e)" % (what, name, err, maxerr, scale)
        return err
    stride = int(d[name + "|stride"])
    ref = d[name + "|samples"]
    got = a.reshape(-1)[::stride][:ref.size]
    err = rel_l2(got, ref)
    maxerr = float(np.abs(got - ref).max())
    assert err <= rtol or maxerr <= atol, "%s %s: sampled rel-L2 %.3e (max-abs %.3e)" % (what, name, err, maxerr)
    l2 = float(np.sqrt((a.astype(np.float64) ** 2).sum()))
    ref_l2 = float(d[name + "|l2"])
    assert abs(l2 - ref_l2) <= max(rtol * ref_l2, atol), "%s %s: L2 %.6e vs %.6e" % (what, name, l2, ref_l2)
    return err


def params_from(shapes, salt, requires_grad=False):
    from spgan import fixture_rng as fr
    p = fr.init_params(shapes, salt=salt)
    if requires_grad:
        for v in p.values():
            v.requires_grad_(True)
    return p
