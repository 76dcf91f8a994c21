"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


# Every comparison made through rel_l2() / check() is logged here; tests/conftest.py writes the log to
# gpurun_out/parity_errors_{gpu,cpu}.json at session end (the committed profiles/r0N_parity.json is that file from a GPU run:
# the MEASURED parity errors the tolerances in the tests are set against).
PARITY_LOG = []


def _log(name, rel, maxabs, scale, rtol=None, atol=None):
    test = os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0]
    PARITY_LOG.append({"test": test, "tensor": name, "rel_l2": rel, "max_abs": maxabs, "ref_max": scale, "rtol": rtol, "atol": atol})


def rel_l2(a, b, name=""):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    r = float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))
    if a.shape == b.shape and a.size:
        _log(name or "rel_l2#%d" % len(PARITY_LOG), r, float(np.abs(a - b).max()), float(np.abs(b).max()))
    return r


def check(d, name, t, rtol=1e-4, atol=1e-6, what=""):
    """Compare tensor `t` against golden entry `name` (full or summarised form).
    rtol is a relative-L2 bound (and an element-wise bound scaled by the max magnitude)."""
    a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    if name + "|full" in d:
        ref = d[name + "|full"]
        assert a.shape == ref.shape, (name, a.shape, ref.shape)
        err = rel_l2(a, ref, name)
        scale = max(float(np.abs(ref).max()), 1e-30)
        maxerr = float(np.abs(a - ref).max())
        PARITY_LOG[-1].update(rtol=rtol, atol=atol)
        assert err <= rtol or maxerr <= atol, "%s %s: rel-L2 %.3e (max-abs %.3e, scale %.3e)" % (what, name, err, maxerr, scale)
        return err
    stride = int(d[name + "|stride"])
    ref = d[name + "|samples"]
    got = a.reshape(-1)[::stride][:ref.size]
    err = rel_l2(got, ref, name + " (sampled)")
    maxerr = float(np.abs(got - ref).max())
    PARITY_LOG[-1].update(rtol=rtol, atol=atol)
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


def install_kernel_models():
    """Replace every op of spgan.ops by its plain-PyTorch model (tests/kernel_model.py) and switch the GPU guard off, process-wide:
    the CPU doubles behind the host-composition tests, for code that runs in a spawned process (gloo workers, `bench.py`'s
    self-test mode) where pytest's monkeypatch fixture is not available.  TEST INFRASTRUCTURE: the product never calls this."""
    import inspect
    import kernel_model as km
    import spgan.modules as modules
    import spgan.ops as ops
    for name, fn in inspect.getmembers(km, inspect.isfunction):
        if not name.startswith("_"):
            setattr(ops, name, fn)
    ops.SparseAffine = km.SparseAffine
    modules._require_gpu = lambda t, what: None
