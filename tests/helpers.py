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


def _entry(d, name, a):
    """(reference values, the matching elements of `a`) for a full or a sampled golden entry."""
    if name + "|full" in d:
        ref = d[name + "|full"]
        assert a.shape == ref.shape, (name, a.shape, ref.shape)
        return ref.reshape(-1), a.reshape(-1)
    stride = int(d[name + "|stride"])
    ref = d[name + "|samples"]
    return ref, a.reshape(-1)[::stride][:ref.size]


def check_bounded_by_reference_noise(d, name32, name64, t, floor, factor=1.5, atol=0.0, what=""):
    """For quantities the reference itself only reproduces to its own float32 rounding amplified by an ill-conditioned or
    discrete step (a near-tied arg-max below D's pool, BatchNorm1d over a batch of nearly equal global features): the golden file
    holds the reference's float32 AND float64 result.  The build's error against the float64 result must not exceed
    max(factor x the reference's own float32 error against it, floor) -- i.e. the build is at least as close to the exact value as
    the reference's float32 path, up to `factor`."""
    a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    r32, got = _entry(d, name32, a)
    r64, _ = _entry(d, name64, a)
    den = max(float(np.sqrt((r64.astype(np.float64) ** 2).sum())), 1e-30)
    e_ref = float(np.sqrt(((r32.astype(np.float64) - r64) ** 2).sum())) / den
    e_own = float(np.sqrt(((got.astype(np.float64) - r64) ** 2).sum())) / den
    maxerr = float(np.abs(got - r64).max()) if got.size else 0.0
    _log(name64 + " (vs float64; reference's own float32 error %.2e)" % e_ref, e_own, maxerr, float(np.abs(r64).max()) if r64.size else 0.0,
         rtol=max(factor * e_ref, floor), atol=atol)
    assert e_own <= max(factor * e_ref, floor) or maxerr <= atol, \
        "%s %s: rel-L2 vs float64 reference %.3e > max(%.1f x %.3e [reference float32 vs float64], %.1e) (max-abs %.3e)" % (what, name64, e_own, factor, e_ref, floor, maxerr)
    return e_own, e_ref


class StepNoise:
    """Golden G18 (tests/golden/make_golden.py::g18): the REFERENCE's own noise on the benchmarked C2 WGAN-GP step, from which the bounds of
    the step tests are derived instead of asserted:
      * `noise32|*`: its float32 run against its float64 run on the same two EdgeConv2 graphs (per D / G gradient tensor, whole-G cosine);
      * `tie|*`:     its float32 run with the n most nearly tied kNN rows of both graphs resolved the other way, n = 1 .. 100 (per tensor
                     movement, whole-G cosine and norm ratio, clouds).
    A build whose own graphs differ from the reference's in n_diff near-tie rows may move by `factor` x what the reference itself moves
    when it lands on the other side of >= n_diff ties."""

    def __init__(self):
        self.d = golden("g18_step_noise_c2.npz")
        self.ns = [int(n) for n in self.d["tie|nflip"]]

    def _pick(self, arr, n_diff):
        for n, v in zip(self.ns, arr):
            if n >= n_diff:
                return float(v)
        return float(arr[-1]) * n_diff / self.ns[-1]

    def tensor_bound(self, kind, name, n_diff, factor):
        """rel-L2 bound for gradient tensor `name` (kind 'dgrad' / 'ggrad') with n_diff differing near-tie rows (0: same graphs)."""
        noise = max(float(self.d["noise32|%s|%s" % (kind, name)]), 4e-7)       # floor: a few float32 ulps (tensors the reference reproduces exactly)
        if n_diff <= 0:
            return factor * noise
        return factor * max(self._pick(self.d["tie|%s|%s" % (kind, name)], n_diff), noise)

    def whole_g_bound(self, n_diff, factor):
        """rel-L2 movement of the whole G gradient: sqrt(1 + r^2 - 2 r cos) of the reference's own (cosine, norm ratio)."""
        def mov(c, r):
            return float(np.sqrt(max(1.0 + r * r - 2.0 * r * c, 0.0)))
        noise = mov(float(self.d["noise32|ggrad_cos"]), float(self.d["noise32|ggrad_ratio"]))
        if n_diff <= 0:
            return factor * noise
        tab = [mov(c, r) for c, r in zip(self.d["tie|ggrad_cos"], self.d["tie|ggrad_ratio"])]
        return factor * max(self._pick(tab, n_diff), noise)


def check_step_gradients_bounded(step_d, noise, kind, grads, n_diff, factor, skip=(), atol=None):
    """Every gradient tensor of the step against golden `step_d` (g17_step_c2: '<kind>|<name>'), bound = StepNoise.tensor_bound."""
    worst = 0.0
    for n, g in grads.items():
        a = g.detach().cpu().numpy()
        ref, got = _entry(step_d, "%s|%s" % (kind, n), a)
        if n.endswith(tuple(skip)):
            # zero-gradient biases: the reference holds rounding noise, the build exact zeros -- absolute bound only
            assert float(np.abs(got - ref).max()) <= (atol or 2e-3), n
            continue
        bound = noise.tensor_bound(kind, n, n_diff, factor)
        err = float(np.sqrt(((got.astype(np.float64) - ref) ** 2).sum()) / max(np.sqrt((ref.astype(np.float64) ** 2).sum()), 1e-30))
        _log("%s|%s (bound = %.1f x the reference's own movement for %d differing tie rows)" % (kind, n, factor, n_diff), err,
             float(np.abs(got - ref).max()), float(np.abs(ref).max()), rtol=bound)
        assert err <= bound, "%s|%s: rel-L2 %.3e > %.1f x the reference's own noise (%.3e) with %d differing tie rows" % (kind, n, err, factor, bound / factor, n_diff)
        worst = max(worst, err / bound)
    return worst


def check_whole_gradient_bounded(step_d, noise, prefix, grads, n_diff, factor, skip=()):
    """The whole G gradient (all golden entries concatenated): rel-L2 movement <= factor x the reference's own for n_diff tie rows."""
    a_all, b_all = [], []
    for n, g in grads.items():
        if n.endswith(tuple(skip)):
            continue
        ref, got = _entry(step_d, prefix + n, g.detach().cpu().numpy())
        a_all.append(got.astype(np.float64)); b_all.append(ref.astype(np.float64))
    a, b = np.concatenate(a_all), np.concatenate(b_all)
    mov = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
    cos = float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))
    bound = noise.whole_g_bound(n_diff, factor)
    _log(prefix + "* (whole gradient rel-L2; cosine %.4f; bound = %.1f x the reference's own movement for %d differing tie rows)" % (cos, factor, n_diff),
         mov, 0.0, float(np.abs(b).max()), rtol=bound)
    assert mov <= bound, "%s*: whole-gradient rel-L2 %.3e (cosine %.4f) > %.1f x the reference's own %.3e" % (prefix, mov, cos, factor, bound / factor)
    return mov, bound


def _ref_flat(d, name):
    """The stored reference values of golden entry `name` (all of them, or its strided samples)."""
    return (d[name + "|full"] if name + "|full" in d else d[name + "|samples"]).reshape(-1)


def check_adam_updates(d, kind, named_params, init, grad_prefixes, grad_rel_err=None, lr=1e-4, k_noise=20.0, rtol=5e-2, min_ok=0.99,
                       min_selected=0.2, skip=(), what="", ref_noise_prefix=None, select_by_gradient=True, atol=1e-8):
    """Post-Adam parameters against the reference, compared as UPDATES p - p0 (round-4 review, weak #1: one Adam step moves an element
    by <= lr, so any tolerance of the size of lr on p itself holds for a wrong-sign or an absent update).

    golden `d` holds '<kind>param|<name>' (the reference's parameters after the step(s)) and, per step, '<prefix><name>' (the reference's
    gradient of that step) in the same full / strided-sample form; `init` = the fixture start values.  Adam's update is homogeneous of
    degree 0 in the gradient, so an element whose gradient is rounding noise moves by +-lr with a sign nobody reproduces; an element is
    therefore compared only when its golden gradient is >= k_noise x the noise level of its tensor IN EVERY STEP (noise level = the
    build's measured rel-L2 error of that gradient tensor x the tensor's rms; `grad_rel_err[name]` = one value per step, or a float),
    and then to `rtol` of the update.  Asserted: >= min_ok of the compared elements agree (overall, and >= min_ok - 0.02 per tensor with
    >= 50 of them), and the compared elements are >= min_selected of all stored ones -- the check is not vacuous.  ref_noise_prefix
    (multi-step golden G19: 'noise_rms|'): the noise level is at least the reference's OWN float32-vs-float64 rms difference of that
    gradient ('<ref_noise_prefix><prefix><name>').  Returns (compared, agreeing, stored)."""
    n_sel = n_ok = n_all = 0
    for n, p in named_params:
        if n.endswith(tuple(skip)):
            continue
        a = p.detach().cpu().numpy()
        ref, got = _entry(d, "%sparam|%s" % (kind, n), a)
        _, p0 = _entry(d, "%sparam|%s" % (kind, n), init[n].detach().cpu().numpy())
        upd_ref = ref.astype(np.float64) - p0.astype(np.float64)
        upd_got = got.astype(np.float64) - p0.astype(np.float64)
        sel = np.abs(upd_ref) > (0.3 if select_by_gradient else 0.05) * lr
        # select_by_gradient=False (a step from a mid-training state, golden G20): the update lr*m_hat/(sqrt(v_hat)+eps) is dominated by
        # the loaded moments, a noise-level gradient no longer decides its sign -- every element that moves at all is compared
        for k, pre in enumerate(grad_prefixes if select_by_gradient else ()):
            g = _ref_flat(d, pre + n).astype(np.float64)
            e = grad_rel_err
            if isinstance(e, dict):
                e = e.get(n, 1e-3)
            if isinstance(e, (list, tuple)):
                e = e[k]
            noise = max(float(e if e is not None else 1e-3), 1e-5) * float(np.sqrt((g ** 2).mean()))
            if ref_noise_prefix is not None:      # multi-step goldens: the reference's own float32-vs-float64 rms difference of this gradient
                noise = max(noise, float(d[ref_noise_prefix + pre + n]))
            sel &= np.abs(g) >= k_noise * noise
        ok = np.abs(upd_got - upd_ref) <= rtol * np.abs(upd_ref) + atol
        s, o = int(sel.sum()), int((ok & sel).sum())
        if s >= 50:
            assert o >= (min_ok - 0.02) * s, "%s %supd|%s: only %d of %d above-noise elements received the reference's update (worst |d| %.2e, lr %.0e)" % (
                what, kind, n, o, s, float(np.abs(upd_got - upd_ref)[sel].max()), lr)
        n_sel += s; n_ok += o; n_all += sel.size
    _log("%supd|* (Adam updates p - p0 on above-noise elements: %d of %d stored compared, %d within %.0e of the update)" % (kind, n_sel, n_all, n_ok, rtol),
         1.0 - n_ok / max(n_sel, 1), 0.0, lr, rtol=1.0 - min_ok)
    assert n_sel >= min_selected * n_all, "%s %supd: only %d of %d stored elements are above the gradient noise floor -- vacuous" % (what, kind, n_sel, n_all)
    if n_sel >= 100 or min_selected > 0:     # a handful of elements (a chaotic multi-step trajectory of G) is no statistic: logged above only
        assert n_ok >= min_ok * n_sel, "%s %supd: %d of %d above-noise elements received the reference's update" % (what, kind, n_ok, n_sel)
    return n_sel, n_ok, n_all


def worst_reference_noise(d, prefix, skip=()):
    """max over the tensors of one network of golden 'noise|<prefix><name>' (the reference's own float32-vs-float64 rel-L2 distance of that
    tensor; zero-gradient biases excluded).  In the chaotic regime of a multi-step trajectory WHICH tensor is hit hardest differs between
    two float32 realisations, so bounds are taken per network and step, not per tensor."""
    vals = [float(d[k]) for k in d.files if k.startswith("noise|" + prefix) and not k.endswith(tuple(skip))]
    return max(vals)


def measured_grad_errors(d, prefix, grads, skip=()):
    """{name: rel-L2 error of the build's gradient against golden '<prefix><name>'} for check_adam_updates' noise floor."""
    out = {}
    for n, g in grads.items():
        if n.endswith(tuple(skip)):
            continue
        ref, got = _entry(d, prefix + n, g.detach().cpu().numpy())
        out[n] = float(np.sqrt(((got.astype(np.float64) - ref) ** 2).sum()) / max(np.sqrt((ref.astype(np.float64) ** 2).sum()), 1e-30))
    return out


def load_mid_state(net, opt, shapes, salt, batches):
    """spgan.fixture_rng.mid_training_state into a module's BatchNorm buffers and its spgan.optim.Adam (what make_golden.py::g20 loads
    into the reference's modules and torch.optim.Adam)."""
    import torch as _t
    from spgan import fixture_rng as fr
    st = fr.mid_training_state(shapes, [n for n, _ in net.named_buffers()], salt=salt, batches=batches)
    net.load_state_dict({**net.state_dict(), **st["buffers"]})
    m, v = _t.zeros_like(opt.m), _t.zeros_like(opt.v)
    for (n, p), off in zip(net.named_parameters(), opt.fp.offsets):
        k = p.numel()
        m[off:off + k].copy_(st["m"][n].reshape(-1)); v[off:off + k].copy_(st["v"][n].reshape(-1))
    opt.load_state_dict({"m": m, "v": v, "t": st["step"], "lr": 1e-4, "betas": (0.5, 0.99), "eps": 1e-8})
    return st


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
    self-test mode) where pytest's monkeypatch fixture is not available.  Returns restore(): a caller inside the pytest process MUST
    call it (or use `kernel_models()` below) -- the patch is process-wide and would otherwise leak into every later test of the
    session (round-4 review, weak #4).  TEST INFRASTRUCTURE: the product never calls this."""
    import inspect
    import kernel_model as km
    import spgan.modules as modules
    import spgan.ops as ops
    missing = object()
    saved = {}

    def put_(obj, name, value):
        saved.setdefault((id(obj), name), (obj, name, getattr(obj, name, missing)))
        setattr(obj, name, value)
    for name, fn in inspect.getmembers(km, inspect.isfunction):
        if not name.startswith("_"):
            put_(ops, name, fn)
    put_(ops, "SparseAffine", km.SparseAffine)
    put_(ops, "Affine2", km.Affine2)
    put_(ops, "ActOperand", km.ActOperand)
    put_(modules, "_require_gpu", lambda t, what: None)

    def restore():
        for obj, name, value in saved.values():
            if value is missing:
                delattr(obj, name)
            else:
                setattr(obj, name, value)
        saved.clear()
    return restore


class kernel_models:
    """with kernel_models(): ... -- install_kernel_models() for a block inside the pytest process, undone on exit."""

    def __enter__(self):
        self._restore = install_kernel_models()
        return self

    def __exit__(self, *exc):
        self._restore()
