"""GPU: the HIP Adam kernels (spgan_adam_step / spgan_adam_step_dev, through the C ABI) pinned against torch.optim.Adam itself
(Generation/model.py:94-97: Adam(lr, betas=(0.5, 0.99))) -- not against the builder's own formula.

What is compared is the UPDATE p - p0, never p: one Adam step moves an element by <= lr = 1e-4, so a tolerance on p of the size of
lr holds for a wrong-sign or an absent update (round-4 review, weak #1).  Two set-ups:
  * p0 = 0: p IS the accumulated update, float32 represents it to 6e-8 of itself, and the kernel must agree with torch's float32
    Adam to 1e-6 of the update, element by element, after 10 steps with a different gradient on every step;
  * realistic p0 (|p| ~ 0.05): p - p0 carries the representational rounding of p (half an ulp of p per step, 3e-9: 3e-5 of one
    update), which no float32 Adam can avoid -- bound: 1e-6 of the update + steps x ulp(p)/2, against torch's float64 Adam.
"""
import numpy as np
import pytest
import torch

from spgan import fixture_rng as fr

pytestmark = pytest.mark.gpu

LR, BETAS, EPS, STEPS = 1e-4, (0.5, 0.99), 1e-8, 10


@pytest.fixture(scope="module")
def ops():
    from spgan import ops as o
    from spgan import _lib
    _lib.load()
    return o


def _grads(n, tag):
    """STEPS fixture gradients: magnitudes over six decades (Adam's update is homogeneous of degree 0 in the gradient's scale), signs
    that change between steps for part of the entries, exact zeros in the first step for a few (m = v = 0: update 0/eps = 0)."""
    gs = []
    scale = torch.pow(10.0, fr.uniform(tag + ".dec", (n,), -6.0, 0.0))
    for s in range(STEPS):
        g = fr.normal("%s.g%d" % (tag, s), (n,)) * scale
        if s == 0:
            g[::97] = 0.0
        gs.append(g.contiguous())
    return gs, scale


def _torch_adam(p0, gs, dtype, grad_scale=1.0):
    p = torch.nn.Parameter(p0.to(dtype).clone())
    opt = torch.optim.Adam([p], lr=LR, betas=BETAS, eps=EPS)
    for g in gs:
        p.grad = (g.to(dtype) * grad_scale).clone()
        opt.step()
    st = opt.state[p]
    return p.detach(), st["exp_avg"], st["exp_avg_sq"]


ATOL = 2e-6 * LR      # where m nearly cancels (a sign change of the gradient) the update is small against lr: absolute floor 2e-6 of ONE step


def _bound(upd_ref, p_ref, rel=1e-6):
    ulp_half = np.abs(p_ref).astype(np.float32).astype(np.float64) * 2.0 ** -24
    return rel * np.abs(upd_ref) + STEPS * ulp_half + ATOL


def _close_state(got, ref, scale, power):
    """Adam's moments against torch's: 2e-6 relative, with an absolute floor of 1e-6 of the entry's gradient scale (cancellation in
    beta1*m + (1-beta1)*g; torch forms it as a lerp)."""
    got, ref, scale = got.cpu().double().numpy(), ref.double().numpy(), scale.double().numpy()
    assert (np.abs(got - ref) <= 2e-6 * np.abs(ref) + 1e-6 * scale ** power).all()


@pytest.mark.parametrize("variant", ["dev", "host"])
def test_adam_update_from_zero_equals_torch_adam(ops, variant):
    n = 50021
    gs, scale = _grads(n, "adam.zero")
    p = torch.zeros(n, device="cuda"); m = torch.zeros_like(p); v = torch.zeros_like(p)
    state = torch.tensor([0.0, 0.0, 0.0, 1.0], device="cuda")
    for s, g in enumerate(gs):
        gd = g.cuda()
        if variant == "dev":
            ops.adam_step_dev(p, gd, m, v, state, LR, BETAS[0], BETAS[1], EPS)
        else:
            ops.adam_step(p, gd, m, v, s + 1, LR, BETAS[0], BETAS[1], EPS)
    ref, rm, rv = _torch_adam(torch.zeros(n), gs, torch.float32)
    upd, upd_ref = p.cpu().double().numpy(), ref.double().numpy()
    # the update of every element is of the order of lr per step (a wrong sign or a missing step is an error of 100 %)
    moved = np.abs(upd_ref) > 1e-7
    assert moved.mean() > 0.95
    err = np.abs(upd - upd_ref)
    assert (err <= 1e-6 * np.abs(upd_ref) + ATOL).all(), "worst relative error of the update %.3e" % (err[moved] / np.abs(upd_ref[moved])).max()
    _close_state(m, rm, scale, 1); _close_state(v, rv, scale, 2)
    if variant == "dev":
        assert int(state[:1].view(torch.int32).item()) == STEPS


def test_adam_update_on_real_weights_against_float64_adam(ops):
    """Realistic start values, gradient scaling (the data-parallel 1/world factor), the fused zero_grad; reference = torch's Adam in
    float64 on the CPU (the exact update), bound = 1e-6 of the update + the unavoidable rounding of p itself."""
    n = 100003
    gs, scale = _grads(n, "adam.real")
    p0 = fr.normal("adam.real.p0", (n,)) * 0.05
    p = p0.cuda().clone(); m = torch.zeros_like(p); v = torch.zeros_like(p)
    state = torch.tensor([0.0, 0.0, 0.0, 1.0], device="cuda")
    for g in gs:
        gd = g.cuda().clone()
        ops.adam_step_dev(p, gd, m, v, state, LR, BETAS[0], BETAS[1], EPS, grad_scale=0.25, zero_grad=True)
        assert not gd.any().item(), "zero_grad=True must leave the gradient buffer zeroed"
    ref, rm, rv = _torch_adam(p0, gs, torch.float64, grad_scale=0.25)
    upd = p.cpu().double().numpy() - p0.double().numpy()
    upd_ref = ref.numpy() - p0.double().numpy()
    err = np.abs(upd - upd_ref)
    bound = _bound(upd_ref, ref.numpy())
    assert (err <= bound).all(), "update off by %.3e (bound %.3e)" % (err.max(), bound[err.argmax()])
    # and the bound bites: an update of the wrong sign, a missing step, or lr off by 1 % fail it on nearly every element
    for wrong in (-upd_ref, upd_ref * 0.9, upd_ref * 1.01):
        assert (np.abs(wrong - upd_ref) > bound).mean() > 0.9
    _close_state(m, rm, 0.25 * scale, 1); _close_state(v, rv, 0.25 * scale, 2)


def test_module_level_adam_equals_torch_adam(ops):
    """spgan.optim.Adam (flat buffer, capturable, zero_grad folded into the step) against torch.optim.Adam on the same module and
    the same per-step gradients: the update of every parameter tensor, 10 steps."""
    import spgan
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Linear(53, 3)).cuda()
    ref = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Linear(53, 3))
    ref.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()})
    init = {k: v.detach().clone() for k, v in ref.named_parameters()}
    ref = ref.double()
    opt = spgan.optim.Adam(net, LR, BETAS, capturable=True, zero_grad_in_step=True)
    ropt = torch.optim.Adam(ref.parameters(), lr=LR, betas=BETAS)
    for s in range(STEPS):
        opt.zero_grad()
        for (n_, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
            g = fr.normal("adam.mod.%s.%d" % (n_, s), tuple(p.shape)) * 10.0 ** (-(s % 4))
            p.grad.add_(g.cuda())                 # into the pre-bound flat gradient slices, as the backward passes do
            q.grad = g.double()
        opt.step(); ropt.step()
    for (n_, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        upd = (p.detach().cpu().double() - init[n_].double()).numpy()
        upd_ref = (q.detach() - init[n_].double()).numpy()
        assert (np.abs(upd - upd_ref) <= _bound(upd_ref, q.detach().numpy())).all(), n_
        assert np.abs(upd_ref).mean() > 2e-4                     # elements move by about lr per step
    assert opt.t == STEPS
