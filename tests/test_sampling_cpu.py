"""CPU: the device-side input samplers (spgan.sampling) keep the shapes / distributions / region semantics of
Generation/model.py:122-180 (run here on the CPU device; they are plain torch)."""
import numpy as np
import torch

from spgan import fixture_rng as fr
from spgan import sampling


class O:
    np = 256; nz = 128; nv = 0.2; n_rand = False; n_mix = False


def test_pc_normalize():
    pc = fr.normal("samp.pc", (100, 3)) * 3 + 5
    out = sampling.pc_normalize(pc)
    assert out.mean(0).abs().max() < 1e-5 and abs(out.norm(dim=1).max().item() - 1.0) < 1e-6


def test_sphere_generator():
    s = sampling.InputSampler(O, device="cpu", seed=1)
    ball = s.sphere_generator(3)
    assert ball.shape == (3, 256, 3) and torch.equal(ball[0], ball[2]) and torch.equal(ball[0], fr.sphere_template(256))
    rnd = s.sphere_generator(2, static=False)
    assert rnd.shape == (2, 256, 3)
    rows = {tuple(r.tolist()) for r in fr.sphere_template(256)}
    assert all(tuple(r.tolist()) in rows for r in rnd[0])                 # drawn from the template, with replacement


def test_noise_generator_default_and_rand():
    s = sampling.InputSampler(O, device="cpu", seed=2)
    z = s.noise_generator(64)
    assert z.shape == (64, 256, 128) and torch.equal(z[:, 0], z[:, 255])   # one vector per shape, tiled over the points
    assert abs(z[:, 0].std().item() - 0.2) < 0.01 and abs(z[:, 0].mean().item()) < 0.01

    class R(O):
        n_rand = True
    z = sampling.InputSampler(R, device="cpu", seed=3).noise_generator(4)
    assert not torch.equal(z[:, 0], z[:, 1]) and abs(z.std().item() - 0.2) < 0.005
    a = sampling.InputSampler(O, device="cpu", seed=7).noise_generator(2)
    b = sampling.InputSampler(O, device="cpu", seed=7).noise_generator(2)
    assert torch.equal(a, b)                                              # seedable (the reference is not)


def test_noise_generator_mix_regions():
    class Mx(O):
        n_mix = True
    s = sampling.InputSampler(Mx, device="cpu", seed=11)
    mixed = 0
    for _ in range(12):
        z = s.noise_generator(8)
        for b in range(8):
            vals = torch.unique(z[b], dim=0)
            assert vals.shape[0] in (1, 2)                                # a shape carries one vector, or two (region mixed in)
            if vals.shape[0] == 2:
                mixed += 1
                inside = (z[b] == z[b, (z[b] != z[b, 0]).any(dim=1).nonzero()[0, 0]]).all(dim=1) if (z[b] != z[b, 0]).any() else None
                n_in = min(int(inside.sum()), 256 - int(inside.sum()))
                assert n_in >= 1
    assert 0 < mixed < 96                                                 # the coin flip (p = 1/2 per call) goes both ways
    # the region is a prefix of the reference's ball_dist ordering around its centre
    s2 = sampling.InputSampler(Mx, device="cpu", seed=5)
    order = s2._region_order(torch.tensor([17]))[0]
    xx = (fr.sphere_template(256) ** 2).sum(1).double()
    dref = -2.0 * xx[17] * xx + xx[17] + xx
    assert np.array_equal(np.sort(dref.numpy()[order.numpy()[:40]]), np.sort(dref.numpy())[:40]) or \
        np.allclose(np.sort(dref.numpy()[order.numpy()[:40]]), np.sort(dref.numpy())[:40], atol=1e-6)


def test_noise_generator_masks_and_xyz(tmp_path):
    s = sampling.InputSampler(O, device="cpu", seed=4)
    masks = torch.randint(0, 3, (2, 256))
    z = s.noise_generator(2, masks=masks)
    for i in range(2):
        for j in range(3):
            part = z[i, masks[i] == j]
            assert (part == part[0]).all()                                # one vector per part
    p = tmp_path / "out" / "a.xyz"
    pts = fr.normal("samp.xyz", (3, 50))
    sampling.save_xyz(str(p), pts)
    back = np.loadtxt(str(p))
    assert back.shape == (50, 3) and np.allclose(back, pts.t().numpy(), atol=1e-6)


def test_deterministic_sampler_parts_match_reference_golden():
    """G15 (tests/golden/make_golden.py::g15, produced by the reference's own pc_normalize / sphere_generator / noise_generator
    compiled from Generation/model.py): normalisation, the sphere prior as the Generator receives it, the ball_dist region
    ordering and the point set a region-mixed latent covers for the (centre, size) the reference drew."""
    from helpers import golden
    d = golden("g15_samplers.npz")
    out = sampling.pc_normalize(torch.from_numpy(d["pc_normalize|in"]))                # float64 in -> the reference's arithmetic
    np.testing.assert_allclose(out.numpy(), d["pc_normalize|out"], rtol=1e-13, atol=1e-15)
    out32 = sampling.pc_normalize(torch.from_numpy(d["pc_normalize|in"]).float())
    np.testing.assert_allclose(out32.numpy(), d["pc_normalize|out"], rtol=0, atol=2e-6)
    for n_pts in (256, 2048):
        class On(O):
            np = n_pts; n_mix = True
        s = sampling.InputSampler(On, device="cpu", seed=1)
        ball = s.sphere_generator(2)
        assert np.array_equal(ball[0].numpy(), d["N%d|ball" % n_pts]) and torch.equal(ball[0], ball[1])   # bit-exact prior
        ids = torch.from_numpy(d["N%d|order_ids" % n_pts])
        assert np.array_equal(s._region_order(ids).numpy(), d["N%d|order" % n_pts])   # the reference's own argsort rows
        mask = s.region_mask(torch.from_numpy(d["N%d|mix_ids" % n_pts]), torch.from_numpy(d["N%d|mix_num" % n_pts]))
        ref = np.unpackbits(d["N%d|mix_mask" % n_pts], axis=1)[:, :n_pts].astype(bool)
        assert np.array_equal(mask.numpy(), ref)
        assert np.array_equal(mask.sum(1).numpy(), d["N%d|mix_num" % n_pts])
