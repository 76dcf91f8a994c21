"""GPU: SURVEY 8(a)9 (noise_generator / sphere_generator / pc_normalize, Generation/model.py:46-52,122-180) and 8(f) N2
(H5DataLoader.__getitem__ semantics, Generation/H5DataLoader.py:97-123 over point_operation.py:21-40,84-112,169-185) ON THE
MI355X against the vectors captured from the reference's own functions: golden G15 (samplers) and G16 (data path).  The CPU
twins of these tests (test_sampling_cpu.py, test_dataset_cpu.py) pin the same code on the CPU device; here every tensor lives
on `cuda` and the comparisons are against the reference's outputs, never against this repo's own functions."""
import numpy as np
import pytest
import torch

from helpers import golden
from spgan import dataset, sampling

pytestmark = pytest.mark.gpu
DEV = "cuda"


class O:
    np = 256; nz = 128; nv = 0.2; n_rand = False; n_mix = False


def test_pc_normalize_on_gpu_matches_reference():
    d = golden("g15_samplers.npz")
    out = sampling.pc_normalize(torch.from_numpy(d["pc_normalize|in"]).to(DEV))             # float64 on the device
    np.testing.assert_allclose(out.cpu().numpy(), d["pc_normalize|out"], rtol=1e-12, atol=1e-14)
    out32 = sampling.pc_normalize(torch.from_numpy(d["pc_normalize|in"]).float().to(DEV))
    np.testing.assert_allclose(out32.cpu().numpy(), d["pc_normalize|out"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("n_pts", [256, 2048])
def test_sphere_prior_and_region_mixing_on_gpu_match_reference(n_pts):
    d = golden("g15_samplers.npz")

    class On(O):
        np = n_pts; n_mix = True
    s = sampling.InputSampler(On, device=DEV, seed=1)
    ball = s.sphere_generator(3)
    assert ball.is_cuda and ball.shape == (3, n_pts, 3)
    assert np.array_equal(ball[0].cpu().numpy(), d["N%d|ball" % n_pts]) and torch.equal(ball[0], ball[2])   # bit-exact prior
    ids = torch.from_numpy(d["N%d|order_ids" % n_pts]).to(DEV)
    order = s._region_order(ids)
    assert order.is_cuda and np.array_equal(order.cpu().numpy(), d["N%d|order" % n_pts])      # the reference's argsort rows
    mask = s.region_mask(torch.from_numpy(d["N%d|mix_ids" % n_pts]).to(DEV), torch.from_numpy(d["N%d|mix_num" % n_pts]).to(DEV))
    ref = np.unpackbits(d["N%d|mix_mask" % n_pts], axis=1)[:, :n_pts].astype(bool)
    assert mask.is_cuda and np.array_equal(mask.cpu().numpy(), ref)
    # the random halves: shapes / tiling / distribution of model.py:128-131, region structure of :133-147
    z = s.noise_generator(64)
    assert z.is_cuda and z.shape == (64, n_pts, 128)
    for b in range(0, 64, 9):
        vals = torch.unique(z[b], dim=0)
        assert vals.shape[0] in (1, 2)
    first = z[:, 0]
    assert abs(first.std().item() - 0.2) < 0.012 and abs(first.mean().item()) < 0.01
    rnd = s.sphere_generator(2, static=False)
    rows = {tuple(r) for r in d["N%d|ball" % n_pts].tolist()}
    assert all(tuple(r) in rows for r in rnd[0].cpu().numpy().tolist()[:64])


def test_noise_generator_modes_on_gpu():
    s = sampling.InputSampler(O, device=DEV, seed=2)
    z = s.noise_generator(32)
    assert z.is_cuda and torch.equal(z[:, 0], z[:, 255])                                     # one latent per shape, tiled
    zc = sampling.InputSampler(O, device=DEV, seed=2).noise_generator(32, compact=True)
    assert zc.shape == (32, 1, 128) and torch.equal(zc.expand(-1, 256, -1), z)                # the un-tiled form is the same draw

    class R(O):
        n_rand = True
    zr = sampling.InputSampler(R, device=DEV, seed=3).noise_generator(4)
    assert not torch.equal(zr[:, 0], zr[:, 1]) and abs(zr.std().item() - 0.2) < 0.005
    masks = torch.randint(0, 3, (2, 256), device=DEV)
    zm = s.noise_generator(2, masks=masks)
    for i in range(2):
        for j in range(3):
            part = zm[i, masks[i] == j]
            assert (part == part[0]).all()


def test_data_path_on_gpu_matches_reference():
    """G16: set normalisation (H5DataLoader.py:107) and the per-item transform (:113-118) given the reference's recorded draws."""
    d = golden("g16_data_path.npz")
    norm = 0.9 * dataset.normalize_point_cloud(torch.from_numpy(d["raw"]).to(DEV))
    np.testing.assert_allclose(norm.cpu().numpy(), d["normalized"], rtol=0, atol=3e-7)
    n6 = dataset.normalize_point_cloud(torch.from_numpy(d["raw6"]).to(DEV))
    np.testing.assert_allclose(n6.cpu().numpy(), d["normalized6"], rtol=0, atol=3e-7)
    pts = torch.from_numpy(d["normalized"])[:, :256].to(DEV)
    perm = torch.from_numpy(d["perm"]).to(DEV)
    out = dataset.item_transform(pts, perm, torch.from_numpy(d["angle_y"]).float().to(DEV), torch.from_numpy(d["scale"]).float().to(DEV))
    assert out.is_cuda
    np.testing.assert_allclose(out.cpu().numpy(), d["items"], rtol=0, atol=6e-7)             # float32 on the device, float64 rotation in the reference
    out64 = dataset.item_transform(pts.double(), perm, torch.from_numpy(d["angle_y"]).to(DEV), torch.from_numpy(d["scale"]).to(DEV))
    np.testing.assert_allclose(out64.cpu().numpy(), d["items"], rtol=0, atol=1.2e-7)


def test_device_dataset_on_gpu_holds_the_reference_normalisation():
    """DeviceDataset(raw) keeps `scale * normalize_point_cloud(data)` in HBM (H5DataLoader.py:107): compared with the reference's
    output; every cloud of every batch is a row-permutation of one of them (H5DataLoader.py:113-116)."""
    d = golden("g16_data_path.npz")
    ds = dataset.DeviceDataset(d["raw"], num_points=256, batch_size=2, scale=0.9, device=DEV, seed=3)
    assert ds.data.is_cuda
    np.testing.assert_allclose(ds.data.cpu().numpy(), d["normalized"][:, :256], rtol=0, atol=3e-7)
    ref = torch.from_numpy(d["normalized"][:, :256]).to(DEV)
    keys = torch.sort(ref[:, :, 0], dim=1)[0]
    seen = []
    for b in ds:
        assert b.is_cuda and b.shape == (2, 256, 3)
        fp = torch.sort(b[:, :, 0], dim=1)[0]
        dist = (fp[:, None, :] - keys[None, :, :]).abs().amax(dim=-1)
        assert float(dist.min(dim=1)[0].max()) < 3e-7
        seen += dist.argmin(dim=1).tolist()
    assert len(seen) == 4 and len(set(seen)) == 4                                            # drop_last, no repeats inside an epoch


def test_host_staged_loader_on_gpu_delivers_the_reference_normalisation():
    """Pinned staging + side-stream H2D: every delivered cloud is a row permutation of the REFERENCE-normalised set (G16)."""
    d = golden("g16_data_path.npz")
    ld = dataset.HostStagedLoader(d["raw"], num_points=256, batch_size=2, scale=0.9, device=DEV, seed=1)
    ref = torch.from_numpy(d["normalized"][:, :256]).to(DEV)
    keys = torch.sort(ref[:, :, 0], dim=1)[0]
    count = 0
    for ep in range(3):
        for b in ld:
            assert b.is_cuda
            fp = torch.sort(b[:, :, 0], dim=1)[0]
            dist = (fp[:, None, :] - keys[None, :, :]).abs().amax(dim=-1)
            assert float(dist.min(dim=1)[0].max()) < 3e-7
            count += 1
    assert count == 6
