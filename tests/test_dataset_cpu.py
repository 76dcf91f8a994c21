"""CPU: the device-resident dataset (spgan.dataset) keeps the semantics of H5DataLoader + DataLoader(shuffle, drop_last)
(plain torch: runs on the CPU device here).  normalize_point_cloud is checked against the reference's numpy formula."""
import numpy as np
import torch

from spgan import dataset
from spgan import fixture_rng as fr


def _raw(S=37, P=96):
    return (fr.normal("ds.raw", (S, P, 3)) * torch.tensor([2.0, 0.5, 1.0]) + torch.tensor([3.0, -1.0, 0.2])).numpy()


def test_normalize_point_cloud_matches_reference_formula():
    raw = _raw()
    out = dataset.normalize_point_cloud(torch.from_numpy(raw)).numpy()
    c = raw.mean(axis=1, keepdims=True)                                     # point_operation.py:155-160
    pc = raw - c
    ref = pc / np.amax(np.sqrt(np.sum(pc ** 2, axis=-1, keepdims=True)), axis=1, keepdims=True)
    np.testing.assert_allclose(out, ref, rtol=1e-5, atol=1e-6)


def test_epoch_semantics(tmp_path):
    raw = _raw()
    np.save(tmp_path / "chair.npy", raw)
    ds = dataset.DeviceDataset(str(tmp_path / "chair.npy"), num_points=64, batch_size=8, scale=1.0, device="cpu", seed=3)
    assert len(ds) == 37 and ds.num_batches == 4
    batches = list(ds)
    assert len(batches) == 4 and all(b.shape == (8, 64, 3) for b in batches)          # drop_last
    norm = dataset.normalize_point_cloud(torch.from_numpy(raw))[:, :64]          # H5DataLoader.py:107 normalises the stored cloud, :113 slices
    keys = {tuple(np.round(np.sort(c.numpy(), axis=0).ravel(), 5)) for c in norm}
    seen = [tuple(np.round(np.sort(c.numpy(), axis=0).ravel(), 5)) for b in batches for c in b]
    assert all(k in keys for k in seen) and len(set(seen)) == 32                        # every item a permuted copy, no repeats
    first = batches[0][0]
    src = [c for c in norm if np.allclose(np.sort(c.numpy(), axis=0), np.sort(first.numpy(), axis=0), atol=1e-6)][0]
    assert not torch.equal(first, src)                                                  # the points were shuffled
    again = [b for b in ds]
    assert not all(torch.equal(a, b) for a, b in zip(batches, again))                   # a new order next epoch


def test_augment_is_rotation_about_y_and_scale():
    raw = _raw(S=16)
    ds = dataset.DeviceDataset(raw, num_points=96, batch_size=16, augment=True, device="cpu", seed=5)
    ds.gen.manual_seed(11)
    idx = torch.arange(16)
    aug = ds.get_batch(idx)
    ds.augment = False
    ds.gen.manual_seed(11)
    plain = ds.get_batch(idx)                                                           # same point permutation (same generator state)
    r_plain, r_aug = plain.norm(dim=-1), aug.norm(dim=-1)
    scale = (r_aug / r_plain).mean(dim=1)
    assert torch.allclose(r_aug, r_plain * scale[:, None], rtol=1e-4)                   # one scale per cloud ...
    assert (scale > 0.8 - 1e-4).all() and (scale < 1.25 + 1e-4).all() and scale.std() > 0.01
    assert torch.allclose(aug[..., 1], plain[..., 1] * scale[:, None], rtol=1e-4, atol=1e-6)   # ... and y only scaled: rotation about y


def test_data_path_matches_reference_golden():
    """G16 (tests/golden/make_golden.py::g16: the reference's normalize_point_cloud / rotate_point_cloud_and_gt /
    random_scale_point_cloud_and_gt and H5DataLoader.__getitem__'s call order, numpy RNG seeded, its draws recorded): the set
    normalisation and the per-item transform given the same draws."""
    from helpers import golden
    d = golden("g16_data_path.npz")
    norm = 0.9 * dataset.normalize_point_cloud(torch.from_numpy(d["raw"]))
    np.testing.assert_allclose(norm.numpy(), d["normalized"], rtol=0, atol=3e-7)
    n6 = dataset.normalize_point_cloud(torch.from_numpy(d["raw6"]))
    np.testing.assert_allclose(n6.numpy(), d["normalized6"], rtol=0, atol=3e-7)     # extra channels pass through
    assert np.array_equal(n6.numpy()[..., 3:], d["raw6"][..., 3:])
    pts = torch.from_numpy(d["normalized"])[:, :256]
    out = dataset.item_transform(pts, torch.from_numpy(d["perm"]), torch.from_numpy(d["angle_y"]).float(), torch.from_numpy(d["scale"]).float())
    np.testing.assert_allclose(out.numpy(), d["items"], rtol=0, atol=5e-7)           # float32 here, float64 rotation in the reference
    out64 = dataset.item_transform(pts.double(), torch.from_numpy(d["perm"]), torch.from_numpy(d["angle_y"]), torch.from_numpy(d["scale"]))
    np.testing.assert_allclose(out64.numpy(), d["items"], rtol=0, atol=1.2e-7)       # float64 maths: only the final float32 rounding differs
    plain = dataset.item_transform(pts, torch.from_numpy(d["perm"]))
    assert torch.equal(plain[2], pts[2][torch.from_numpy(d["perm"][2])])


def test_host_staged_loader_semantics():
    """HostStagedLoader (pinned staging + side-stream H2D on the GPU; here on the CPU device): same epoch contract as DeviceDataset."""
    raw = _raw()
    ld = dataset.HostStagedLoader(raw, num_points=64, batch_size=8, device="cpu", seed=5)
    assert len(ld) == 37 and ld.num_batches == 4
    norm = dataset.normalize_point_cloud(torch.from_numpy(raw))[:, :64]
    seen = []
    for b in ld:
        assert b.shape == (8, 64, 3)
        for cloud in b.clone():
            hit = [i for i in range(37) if torch.allclose(torch.sort(cloud[:, 0])[0], torch.sort(norm[i][:, 0])[0], atol=1e-6)]
            assert len(hit) == 1
            seen.append(hit[0])
    assert len(set(seen)) == 32                                                       # 4 batches x 8 distinct shapes, drop_last
    aug = dataset.HostStagedLoader(raw[:, :64], num_points=64, batch_size=8, augment=True, device="cpu", seed=6)    # the stored cloud has exactly num_points (as the reference's poisson_<np> sets)
    b = next(iter(aug))
    r = b.norm(dim=-1).amax(dim=1)
    assert float(r.min()) >= 0.8 - 1e-5 and float(r.max()) <= 1.25 + 1e-5             # unit-radius clouds scaled by U[0.8, 1.25]
