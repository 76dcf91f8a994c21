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
    norm = dataset.normalize_point_cloud(torch.from_numpy(raw)[:, :64])
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
