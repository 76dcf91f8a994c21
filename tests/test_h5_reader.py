"""CPU: spgan.h5lite (pure numpy + zlib HDF5 reader) against files written by the REAL HDF5 library (tests/golden/make_h5_fixtures.py
drives libhdf5 1.10.6 through ctypes; expected arrays in tests/golden/h5/expected.npz): every layout h5py can give the reference's
`poisson_<np>` datasets, and the dataset path on top of it (Generation/H5DataLoader.py:14-17,107)."""
import os

import numpy as np
import pytest
import torch

from spgan import dataset, h5lite

H5 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "h5")
EXP = np.load(os.path.join(H5, "expected.npz"))


@pytest.mark.parametrize("key", EXP.files)
def test_dataset_bit_exact(key):
    fname, name = key.split("|")
    got = h5lite.read(os.path.join(H5, fname), name)
    assert got.dtype == EXP[key].dtype and got.shape == EXP[key].shape
    assert np.array_equal(got, EXP[key])                                   # stored bytes, decoded: no tolerance


def test_group_listing_and_errors(tmp_path):
    f = h5lite.File(os.path.join(H5, "many_datasets.h5"))
    assert "poisson_64" in f and "grp/poisson_64" in f and "nope" not in f
    assert f.keys() == sorted(["grp", "labels"] + ["poisson_%d" % n for n in (8, 16, 24, 32, 40, 48, 56, 64, 72, 80, 96, 128)])
    assert f.keys("grp") == ["poisson_64"]
    with pytest.raises(KeyError):
        f["poisson_999"]
    with pytest.raises(h5lite.H5Error):
        f["grp"]                                                           # a group, not a dataset
    bad = tmp_path / "x.h5"
    bad.write_bytes(b"not hdf5" * 100)
    with pytest.raises(h5lite.H5Error):
        h5lite.File(str(bad))


@pytest.mark.parametrize("fname", ["chair_contiguous.h5", "chair_gzip.h5", "latest_format.h5"])
def test_device_dataset_reads_reference_style_h5(fname):
    """`<data_root>/<np>/<choice>.h5` -> DeviceDataset: load_h5 + `opts.scale * normalize_point_cloud(data)` (H5DataLoader.py:97-107)."""
    ds = dataset.DeviceDataset(os.path.join(H5, fname), num_points=64, batch_size=2, scale=0.9, device="cpu", seed=1)
    raw = torch.from_numpy(EXP[fname + "|poisson_64"])
    assert torch.allclose(ds.data, 0.9 * dataset.normalize_point_cloud(raw), atol=1e-7)
    assert sum(1 for _ in ds) == len(raw) // 2
