"""Device-resident training data: the work of Generation/H5DataLoader.py:43-123 + torch DataLoader
(model.py:209-212: shuffle=True, drop_last=True, pin_memory) without a host loader.

A ShapeNet category at 2048 points is ~100 MB; MI355X has 288 GB.  The whole set is normalised once and kept in HBM;
an epoch is a device-side permutation, a batch is a gather, and the per-item work of `__getitem__` (point shuffle, optional
rotation about the up axis and random scale: H5DataLoader.py:113-118, point_operation.py:207-308) runs on the device for the
whole batch.  No worker processes, no pinned staging, no PCIe traffic per step.

Sources: a numpy array / torch tensor [S, P, >=3], an `.npy` / `.npz` file, or (when h5py is importable -- it is not in this
image) the reference's `<data_root>/<np>/<choice>.h5` with its `poisson_<np>` dataset.
"""
from __future__ import annotations

import math
import os
from typing import Iterator, Optional, Union

import numpy as np
import torch

Tensor = torch.Tensor


def normalize_point_cloud(pc: Tensor) -> Tensor:
    """point_operation.py:144-163 for [S,P,3(+C)]: centre every cloud on its centroid and scale its farthest point to radius 1
    (extra channels pass through)."""
    xyz = pc[..., :3]
    xyz = xyz - xyz.mean(dim=1, keepdim=True)
    xyz = xyz / xyz.norm(dim=-1, keepdim=True).amax(dim=1, keepdim=True)
    return xyz if pc.shape[-1] == 3 else torch.cat([xyz, pc[..., 3:]], dim=-1)


def load_points(source: Union[str, np.ndarray, Tensor], num_points: int) -> Tensor:
    """-> float32 [S, P, 3+] on the CPU.  H5 files follow H5DataLoader.load_h5 (dataset 'poisson_<num_points>')."""
    if isinstance(source, torch.Tensor):
        return source.detach().float().cpu()
    if isinstance(source, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(source, dtype=np.float32))
    ext = os.path.splitext(source)[1].lower()
    if ext == ".npy":
        return torch.from_numpy(np.load(source).astype(np.float32))
    if ext == ".npz":
        z = np.load(source)
        key = "poisson_%d" % num_points if "poisson_%d" % num_points in z.files else z.files[0]
        return torch.from_numpy(z[key].astype(np.float32))
    if ext in (".h5", ".hdf5"):
        try:
            import h5py
        except ImportError as e:                                            # pragma: no cover (h5py is absent in the build image)
            raise RuntimeError("reading %s needs h5py (H5DataLoader.py:14-17); convert it to .npy/.npz or install h5py" % source) from e
        with h5py.File(source, "r") as f:
            return torch.from_numpy(f["poisson_%d" % num_points][:].astype(np.float32))
    raise ValueError("unsupported point source: %s" % source)


class DeviceDataset:
    """All shapes of a category in HBM, normalised as H5DataLoader.py:107 (`opts.scale * normalize_point_cloud(data)`).
    Iterating yields `len(self) // bs` batches [bs, np, 3] per epoch in a fresh random order (shuffle=True, drop_last=True),
    every cloud with its points permuted and, with `augment`, rotated about y and scaled by U[0.8, 1.25]."""

    def __init__(self, source, num_points: int = 2048, batch_size: int = 32, scale: float = 1.0, augment: bool = False,
                 device="cuda", seed: Optional[int] = None):
        pts = load_points(source, num_points)[:, :num_points, :3]
        if pts.dim() != 3 or pts.shape[1] < num_points:
            raise ValueError("need [S, >=%d, >=3] points, got %s" % (num_points, tuple(pts.shape)))
        self.device = torch.device(device)
        self.data = (scale * normalize_point_cloud(pts)).to(self.device).contiguous()
        self.num_points, self.batch_size, self.augment = num_points, batch_size, augment
        self.gen = torch.Generator(device=self.device)
        if seed is not None:
            self.gen.manual_seed(seed)

    def __len__(self) -> int:
        return self.data.shape[0]

    @property
    def num_batches(self) -> int:
        return len(self) // self.batch_size                                    # model.py:213

    def get_batch(self, index: Tensor) -> Tensor:
        """The batch of `__getitem__(i) for i in index` (H5DataLoader.py:113-123)."""
        dev, g = self.device, self.gen
        B, P = index.numel(), self.num_points
        perm = torch.rand((B, P), generator=g, device=dev).argsort(dim=1)      # np.random.shuffle(point_set), per cloud
        batch = self.data[index][torch.arange(B, device=dev)[:, None], perm]
        if self.augment:
            ang = torch.rand((B,), generator=g, device=dev) * (2 * math.pi)    # rotate_point_cloud_and_gt: y_rotated=True -> Ry
            c, s = torch.cos(ang), torch.sin(ang)
            R = torch.zeros((B, 3, 3), device=dev)
            R[:, 0, 0] = c; R[:, 0, 2] = s; R[:, 1, 1] = 1.0; R[:, 2, 0] = -s; R[:, 2, 2] = c
            batch = torch.bmm(batch, R)                                        # pc @ rotation_matrix
            batch = batch * (0.8 + 0.45 * torch.rand((B, 1, 1), generator=g, device=dev))   # random_scale_point_cloud_and_gt
        return batch

    def __iter__(self) -> Iterator[Tensor]:
        order = torch.randperm(len(self), generator=self.gen, device=self.device)
        for b in range(self.num_batches):
            yield self.get_batch(order[b * self.batch_size:(b + 1) * self.batch_size])
