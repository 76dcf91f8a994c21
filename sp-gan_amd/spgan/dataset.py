"""Device-resident training data: the work of Generation/H5DataLoader.py:43-123 + torch DataLoader
(model.py:209-212: shuffle=True, drop_last=True, pin_memory) without a host loader.

A ShapeNet category at 2048 points is ~100 MB; MI355X has 288 GB.  The whole set is normalised once and kept in HBM;
an epoch is a device-side permutation, a batch is a gather, and the per-item work of `__getitem__` (point shuffle, optional
rotation about the up axis and random scale: H5DataLoader.py:113-118, point_operation.py:207-308) runs on the device for the
whole batch.  No worker processes, no pinned staging, no PCIe traffic per step.

Sources: a numpy array / torch tensor [S, P, >=3], an `.npy` / `.npz` file, or the reference's `<data_root>/<np>/<choice>.h5` with
its `poisson_<np>` dataset -- through h5py when it is importable, otherwise through the built-in reader spgan.h5lite.
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
        key = "poisson_%d" % num_points                                      # H5DataLoader.py:14-17: f['poisson_%d' % num_points][:]
        try:
            import h5py
        except ImportError:                                                  # h5py is absent in the build image: the built-in reader
            from . import h5lite
            return torch.from_numpy(np.ascontiguousarray(h5lite.read(source, key), dtype=np.float32))
        with h5py.File(source, "r") as f:
            return torch.from_numpy(f[key][:].astype(np.float32))
    raise ValueError("unsupported point source: %s" % source)


def item_transform(points: Tensor, perm: Tensor, angle_y: Optional[Tensor] = None, scale: Optional[Tensor] = None) -> Tensor:
    """The deterministic part of `H5DataLoader.__getitem__` (H5DataLoader.py:113-118) for a batch, given the random draws:
    rows permuted (np.random.shuffle), then `pc @ Ry(angle)` (point_operation.py:215-230, y_rotated) and `pc * scale`
    (point_operation.py:300-301).  points [B,P,3], perm int64 [B,P], angle_y / scale [B] or None (no augmentation).
    Pinned by tests/golden/g16_data_path.npz (captured from the reference functions)."""
    B = points.shape[0]
    out = points[torch.arange(B, device=points.device)[:, None], perm]
    if angle_y is not None:
        c, s = torch.cos(angle_y), torch.sin(angle_y)
        R = torch.zeros((B, 3, 3), device=points.device, dtype=points.dtype)
        R[:, 0, 0] = c; R[:, 0, 2] = s; R[:, 1, 1] = 1.0; R[:, 2, 0] = -s; R[:, 2, 2] = c
        out = torch.bmm(out, R)                                                # pc @ rotation_matrix
    if scale is not None:
        out = out * scale.view(B, 1, 1).to(points.dtype)
    return out


class DeviceDataset:
    """All shapes of a category in HBM, normalised as H5DataLoader.py:107 (`opts.scale * normalize_point_cloud(data)`).
    Iterating yields `len(self) // bs` batches [bs, np, 3] per epoch in a fresh random order (shuffle=True, drop_last=True),
    every cloud with its points permuted and, with `augment`, rotated about y and scaled by U[0.8, 1.25]."""

    def __init__(self, source, num_points: int = 2048, batch_size: int = 32, scale: float = 1.0, augment: bool = False,
                 device="cuda", seed: Optional[int] = None):
        pts = load_points(source, num_points)[:, :, :3]
        if pts.dim() != 3 or pts.shape[1] < num_points:
            raise ValueError("need [S, >=%d, >=3] points, got %s" % (num_points, tuple(pts.shape)))
        self.device = torch.device(device)
        # H5DataLoader.py:107 normalises the clouds as stored; :113 takes the first num_points of the normalised cloud
        self.data = (scale * normalize_point_cloud(pts))[:, :num_points].to(self.device).contiguous()
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
        angle = scale = None
        if self.augment:
            angle = torch.rand((B,), generator=g, device=dev) * (2 * math.pi)  # rotate_point_cloud_and_gt: y_rotated=True -> Ry
            scale = 0.8 + 0.45 * torch.rand((B,), generator=g, device=dev)     # random_scale_point_cloud_and_gt: U[0.8, 1.25]
        return item_transform(self.data[index], perm, angle, scale)

    def __iter__(self) -> Iterator[Tensor]:
        order = torch.randperm(len(self), generator=self.gen, device=self.device)
        for b in range(self.num_batches):
            yield self.get_batch(order[b * self.batch_size:(b + 1) * self.batch_size])


class HostStagedLoader:
    """The classic staging path for a point set that shall NOT live in HBM ("pinned host buffers -> async H2D"):
    the normalised set stays in pinned host memory; a batch is gathered / shuffled / augmented on the host (numpy, as the reference's
    DataLoader workers do, H5DataLoader.py:113-118) straight into one of two pinned staging buffers and copied to the device on a
    side stream while the previous batch is being consumed; the consumer's stream waits on the copy's event, never the host.
    Same iteration contract as DeviceDataset (shuffle=True, drop_last=True); yields [bs, np, 3] device tensors.  A yielded batch is
    valid only UNTIL THE NEXT ONE IS REQUESTED: asking for batch b+1 stages batch b+2 into the device slot batch b was handed out from
    (the copy waits for the consumer-stream work enqueued before that request, not for work enqueued later) -- clone a batch that
    has to outlive the next `next()`."""

    def __init__(self, source, num_points: int = 2048, batch_size: int = 32, scale: float = 1.0, augment: bool = False,
                 device="cuda", seed: Optional[int] = None):
        pts = load_points(source, num_points)[:, :, :3]
        if pts.dim() != 3 or pts.shape[1] < num_points:
            raise ValueError("need [S, >=%d, >=3] points, got %s" % (num_points, tuple(pts.shape)))
        self.device = torch.device(device)
        self.data = (scale * normalize_point_cloud(pts))[:, :num_points].contiguous().numpy()        # H5DataLoader.py:107, then :113
        self.num_points, self.batch_size, self.augment = num_points, batch_size, augment
        self.rng = np.random.default_rng(seed)
        cuda = self.device.type == "cuda"
        self._host = [torch.empty((batch_size, num_points, 3), dtype=torch.float32, pin_memory=cuda) for _ in range(2)]
        self._dev = [torch.empty((batch_size, num_points, 3), dtype=torch.float32, device=self.device) for _ in range(2)]
        self._copy = torch.cuda.Stream(device=self.device) if cuda else None
        self._done = [None, None]         # event: H2D copy into slot i finished
        self._free = [None, None]         # event: the consumer no longer reads _dev[i]

    def __len__(self) -> int:
        return self.data.shape[0]

    @property
    def num_batches(self) -> int:
        return len(self) // self.batch_size

    def _stage(self, slot: int, index: np.ndarray) -> None:
        B, P = self.batch_size, self.num_points
        perm = self.rng.random((B, P)).argsort(axis=1)
        angle = scale = None
        if self.augment:
            angle = torch.from_numpy(self.rng.random(B) * (2 * math.pi))
            scale = torch.from_numpy(0.8 + 0.45 * self.rng.random(B))
        out = item_transform(torch.from_numpy(self.data[index]), torch.from_numpy(perm),
                             None if angle is None else angle.float(), None if scale is None else scale.float())
        if self._done[slot] is not None:
            self._done[slot].synchronize()                                     # the DMA that last read this pinned buffer has finished (host-side wait; two batches old)
        self._host[slot].copy_(out)
        if self._copy is None:
            self._dev[slot].copy_(self._host[slot])
            return
        if self._free[slot] is not None:
            self._copy.wait_event(self._free[slot])                            # the batch last handed out from this slot was consumed
        with torch.cuda.stream(self._copy):
            self._dev[slot].copy_(self._host[slot], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy)
        self._done[slot] = ev

    def __iter__(self) -> Iterator[Tensor]:
        order = self.rng.permutation(len(self))
        nb, bs = self.num_batches, self.batch_size
        if nb == 0:
            return
        self._stage(0, order[:bs])
        for b in range(nb):
            slot = b & 1
            if b + 1 < nb:
                self._stage(slot ^ 1, order[(b + 1) * bs:(b + 2) * bs])        # next batch: host work + H2D overlap the consumer
            if self._copy is not None:
                torch.cuda.current_stream().wait_event(self._done[slot])
            try:
                yield self._dev[slot]
            finally:
                # also when the consumer leaves the loop early (break / exception / generator close): the next epoch's copy into
                # this slot must still wait for the work that read it
                if self._copy is not None:
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream())
                    self._free[slot] = ev
