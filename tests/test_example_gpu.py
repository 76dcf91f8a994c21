"""GPU: examples/train.py runs end to end (device dataset -> samplers -> graph-replayed TrainStep -> reference-layout
checkpoints -> eval-mode samples) and its checkpoints load back into fresh modules."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_example(tmp_path):
    out = str(tmp_path / "run")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "train.py"), "--synthetic", "64", "--np", "256", "--bs", "8",
                        "--epochs", "2", "--gan", "wgan", "--gp", "--out", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "epoch 1" in r.stdout
    ck = torch.load(os.path.join(out, "1_chair_G.pth"))
    assert set(ck) == {"G_model", "G_optimizer", "G_epoch"} and ck["G_optimizer"]["t"] == 16

    import spgan

    class Opts:
        np = 256; nk = 20; nz = 128; softmax = True; off = False; attn = False; use_head = False; eql = False; z_norm = False; small_d = False
    G = spgan.Generator(Opts)
    G.load_state_dict(ck["G_model"])
    assert int(G.global_conv[1].num_batches_tracked) == 2 * 16 + 0      # two G forwards per step, 16 steps
    pts = open(os.path.join(out, "sample", "0.xyz")).read().split("\n")
    assert len([l for l in pts if l.strip()]) == 256


def test_host_staged_loader_on_gpu():
    """Pinned staging buffers + H2D on a side stream: every batch arrives intact although the next one is being staged and copied
    while it is consumed (two slots, event hand-over in both directions)."""
    import numpy as np
    import torch
    from spgan import dataset, fixture_rng as fr
    raw = fr.synthetic_real(70, 512, seed=9).numpy() * 2.0 + 1.0
    ld = dataset.HostStagedLoader(raw, num_points=512, batch_size=16, device="cuda", seed=1)
    ref = dataset.normalize_point_cloud(torch.from_numpy(raw)).cuda()
    keys = torch.sort(ref[:, :, 0], dim=1)[0]                                   # [S,P] fingerprint of every source cloud
    count = 0
    acc = torch.zeros((), device="cuda")
    for ep in range(2):
        for b in ld:
            assert b.is_cuda and b.shape == (16, 512, 3)
            acc = acc + (b @ torch.randn(3, 64, device="cuda")).sum() * 0          # consumer work on the current stream
            fp = torch.sort(b[:, :, 0], dim=1)[0]
            dist = (fp[:, None, :] - keys[None, :, :]).abs().amax(dim=-1)         # [16, S]: exact element-wise distance of the fingerprints
            assert float(dist.min(dim=1)[0].max()) < 1e-6, "a batch does not consist of (permuted) source clouds"
            count += 1
    torch.cuda.synchronize()
    assert count == 2 * (70 // 16)
