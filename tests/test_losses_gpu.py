"""GPU: the loss kernel (csrc/loss.hip::gan_loss_kernel through spgan.dis_loss / gen_loss) against golden G6, captured from the
reference's own dis_loss / gen_loss (Common/loss_utils.py:727-802, 854-972): all four modes, values and logit gradients, plus the
noisy-label path with its [B,1] x [B] broadcast (loss_utils.py:753-755, 897-901) on both the D and the G side, with labels that
really flipped."""
import numpy as np
import pytest
import torch

from helpers import golden, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sp():
    import spgan
    from spgan import _lib
    _lib.load()
    return spgan


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("gan", ["ls", "wgan", "hinge", "gan"])
def test_gan_loss_kernel_golden(sp, gan):
    d = golden("g6_losses.npz")
    dr, df = _t(d["d_real"]).requires_grad_(True), _t(d["d_fake"]).requires_grad_(True)
    l, info = sp.dis_loss(dr, df, gan=gan)
    l.backward()
    np.testing.assert_allclose(l.item(), float(d["dis|%s|loss" % gan]), rtol=2e-6)
    assert rel_l2(dr.grad.cpu().numpy(), d["dis|%s|g_real" % gan], "dis|%s|g_real" % gan) <= 2e-6
    assert rel_l2(df.grad.cpu().numpy(), d["dis|%s|g_fake" % gan], "dis|%s|g_fake" % gan) <= 2e-6
    assert 0.0 <= info["real_acc"].item() <= 1.0 and 0.0 <= info["fake_acc"].item() <= 1.0
    df2 = _t(d["d_fake"]).requires_grad_(True)
    l, _ = sp.gen_loss(dr.detach(), df2, gan=gan)
    l.backward()
    np.testing.assert_allclose(l.item(), float(d["gen|%s|loss" % gan]), rtol=2e-6)
    assert rel_l2(df2.grad.cpu().numpy(), d["gen|%s|g_fake" % gan], "gen|%s|g_fake" % gan) <= 2e-6
    # a weight scales value and gradients alike (loss_utils.py: `weight *`)
    df3 = _t(d["d_fake"]).requires_grad_(True)
    l3, _ = sp.gen_loss(None, df3, gan=gan, weight=0.25)
    l3.backward()
    np.testing.assert_allclose(l3.item(), 0.25 * float(d["gen|%s|loss" % gan]), rtol=2e-6)
    assert rel_l2(df3.grad.cpu().numpy(), 0.25 * d["gen|%s|g_fake" % gan]) <= 2e-6


@pytest.mark.parametrize("pre", ["", "b40|"])
def test_noisy_label_broadcast_golden(sp, pre):
    d = golden("g6_losses.npz")
    dr = _t(d[(pre or "") + "d_real"]).requires_grad_(True)
    df = _t(d[(pre or "") + "d_fake"]).requires_grad_(True)
    rl = _t(d[pre + "dis|ls_noisy|real_label"])
    l, info = sp.dis_loss(dr, df, gan="ls", real_label=rl)
    l.backward()
    np.testing.assert_allclose(l.item(), float(d[pre + "dis|ls_noisy|loss"]), rtol=2e-6)
    assert rel_l2(dr.grad.cpu().numpy(), d[pre + "dis|ls_noisy|g_real"], pre + "dis|ls_noisy|g_real") <= 2e-6
    assert rel_l2(df.grad.cpu().numpy(), d[pre + "dis|ls_noisy|g_fake"], pre + "dis|ls_noisy|g_fake") <= 2e-6
    assert torch.equal(info["real_label"], rl)
    if pre:
        assert (rl < 0.5).any(), "the golden case must contain a flipped label"
        fl = _t(d[pre + "gen|ls_noisy|fake_label"])
        df2 = _t(d[pre + "d_fake"]).requires_grad_(True)
        l, _ = sp.gen_loss(None, df2, gan="ls", fake_label=fl)
        l.backward()
        np.testing.assert_allclose(l.item(), float(d[pre + "gen|ls_noisy|loss"]), rtol=2e-6)
        assert rel_l2(df2.grad.cpu().numpy(), d[pre + "gen|ls_noisy|g_fake"], pre + "gen|ls_noisy|g_fake") <= 2e-6


def test_noise_label_draws_on_device_and_under_graph_capture(sp):
    """noise_label=True draws the labels with torch's device generator (loss_utils.py:698-725 semantics): a captured graph draws
    new labels on every replay instead of freezing the capture-time draw."""
    B = 40
    dr = torch.randn(B, 1, device="cuda"); df = torch.randn(B, 1, device="cuda")
    torch.manual_seed(3)
    l, info = sp.dis_loss(dr, df, gan="ls", noise_label=True)
    rl = info["real_label"]
    assert rl.is_cuda and rl.shape == (B,)
    flipped = rl < 0.5
    assert 1 <= int(flipped.sum()) <= 2                                                   # int(0.05*40) = 2 draws with replacement
    assert bool(((rl[~flipped] >= 0.9) & (rl[~flipped] < 1.0)).all()) and bool((rl[flipped] <= 0.1).all())
    l2, info2 = sp.gen_loss(None, df, gan="ls", noise_label=True)
    fl = info2["fake_label"]
    assert set(fl.unique().tolist()) <= {0.0, 1.0} and 1 <= int((fl == 0).sum()) <= 2
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        sp.dis_loss(dr, df, gan="ls", noise_label=True)                                   # warm-up outside the capture
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        lg, ig = sp.dis_loss(dr, df, gan="ls", noise_label=True)
    seen = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        seen.append((ig["real_label"].clone(), lg.item()))
    assert not torch.equal(seen[0][0], seen[1][0]) and not torch.equal(seen[1][0], seen[2][0])
    assert len({s[1] for s in seen}) == 3
