"""GPU: ball-query / grouping family (spgan.pointnet_util, HIP) against the vectors captured from the reference's
Common/pointnet_util.py and Common/pointconv_util.py (golden G9) and against the oracle on other sizes."""
import numpy as np
import pytest
import torch

from helpers import golden
from oracle import spgan_oracle as orc
from spgan import fixture_rng as fr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pu():
    from spgan import pointnet_util
    return pointnet_util


def test_golden_g9(pu):
    d = golden("g9_ball_group.npz")
    B, N, S = 2, 256, 32
    xyz = fr.synthetic_real(B, N, seed=91).cuda()
    feat = fr.normal("g9.feat", (B, N, 5)).cuda()
    new_xyz = xyz[:, ::N // S][:, :S].contiguous()
    np.testing.assert_allclose(pu.square_distance(new_xyz, xyz).cpu().numpy(), d["square_distance"], rtol=0, atol=1e-6)
    for r, ns in ((0.3, 16), (0.15, 32), (0.02, 8)):
        assert np.array_equal(pu.query_ball_point(r, ns, xyz, new_xyz).cpu().numpy(), d["query_ball|%g|%d" % (r, ns)])
    idx = pu.query_ball_point(0.3, 16, xyz, new_xyz)
    assert np.array_equal(pu.index_points(feat, idx).cpu().numpy(), d["index_points3"])
    assert np.array_equal(pu.index_points(feat, idx[:, :, 0].contiguous()).cpu().numpy(), d["index_points2"])
    start = torch.from_numpy(d["fps_start"].astype(np.int64)).cuda()
    assert np.array_equal(pu.farthest_point_sample(xyz, 24, start).cpu().numpy(), d["fps"])
    assert np.array_equal(pu.farthest_point_sample(xyz, 24, torch.zeros(B, dtype=torch.long)).cpu().numpy(), d["fps0"])
    nx, npts = pu.sample_and_group(24, 0.3, 16, xyz, feat, start=start)
    assert np.array_equal(nx.cpu().numpy(), d["sag|new_xyz"]) and np.array_equal(npts.cpu().numpy(), d["sag|new_points"])
    knn = pu.knn_point(10, xyz, xyz)
    assert np.array_equal(torch.sort(knn, dim=-1)[0].cpu().numpy(), d["knn_point_sorted"])
    new_points, gxn = pu.group(10, xyz, feat)
    # `group` inherits knn_point's unspecified neighbour order in the reference: compare per-row as sets via a canonical sort
    gi = torch.from_numpy(d["group|idx"].astype(np.int64))
    ref_np = torch.from_numpy(d["group|new_points"])
    order_ref = torch.argsort(gi, dim=-1)
    order_got = torch.argsort(knn.cpu(), dim=-1)
    a = torch.gather(ref_np, 2, order_ref.unsqueeze(-1).expand_as(ref_np))
    b = torch.gather(new_points.cpu(), 2, order_got.unsqueeze(-1).expand_as(ref_np))
    assert torch.equal(a, b)
    assert torch.equal(gxn.cpu(), new_points.cpu()[..., :3])


@pytest.mark.parametrize("B,N,S", [(3, 2048, 512), (2, 777, 100)])
def test_vs_oracle_other_sizes(pu, B, N, S):
    xyz = fr.synthetic_real(B, N, seed=5)
    feat = fr.normal("pn.feat", (B, N, 7))
    start = torch.arange(B) * 3
    fps_ref = orc.farthest_point_sample(xyz, S, start)
    fps = pu.farthest_point_sample(xyz.cuda(), S, start.cuda())
    assert torch.equal(fps.cpu(), fps_ref)
    new_xyz = orc.index_points(xyz, fps_ref)
    for r, ns in ((0.2, 32), (0.05, 16)):
        ref = orc.query_ball_point(r, ns, xyz, new_xyz)
        got = pu.query_ball_point(r, ns, xyz.cuda(), new_xyz.cuda()).cpu()
        # a point within 1 ulp-class rounding of the sphere surface may flip: allow <0.1% differing rows
        assert (got == ref).all(dim=-1).float().mean().item() >= 0.999
    ref = orc.knn_point(16, xyz, new_xyz)
    got = pu.knn_point(16, xyz.cuda(), new_xyz.cuda()).cpu()
    assert (got == ref).all(dim=-1).float().mean().item() >= 0.995
    gi = orc.query_ball_point(0.2, 32, xyz, new_xyz)
    _, np_ref = orc.sample_and_group(S, 0.2, 32, xyz, feat, start)
    _, np_got = pu.sample_and_group(S, 0.2, 32, xyz.cuda(), feat.cuda(), start=start.cuda())
    same = (np_got.cpu() == np_ref).flatten(2).all(dim=-1).float().mean().item()
    assert same >= 0.999, same


# ---------------------------------------------------------------- adjoints of the gathers (group_points fwd/bwd, edge_gather fwd/bwd)
def _rel(a, b):
    return float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-30))


@pytest.mark.parametrize("B,N,S,K", [(2, 256, 40, 16), (3, 1000, 1000, 10)])
def test_index_points_and_group_backward_vs_oracle_autograd(pu, B, N, S, K):
    """index_points / group / sample_and_group are differentiable in points / xyz like the reference's torch indexing
    (pointnet_util.py:43-60, pointconv_util.py:174-197): gradients against torch autograd through the oracle on the CPU, identical
    index tensors on both sides; repeated indices (ball-query padding) included; two runs bit-identical (no float atomics)."""
    xyz = fr.synthetic_real(B, N, seed=51)
    feat = fr.normal("pnb.feat", (B, N, 6))
    idx3 = orc.query_ball_point(0.25, K, xyz, xyz[:, :S].contiguous())                     # [B,S,K] with padding repeats
    w = fr.normal("pnb.w", (B, S, K, 6))
    # index_points, 3-D idx
    f_ref = feat.clone().requires_grad_(True)
    (orc.index_points(f_ref, idx3) * w).sum().backward()
    f = feat.cuda().requires_grad_(True)
    out = pu.index_points(f, idx3.cuda())
    (out * w.cuda()).sum().backward()
    assert _rel(f.grad.cpu(), f_ref.grad) < 2e-6
    g1 = f.grad.clone(); f.grad = None
    (pu.index_points(f, idx3.cuda()) * w.cuda()).sum().backward()
    assert torch.equal(g1, f.grad)
    # index_points, 2-D idx
    f_ref.grad = None; f.grad = None
    (orc.index_points(f_ref, idx3[:, :, 0]) * w[:, :, 0]).sum().backward()
    (pu.index_points(f, idx3[:, :, 0].contiguous().cuda()) * w[:, :, 0].cuda()).sum().backward()
    assert _rel(f.grad.cpu(), f_ref.grad) < 2e-6
    # group (kNN around every point, centre = the point itself): gradients for xyz (gathered AND centre role) and for the features
    x_ref, f_ref = xyz.clone().requires_grad_(True), feat.clone().requires_grad_(True)
    x, f = xyz.cuda().requires_grad_(True), feat.cuda().requires_grad_(True)
    new_points, gxn = pu.group(K, x, f)
    idx = pu.knn_point(K, xyz.cuda(), xyz.cuda()).cpu()
    ref_points, ref_gxn = orc.group(K, x_ref, f_ref, idx=idx)
    wg, wn = fr.normal("pnb.wg", tuple(ref_points.shape)), fr.normal("pnb.wn", tuple(ref_gxn.shape))
    ((ref_points * wg).sum() + (ref_gxn * wn).sum()).backward()
    ((new_points * wg.cuda()).sum() + (gxn * wn.cuda()).sum()).backward()
    assert _rel(new_points.detach().cpu(), ref_points.detach()) < 1e-6
    assert _rel(x.grad.cpu(), x_ref.grad) < 3e-6 and _rel(f.grad.cpu(), f_ref.grad) < 3e-6


def test_sample_and_group_backward_vs_oracle_autograd(pu):
    B, N, S, ns = 2, 512, 64, 16
    xyz, feat = fr.synthetic_real(B, N, seed=53), fr.normal("pnb.sf", (B, N, 4))
    start = torch.arange(B) * 7
    x_ref, f_ref = xyz.clone().requires_grad_(True), feat.clone().requires_grad_(True)
    nx_ref, np_ref = orc.sample_and_group(S, 0.3, ns, x_ref, f_ref, start=start)
    w1, w2 = fr.normal("pnb.s1", tuple(nx_ref.shape)), fr.normal("pnb.s2", tuple(np_ref.shape))
    ((nx_ref * w1).sum() + (np_ref * w2).sum()).backward()
    x, f = xyz.cuda().requires_grad_(True), feat.cuda().requires_grad_(True)
    nx, npts = pu.sample_and_group(S, 0.3, ns, x, f, start=start.cuda())
    ((nx * w1.cuda()).sum() + (npts * w2.cuda()).sum()).backward()
    assert torch.equal(npts.detach().cpu(), np_ref.detach())
    assert _rel(x.grad.cpu(), x_ref.grad) < 3e-6 and _rel(f.grad.cpu(), f_ref.grad) < 3e-6


@pytest.mark.parametrize("B,C,N", [(2, 3, 256), (2, 64, 512), (1, 5, 300)])
def test_get_edge_features_backward_vs_oracle_autograd(B, C, N):
    """spgan.get_edge_features is differentiable in x like Generation/modules.py:708-720 (central term, the '-central' term and the
    in-edge scatter, with variable in-degree): against torch autograd through the oracle with the same indices."""
    import spgan
    k = 10
    x = fr.normal("gefb.x%d" % C, (B, C, N), 0.6)
    xg = x.cuda().requires_grad_(True)
    ee, idx = spgan.get_edge_features(xg, k, return_idx=True)
    x_ref = x.clone().requires_grad_(True)
    ee_ref = orc.get_edge_features(x_ref, k, idx=idx.cpu())
    assert torch.equal(ee.detach().cpu(), ee_ref.detach())
    w = fr.normal("gefb.w%d" % C, tuple(ee_ref.shape))
    (ee_ref * w).sum().backward()
    (ee * w.cuda()).sum().backward()
    assert _rel(xg.grad.cpu(), x_ref.grad) < 3e-6
    g1 = xg.grad.clone(); xg.grad = None
    (spgan.get_edge_features(xg, k, idx=idx) * w.cuda()).sum().backward()
    assert torch.equal(g1, xg.grad)                                               # deterministic: slot lists, no float atomics
