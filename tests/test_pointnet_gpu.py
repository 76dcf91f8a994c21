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
