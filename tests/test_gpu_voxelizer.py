"""GPU voxelizer vs the CPU oracle (bit-exact: indices, counts, and copied point floats)."""
import numpy as np
import pytest
import torch

from airv2x_perception_amd import synth
from oracle import voxelize_oracle as vox

pytestmark = pytest.mark.gpu


def _cmp(pts, rng, vs, mp, mv, range_filter=False):
    from airv2x_perception_amd.opencood_iface.voxelizer import voxelize_points
    v, c, n = voxelize_points(torch.from_numpy(pts).cuda(), rng, vs, mp, mv, range_filter=range_filter)
    ref_pts = vox.mask_points_by_range(pts, rng) if range_filter else pts
    rv, rc, rn = vox.points_to_voxels(ref_pts, rng, vs, mp, mv)
    assert np.array_equal(c.cpu().numpy(), rc), "voxel coordinates / order"
    assert np.array_equal(n.cpu().numpy(), rn), "points per voxel"
    assert np.array_equal(v.cpu().numpy(), rv), "voxel contents (bit-exact copies, zero padding)"
    return v.shape[0]


def test_uniform_default_grid():
    pts = synth.synthetic_cloud(0, 8192)
    m = _cmp(pts, synth.DEFAULT_RANGE, [0.4, 0.4, 4.0], 32, 70000, range_filter=True)
    assert 7000 < m < 8192


def test_dense_cloud_overflows_point_cap_and_boundary_points():
    pts = synth.clustered_cloud(1, 108000)
    r = synth.DEFAULT_RANGE
    pts[:7] = np.array([[r[0], 0, 0, 1], [r[3], 0, 0, 1], [0, r[1], 0, 1], [0, r[4], 0, 1], [0, 0, r[2], 1],
                        [0, 0, r[5], 1], [r[3] - 1e-4, r[4] - 1e-4, r[5] - 1e-4, 1]], np.float32)
    m = _cmp(pts, r, [0.4, 0.4, 4.0], 32, 70000)
    assert m > 15000
    _cmp(pts, r, [0.4, 0.4, 4.0], 32, 70000, range_filter=True)


def test_voxel_cap_and_small_point_cap():
    rng = [-6.4, -3.2, -3.0, 6.4, 3.2, 1.0]
    pts = synth.clustered_cloud(2, 5000, rng)
    pts[::37, 0] += 50.0
    assert _cmp(pts, rng, [0.4, 0.4, 4.0], 32, 50) == 50
    _cmp(pts, rng, [0.4, 0.4, 4.0], 3, 17)
    _cmp(pts, rng, [0.4, 0.4, 2.0], 5, 70000)  # nz = 2


def test_empty_cloud_uses_the_reference_dummy_points():
    from airv2x_perception_amd.opencood_iface.voxelizer import voxelize_points
    rng = [-140.8, -40.0, -150.0, 140.8, 40.0, -6.0]  # drone range contains the second dummy point
    v, c, n = voxelize_points(torch.zeros((0, 4), device="cuda"), rng, [0.4, 0.4, 144.0])
    assert v.shape == (1, 32, 4) and int(n[0]) == 1


def test_fused_prepare_voxelize_frame_equals_the_two_step_path():
    """voxelize_frame (av2x_prepare_voxelize, one host read-back per frame) == prepare_points + voxelize_points per agent
    == the oracle chain, bit for bit, including an agent whose cloud is empty after the crop."""
    from airv2x_perception_amd.opencood_iface.voxelizer import prepare_points, voxelize_frame, voxelize_points
    rng = [-25.6, -12.8, -3.0, 25.6, 12.8, 1.0]
    vs = [0.4, 0.4, 4.0]
    clouds = [synth.clustered_cloud(i, 3000, [-40, -20, -4, 40, 20, 2]) for i in range(3)]
    clouds.append(np.full((50, 4), 500.0, np.float32))                      # everything outside the range
    poses = [None, np.array([[0.96, -0.28, 0, 3.0], [0.28, 0.96, 0, -1.5], [0, 0, 1, 0.2], [0, 0, 0, 1]], np.float32),
             np.array([[1, 0, 0, -7.25], [0, 1, 0, 2.0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32), None]
    perms = [None, torch.from_numpy(np.random.default_rng(1).permutation(3000).astype(np.int32)), None, None]
    dev = [torch.from_numpy(c).cuda() for c in clouds]
    fused = voxelize_frame(dev, rng, vs, poses=poses, mask_ego=True, perms=perms)
    for i, c in enumerate(clouds):
        two = voxelize_points(prepare_points(dev[i], rng, poses[i], mask_ego=True, perm=perms[i]), rng, vs)
        p = vox.prepare_points(c, rng, poses[i], True, None if perms[i] is None else perms[i].numpy())
        if p.shape[0] == 0:
            assert fused[i][0].shape[0] == two[0].shape[0] and torch.equal(fused[i][0], two[0])
            continue
        ref = vox.points_to_voxels(p, rng, vs)
        for a, b, r in zip(fused[i], two, ref):
            assert torch.equal(a, b)
            assert np.array_equal(a.cpu().numpy(), r)
