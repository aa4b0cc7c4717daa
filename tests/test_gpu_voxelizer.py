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
