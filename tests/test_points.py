"""Row a1: point preparation (shuffle -> ego mask -> projection -> range crop); bit-exact everywhere."""
import numpy as np
import pytest
import torch

from oracle import voxelize_oracle as vox
from tests.helpers import load_fixture

CASES = (("full", True, True, True), ("noproj", False, True, False), ("rangeonly", False, False, False))


@pytest.mark.parametrize("tag,use_perm,ego,proj", CASES)
def test_oracle_matches_reference_golden(tag, use_perm, ego, proj):
    fx = load_fixture("points_small")
    out = vox.prepare_points(fx["points"], fx["lidar_range"], fx["T"] if proj else None, ego, fx["perm"] if use_perm else None)
    assert out.dtype == np.float32 and np.array_equal(out, fx[f"out_{tag}"])


def test_oracle_edge_cases():
    rng = [-10, -10, -3, 10, 10, 1]
    p = np.array([[-10, 0, 0, 1], [10, 0, 0, 1], [0, 0, -3, 1], [0, 0, 1, 1], [9.999999, 0, 0, 1],    # faces are excluded
                  [-1.95, 0, 0, 1], [2.95, 1.1, 0, 1], [2.9500003, 0, 0, 1]], np.float32)             # ego box is closed
    out = vox.prepare_points(p, rng, None, True)
    assert np.array_equal(out, p[[4, 7]])
    assert vox.prepare_points(np.zeros((0, 4), np.float32), rng).shape == (0, 4)
    # identity projection leaves the coordinates bit-identical
    q = np.random.default_rng(0).uniform(-9, 9, (100, 4)).astype(np.float32)
    assert np.array_equal(vox.prepare_points(q, rng, np.eye(4), False), vox.prepare_points(q, rng, None, False))


@pytest.mark.gpu
@pytest.mark.parametrize("tag,use_perm,ego,proj", CASES)
def test_gpu_prepare_points_bit_exact(tag, use_perm, ego, proj):
    from airv2x_perception_amd.opencood_iface.voxelizer import prepare_points
    fx = load_fixture("points_small")
    pts = torch.from_numpy(fx["points"]).cuda()
    out = prepare_points(pts, fx["lidar_range"].tolist(), fx["T"] if proj else None, mask_ego=ego,
                         perm=torch.from_numpy(fx["perm"]).cuda() if use_perm else None)
    assert np.array_equal(out.cpu().numpy(), fx[f"out_{tag}"])


@pytest.mark.gpu
def test_gpu_prepare_points_large_and_empty():
    from airv2x_perception_amd.opencood_iface.voxelizer import prepare_points
    rng = np.random.default_rng(5)
    P = 300_000   # several passes of the single-workgroup scan
    pts = np.empty((P, 4), np.float32)
    pts[:, 0], pts[:, 1] = rng.uniform(-200, 200, P), rng.uniform(-70, 70, P)
    pts[:, 2], pts[:, 3] = rng.uniform(-5, 3, P), rng.uniform(0, 1, P)
    T = np.eye(4, dtype=np.float32)
    T[:2, :2] = [[np.cos(1.1), -np.sin(1.1)], [np.sin(1.1), np.cos(1.1)]]
    T[:3, 3] = [-20.5, 7.25, 0.1]
    r = [-140.8, -40, -3, 140.8, 40, 1]
    out = prepare_points(torch.from_numpy(pts).cuda(), r, T, mask_ego=True)
    assert np.array_equal(out.cpu().numpy(), vox.prepare_points(pts, r, T, True))
    assert prepare_points(torch.zeros((0, 4), device="cuda"), r, T).shape == (0, 4)
    far = torch.full((10, 4), 1e6, device="cuda")
    assert prepare_points(far, r, None).shape == (0, 4)
