"""CPU: the oracle restatement reproduces the REAL reference's outputs (golden
vectors written by tools/gen_golden.py from /root/reference)."""
import numpy as np
import pytest
import torch

from oracle import where2comm_oracle as orc
from tests.helpers import assert_close, case_from_fixture, load_fixture, sample

# the oracle runs the same ATen CPU ops as the reference, so agreement is to rounding of the
# thread-count-dependent reductions only
RTOL, ATOL = 1e-5, 1e-5


@pytest.mark.parametrize("name", ["w2c_small_n3", "w2c_small_n1", "w2c_full_n4", "w2c_full_n8"])
def test_oracle_matches_reference_golden(name):
    fx = load_fixture(name)
    hy, args, sd, dd, voxd, types = case_from_fixture(fx)
    s, bs = int(fx["sample_stride"]), int(fx["big_stride"])
    tr = {}
    with torch.no_grad():
        out = orc.where2com_forward(dd, sd, args, trace=tr)
    for k in ("psm", "rm", "obj"):
        assert list(out[k].shape) == list(fx[k + "_shape"])
        assert_close(sample(out[k], s), fx[k], RTOL, ATOL, k)
        assert abs(out[k].double().sum().item() - float(fx[k + "_sum"])) <= 1e-4 * float(fx[k + "_abssum"])
    assert out["comm_rate"] == int(fx["comm_rate"])
    assert abs(float(out["com"]) - float(fx["com"])) < 1e-6
    assert_close(sample(tr["psm_single"], s), fx["psm_single"], RTOL, ATOL, "psm_single")
    assert np.array_equal(sample(tr["comm_mask"], s), fx["comm_mask"])
    assert_close(sample(tr["comm_map"], s), fx["comm_map"], RTOL, 1e-7, "comm_map")
    for i in range(3):
        assert_close(sample(tr[f"fused{i}"][0], s), fx[f"fused{i}"], RTOL, ATOL, f"fused{i}")
        assert_close(sample(tr[f"block{i}"], bs), fx[f"block{i}"], RTOL, ATOL, f"block{i}")
    assert_close(sample(tr["spatial_features_2d"], bs), fx["spatial_features_2d"], RTOL, ATOL, "sf2d")
    assert_close(sample(tr["shrink"], bs), fx["shrink"], RTOL, ATOL, "shrink")
    assert_close(sample(tr["fused_shrink"], bs), fx["fused_shrink"], RTOL, ATOL, "fused_shrink")


def test_voxelizer_vectorised_equals_sequential():
    from airv2x_perception_amd import synth
    from oracle import voxelize_oracle as vox
    rng = [-6.4, -3.2, -3.0, 6.4, 3.2, 1.0]
    pts = synth.clustered_cloud(0, 4000, rng)
    pts[::50, 0] += 100.0  # some out-of-range points
    for mp, mv in ((32, 70000), (4, 70000), (32, 50), (3, 17)):
        a = vox.points_to_voxels(pts, rng, [0.4, 0.4, 4.0], mp, mv)
        b = vox.points_to_voxels_sequential(pts, rng, [0.4, 0.4, 4.0], mp, mv)
        for x, y in zip(a, b):
            assert x.dtype == y.dtype and np.array_equal(x, y)
    e = vox.points_to_voxels(np.zeros((0, 4), np.float32), rng, [0.4, 0.4, 4.0])
    assert e[0].shape == (0, 32, 4) and e[1].shape == (0, 3) and e[2].shape == (0,)
