"""SpVoxelPreprocessor mirror (sp_voxel_preprocessor.py): collate semantics on the CPU, preprocess on the GPU against
the voxelizer restatement (integer-exact coordinates / counts, exact copies of the point rows)."""
import numpy as np
import pytest
import torch

from airv2x_perception_amd import synth
from airv2x_perception_amd.opencood_iface.voxelizer import SpVoxelPreprocessor
from oracle import voxelize_oracle as vox

RNG = [-12.8, -6.4, -3.0, 12.8, 6.4, 1.0]


def _pre(numpy=False):
    pp = synth.default_hypes(RNG)["preprocess"]
    return SpVoxelPreprocessor(pp, train=False, numpy=numpy), pp


def test_constructor_and_collate_batch():
    pre, pp = _pre()
    assert pre.max_voxels == pp["args"]["max_voxel_test"] and list(pre.grid_size) == [64, 32, 1]
    assert SpVoxelPreprocessor(pp, train=True).max_voxels == pp["args"]["max_voxel_train"]
    vs = [vox.points_to_voxels(synth.synthetic_cloud(i, 300, RNG), RNG, [0.4, 0.4, 4.0]) for i in range(3)]
    as_dict = {"voxel_features": [v[0] for v in vs], "voxel_coords": [v[1] for v in vs], "voxel_num_points": [v[2] for v in vs]}
    out = pre.collate_batch(as_dict)
    # what the reference's collate computes (:142-175): concat + agent index column
    want_c = np.concatenate([np.hstack((np.full((v[1].shape[0], 1), i, dtype=v[1].dtype), v[1])) for i, v in enumerate(vs)])
    np.testing.assert_array_equal(out["voxel_coords"].numpy(), want_c)
    np.testing.assert_array_equal(out["voxel_features"].numpy(), np.concatenate([v[0] for v in vs]))
    np.testing.assert_array_equal(out["voxel_num_points"].numpy(), np.concatenate([v[2] for v in vs]))
    assert out["voxel_coords"].dtype == torch.from_numpy(vs[0][1]).dtype
    as_list = [{"voxel_features": v[0], "voxel_coords": v[1], "voxel_num_points": v[2]} for v in vs]
    out2 = pre.collate_batch(as_list)
    for k in out:
        assert torch.equal(out[k], out2[k])
    with pytest.raises(TypeError):
        pre.collate_batch(3)


@pytest.mark.gpu
def test_preprocess_matches_voxelizer_restatement_and_model_input_contract():
    pre, pp = _pre()
    clouds = [synth.synthetic_cloud(i, 2000, RNG) for i in range(2)] + [np.zeros((0, 4), np.float32)]
    outs = [pre.preprocess(c) for c in clouds]
    for c, o in zip(clouds, outs):
        if len(c) == 0:   # the reference's dummy points (:80-90)
            c = np.array([[0, 0, 0, 0], [-0.218277, -11.13425732, -80.05884552, 1.230595649e-38]], dtype=np.float32)
        f, k, n = vox.points_to_voxels(c, RNG, pp["args"]["voxel_size"], 32, pp["args"]["max_voxel_test"])
        np.testing.assert_array_equal(o["voxel_coords"].cpu().numpy(), k)
        np.testing.assert_array_equal(o["voxel_num_points"].cpu().numpy(), n)
        np.testing.assert_array_equal(o["voxel_features"].cpu().numpy(), f)
    batch = pre.collate_batch(outs)
    assert batch["voxel_features"].is_cuda and batch["voxel_coords"].shape[1] == 4
    assert batch["voxel_coords"][:, 0].unique().tolist() == [0, 1, 2]
    host = SpVoxelPreprocessor(pp, train=False, numpy=True).preprocess(clouds[0])
    assert isinstance(host["voxel_features"], np.ndarray)
    np.testing.assert_array_equal(host["voxel_coords"], outs[0]["voxel_coords"].cpu().numpy())
