"""Shared test helpers: rebuild the seeded inputs a golden fixture was generated from."""
import os

import numpy as np
import torch

from airv2x_perception_amd import synth
from oracle import voxelize_oracle as vox

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def case_from_fixture(fx):
    """-> (hypes, args, state_dict, data_dict, voxelized, types)"""
    rng = [float(v) for v in fx["lidar_range"]]
    types = [str(t) for t in fx["types"]]
    hy = synth.default_hypes(rng)
    args = hy["model"]["args"]
    spec = synth.where2com_param_spec(args)
    assert [k for k, _, _ in spec] == [str(k) for k in fx["spec_keys"]]
    sd = synth.synthetic_state_dict(spec, seed=int(fx["seed"]))
    pp = hy["preprocess"]
    voxd = []
    for i, _ in enumerate(types):
        gen = synth.synthetic_cloud if str(fx["cloud"]) == "uniform" else synth.clustered_cloud
        p = vox.mask_points_by_range(gen(i, int(fx["n_points"]), rng), pp["cav_lidar_range"])
        v = vox.points_to_voxels(p, pp["cav_lidar_range"], pp["args"]["voxel_size"],
                                 pp["args"]["max_points_per_voxel"], pp["args"]["max_voxel_test"])
        assert np.array_equal(v[1], fx[f"vox_coords_{i}"]) and np.array_equal(v[2], fx[f"vox_num_{i}"])
        voxd.append(v)
    dd = synth.build_data_dict(voxd, types, max_cav_num=args["max_cav_num"])
    return hy, args, sd, dd, voxd, types


def sample(t, s):
    t = t.detach().float().cpu() if isinstance(t, torch.Tensor) else torch.as_tensor(t)
    return (t[..., ::s, ::s] if s > 1 else t).numpy()


def assert_close(a, b, rtol, atol, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    bad = err > tol
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.size} out of tol; max err {err.max():.3e} at |ref| {np.abs(b).flat[err.argmax()]:.3e}"


def train_case_from_fixture(fx):
    """-> (hypes, args, state_dict, data_dict, loss targets) of a ``train_*`` fixture (tools/gen_golden.py:train_golden)."""
    rng = [float(v) for v in fx["lidar_range"]]
    types = [str(t) for t in fx["types"]]
    hy = synth.default_hypes(rng)
    args = hy["model"]["args"]
    if "multi_scale" in fx:         # round 6: the single-scale / compressed variants (tools/gen_golden.py train_variants)
        args["where2com_fusion"]["multi_scale"] = bool(int(fx["multi_scale"]))
        args["modality_fusion"]["compression"] = int(fx["compression"])
        if int(fx["compression"]):
            args["compression"] = int(fx["compression"])
    sd = synth.synthetic_state_dict(synth.where2com_param_spec(args), seed=int(fx["seed"]))
    pp = hy["preprocess"]
    voxd = []
    for i, _ in enumerate(types):
        p = vox.mask_points_by_range(synth.synthetic_cloud(i, int(fx["n_points"]), rng), pp["cav_lidar_range"])
        voxd.append(vox.points_to_voxels(p, pp["cav_lidar_range"], pp["args"]["voxel_size"], pp["args"]["max_points_per_voxel"],
                                         pp["args"]["max_voxel_train"]))
    dd = synth.build_data_dict(voxd, types, max_cav_num=args["max_cav_num"])
    H, W = [int(v) for v in fx["psm_shape"][-2:]]
    lc = synth.loss_case(int(fx["seed"]) + 100, B=1, H=H, W=W, A=args["anchor_number"], C=args["num_class"], pos_frac=float(fx["pos_frac"]))
    tgt = {k: torch.from_numpy(lc[k]) for k in ("targets", "pos_equal_one", "neg_equal_one", "class_ids")}
    return hy, args, sd, dd, tgt
