"""CPU: the train-mode oracles of the transformer heads (oracle/cobevt_oracle.py, oracle/v2xvit_oracle.py under ``train_mode()`` +
loss_oracle.pp_loss + torch autograd) reproduce one training step of the REFERENCE's own Airv2xCoBEVT / Airv2xV2XVit
(tests/golden/train_cobevt_small_*.npz, train_v2xvit_small_*.npz; tools/gen_golden.py train_cobevt_golden / train_v2xvit_golden): heads,
losses, the gradient of every parameter and every BatchNorm buffer after the step.  (The GPU tests hold the device step to the same
fixtures: tests/test_gpu_train_cobevt.py, test_gpu_train_v2xvit.py.)"""
import numpy as np
import pytest
import torch

from airv2x_perception_amd import synth
from oracle import cobevt_oracle as cob
from oracle import loss_oracle as lo
from oracle import v2xvit_oracle as vit
from oracle import voxelize_oracle as vox
from oracle import where2comm_oracle as orc
from tests.helpers import load_fixture


def _frame(fx, hy):
    rng = [float(v) for v in fx["lidar_range"]]
    types = [str(t) for t in fx["types"]]
    pp = hy["preprocess"]
    voxd = [vox.points_to_voxels(vox.mask_points_by_range(synth.synthetic_cloud(i, int(fx["n_points"]), rng), pp["cav_lidar_range"]),
                                 pp["cav_lidar_range"], pp["args"]["voxel_size"], pp["args"]["max_points_per_voxel"], pp["args"]["max_voxel_train"])
            for i in range(len(types))]
    args = hy["model"]["args"]
    dd = synth.build_data_dict(voxd, types, max_cav_num=args["max_cav_num"])
    H, W = fx["psm"].shape[-2:]
    lc = synth.loss_case(int(fx["seed"]) + 100, B=1, H=H, W=W, A=args["anchor_number"], C=args["num_class"], pos_frac=float(fx["pos_frac"]))
    return dd, {k: torch.from_numpy(lc[k]) for k in ("targets", "pos_equal_one", "neg_equal_one", "class_ids")}


def _check(fx, forward, args, sd, dd, tgt):
    names = set(str(k) for k in fx["grad_keys"])
    sd = {k: v.clone() for k, v in sd.items()}
    for k, v in sd.items():
        if k in names:
            v.requires_grad_(True)
    with orc.train_mode():
        o = forward(dd, sd, args)
    losses = lo.pp_loss(o["psm"], o["rm"], o["obj"], tgt["targets"], tgt["pos_equal_one"], tgt["class_ids"], 7, 1.0, 2.0)
    losses[0].backward()
    for k in ("psm", "rm", "obj"):
        assert np.abs(o[k].detach().numpy() - fx[k]).max() <= 1e-5 * max(1.0, np.abs(fx[k]).max()), k
    for i in range(3):
        assert abs(float(losses[i].detach()) - fx["losses"][i]) < 1e-4 * abs(fx["losses"][i]), i
    for k in names:
        g = sd[k].grad.reshape(-1)
        stride = max(1, g.numel() // 4096)
        gmax = fx["gsum:" + k][2]
        assert np.abs(g[::stride].numpy() - fx["g:" + k]).max() <= 2e-4 * gmax + 1e-9, k
    for k in fx.files:
        if k.startswith("b:"):
            ref = fx[k]
            assert np.abs(sd[k[2:]].detach().numpy().astype(np.float64) - ref.astype(np.float64)).max() <= 1e-5 * max(1.0, np.abs(ref).max()), k


@pytest.mark.parametrize("name", ["train_cobevt_small_n3", "train_cobevt_small_n2"])
def test_cobevt_train_oracle_reproduces_the_reference_step(name):
    fx = load_fixture(name)
    hy = synth.default_hypes_cobevt([float(v) for v in fx["lidar_range"]], tuple(int(v) for v in fx["max_cav"]))
    hy["model"]["args"]["fax_fusion"]["drop_out"] = 0.0
    args = hy["model"]["args"]
    sd = synth.synthetic_state_dict(synth.cobevt_param_spec(args), seed=int(fx["seed"]))
    dd, tgt = _frame(fx, hy)
    _check(fx, cob.cobevt_forward, args, sd, dd, tgt)


@pytest.mark.parametrize("name", ["train_v2xvit_small_n3", "train_v2xvit_small_n2"])
def test_v2xvit_train_oracle_reproduces_the_reference_step(name):
    fx = load_fixture(name)
    hy = synth.default_hypes_v2xvit([float(v) for v in fx["lidar_range"]], tuple(int(v) for v in fx["max_cav"]))
    e = hy["model"]["args"]["transformer"]["encoder"]
    e["cav_att_config"]["dropout"] = e["pwindow_att_config"]["dropout"] = e["feed_forward"]["dropout"] = 0.0
    args = hy["model"]["args"]
    sd = synth.synthetic_state_dict(synth.v2xvit_param_spec(args), seed=int(fx["seed"]))
    dd, tgt = _frame(fx, hy)
    dd["spatial_correction_matrix"] = torch.from_numpy(fx["spatial_correction_matrix"])
    dd["prior_encoding"] = torch.from_numpy(fx["prior_encoding"])
    _check(fx, vit.v2xvit_forward, args, sd, dd, tgt)


def test_model_classes_are_trainable_modules():
    """.train() no longer raises at construction level; parameters are trainable; backbone_fix leaves only the fusion net (CPU: no forward)."""
    from airv2x_perception_amd.opencood_iface import Airv2xCoBEVT, Airv2xV2XVit
    rng = [-12.8, -6.4, -3.0, 12.8, 6.4, 1.0]
    for cls, hy in ((Airv2xCoBEVT, synth.default_hypes_cobevt(rng)), (Airv2xV2XVit, synth.default_hypes_v2xvit(rng))):
        m = cls(hy["model"]["args"]).train()
        assert all(p.requires_grad for p in m.parameters())
        a2 = synth.clone_hypes(hy)["model"]["args"]
        a2["backbone_fix"] = True
        m2 = cls(a2)
        assert all(p.requires_grad == k.startswith("fusion_net.") for k, p in m2.named_parameters())
        with pytest.raises(RuntimeError, match="no CPU path"):
            m(synth.build_data_dict([], []))
