"""GPU, whole chain a1 -> a20 on seeded frames: raw clouds -> av2x_prepare_points -> av2x_voxelize -> Airv2xWhere2com ->
av2x_postprocess -> av2x_eval_tp_fp / AP, against the same chain built from the oracles on the CPU.  The model stage is
compared within the fp32 tolerance; the post-process and the AP evaluation are discontinuous (score order, IoU
thresholds), so their parity is EXACT on identical inputs (the oracle post-process / evaluation of the device's own head
maps: same boxes, same order, same TP/FP lists, same AP@0.3/0.5/0.7) and the two complete chains must produce the same
boxes up to threshold-sitting decisions (>= 95 % of the oracle chain's boxes within 5 cm).  This is the synthetic-data
form of the north star's "AP within +-0.5 pt of the reference" (no dataset on the box)."""
import numpy as np
import pytest
import torch

from airv2x_perception_amd import synth
from oracle import eval_oracle as eo
from oracle import postprocess_oracle as po
from oracle import voxelize_oracle as vox
from oracle import where2comm_oracle as orc

pytestmark = pytest.mark.gpu
RNG = [-25.6, -12.8, -3.0, 25.6, 12.8, 1.0]
TYPES = ["vehicle", "rsu", "drone"]
THS = (0.3, 0.5, 0.7)


def _pose(i, frame):
    """agent i's sensor pose in the ego frame (yaw + translation): the `transformation_matrix` of proj_first."""
    if i == 0:
        return np.eye(4, dtype=np.float32)
    yaw, tx, ty = 0.2 * i + 0.05 * frame, 3.0 * i, -2.0 * i + frame
    T = np.eye(4, dtype=np.float32)
    T[:2, :2] = [[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]]
    T[:3, 3] = [tx, ty, 0.1 * i]
    return T


def test_points_to_ap_chain_matches_oracle_chain():
    from airv2x_perception_amd.opencood_iface import Airv2xWhere2com
    from airv2x_perception_amd.opencood_iface import eval_utils as ev
    from airv2x_perception_amd.opencood_iface.voxel_postprocessor import VoxelPostprocessor
    from airv2x_perception_amd.opencood_iface.voxelizer import prepare_points, voxelize_points
    hy = synth.default_hypes(RNG)
    args, pp = hy["model"]["args"], hy["preprocess"]
    sd = synth.synthetic_state_dict(synth.where2com_param_spec(args), seed=5)
    model = Airv2xWhere2com(args)
    model.load_state_dict(sd)
    model = model.to("cuda").eval()
    post = VoxelPostprocessor(hy["postprocess"], dataset="airv2x", train=False)
    anchors = post.generate_anchor_box()
    T = torch.eye(4)
    stat_gpu = {t: {"tp": [], "fp": [], "gt": 0, "score": []} for t in THS}
    stat_cpu = {t: {"tp": [], "fp": [], "gt": 0, "score": []} for t in THS}
    n_boxes = 0
    for frame in range(3):
        clouds = [synth.clustered_cloud(10 * frame + i, 1500, [-32, -18, -3.5, 32, 18, 1.5]) for i in range(len(TYPES))]
        perms = [np.random.default_rng(frame * 7 + i).permutation(c.shape[0]).astype(np.int32) for i, c in enumerate(clouds)]
        # ---- device chain
        voxd_gpu = []
        for i, c in enumerate(clouds):
            p = prepare_points(torch.from_numpy(c).cuda(), RNG, _pose(i, frame), mask_ego=True, perm=torch.from_numpy(perms[i]).cuda())
            voxd_gpu.append(voxelize_points(p, pp["cav_lidar_range"], pp["args"]["voxel_size"], 32, pp["args"]["max_voxel_test"]))
        dd_gpu = synth.build_data_dict_device(voxd_gpu, TYPES, "cuda", max_cav_num=args["max_cav_num"])
        out = model(dd_gpu)
        data = {"ego": {"transformation_matrix": T, "anchor_box": torch.from_numpy(np.array(anchors))}}
        corners, scores, labels, boxes3d = post.post_process_airv2x(data, {"ego": out})
        # ---- oracle chain
        voxd = []
        for i, c in enumerate(clouds):
            p = vox.prepare_points(c, RNG, _pose(i, frame), True, perms[i])
            voxd.append(vox.points_to_voxels(p, pp["cav_lidar_range"], pp["args"]["voxel_size"], 32, pp["args"]["max_voxel_test"]))
        for (a, b, c_), (d, e, f) in zip(voxd, voxd_gpu):   # bit-exact up to the model input
            assert np.array_equal(a, d.cpu().numpy()) and np.array_equal(b, e.cpu().numpy()) and np.array_equal(c_, f.cpu().numpy())
        dd = synth.build_data_dict(voxd, TYPES, max_cav_num=args["max_cav_num"])
        with torch.no_grad():
            ref = orc.where2com_forward(dd, sd, args)
        # (1) the model: head maps of the device chain vs the oracle chain, fp32 tolerance
        for k in ("psm", "rm", "obj"):
            np.testing.assert_allclose(out[k].cpu().numpy(), ref[k].numpy(), rtol=2e-4, atol=2e-4)
        # (2) the post-process is discontinuous (score order, IoU > 0.15), so its EXACT parity is checked on identical
        # inputs: the oracle post-process of the DEVICE's head maps must return the same boxes in the same order
        rc, rs, rl, rb = po.post_process(out["psm"].cpu(), out["rm"].cpu(), out["obj"].cpu(), torch.from_numpy(anchors), T,
                                         hy["postprocess"], hy["postprocess"]["anchor_args"]["cav_lidar_range"])
        assert (corners is None) == (rc is None)
        if rc is None:
            continue
        assert corners.shape == rc.shape, (corners.shape, rc.shape)        # the same boxes survive
        assert torch.equal(labels.cpu(), rl)
        np.testing.assert_allclose(scores.cpu().numpy(), rs.numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(corners.cpu().numpy(), rc.numpy(), rtol=1e-4, atol=2e-4)
        # (3) end to end against the oracle chain (its own head maps): up to decisions that sit on a threshold the same
        # boxes come out -- at least 95 % of the oracle's boxes have a device box within 5 cm
        oc = po.post_process(ref["psm"], ref["rm"], ref["obj"], torch.from_numpy(anchors), T, hy["postprocess"],
                             hy["postprocess"]["anchor_args"]["cav_lidar_range"])[0]
        if oc is not None:
            dist = torch.cdist(oc.mean(1)[:, :2], corners.cpu().mean(1)[:, :2])
            assert float((dist.min(1).values < 0.05).float().mean()) >= 0.95
        n_boxes += rc.shape[0]
        # ---- ground truth: some detections (jittered) + some unrelated boxes
        g = np.random.default_rng(100 + frame)
        k = max(1, rc.shape[0] // 2)
        gt = rc.numpy()[g.choice(rc.shape[0], k, replace=False)].copy()
        gt[:, :, :2] += g.normal(0, 0.25, (k, 1, 2)).astype(np.float32)
        far = rc.numpy()[:2].copy()
        far[:, :, 0] += 500.0
        gt = np.concatenate([gt, far])
        for t in THS:
            ev.caluclate_tp_fp(corners, scores, torch.from_numpy(gt), stat_gpu, t)
            eo.caluclate_tp_fp(rc.numpy(), rs.numpy(), gt, stat_cpu, t)
    assert n_boxes >= 10
    for t in THS:
        assert stat_gpu[t]["tp"] == stat_cpu[t]["tp"] and stat_gpu[t]["gt"] == stat_cpu[t]["gt"], t
        assert ev.calculate_ap(stat_gpu, t, False)[0] == eo.calculate_ap(stat_cpu, t, False)[0]
    assert 0.0 < ev.calculate_ap(stat_gpu, 0.3, False)[0] <= 1.0
