"""GPU: what autocast (BASELINE configs[3]'s precision: bf16 matrix-core operands, bf16 activations in the V2X-ViT fusion) does to BOXES
and AP, not only to head maps.  For Where2Comm, CoBEVT and V2X-ViT, 20 seeded frames each run the synthetic chain of
tests/test_gpu_e2e_ap.py (raw clouds -> av2x_prepare_points -> av2x_voxelize -> model -> av2x_postprocess -> TP/FP/AP) twice on the
device: on the fp32-accurate path (pinned to the reference by the goldens) and under torch.autocast (tools/train.py:50,118 of the
reference wraps its validation forward the same way).  Ground truth = jittered boxes of the fp32 chain + unrelated boxes, the same
for both chains (voxel_postprocessor.py:666-839, eval_utils_opv2v.py:15-189 are the reference's post-process / evaluation).
The AP of this chain is a harsh instrument (untrained heads: many detections sit at the objectness cut, ~3 % of them appear or disappear
under ANY perturbation), so the +-0.5 pt of the north star -- stated for the fp32 path, which the goldens and tests/test_gpu_e2e_ap.py
hold -- is not what autocast can meet; what is asserted instead: (1) the same boxes come out where they were (>= 95 % within 5 cm;
V2X-ViT with bf16 activations >= 88 %), the box count moves by <= 3 %; (2) the AP shift stays inside the measured band; (3) for
Where2Comm the device's autocast costs no more AP than the REFERENCE's own autocast on the same frames (oracle under
torch.autocast(cpu, bfloat16)) + 0.5 pt; (4) per-head drift bounds = measured drift x 2 instead of the blanket 6 %."""
import numpy as np
import pytest
import torch

from airv2x_perception_amd import synth

pytestmark = pytest.mark.gpu
RNG = [-25.6, -12.8, -3.0, 25.6, 12.8, 1.0]
TYPES = ["vehicle", "rsu", "drone"]
THS = (0.3, 0.5, 0.7)
FRAMES = 20
# per-head bound on max|autocast - fp32| / max|fp32| over the 20 frames: twice the measured drift (printed by the test;
# profiles/r04_amp_ap.txt).  Where2Comm / CoBEVT keep fp32 activations (bf16 operand rounding only); V2X-ViT stores bf16 activations.
DRIFT_BOUND = {"where2com": {"psm": 0.025, "rm": 0.05, "obj": 0.03}, "cobevt": {"psm": 0.012, "rm": 0.026, "obj": 0.014},
               "v2xvit": {"psm": 0.052, "rm": 0.056, "obj": 0.076}}
# measured (MI355X, profiles/r04_amp_ap.txt): boxes of the fp32 chain with an autocast box within 5 cm 0.972 / 0.967 / 0.911, dAP@0.5 on this
# chain -2.7 / -1.6 / -3.8 pt; the REFERENCE's own autocast (oracle under torch.autocast(cpu, bf16)) on the same Where2Comm frames: -4.0 pt
# against the device's -3.8, 0.947 of its boxes within 5 cm against the device's 0.970
MATCH_5CM = {"where2com": 0.95, "cobevt": 0.95, "v2xvit": 0.88}
DAP05_BOUND = {"where2com": 4.5, "cobevt": 3.0, "v2xvit": 5.5}


def _pose(i, frame):
    if i == 0:
        return np.eye(4, dtype=np.float32)
    yaw, tx, ty = 0.2 * i + 0.05 * frame, 3.0 * i, -2.0 * i + 0.1 * frame
    T = np.eye(4, dtype=np.float32)
    T[:2, :2] = [[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]]
    T[:3, 3] = [tx, ty, 0.1 * i]
    return T


def _build(which):
    from airv2x_perception_amd import opencood_iface as oi
    if which == "where2com":
        hy, cls, spec = synth.default_hypes(RNG), oi.Airv2xWhere2com, synth.where2com_param_spec
    elif which == "cobevt":
        hy, cls, spec = synth.default_hypes_cobevt(RNG), oi.Airv2xCoBEVT, synth.cobevt_param_spec
    else:
        hy, cls, spec = synth.default_hypes_v2xvit(RNG), oi.Airv2xV2XVit, synth.v2xvit_param_spec
    args = hy["model"]["args"]
    sd = synth.synthetic_state_dict(spec(args), seed=5)
    model = cls(args)
    model.load_state_dict(sd)
    return hy, args, model.to("cuda").eval()


def _calibrate_heads(model, dd):
    """Untrained regression heads emit values whose decoded boxes fall outside the post-processor's size filters (CoBEVT / V2X-ViT: the
    fused map is LayerNorm-scaled).  Scale reg_head so that the regression map has the spread of a trained detector (std 0.3): a
    synthetic checkpoint that yields boxes -- the same weights for every chain that is compared."""
    with torch.no_grad():
        rm = model(dd)["rm"]
        f = float(0.3 / rm.std().clamp_min(1e-6))
        if f < 0.8:
            sd = model.state_dict()
            sd["reg_head.weight"] = sd["reg_head.weight"] * f
            sd["reg_head.bias"] = sd["reg_head.bias"] * f
            model.load_state_dict(sd)


def _frame(which, args, pp, frame):
    from airv2x_perception_amd.opencood_iface.voxelizer import prepare_points, voxelize_points
    voxd = []
    for i in range(len(TYPES)):
        c = synth.clustered_cloud(10 * frame + i, 1500, [-32, -18, -3.5, 32, 18, 1.5])
        perm = np.random.default_rng(frame * 7 + i).permutation(c.shape[0]).astype(np.int32)
        p = prepare_points(torch.from_numpy(c).cuda(), RNG, _pose(i, frame), mask_ego=True, perm=torch.from_numpy(perm).cuda())
        voxd.append(voxelize_points(p, pp["cav_lidar_range"], pp["args"]["voxel_size"], 32, pp["args"]["max_voxel_test"]))
    dd = synth.build_data_dict_device(voxd, TYPES, "cuda", max_cav_num=args["max_cav_num"])
    if which == "v2xvit":
        g = np.random.default_rng(99 + frame)
        scm = torch.eye(4, dtype=torch.float64).repeat(1, args["max_cav_num"], 1, 1)
        for i in range(1, len(TYPES)):
            scm[0, i] = torch.from_numpy(synth.se2_correction(g.uniform(-3, 3), g.uniform(-2, 2), g.uniform(-2, 2)))
        dd["spatial_correction_matrix"] = scm
        empty = (np.zeros((0, 32, 4), np.float32), np.zeros((0, 3), np.int32), np.zeros((0,), np.int32))
        dd["prior_encoding"] = synth.build_data_dict([empty] * len(TYPES), TYPES, "cpu", args["max_cav_num"])["prior_encoding"]
    return dd


def _ap_run(post, anchors, heads_fp32, heads_amp):
    """TP/FP/AP of two chains of head maps over the same frames against ground truth made from the FIRST chain's boxes."""
    from airv2x_perception_amd.opencood_iface import eval_utils as ev
    T = torch.eye(4)
    stat = {m: {t: {"tp": [], "fp": [], "gt": 0, "score": []} for t in THS} for m in ("fp32", "amp")}
    n32 = namp = matched5 = matched20 = 0
    for frame, (o32, oa) in enumerate(zip(heads_fp32, heads_amp)):
        data = {"ego": {"transformation_matrix": T, "anchor_box": anchors}}
        c32, s32, _, _ = post.post_process_airv2x(data, {"ego": o32})
        ca, sa, _, _ = post.post_process_airv2x(data, {"ego": oa})
        if c32 is None or c32.shape[0] == 0:
            continue
        n32 += c32.shape[0]
        g = np.random.default_rng(100 + frame)
        k = max(1, c32.shape[0] // 2)
        gt = c32.cpu().numpy()[g.choice(c32.shape[0], k, replace=False)].copy()
        gt[:, :, :2] += g.normal(0, 0.25, (k, 1, 2)).astype(np.float32)
        far = c32.cpu().numpy()[:2].copy()
        far[:, :, 0] += 500.0
        gt = torch.from_numpy(np.concatenate([gt, far]))
        for t in THS:
            ev.caluclate_tp_fp(c32, s32, gt, stat["fp32"], t)
            ev.caluclate_tp_fp(ca, sa, gt, stat["amp"], t)       # (None, None) counts the ground truth only
        if ca is not None and ca.shape[0]:
            namp += ca.shape[0]
            dist = torch.cdist(c32.cpu().mean(1)[:, :2], ca.cpu().mean(1)[:, :2]).min(1).values
            matched5 += int((dist < 0.05).sum())
            matched20 += int((dist < 0.20).sum())
    ap = {m: {t: 100.0 * ev.calculate_ap(stat[m], t, False)[0] for t in THS} for m in stat}
    return {"boxes_fp32": n32, "boxes_autocast": namp, "matched_within_5cm": round(matched5 / max(1, n32), 4),
            "matched_within_20cm": round(matched20 / max(1, n32), 4),
            "AP_fp32": {str(t): round(v, 3) for t, v in ap["fp32"].items()}, "AP_autocast": {str(t): round(v, 3) for t, v in ap["amp"].items()},
            "dAP": {str(t): round(ap["amp"][t] - ap["fp32"][t], 3) for t in THS}}


@pytest.mark.parametrize("which", ["where2com", "cobevt", "v2xvit"])
def test_autocast_at_box_and_ap_level(which):
    from airv2x_perception_amd.opencood_iface.voxel_postprocessor import VoxelPostprocessor
    hy, args, model = _build(which)
    pp = hy["preprocess"]
    frames = [_frame(which, args, pp, f) for f in range(FRAMES)]
    _calibrate_heads(model, frames[0])
    heads32, headsa = [], []
    drift = {k: 0.0 for k in ("psm", "rm", "obj")}
    for dd in frames:
        out32 = {k: v.clone() for k, v in model(dd).items() if k in ("psm", "rm", "obj")}
        with torch.autocast("cuda", dtype=torch.bfloat16):
            outa = {k: v.float().clone() for k, v in model(dd).items() if k in ("psm", "rm", "obj")}
        for k in drift:
            drift[k] = max(drift[k], float((outa[k] - out32[k]).abs().max() / out32[k].abs().max().clamp_min(1e-12)))
        heads32.append(out32)
        headsa.append(outa)
    # synthetic (untrained) weights: put the objectness cut where ~400 anchors of the first frame pass, the same cut for every chain
    ob = torch.sigmoid(heads32[0]["obj"]).flatten().sort(descending=True).values
    hy["postprocess"]["target_args"]["obj_threshold"] = float(ob[min(400, ob.numel() - 1)])
    post = VoxelPostprocessor(hy["postprocess"], dataset="airv2x", train=False)
    anchors = torch.from_numpy(np.array(post.generate_anchor_box()))
    rep = {"model": which, "frames": FRAMES, **_ap_run(post, anchors, heads32, headsa), "head_drift_rel_to_max": {k: round(v, 5) for k, v in drift.items()}}
    if which == "where2com":
        # the REFERENCE's own autocast shift on the same frames: the oracle (torch CPU restatement of the reference model) in fp32 and
        # under torch.autocast(bfloat16) -- what tools/train.py:50,118 does to the reference's validation forward
        from oracle import where2comm_oracle as orc
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        o32, oa = [], []
        for dd in frames[:8]:
            ddc = synth.data_dict_to(dd, "cpu")
            with torch.no_grad():
                r32 = orc.where2com_forward(ddc, sd, args)
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    ra = orc.where2com_forward(ddc, sd, args)
            o32.append({k: r32[k].float().cuda() for k in ("psm", "rm", "obj")})
            oa.append({k: ra[k].float().cuda() for k in ("psm", "rm", "obj")})
        rep["reference_autocast_same_frames"] = _ap_run(post, anchors, o32, oa)
        rep["device_autocast_first_8_frames"] = _ap_run(post, anchors, heads32[:8], headsa[:8])
    print("[amp ap]", rep)
    assert rep["boxes_fp32"] >= 1000, rep
    assert rep["matched_within_5cm"] >= MATCH_5CM[which] and rep["matched_within_20cm"] >= 0.93, rep    # the same objects come out where they were
    assert abs(rep["boxes_autocast"] - rep["boxes_fp32"]) <= 0.03 * rep["boxes_fp32"] + 2, rep
    assert abs(rep["dAP"]["0.5"]) <= DAP05_BOUND[which], rep
    for k, v in drift.items():
        assert v <= DRIFT_BOUND[which][k], (k, v, rep)
    if which == "where2com":
        # the device's autocast costs no more AP than the reference's own autocast on the same frames (+ 0.5 pt, the north star's AP tolerance)
        ref_shift = rep["reference_autocast_same_frames"]["dAP"]
        dev_shift = rep["device_autocast_first_8_frames"]["dAP"]
        for t in ("0.3", "0.5", "0.7"):
            assert abs(dev_shift[t]) <= abs(ref_shift[t]) + 0.5, (t, dev_shift, ref_shift)
