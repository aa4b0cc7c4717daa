"""GPU: on-device post-process (av2x_postprocess through the VoxelPostprocessor mirror) vs the
reference golden vectors and the oracle.  Index bookkeeping (candidate order, keep flags, NMS picks,
labels) must be exact; box floats agree to fp32 rounding of exp/sin/cos (1e-5 relative)."""
import numpy as np
import pytest
import torch

from airv2x_perception_amd import synth
from oracle import postprocess_oracle as po
from tests.helpers import assert_close, load_fixture

pytestmark = pytest.mark.gpu


def _run(fx, psm=None, rm=None, obj=None):
    from airv2x_perception_amd.opencood_iface.voxel_postprocessor import VoxelPostprocessor
    hy = synth.default_hypes([float(v) for v in fx["lidar_range"]])
    post = VoxelPostprocessor(hy["postprocess"], dataset="airv2x", train=False)
    anchors = post.generate_anchor_box()
    assert np.array_equal(anchors, po.generate_anchor_box(hy["postprocess"]))
    T = torch.from_numpy(np.identity(4)).float()
    data = {"ego": {"transformation_matrix": T, "anchor_box": torch.from_numpy(np.array(anchors))}}
    g = lambda k, v: torch.from_numpy(fx[k] if v is None else v).cuda()
    outd = {"ego": {"psm": g("psm", psm), "rm": g("rm", rm), "obj": g("obj", obj)}}
    return post, post.post_process_airv2x(data, outd, return_counts=True), hy, anchors


@pytest.mark.parametrize("name", ["w2c_small_n3", "w2c_small_n1"])
def test_postprocess_matches_reference_golden(name):
    fx = load_fixture(name)
    post, (corners, scores, labels, boxes, counts, index), hy, anchors = _run(fx)
    assert counts[0] == fx["pp_cand_index"].shape[0]            # obj > 0.2 candidates
    assert counts[1] == int(fx["pp_cand_keep"].sum())           # after size / z filters
    assert counts[3] == fx["pp_nms_keep"].shape[0]              # NMS picks
    assert counts[4] == fx["pp_scores"].shape[0]                # inside the range
    # the anchor index of every returned box = reference's candidate -> kept -> picked -> in-range chain
    ref_idx = fx["pp_cand_index"][fx["pp_cand_keep"]][fx["pp_nms_keep"]]
    lo, hi = np.asarray(fx["lidar_range"][:2], np.float32), np.asarray(fx["lidar_range"][3:5], np.float32)
    c = fx["pp_nms_in_corners"][fx["pp_nms_keep"]]
    inr = np.all((c[:, :, :2] >= lo) & (c[:, :, :2] <= hi), axis=(1, 2))
    assert np.array_equal(index.cpu().numpy(), ref_idx[inr])
    assert np.array_equal(labels.cpu().numpy(), fx["pp_labels"]) and labels.dtype == torch.int64
    assert_close(scores.cpu(), fx["pp_scores"], 1e-6, 1e-7, "scores")
    assert_close(boxes.cpu(), fx["pp_boxes3d"], 1e-5, 1e-5, "boxes3d")
    assert_close(corners.cpu(), fx["pp_corners"], 1e-5, 1e-4, "corners")
    # descending score order (NMS pick order)
    s = scores.cpu().numpy()
    assert np.all(s[:-1] >= s[1:])


def test_postprocess_dense_overlaps_against_oracle():
    """Many overlapping boxes (small regression deltas -> the anchors themselves): stresses the NMS."""
    fx = load_fixture("w2c_small_n1")
    g = np.random.default_rng(3)
    rm = (g.standard_normal(fx["rm"].shape) * 0.05).astype(np.float32)
    obj = (g.standard_normal(fx["obj"].shape) * 2.0).astype(np.float32)
    post, (corners, scores, labels, boxes, counts, index), hy, anchors = _run(fx, rm=rm, obj=obj)
    pp = hy["postprocess"]
    st = {}
    ref = po.post_process(torch.from_numpy(fx["psm"]), torch.from_numpy(rm), torch.from_numpy(obj), torch.from_numpy(anchors),
                          torch.eye(4), pp, pp["anchor_args"]["cav_lidar_range"], stages=st)
    assert counts[0] == st["cand_scores"].numel() and counts[1] == int(st["cand_keep"].sum()) and counts[1] > 1000
    assert counts[2] == 1000                                     # top-1000 cut of nms_rotated
    assert counts[3] == st["nms_keep"].numel() and counts[4] == ref[1].numel()
    assert np.array_equal(labels.cpu().numpy(), ref[2].numpy())
    assert_close(corners.cpu(), ref[0], 1e-5, 1e-4, "corners")
    assert_close(scores.cpu(), ref[1], 1e-6, 1e-7, "scores")


@pytest.mark.parametrize("ties", [False, True])
def test_every_anchor_a_candidate_takes_the_preselect_path(ties):
    """An untrained model / a low threshold: all 70 400 anchors of the default grid pass obj > 0.2, far above the 4 096 the
    O(K^2) ranking handles directly -> pp_select's exact rank-1000 cut (bisection over the score bits + a scan for the
    candidates that tie with the cut).  ``ties``: scores quantised to 17 values, so the cut falls inside a huge tie group
    (ties -> higher position first, as a stable descending argsort).  Must equal the oracle's full sort."""
    hy = synth.default_hypes()
    from airv2x_perception_amd.opencood_iface.voxel_postprocessor import VoxelPostprocessor
    post = VoxelPostprocessor(hy["postprocess"], dataset="airv2x", train=False)
    anchors = post.generate_anchor_box()
    g = np.random.default_rng(11)
    psm = g.standard_normal((1, 14, 100, 352)).astype(np.float32)
    rm = (g.standard_normal((1, 14, 100, 352)) * 0.3).astype(np.float32)
    obj = (g.standard_normal((1, 2, 100, 352)) * 0.8 + 2.5).astype(np.float32)
    if ties:
        obj = (np.round(obj * 4) / 4).astype(np.float32)
    obj = np.maximum(obj, -1.0).astype(np.float32)           # sigmoid(-1) = 0.27 > 0.2: every anchor is a candidate
    data = {"ego": {"transformation_matrix": torch.eye(4), "anchor_box": torch.from_numpy(np.array(anchors))}}
    outd = {"ego": {"psm": torch.from_numpy(psm).cuda(), "rm": torch.from_numpy(rm).cuda(), "obj": torch.from_numpy(obj).cuda()}}
    corners, scores, labels, boxes, counts, index = post.post_process_airv2x(data, outd, return_counts=True)
    pp = hy["postprocess"]
    st = {}
    ref = po.post_process(torch.from_numpy(psm), torch.from_numpy(rm), torch.from_numpy(obj), torch.from_numpy(anchors),
                          torch.eye(4), pp, pp["anchor_args"]["cav_lidar_range"], stages=st)
    assert counts[0] == 70400 and counts[1] == int(st["cand_keep"].sum()) and counts[1] > 4096
    assert counts[2] == 1000 and counts[3] == st["nms_keep"].numel() and counts[4] == ref[1].numel()
    assert np.array_equal(labels.cpu().numpy(), ref[2].numpy())
    assert_close(scores.cpu(), ref[1], 1e-6, 1e-7, "scores")
    assert_close(corners.cpu(), ref[0], 1e-5, 1e-4, "corners")


def test_no_candidates():
    fx = load_fixture("w2c_small_n1")
    obj = np.full(fx["obj"].shape, -10.0, np.float32)
    post, res, hy, anchors = _run(fx, obj=obj)
    assert res[:4] == (None, None, None, None) and res[4][0] == 0


def test_pipelined_frames_with_postprocess_equal_sequential():
    """FramePipeline.submit(after=post.launch) keeps model + post-process of several frames in flight (one buffer set per
    slot, `finish` reads the counts one lap later); every frame's boxes must equal the sequential post_process_airv2x."""
    from airv2x_perception_amd.opencood_iface import Airv2xWhere2com
    from airv2x_perception_amd.opencood_iface.engine import FramePipeline
    from airv2x_perception_amd.opencood_iface.voxel_postprocessor import VoxelPostprocessor
    from tests.helpers import case_from_fixture
    fx = load_fixture("w2c_small_n3")
    hy, args, sd, dd, voxd, types = case_from_fixture(fx)
    model = Airv2xWhere2com(args)
    model.load_state_dict(sd)
    model = model.to("cuda").eval()
    eng = model.engine()
    eng.stream_k = False
    post = VoxelPostprocessor(hy["postprocess"], dataset="airv2x", train=False)
    data = {"ego": {"transformation_matrix": torch.eye(4), "anchor_box": torch.from_numpy(np.array(post.generate_anchor_box()))}}
    frames = [dd, synth.build_data_dict([voxd[1], voxd[0], voxd[2]], types, max_cav_num=args["max_cav_num"])]
    want = []
    for f in frames:
        o = model(f)
        want.append(post.post_process_airv2x(data, {"ego": o}, return_counts=True))
    assert want[0][4] != want[1][4] or not torch.equal(want[0][1], want[1][1])       # the two frames differ
    pipe = FramePipeline(eng, 3)
    pend, got = [], []
    for it in range(7):
        if len(pend) == 3:
            h, stream, which = pend.pop(0)
            stream.synchronize()
            got.append((which, post.finish(h, return_counts=True)))
        which = it % 2
        h, stream = pipe.submit(frames[which], after=lambda o, slot: post.launch(data, {"ego": o}, slot=slot))
        pend.append((h, stream, which))
    for h, stream, which in pend:
        stream.synchronize()
        got.append((which, post.finish(h, return_counts=True)))
    assert len(got) == 7
    for which, res in got:
        ref = want[which]
        assert res[4] == ref[4]                                                       # the five counters
        for a, b in zip(res[:4], ref[:4]):
            assert torch.equal(a, b)
        assert torch.equal(res[5], ref[5])                                            # anchor indices of the boxes
