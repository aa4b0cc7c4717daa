"""GPU: the raw-cloud input (voxelizer.points_frame -> engine.encode_points: prepare + voxelize + PFN + scatter with the
pillar counts staying in HBM) gives bit-identical outputs to the reference's input contract (voxelize_frame -> exact-shape
(M,32,4) tensors -> model), for every model, incl. an empty cloud (dummy-point branch), a pose and a shuffle."""
import numpy as np
import pytest
import torch

from airv2x_perception_amd import synth

pytestmark = pytest.mark.gpu
RNG = [-25.6, -12.8, -3.0, 25.6, 12.8, 1.0]


def _clouds(types, empty=None):
    out = []
    for i, _ in enumerate(types):
        c = synth.synthetic_cloud(i, 3000, [RNG[0] - 3, RNG[1] - 3, -4.0, RNG[3] + 3, RNG[4] + 3, 2.0])   # some points outside the crop
        if empty == i:
            c = c[:0]
        out.append(torch.from_numpy(c).cuda())
    return out


def _build(which):
    from airv2x_perception_amd import opencood_iface as oi
    types = ["vehicle", "vehicle", "rsu", "drone"]
    extra = {}
    if which == "where2com":
        hy = synth.default_hypes(RNG); spec = synth.where2com_param_spec(hy["model"]["args"]); M = oi.Airv2xWhere2com
    elif which == "cobevt":
        hy = synth.default_hypes_cobevt(RNG); spec = synth.cobevt_param_spec(hy["model"]["args"]); M = oi.Airv2xCoBEVT
    elif which == "v2xvit":
        hy = synth.default_hypes_v2xvit(RNG); spec = synth.v2xvit_param_spec(hy["model"]["args"]); M = oi.Airv2xV2XVit
    else:
        hy = synth.default_hypes_when2com(RNG); spec = synth.when2com_param_spec(hy["model"]["args"]); M = oi.Airv2xWhen2com
    args = hy["model"]["args"]
    model = M(args)
    model.load_state_dict(synth.synthetic_state_dict(spec, seed=1))
    model = model.cuda().eval()
    model.engine().stream_k = False
    return hy, args, model, types


@pytest.mark.parametrize("which,empty", [("where2com", None), ("where2com", 2), ("cobevt", None), ("v2xvit", 3), ("when2com", None)])
def test_points_input_equals_voxel_input(which, empty):
    from airv2x_perception_amd.opencood_iface.voxelizer import points_frame, voxelize_frame
    hy, args, model, types = _build(which)
    pp = hy["preprocess"]
    clouds = _clouds(types, empty)
    g = torch.Generator().manual_seed(5)
    perms = [torch.randperm(c.shape[0], generator=g) if i == 1 else None for i, c in enumerate(clouds)]
    poses = [None, synth.se2_correction(5.0, 0.8, -0.4).astype(np.float32), None, synth.se2_correction(-3.0, -1.0, 0.5).astype(np.float32)]
    vv = voxelize_frame(clouds, pp["cav_lidar_range"], pp["args"]["voxel_size"], poses=poses, mask_ego=True, perms=perms,
                        max_points=pp["args"]["max_points_per_voxel"], max_voxels=pp["args"]["max_voxel_test"])
    dd = synth.build_data_dict_device(vv, types, "cuda", max_cav_num=args["max_cav_num"])
    meta = {}
    if which == "v2xvit":
        meta = {k: dd[k] for k in ("prior_encoding", "spatial_correction_matrix")}
    if which == "when2com":
        meta = {"img_pairwise_t_matrix_collab": synth.when2com_pairwise(len(types), args["max_cav_num"])}
        dd.update(meta)
    ref = model(dd)
    out = model(points_frame(clouds, types, pp, poses=poses, perms=perms, mask_ego=True, **meta))
    for k in ("psm", "rm", "obj"):
        assert torch.equal(out[k], ref[k]), (which, k)
    if "comm_rate" in ref:
        assert out["comm_rate"] == ref["comm_rate"]
    if empty is not None:
        assert vv[empty][0].shape[0] == 1 and int(vv[empty][2][0]) == 1     # the dummy point's pillar


def test_points_frame_order_and_types_are_checked():
    from airv2x_perception_amd.opencood_iface.voxelizer import points_frame
    hy, args, model, types = _build("where2com")
    clouds = _clouds(types)
    with pytest.raises(ValueError, match="frame order"):
        model(points_frame(clouds, ["rsu", "vehicle", "vehicle", "drone"], hy["preprocess"]))
    with pytest.raises(ValueError):
        model(points_frame(clouds[:2], types, hy["preprocess"]))


def test_points_input_in_the_frame_pipeline():
    from airv2x_perception_amd.opencood_iface.engine import FramePipeline
    from airv2x_perception_amd.opencood_iface.voxelizer import points_frame
    hy, args, model, types = _build("where2com")
    clouds = _clouds(types)
    dd = points_frame(clouds, types, hy["preprocess"])
    ref = {k: v.clone() for k, v in model(dd).items() if torch.is_tensor(v) and v.dim() > 0}
    pipe = FramePipeline(model.engine(), 3)
    outs = [pipe.submit(dd)[0] for _ in range(6)]
    pipe.drain()
    torch.cuda.synchronize()
    for o in outs[-3:]:
        for k in ("psm", "rm", "obj"):
            assert torch.equal(o[k], ref[k]), k
