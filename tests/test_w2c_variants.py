"""Airv2xWhere2com in the configurations no shipped AirV2X YAML selects, against fixtures of the REFERENCE's own model built with those keys
(tools/gen_golden.py w2c_variants): ``multi_scale: false`` (where2comm_fuse.py:264-286, airv2x_where2com.py:163-166), ``modality_fusion.
compression > 0`` with the top-level ``compression`` ratio (airv2x_where2com.py:50-52, 147-150; naive_compress.py:5-42) and ``fully: true``
(where2comm_fuse.py:222-223, 266-267).  CPU: the oracle reproduces the fixtures, the module's state_dict has the reference's keys.
GPU (-m gpu): the HIP path reproduces them (2e-4 relative + absolute on the heads, mask bit-exact away from the threshold), alone and through
the agent-sharded frame where that is built."""
import numpy as np
import pytest
import torch

from airv2x_perception_amd import synth
from oracle import voxelize_oracle as vox
from oracle import where2comm_oracle as orc
from tests.helpers import assert_close, load_fixture

NAMES = ["w2c_small_single_c2", "w2c_small_single", "w2c_small_multi_c4", "w2c_small_single_fully", "w2c_small_multi_fully",
         # BaseBEVBackbone variants inside the full model (base_bev_backbone.py:87-121): the extra deblock on the concatenated map, deblocks that
         # down-sample -- the shared map is at twice / half the first block's resolution and the mask takes where2comm_fuse.py:229-235
         "w2c_small_final_deblock", "w2c_small_down_deblock"]
RTOL, ATOL = 2e-4, 2e-4


def case(fx):
    rng = [float(v) for v in fx["lidar_range"]]
    types = [str(t) for t in fx["types"]]
    hy = synth.default_hypes(rng)
    a = hy["model"]["args"]
    a["where2com_fusion"]["multi_scale"] = bool(int(fx["multi_scale"]))
    a["where2com_fusion"]["fully"] = bool(int(fx["fully"]))
    a["modality_fusion"]["compression"] = int(fx["compression"])
    if int(fx["compression"]):
        a["compression"] = int(fx["compression"])
    if "upsample_strides" in fx:
        us = [float(v) for v in fx["upsample_strides"]]
        a["modality_fusion"]["base_bev_backbone"]["upsample_strides"] = [int(v) if v >= 1 else v for v in us]
        a["modality_fusion"]["base_bev_backbone"]["num_upsample_filter"] = [int(v) for v in fx["num_upsample_filter"]]
    spec = synth.where2com_param_spec(a)
    assert [k for k, _, _ in spec] == [str(k) for k in fx["spec_keys"]]
    sd = synth.synthetic_state_dict(spec, seed=int(fx["seed"]))
    pp = hy["preprocess"]
    voxd = [vox.points_to_voxels(vox.mask_points_by_range(synth.synthetic_cloud(i, int(fx["n_points"]), rng), pp["cav_lidar_range"]),
                                 pp["cav_lidar_range"], pp["args"]["voxel_size"], pp["args"]["max_points_per_voxel"], pp["args"]["max_voxel_test"])
            for i in range(len(types))]
    return hy, a, sd, synth.build_data_dict(voxd, types, max_cav_num=a["max_cav_num"]), voxd, types


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_the_reference_fixture(name):
    fx = load_fixture(name)
    hy, args, sd, dd, _, _ = case(fx)
    tr = {}
    with torch.no_grad():
        out = orc.where2com_forward(dd, sd, args, trace=tr)
    for k in ("psm", "rm", "obj"):
        assert_close(out[k].numpy(), fx[k], 1e-5, 1e-5, k)
    assert out["comm_rate"] == int(fx["comm_rate"]) and abs(float(out["com"]) - float(fx["com"])) < 1e-6
    if "comm_mask" in fx:
        assert np.array_equal(tr["comm_mask"].numpy(), fx["comm_mask"])
    n_comp = sum(1 for k in fx["spec_keys"] if str(k).startswith("naive_compressor."))
    assert n_comp == (21 if int(fx["compression"]) else 0)           # 3 x (conv weight, bias + 5 BatchNorm entries)
    if int(fx["compression"]):
        assert int(fx["message_shape"][1]) == 256 // int(fx["compression"])     # what a sharded deployment would put on the wire


def test_compression_needs_both_keys_as_in_the_reference():
    """airv2x_where2com.py:50-52 guards on modality_fusion.compression and reads args["compression"]: a KeyError with only the first."""
    hy = synth.default_hypes([-25.6, -12.8, -3.0, 25.6, 12.8, 1.0])
    hy["model"]["args"]["modality_fusion"]["compression"] = 2
    with pytest.raises(KeyError):
        synth.where2com_param_spec(hy["model"]["args"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_model_matches_the_reference_fixture(name):
    from airv2x_perception_amd.opencood_iface import Airv2xWhere2com
    fx = load_fixture(name)
    hy, args, sd, dd, voxd, types = case(fx)
    model = Airv2xWhere2com(args)
    assert list(model.state_dict().keys()) == [str(k) for k in fx["spec_keys"]]
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda").eval()
    tr = {}
    out = model.engine().forward(dd, trace=tr, sync_comm_rate=True)
    torch.cuda.synchronize()
    assert int(out["comm_rate"]) == int(fx["comm_rate"])
    s = int(fx["big_stride"])
    assert_close(tr["shrink"].cpu().numpy()[..., ::s, ::s], fx["shrink"], RTOL, ATOL, "shrink")
    assert_close(tr["psm_single"].cpu().numpy(), fx["psm_single"], RTOL, ATOL, "psm_single")
    flips = 0
    if "comm_mask" in fx:
        got, ref = tr["comm_mask"].cpu().numpy(), fx["comm_mask"]
        near = np.abs(fx["comm_map"].astype(np.float64) - 0.01) < 1e-6
        assert not ((got != ref) & ~near).any()
        flips = int((got != ref).sum())
        assert flips <= 4
    else:
        assert int(out["com"]) == 1
    if "compressed" in fx and not bool(int(fx["multi_scale"])):
        assert_close(tr["compressed"].cpu().numpy()[..., ::s, ::s], fx["compressed"], RTOL, ATOL, "compressor output")
    if flips == 0:
        if "fused" in fx:
            assert_close(tr["fused_2d"].cpu().numpy()[..., ::s, ::s], fx["fused"], RTOL, ATOL, "fused (single scale)")
        for k in ("psm", "rm", "obj"):
            assert_close(out[k].cpu().numpy(), fx[k], RTOL, ATOL, k)
        assert abs(float(out["com"]) - float(fx["com"])) < 1e-6
    else:       # a cell on the threshold: replay the device's mask through the oracle (as tests/test_gpu_forward.py)
        with torch.no_grad():
            ref = orc.where2com_forward(dd, sd, args, comm_mask=tr["comm_mask"].cpu())
        for k in ("psm", "rm", "obj"):
            assert_close(out[k].cpu().numpy(), ref[k].numpy(), RTOL, ATOL, k + " (device mask replayed)")
    # the module call and a B = 2 batch of the same frame give the same bits
    o1 = model(dd)
    assert torch.equal(o1["psm"], out["psm"])
    o2 = model(synth.merge_frames([dd, dd]))
    for k in ("psm", "rm", "obj"):
        assert torch.equal(o2[k][0], out[k][0]) and torch.equal(o2[k][1], out[k][0]), k


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["w2c_small_multi_fully", "w2c_small_single_c2", "w2c_small_single", "w2c_small_single_fully",
                                  "w2c_small_final_deblock", "w2c_small_down_deblock"])
def test_variants_through_the_agent_sharded_stages(name):
    """The variants in the agent-sharded frame (engine.shard_local_stage / shard_ego_stage): ``fully: true`` (no mask, the unmasked block
    outputs are the message), ``multi_scale: false`` (one level: the masked, decompressed 256-channel map), the backbone variants (shared
    map at twice / half the first block's resolution, mask resized on the sender).  Two emulated ranks equal the single-GPU forward bit
    for bit."""
    from airv2x_perception_amd.opencood_iface import Airv2xWhere2com
    from airv2x_perception_amd.opencood_iface.sharded import partition_agents
    fx = load_fixture(name)
    hy, args, sd, dd, voxd, types = case(fx)
    model = Airv2xWhere2com(args)
    model.load_state_dict(sd)
    model = model.to("cuda").eval()
    eng = model.engine()
    eng.stream_k = False
    ref = eng.forward(dd, sync_comm_rate=True)
    sends, stats, meta = [], None, None
    parts = partition_agents(len(types), 2)
    counts = [len(p) for p in parts]
    for r, mine in enumerate(parts):
        dd_local = synth.build_data_dict([voxd[i] for i in mine], [types[i] for i in mine], max_cav_num=args["max_cav_num"])
        send, st, meta = eng.shard_local_stage(dd_local, has_ego=(r == 0), n_pad=max(counts))
        sends.append(send.clone())
        stats = st.clone() if stats is None else stats + st
    out = eng.shard_ego_stage(torch.cat(sends), stats, dict(meta, counts=counts, n_pad=max(counts)), world=2, sync_comm_rate=True)
    for k in ("psm", "rm", "obj"):
        assert torch.equal(out[k], ref[k]), k
    assert out["comm_rate"] == ref["comm_rate"] == int(fx["comm_rate"])
    assert abs(float(out["com"]) - float(ref["com"])) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("n,hi,wi,ho,wo", [(2, 64, 128, 32, 64), (3, 16, 32, 32, 64), (1, 25, 88, 100, 352), (2, 7, 9, 13, 5), (1, 1, 1, 4, 3)])
def test_mask_resize_kernel_matches_interpolate(n, hi, wi, ho, wo):
    """av2x_mask_resize_bilinear = F.interpolate(mode="bilinear", align_corners=False) (where2comm_fuse.py:229-235): bit-exact for binary
    masks at power-of-two ratios (every weight and product is exact), 1e-6 for arbitrary values and sizes."""
    import torch.nn.functional as F
    from ctypes import c_void_p

    from airv2x_perception_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(n * 1000 + hi + wo)
    for binary in (True, False):
        m = (torch.rand(n, hi, wi, generator=g) > 0.4).float() if binary else torch.rand(n, hi, wi, generator=g)
        want = F.interpolate(m.unsqueeze(1), size=(ho, wo), mode="bilinear", align_corners=False)[:, 0]
        md = m.cuda()
        out = torch.full((n, ho, wo), float("nan"), device="cuda")
        _lib.check(lib.av2x_mask_resize_bilinear(c_void_p(md.data_ptr()), n, hi, wi, ho, wo, c_void_p(out.data_ptr()),
                                                 c_void_p(torch.cuda.current_stream().cuda_stream)), "av2x_mask_resize_bilinear")
        pow2 = (hi * 2 == ho or ho * 2 == hi) and (wi * 2 == wo or wo * 2 == wi)
        if binary and pow2:
            assert torch.equal(out.cpu(), want)
        else:
            assert float((out.cpu() - want).abs().max()) <= 1e-6
