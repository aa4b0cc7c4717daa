"""V2VNet (models/airv2x_v2vnet.py + v2vnet_modules/v2v_fuse.py): oracle vs the reference's golden vectors
(CPU) and the MI355X drop-in module vs the same vectors (GPU).  The goldens come from the reference's own classes
(tools/gen_golden.py `v2vnet`; see there for the base-class note)."""
import numpy as np
import pytest
import torch

from airv2x_perception_amd import synth
from oracle import v2vnet_oracle as v2v
from oracle import voxelize_oracle as vox
from tests.helpers import assert_close, load_fixture


def _case(fx):
    rng = [float(v) for v in fx["lidar_range"]]
    types = [str(t) for t in fx["types"]]
    hy = synth.default_hypes_v2vnet(rng, agg=str(fx["agg"]))
    args = hy["model"]["args"]
    if "compression" in fx and int(fx["compression"]):      # NaiveCompressor behind the shrink header (airv2x_v2vnet.py:42-44, 180-181)
        args["modality_fusion"]["compression"] = args["compression"] = int(fx["compression"])
    spec = synth.v2vnet_param_spec(args)
    assert [k for k, _, _ in spec] == [str(k) for k in fx["spec_keys"]]
    sd = synth.synthetic_state_dict(spec, seed=int(fx["seed"]))
    pp = hy["preprocess"]
    voxd = [vox.points_to_voxels(vox.mask_points_by_range(synth.synthetic_cloud(i, int(fx["n_points"]), rng), rng), rng,
                                 pp["args"]["voxel_size"]) for i in range(len(types))]
    for i, v in enumerate(voxd):
        assert np.array_equal(v[1], fx[f"vox_coords_{i}"]) and np.array_equal(v[2], fx[f"vox_num_{i}"])
    dd = synth.build_data_dict(voxd, types, max_cav_num=args["max_cav_num"])
    dd["img_pairwise_t_matrix_collab"] = synth.v2vnet_pairwise(len(types), args["max_cav_num"])
    return hy, args, sd, dd


@pytest.mark.parametrize("name", ["v2vnet_small_n3", "v2vnet_small_n2_max", "v2vnet_small_n2_c2"])
def test_oracle_matches_reference_golden(name):
    fx = load_fixture(name)
    hy, args, sd, dd = _case(fx)
    tr = {}
    with torch.no_grad():
        out = v2v.v2vnet_forward(dd, sd, args, trace=tr)
    for k in ("psm", "rm", "obj"):
        assert_close(out[k], fx[k], 1e-5, 1e-5, k)
    bs = int(fx["big_stride"])
    assert_close(tr["fused"][..., ::bs, ::bs], fx["fused"], 1e-5, 1e-5, "fused")
    # comm_rate counts the non-zeros of the node maps of every round (v2v_fuse.py:139); after the first ConvGRU update a map
    # is sigmoid * tanh, zero only where fp32 happens to round to it -> equal up to a handful of elements across machines
    assert abs(float(out["comm_rate"]) - float(fx["comm_rate"])) <= 1e-5 * float(fx["comm_rate"])
    # state_dict contract: trunk + msg_cnn / ConvGRU cell / mlp + heads
    assert sd["fusion_net.conv_gru.cell_list.0.conv_gates.weight"].shape == (512, 768, 3, 3)
    assert sd["fusion_net.msg_cnn.weight"].shape == (256, 512, 3, 3) and sd["fusion_net.mlp.weight"].shape == (256, 256)


def test_zero_hidden_state_reduces_the_conv_gru_to_two_half_convolutions():
    """The identity the device path relies on: with h = 0 the cell is update * tanh(candidate) on the x-channels only."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(3)
    c = 8
    sd = {"p.conv_gates.weight": torch.randn(2 * c, 3 * c, 3, 3, generator=g) * 0.1, "p.conv_gates.bias": torch.randn(2 * c, generator=g),
          "p.conv_can.weight": torch.randn(c, 3 * c, 3, 3, generator=g) * 0.1, "p.conv_can.bias": torch.randn(c, generator=g)}
    x = torch.randn(1, 2 * c, 9, 11, generator=g)
    full = v2v.conv_gru_step(x, sd, "p")
    u = torch.sigmoid(F.conv2d(x, sd["p.conv_gates.weight"][c:, :2 * c], sd["p.conv_gates.bias"][c:], padding=1))
    h = u * torch.tanh(F.conv2d(x, sd["p.conv_can.weight"][:, :2 * c], sd["p.conv_can.bias"], padding=1))
    assert torch.allclose(full, h, rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["v2vnet_small_n3", "v2vnet_small_n2_max", "v2vnet_full_n3", "v2vnet_small_n2_c2"])
def test_gpu_forward_matches_golden(name):
    from airv2x_perception_amd.opencood_iface import Airv2xV2VNet, create_model
    fx = load_fixture(name)
    hy, args, sd, dd = _case(fx)
    model = create_model(hy)
    assert isinstance(model, Airv2xV2VNet) and list(model.state_dict().keys()) == [str(k) for k in fx["spec_keys"]]
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda").eval()
    tr = {}
    out = model.engine().forward(dd, trace=tr, sync_comm_rate=True)
    torch.cuda.synchronize()
    bs, hs = int(fx["big_stride"]), int(fx["head_stride"])
    for it in range(args["v2vfusion"]["num_iteration"]):
        assert_close(tr[f"agg_it{it}"].cpu()[..., ::bs, ::bs], fx[f"agg_it{it}"], 3e-4, 3e-4, f"agg_it{it}")
        assert_close(tr[f"node0_it{it}"].cpu()[..., ::bs, ::bs], fx[f"node0_it{it}"], 3e-4, 3e-4, f"node0_it{it}")
    assert_close(tr["fused"].cpu()[..., ::bs, ::bs], fx["fused"], 3e-4, 3e-4, "fused")
    for k in ("psm", "rm", "obj"):
        assert_close(out[k].cpu()[..., ::hs, ::hs], fx[k], 3e-4, 3e-4, k)
        tot, ref = float(out[k].double().sum()), float(fx[k + "_sum"])
        assert abs(tot - ref) <= 1e-5 * max(1.0, float(out[k].double().abs().sum())), (k, tot, ref)
    assert abs(float(out["comm_rate"]) - float(fx["comm_rate"])) <= 1e-5 * float(fx["comm_rate"])   # see the oracle test
    assert set(out.keys()) == {"psm", "rm", "obj", "mask", "comm_rate"} and out["mask"] == 0
    again = model(dd)
    assert torch.equal(again["psm"], out["psm"]) and isinstance(again["comm_rate"], float)


@pytest.mark.gpu
def test_gpu_v2v_aggregate_kernel_matches_torch():
    """av2x_v2v_aggregate: (A_j + B) * roi_mask_j, mean / max over j, with the ROI mask = warp_affine_simple(ones)."""
    from ctypes import c_void_p

    from airv2x_perception_amd import _lib
    from oracle.when2com_oracle import warp_affine_simple
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    n, h, w, c = 3, 20, 36, 64
    a, b = torch.randn(n, h, w, c, generator=g), torch.randn(h, w, c, generator=g)
    theta = torch.tensor([[[1, 0, 0], [0, 1, 0]], [[0.96, -0.2, 0.15], [0.25, 0.95, -0.1]], [[1.0, 0.1, -0.4], [-0.08, 1.0, 0.3]]], dtype=torch.float32)
    mask = warp_affine_simple(torch.ones(n, 1, h, w), theta, (h, w)).permute(0, 2, 3, 1)
    msg = (a + b) * mask
    P = lambda t: c_void_p(t.data_ptr())
    for op, ref in ((0, msg.mean(0)), (1, msg.max(0)[0])):
        out = torch.full((h, w, c), float("nan"), device="cuda")
        ad, bd, td = a.cuda(), b.cuda(), theta.cuda()
        _lib.check(lib.av2x_v2v_aggregate(P(ad), P(bd), P(td), n, h, w, c, op, P(out), c_void_p(torch.cuda.current_stream().cuda_stream)),
                   "av2x_v2v_aggregate")
        assert_close(out.cpu(), ref, 1e-5, 1e-5, f"v2v aggregate op {op}")
