"""CPU: host-side logic of the mirror (weight packing, BN folding, frame layout, tile choice,
module parameter tree, loud failure without a GPU)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from airv2x_perception_amd import synth
from airv2x_perception_amd.opencood_iface import Airv2xWhere2com
from airv2x_perception_amd.opencood_iface.engine import Where2ComEngine, frame_layout
from airv2x_perception_amd.opencood_iface.packing import fold_bn, pack_conv_weight, pack_deconv_weight


def _gemm_from_packed(wp, cin):
    """[tap][cin/4][coutp][4] -> B[tap*cin + k][n] as the kernel consumes it."""
    taps, q, coutp, _ = wp.shape
    return wp.permute(0, 1, 3, 2).reshape(taps * cin, coutp)


def test_conv_packing_is_the_im2col_gemm():
    g = torch.Generator().manual_seed(0)
    w = torch.randn(30, 64, 3, 3, generator=g)
    x = torch.randn(1, 64, 6, 7, generator=g)
    wp, coutp = pack_conv_weight(w)
    assert coutp == 32 and wp.shape == (9, 16, 32, 4) and float(wp[:, :, 30:].abs().max()) == 0
    Bm = _gemm_from_packed(wp, 64)
    cols = F.unfold(x, 3, padding=1).view(64, 9, 42).permute(1, 0, 2).reshape(9 * 64, 42)  # [tap*cin+k][pixel]
    got = (cols.t() @ Bm)[:, :30].t().reshape(1, 30, 6, 7)
    assert torch.allclose(got, F.conv2d(x, w, padding=1), atol=1e-4)


@pytest.mark.parametrize("s", [1, 2, 4])
def test_deconv_packing_is_a_gemm_with_scatter(s):
    g = torch.Generator().manual_seed(s)
    w = torch.randn(64, 32, s, s, generator=g)
    x = torch.randn(1, 64, 3, 5, generator=g)
    wp, ncol = pack_deconv_weight(w)
    assert ncol == s * s * 32
    Bm = _gemm_from_packed(wp, 64)
    y = x.permute(0, 2, 3, 1).reshape(15, 64) @ Bm  # [pixel][ (i*s+j)*32 + co ]
    out = torch.zeros(1, 32, 3 * s, 5 * s)
    y = y.view(3, 5, s, s, 32)
    for i in range(s):
        for j in range(s):
            out[0, :, i::s, j::s] = y[:, :, i, j].permute(2, 0, 1)
    assert torch.allclose(out, F.conv_transpose2d(x, w, stride=s), atol=1e-4)


def test_fold_bn_matches_eval_batchnorm():
    g = torch.Generator().manual_seed(1)
    sd = {"p.weight": torch.rand(8, generator=g) + 0.5, "p.bias": torch.randn(8, generator=g),
          "p.running_mean": torch.randn(8, generator=g), "p.running_var": torch.rand(8, generator=g) + 0.1}
    x = torch.randn(4, 8, 3, 3, generator=g)
    sc, sh = fold_bn(sd, "p")
    ref = F.batch_norm(x, sd["p.running_mean"], sd["p.running_var"], sd["p.weight"], sd["p.bias"], False, 0.0, 1e-3)
    assert torch.allclose(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1), ref, atol=1e-5)


def test_frame_layout_follows_reference_repack_order():
    # B = 2: sample 0 has 2 veh + 1 drone, sample 1 has 1 veh + 2 rsu
    dd = {"vehicle": {"record_len": torch.tensor([2, 1]), "batch_idxs": [0, 1]},
          "rsu": {"record_len": torch.tensor([0, 2]), "batch_idxs": [1]},
          "drone": {"record_len": torch.tensor([1, 0]), "batch_idxs": [0]}}
    rl, slots = frame_layout(["vehicle", "rsu", "drone"], dd)
    assert rl == [3, 3]
    assert slots == {"vehicle": [0, 1, 3], "rsu": [4, 5], "drone": [2]}
    # a type that is not a collaborator, or has no agents, is skipped
    rl, slots = frame_layout(["vehicle"], dd)
    assert rl == [2, 1] and slots == {"vehicle": [0, 1, 2]}
    dd1 = synth.build_data_dict([(np.zeros((1, 32, 4), np.float32), np.zeros((1, 3), np.int32), np.ones(1, np.int32))] * 4,
                                ["vehicle", "vehicle", "rsu", "drone"])
    rl, slots = frame_layout(["vehicle", "rsu", "drone"], dd1)
    assert rl == [4] and slots == {"vehicle": [0, 1], "rsu": [2], "drone": [3]}


def test_tile_choice_keeps_the_chip_full():
    for m, coutp in [(140800, 64), (35200, 128), (8800, 256), (140800, 256), (35200, 32), (2200, 256), (8800, 2048)]:
        bm, bn = Where2ComEngine.pick_tile(m, coutp)
        assert coutp % bn == 0 and bm in (64, 128) and bn in (32, 64, 128)
        wgs = -(-m // bm) * (coutp // bn)
        assert wgs >= 512 or (bm, bn) in ((64, 64), (128, 32)), (m, coutp, bm, bn, wgs)


def test_module_tree_and_loud_failure_on_cpu():
    hy = synth.default_hypes([-12.8, -6.4, -3, 12.8, 6.4, 1])
    args = hy["model"]["args"]
    m = Airv2xWhere2com(args).eval()
    spec = synth.where2com_param_spec(args)
    assert [k for k, _, _ in spec] == list(m.state_dict().keys()) and len(spec) == 162
    m.load_state_dict(synth.synthetic_state_dict(spec, 0), strict=True)
    assert m.backbone.blocks[1][4].weight.shape == (128, 128, 3, 3)  # reference indexing style still works
    with pytest.raises(RuntimeError, match="no CPU path"):
        m({})
    with pytest.raises(RuntimeError):
        Where2ComEngine(args, "cpu")
    bad = synth.clone_hypes(hy)["model"]["args"]
    bad["task"] = "seg"
    with pytest.raises(NotImplementedError):
        Airv2xWhere2com(bad)


def test_synthetic_inputs_are_reproducible():
    a, b = synth.synthetic_cloud(2, 100), synth.synthetic_cloud(2, 100)
    assert np.array_equal(a, b) and a.dtype == np.float32
    assert synth.agent_types_for(6) == ["vehicle", "vehicle", "rsu", "drone", "vehicle", "vehicle"]
    idx, ts = synth.sort_types(synth.agent_types_for(6))
    assert ts == ["vehicle"] * 4 + ["rsu", "drone"] and idx == [0, 1, 4, 5, 2, 3]


def test_create_model_follows_the_reference_name_registry():
    """tools/train_utils.py:288-325: module = core_method, class = core_method without underscores, case-insensitive."""
    from airv2x_perception_amd.opencood_iface import Airv2xCoBEVT, Airv2xV2XVit, create_model
    rng = [-12.8, -6.4, -3, 12.8, 6.4, 1]
    hy = synth.default_hypes(rng)
    assert hy["model"]["core_method"] == "airv2x_where2com"
    assert isinstance(create_model(hy), Airv2xWhere2com)
    assert isinstance(create_model(synth.default_hypes_cobevt(rng)), Airv2xCoBEVT)
    assert isinstance(create_model(synth.default_hypes_v2xvit(rng, (2, 1, 1))), Airv2xV2XVit)
    with pytest.raises(ValueError):
        create_model({"model": {"core_method": "point_pillar_intermediate", "args": {}}})


def test_winograd_class_rules_are_functions_of_layer_map_and_mode_only():
    """engine.wino4_rule / wino_x3_rule decide which Winograd class a 3x3 layer runs in -- host logic, no GPU needed: never a function of the
    number of agents in the launch (a batch, an agent group and the sharded frame must pick what the single frame picks); latency mode, the
    throughput mode of FramePipeline, and the agent-sharded frame that keeps the latency-mode classes even with frames in flight."""
    from airv2x_perception_amd import _lib
    from airv2x_perception_amd.opencood_iface.engine import ConvLayer, Where2ComEngine
    from airv2x_perception_amd.opencood_iface.when2com_engine import When2comEngine
    mk = lambda cin, cout, ks=3, stride=1: ConvLayer(None, None, None, cin, cout, cout, ks, stride, 1 if ks == 3 else 0, 1, _lib.AV2X_CONV)
    lat = object.__new__(Where2ComEngine)
    thr = object.__new__(Where2ComEngine)
    thr.throughput_mode = True
    shd = object.__new__(Where2ComEngine)
    shd.throughput_mode, shd.sharded_frame = True, True
    w2c = object.__new__(When2comEngine)
    w2c.throughput_mode = True
    shrink, b1, b2, b0 = (mk(256, 256), 100, 352), (mk(128, 128), 50, 176), (mk(256, 256), 25, 88), (mk(64, 64), 100, 352)
    for n in range(1, 17):
        assert lat.wino4_rule(shrink[0], n, *shrink[1:]) and not lat.wino4_rule(b1[0], n, *b1[1:]) and not lat.wino4_rule(b2[0], n, *b2[1:])
        assert thr.wino4_rule(shrink[0], n, *shrink[1:]) and thr.wino4_rule(b1[0], n, *b1[1:]) and thr.wino4_rule(b2[0], n, *b2[1:])
        assert not thr.wino4_rule(b0[0], n, *b0[1:])                                  # the 64-channel layers stay on F(2x2) in both modes
        for e in (shd, w2c):                                                          # sharded frames / When2com: the latency-mode classes
            assert e.wino4_rule(shrink[0], n, *shrink[1:]) and not e.wino4_rule(b1[0], n, *b1[1:]) and not e.wino4_rule(b2[0], n, *b2[1:])
    assert Where2ComEngine.wino_x3_rule(b0[0]) and Where2ComEngine.wino_x3_rule(b1[0])
    assert not Where2ComEngine.wino_x3_rule(mk(32, 64)) and not Where2ComEngine.wino_x3_rule(mk(128, 128, stride=2))
