"""OPV2V-style Where2comm (reference: models/where2comm_modules/where2comm_attn.py + where2comm.py's Communication).

tests/golden/w2c_attn.npz holds the outputs of the REAL reference modules (tools/gen_golden.py w2c_attn: reference
Where2comm + reference BaseBEVBackbone on seeded inputs and weights, with per-agent SE(2) motions in the ego's row of the
pairwise matrix).  CPU: the oracle restatement against those outputs.  GPU: the drop-in sub-module
(opencood_iface/where2comm_attn.py -> av2x_warp_fuse / av2x_comm_mask / av2x_count_nonzero_where through the C-ABI)
against the same outputs.

Tolerance (fp32): 2e-4 * max(1, max|ref|) on the fused maps (conv chains with a different summation order feed them);
the communication volume is an integer count and must be EQUAL -- the fixture records how far the smoothed confidence
stays from the threshold (>= 4e-6, three orders above the arithmetic noise of the mask kernel)."""
import os

import numpy as np
import pytest
import torch

from airv2x_perception_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "w2c_attn.npz")
CFG = synth.w2c_attn_configs()
H, W = 32, 48
MS_CASES = (("ms_atten", [3, 2], 41), ("ms_max", [3], 42), ("ms_atten_n5", [5], 43))
SS_CASES = (("ss_atten", [2, 2], 256, 51), ("ss_max", [3], 64, 52))


def _gauss_sd(cfg, seed):
    """The 'trained' smoothing filter of the fixture (tools/gen_golden.py w2c_attn_golden.gauss_sd): the constructor's
    gaussian scaled per tap + a small bias.  {} when the configuration does not smooth."""
    comm = cfg.get("communication", {})
    if "gaussian_smooth" not in comm:
        return {}
    k, s = comm["gaussian_smooth"]["k_size"], comm["gaussian_smooth"]["c_sigma"]
    c = k // 2
    gx, gy = np.mgrid[0 - c:k - c, 0 - c:k - c]
    g = torch.Tensor(1 / (2 * np.pi * s) * np.exp(-(np.square(gx) + np.square(gy)) / (2 * np.square(s)))).view(1, 1, k, k)
    return {"naive_communication.gaussian_filter.weight": g * torch.from_numpy(synth.seeded_uniform(seed, (1, 1, k, k), 0.8, 1.2)),
            "naive_communication.gaussian_filter.bias": torch.tensor([1e-4])}


def _ms_inputs(tag, rl, seed):
    n = sum(rl)
    return (torch.from_numpy(synth.w2c_attn_features(seed, n, 64, H, W)),
            torch.from_numpy(synth.w2c_attn_psm(seed + 1, n, H // 2, W // 2)), synth.w2c_attn_pairwise(rl))


def _ss_inputs(rl, ch, seed):
    n = sum(rl)
    return (torch.from_numpy(synth.w2c_attn_features(seed, n, ch, H // 2, W // 2, keep=0.6)),
            torch.from_numpy(synth.w2c_attn_psm(seed + 1, n, H // 2, W // 2)), synth.w2c_attn_pairwise(rl))


def _close(a, ref, rel=2e-4):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    assert a.shape == ref.shape, (a.shape, ref.shape)
    tol = rel * max(1.0, float(np.abs(ref).max()))
    err = float(np.abs(a - ref).max())
    assert err <= tol, f"max abs err {err:.3e} > {tol:.3e}"


# ---------------------------------------------------------------------------------------------- CPU: oracle vs reference
@pytest.mark.parametrize("tag,rl,seed", MS_CASES)
def test_oracle_multi_scale_matches_reference(tag, rl, seed):
    from oracle import where2comm_attn_oracle as wa
    g = np.load(GOLD)
    c = CFG[tag.replace("_n5", "")]
    bsd = {"backbone." + k: v for k, v in synth.synthetic_state_dict(synth.backbone_param_spec(CFG["backbone"], 64, ""), seed=31).items()}
    x, rm, pw = _ms_inputs(tag, rl, seed)
    with torch.no_grad():
        fused, vol = wa.where2comm_attn(x, rm, rl, pw, _gauss_sd(c, seed + 500), c, bsd, CFG["backbone"])
    _close(fused, g[f"{tag}_fused"], rel=2e-5)
    assert float(vol) == float(g[f"{tag}_vol"])


@pytest.mark.parametrize("tag,rl,ch,seed", SS_CASES)
def test_oracle_single_scale_matches_reference(tag, rl, ch, seed):
    from oracle import where2comm_attn_oracle as wa
    g = np.load(GOLD)
    x, rm, pw = _ss_inputs(rl, ch, seed)
    with torch.no_grad():
        fused, vol = wa.where2comm_attn(x, rm, rl, pw, _gauss_sd(CFG[tag], seed + 500), CFG[tag])
    _close(fused, g[f"{tag}_fused"], rel=2e-5)
    assert float(vol) == float(g[f"{tag}_vol"])


def test_oracle_fusion_operators_match_reference():
    from oracle import where2comm_attn_oracle as wa
    g = np.load(GOLD)
    xa = torch.from_numpy(synth.seeded_uniform(61, (3, 128, 6, 10)))
    _close(wa.atten_fusion(xa), g["atten_fusion"], rel=1e-6)
    np.testing.assert_array_equal(wa.max_fusion(xa).numpy(), g["max_fusion"])


def test_drop_in_contract_on_cpu():
    from airv2x_perception_amd.opencood_iface import where2comm_attn as wm
    m = wm.Where2comm(CFG["ms_atten"])
    assert list(m.state_dict().keys()) == ["naive_communication.gaussian_filter.weight", "naive_communication.gaussian_filter.bias"]
    ref_init = _gauss_sd({"communication": {"gaussian_smooth": {"k_size": 5, "c_sigma": 1.0}}}, 0)
    # constructor default = init_gaussian_filter's values (where2comm.py:28-45), bias zero
    k = m.state_dict()["naive_communication.gaussian_filter.weight"]
    g = ref_init["naive_communication.gaussian_filter.weight"] / torch.from_numpy(synth.seeded_uniform(0, (1, 1, 5, 5), 0.8, 1.2))
    np.testing.assert_allclose(k.numpy(), g.numpy(), rtol=1e-6)
    assert float(m.state_dict()["naive_communication.gaussian_filter.bias"]) == 0.0
    assert list(wm.Where2comm(CFG["ms_max"]).state_dict().keys()) == []
    assert len(m.fuse_modules) == 3 and isinstance(m.fuse_modules[0], wm.AttenFusion)
    assert isinstance(wm.Where2comm(CFG["ss_max"]).fuse_modules, wm.MaxFusion)
    with pytest.raises(NotImplementedError, match="Transformer"):
        wm.Where2comm(dict(CFG["ms_atten"], agg_operator={"mode": "Transformer", "n_head": 8, "with_spe": True, "with_scm": True}))
    # the host-side matrix preparation (where2comm_attn.py:293-307) against the oracle's, and the caller's tensor is kept
    from oracle.when2com_oracle import normalized_pairwise
    pw = synth.w2c_attn_pairwise([3, 2])
    keep = pw.clone()
    mine = wm.normalized_pairwise(pw, H, W, 0.4, 2)
    np.testing.assert_array_equal(mine, normalized_pairwise(pw, H, W, 0.4, 2).numpy())
    assert torch.equal(pw, keep)
    with pytest.raises(RuntimeError, match="no CPU path"):
        wm.Where2comm(CFG["ss_max"]).eval()(torch.zeros(1, 64, 4, 4), torch.zeros(1, 2, 4, 4), [1], torch.eye(4).view(1, 1, 1, 4, 4))


# ---------------------------------------------------------------------------------------------- GPU: drop-in vs reference
def _gpu_backbone():
    from airv2x_perception_amd.opencood_iface import submodules as sm
    bb = sm.BaseBEVBackbone(CFG["backbone"], 64)
    bb.load_state_dict(synth.synthetic_state_dict(synth.backbone_param_spec(CFG["backbone"], 64, ""), seed=31), strict=True)
    return bb.eval().cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("tag,rl,seed", MS_CASES)
def test_gpu_multi_scale_matches_reference(tag, rl, seed):
    from airv2x_perception_amd.opencood_iface import where2comm_attn as wm
    g = np.load(GOLD)
    c = CFG[tag.replace("_n5", "")]
    mod = wm.Where2comm(c)
    sd = _gauss_sd(c, seed + 500)
    mod.load_state_dict(sd, strict=True)
    mod = mod.eval().cuda()
    x, rm, pw = _ms_inputs(tag, rl, seed)
    keep = x.clone()
    xg = x.cuda()
    fused, vol, extra = mod(xg, rm.cuda(), torch.tensor(rl).cuda(), pw.cuda(), _gpu_backbone(), None)
    assert extra == {} and isinstance(vol, np.floating)
    assert float(vol) == float(g[f"{tag}_vol"]), (float(vol), float(g[f"{tag}_vol"]), float(g[f"{tag}_margin"]))
    _close(fused, g[f"{tag}_fused"])
    assert torch.equal(xg.cpu(), keep), "the caller's features must not be modified"
    # a second call on the same module (workspaces reused) gives the same bits
    fused2, vol2, _ = mod(xg, rm.cuda(), rl, pw.cuda(), _gpu_backbone(), None)
    assert float(vol2) == float(vol) and torch.equal(fused2, fused)


def _resnet_inputs():
    rl, seed = [3, 2], 45
    return (rl, seed, torch.from_numpy(synth.w2c_attn_features(seed, 5, 64, H, W)), torch.from_numpy(synth.w2c_attn_psm(seed + 1, 5, H // 2, W // 2)),
            synth.w2c_attn_pairwise(rl))


def test_oracle_resnet_backbone_variant_matches_reference():
    """where2comm_attn.py:312-314 with the reference's ResNetBEVBackbone (fixture: its own modules)."""
    from oracle import where2comm_attn_oracle as wa
    g = np.load(GOLD)
    rbc = CFG["resnet_backbone"]
    rsd = {"backbone." + k: v for k, v in synth.synthetic_state_dict(synth.resnet_backbone_param_spec(rbc, ""), seed=33).items()}
    rl, seed, x, rm, pw = _resnet_inputs()
    c = CFG["ms_atten"]
    with torch.no_grad():
        fused, vol = wa.where2comm_attn(x, rm, rl, pw, _gauss_sd(c, seed + 500), c, rsd, rbc, with_resnet=True)
        feats = wa.resnet_features(x, rsd, rbc)
    _close(fused, g["ms_resnet_fused"], rel=2e-5)
    assert float(vol) == float(g["ms_resnet_vol"])
    _close(feats[2][:, ::8], g["resnet_level2"], rel=2e-5)


@pytest.mark.gpu
def test_gpu_resnet_backbone_variant_matches_reference():
    from airv2x_perception_amd.opencood_iface import submodules as sm
    from airv2x_perception_amd.opencood_iface import where2comm_attn as wm
    g = np.load(GOLD)
    rbc = CFG["resnet_backbone"]
    spec = synth.resnet_backbone_param_spec(rbc, "")
    bb = sm.ResNetBEVBackbone(rbc, 64)
    assert list(bb.state_dict().keys()) == [k for k, _, _ in spec] and hasattr(bb, "resnet")
    bb.load_state_dict(synth.synthetic_state_dict(spec, seed=33), strict=True)
    bb = bb.eval().cuda()
    rl, seed, x, rm, pw = _resnet_inputs()
    c = CFG["ms_atten"]
    mod = wm.Where2comm(c)
    mod.load_state_dict(_gauss_sd(c, seed + 500), strict=True)
    mod = mod.eval().cuda()
    xg = x.cuda()
    fused, vol, _ = mod(xg, rm.cuda(), torch.tensor(rl), pw.cuda(), bb, None)
    assert float(vol) == float(g["ms_resnet_vol"])
    _close(fused, g["ms_resnet_fused"])
    # the backbone on its own: forward(data_dict) and the `resnet` callable other reference code uses
    d = bb({"spatial_features": xg})
    _close(d["spatial_features_2d"][:, ::2], g["resnet_backbone_2d"])
    feats = bb.resnet(xg)
    assert len(feats) == 3 and feats[2].shape == (5, 256, H // 8, W // 8)
    _close(feats[2][:, ::8], g["resnet_level2"])
    assert torch.equal(xg.cpu(), x)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,rl,ch,seed", SS_CASES)
def test_gpu_single_scale_matches_reference(tag, rl, ch, seed):
    from airv2x_perception_amd.opencood_iface import where2comm_attn as wm
    g = np.load(GOLD)
    mod = wm.Where2comm(CFG[tag])
    mod.load_state_dict(_gauss_sd(CFG[tag], seed + 500), strict=True)
    mod = mod.eval().cuda()
    x, rm, pw = _ss_inputs(rl, ch, seed)
    xg = x.cuda()
    fused, vol, _ = mod(xg, rm.cuda(), torch.tensor(rl), pw.cuda())
    _close(fused, g[f"{tag}_fused"], rel=2e-5)
    if "communication" in CFG[tag]:
        assert float(vol) == float(g[f"{tag}_vol"])
    else:
        assert isinstance(vol, torch.Tensor) and int(vol) == 0 and vol.device.type == "cuda"
    assert torch.equal(xg.cpu(), x)


@pytest.mark.gpu
def test_gpu_fusion_operators_match_reference():
    from airv2x_perception_amd.opencood_iface import where2comm_attn as wm
    g = np.load(GOLD)
    xa = torch.from_numpy(synth.seeded_uniform(61, (3, 128, 6, 10))).cuda()
    _close(wm.AttenFusion(128)(xa), g["atten_fusion"], rel=2e-6)
    np.testing.assert_array_equal(wm.MaxFusion()(xa).cpu().numpy(), g["max_fusion"])


@pytest.mark.gpu
def test_gpu_fused_warp_equals_warp_then_fuse():
    """The fused kernel against the two-step form on the device's own kernels (av2x_warp_affine_simple, pinned by the
    When2com fixtures, then av2x_pixel_attn_fuse / av2x_agent_max) at the full OPV2V map size, 5 agents: the taps use the
    same arithmetic, so MAX is bit-equal and ATTEN agrees to rounding of the online softmax (same order -> also equal)."""
    from ctypes import c_void_p
    from airv2x_perception_amd import _lib
    from airv2x_perception_amd.opencood_iface import where2comm_attn as wm
    from airv2x_perception_amd.opencood_iface.warp import warp_affine_simple
    lib = _lib.load()
    n, c, h, w = 5, 128, 50, 176
    x = torch.from_numpy(synth.w2c_attn_features(71, n, c, h, w, keep=0.5)).cuda()
    th = wm.normalized_pairwise(synth.w2c_attn_pairwise([n]), h, w, 0.4, 2)[0, 0, :n]
    xn = x.permute(0, 2, 3, 1).contiguous()
    warped = warp_affine_simple(x, torch.from_numpy(th), (h, w))
    ptrs = (c_void_p * n)(*[xn[j].data_ptr() for j in range(n)])
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    for mode, two_step in ((0, wm.AttenFusion(c)), (1, wm.MaxFusion())):
        out = torch.empty((h, w, c), device="cuda")
        _lib.check(lib.av2x_warp_fuse(ptrs, th.ctypes.data_as(c_void_p), n, h, w, c, mode, c_void_p(out.data_ptr()), st), "av2x_warp_fuse")
        ref = two_step(warped).permute(1, 2, 0)
        assert torch.equal(out, ref), (mode, float((out - ref).abs().max()))
    # argument checks fail loudly
    assert lib.av2x_warp_fuse(ptrs, th.ctypes.data_as(c_void_p), n, h, w, 96, 0, c_void_p(out.data_ptr()), st) != 0
    assert lib.av2x_warp_fuse(ptrs, th.ctypes.data_as(c_void_p), 33, h, w, c, 0, c_void_p(out.data_ptr()), st) != 0


@pytest.mark.gpu
def test_gpu_multi_scale_full_canvas_five_agents():
    """The 200 x 704 AirV2X canvas, 5 agents (BASELINE-sized maps: 100x352 / 50x176 / 25x88 levels): strided samples and the
    sums of the fused map against the reference, the communication volume equal."""
    from airv2x_perception_amd.opencood_iface import where2comm_attn as wm
    g = np.load(GOLD)
    c, rl, seed = CFG["ms_atten"], [5], 44
    mod = wm.Where2comm(c)
    mod.load_state_dict(_gauss_sd(c, seed + 500), strict=True)
    mod = mod.eval().cuda()
    x = torch.from_numpy(synth.w2c_attn_features(seed, 5, 64, 200, 704, keep=0.08)).cuda()
    rm = torch.from_numpy(synth.w2c_attn_psm(seed + 1, 5, 100, 352)).cuda()
    fused, vol, _ = mod(x, rm, torch.tensor(rl), synth.w2c_attn_pairwise(rl).cuda(), _gpu_backbone(), None)
    assert tuple(fused.shape) == (1, 96, 100, 352)
    # an integer count of non-zero cells behind a threshold on a smoothed map: a handful of cells may sit on the threshold
    assert abs(float(vol) - float(g["ms_atten_full_vol"])) <= 1e-5 * float(g["ms_atten_full_vol"])
    _close(fused[:, ::4, ::3, ::5], g["ms_atten_full_fused"])
    tot, ab = float(fused.double().sum()), float(fused.double().abs().sum())
    assert abs(tot - float(g["ms_atten_full_sum"])) <= 1e-5 * float(g["ms_atten_full_abs_sum"])
    assert abs(ab - float(g["ms_atten_full_abs_sum"])) <= 1e-5 * float(g["ms_atten_full_abs_sum"])


# ---------------------------------------------------------------------------------------------------- train mode
TRAIN_GOLD = os.path.join(os.path.dirname(__file__), "golden", "train_w2c_attn.npz")


@pytest.mark.gpu
@pytest.mark.parametrize("tag,single", [("ms_atten", False), ("ms_max", False), ("ss_atten", True), ("ms_resnet", False), ("resnet_alone", False),
                                        ("variant_alone", False), ("resnet_variant_ms", False)])
def test_gpu_train_mode_forward_and_gradients_match_the_reference(tag, single):
    """`Where2comm.train()` (where2comm_attn.py:275-404 under autograd, the reference's BaseBEVBackbone with batch statistics):
    fused map, dL/dx, every backbone parameter's gradient and the BatchNorm buffers of one step against
    tests/golden/train_w2c_attn.npz (tools/gen_golden.py train_w2c_attn: the reference's own modules; gradients from its float64 pass)."""
    from airv2x_perception_amd.opencood_iface import where2comm_attn as wm
    from airv2x_perception_amd.opencood_iface.submodules import BaseBEVBackbone, ResNetBEVBackbone
    g = np.load(TRAIN_GOLD)
    resnet, alone = tag in ("ms_resnet", "resnet_alone", "resnet_variant_ms"), tag in ("resnet_alone", "variant_alone")
    variant = tag in ("variant_alone", "resnet_variant_ms")
    c = CFG["ms_atten2" if tag == "resnet_variant_ms" else "ms_atten" if resnet or variant else tag]
    rl, seed = [int(v) for v in g[f"{tag}_rl"]], int(g[f"{tag}_seed"])
    mod = wm.Where2comm(c)
    mod.load_state_dict(_gauss_sd(c, seed + 500), strict=True)
    mod = mod.cuda().train()
    n = sum(rl)
    if single:
        x = torch.from_numpy(synth.w2c_attn_features(seed, n, 256, H // 2, W // 2, keep=0.6))
        rm = torch.from_numpy(synth.w2c_attn_psm(seed + 1, n, H // 2, W // 2))
        bb = None
    else:
        x = torch.from_numpy(synth.w2c_attn_features(seed, n, 64, H, W))
        rm = torch.from_numpy(synth.w2c_attn_psm(seed + 1, n, H // 2, W // 2))
        if variant and resnet:   # base_bev_backbone_resnet.py:57-110 under the two-level fusion; level 0 has stride 1, so the confidence map is H x W
            rm = torch.from_numpy(synth.w2c_attn_psm(seed + 1, n, H, W))
            bb = ResNetBEVBackbone(CFG["resnet_backbone_variant"], 64)
            bb.load_state_dict(synth.synthetic_state_dict(synth.resnet_backbone_param_spec(CFG["resnet_backbone_variant"], ""), seed=37), strict=True)
        elif variant:  # base_bev_backbone.py:87-121: a deblock that DOWN-samples (Conv2d(2, stride 2)) and the final deblock on the concatenation
            vbc = synth.submodule_configs()["backbone_variant"]
            bb = BaseBEVBackbone(vbc, 64)
            bb.load_state_dict(synth.synthetic_state_dict(synth.backbone_param_spec(vbc, 64, ""), seed=35), strict=True)
        elif resnet:   # base_bev_backbone_resnet.py + resblock.py in train mode
            bb = ResNetBEVBackbone(CFG["resnet_backbone"], 64)
            bb.load_state_dict(synth.synthetic_state_dict(synth.resnet_backbone_param_spec(CFG["resnet_backbone"], ""), seed=33), strict=True)
        else:
            bb = BaseBEVBackbone(CFG["backbone"], 64)
            bb.load_state_dict(synth.synthetic_state_dict(synth.backbone_param_spec(CFG["backbone"], 64, ""), seed=31), strict=True)
        bb = bb.cuda().train()
        for p_ in bb.parameters():
            p_.requires_grad_(True)
    xg = x.cuda().requires_grad_(True)
    pw = synth.w2c_attn_pairwise(rl).cuda()
    if alone:
        fused, vol, extra = bb({"spatial_features": xg})["spatial_features_2d"], 0.0, {}
    else:
        fused, vol, extra = mod(xg, rm.cuda(), torch.tensor(rl).cuda(), pw) if single else mod(xg, rm.cuda(), torch.tensor(rl).cuda(), pw, bb, None)
    assert extra == {} and fused.requires_grad and float(vol) == float(g[f"{tag}_vol"])
    _close(fused, g[f"{tag}_fused"])
    G = torch.from_numpy(synth.seeded_uniform(seed + 9, tuple(fused.shape), -1.0, 1.0)).cuda()
    (fused * G).sum().backward()
    got = {"x": xg.grad}
    if bb is not None:
        got.update({k: p_.grad for k, p_ in bb.named_parameters()})
    keys = [str(k) for k in g[f"{tag}_grad_keys"]]
    assert len(keys) == (1 if single else 25 if resnet and variant else 49 if resnet else 22 if variant else 28)
    worst = 0.0
    for k in keys:
        assert got[k] is not None, k
        gm = max(float(g[f"{tag}_g64max:{k}"]), 1e-30)
        v = got[k].reshape(-1)
        stride = max(1, v.numel() // 4096)
        d = np.abs(v[::stride].cpu().numpy().astype(np.float64) - g[f"{tag}_g64:{k}"].astype(np.float64)).max() / gm
        a = abs(float(v.double().abs().sum()) - float(g[f"{tag}_g64abs:{k}"])) / max(float(g[f"{tag}_g64abs:{k}"]), 1e-30)
        worst = max(worst, d)
        assert d <= 3e-4 and a <= 3e-4, (k, d, a, float(g[f"{tag}_gdev:{k}"]))
    print(f"{tag}: {len(keys)} gradients, worst deviation from the float64 step {worst:.2e} of the gradient's maximum")
    if bb is not None:
        for k, b in bb.named_buffers():
            ref = g[f"{tag}_b:{k}"].astype(np.float64)
            assert np.abs(b.detach().cpu().numpy().astype(np.float64) - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), k
