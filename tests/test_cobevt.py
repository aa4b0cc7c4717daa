"""CoBEVT: CPU oracle vs reference golden; GPU engine vs golden + oracle."""
import numpy as np
import pytest
import torch

from airv2x_perception_amd import synth
from oracle import cobevt_oracle as cob
from oracle import voxelize_oracle as vox
from tests.helpers import assert_close, load_fixture, sample


def _case(fx):
    rng = [float(v) for v in fx["lidar_range"]]
    types = [str(t) for t in fx["types"]]
    hy = synth.default_hypes_cobevt(rng, tuple(int(v) for v in fx["max_cav"]),
                                    compression=int(fx["compression"]) if "compression" in fx else 0)
    args = hy["model"]["args"]
    spec = synth.cobevt_param_spec(args)
    assert [k for k, _, _ in spec] == [str(k) for k in fx["spec_keys"]]
    sd = synth.synthetic_state_dict(spec, seed=int(fx["seed"]))
    pp = hy["preprocess"]
    voxd = [vox.points_to_voxels(vox.mask_points_by_range(synth.synthetic_cloud(i, int(fx["n_points"]), rng), rng), rng,
                                 pp["args"]["voxel_size"]) for i in range(len(types))]
    for i, v in enumerate(voxd):
        assert np.array_equal(v[1], fx[f"vox_coords_{i}"])
    return hy, args, sd, synth.build_data_dict(voxd, types, max_cav_num=args["max_cav_num"])


@pytest.mark.parametrize("name", ["cobevt_small_n3", "cobevt_small_n2_c4"])
def test_oracle_matches_reference_golden(name):
    fx = load_fixture(name)
    hy, args, sd, dd = _case(fx)
    tr = {}
    with torch.no_grad():
        out = cob.cobevt_forward(dd, sd, args, trace=tr)
    for k in ("psm", "rm", "obj"):
        assert_close(out[k], fx[k], 1e-5, 1e-5, k)
    bs = int(fx["big_stride"])
    for i in range(3):
        assert_close(sample(tr[f"fax_block{i}"], bs), fx[f"fax_block{i}"], 1e-5, 1e-5, f"fax_block{i}")
    assert_close(sample(tr["fused"], 2), fx["fused"], 1e-5, 1e-5, "fused")
    nv = len(fx["types"])
    assert tr["mask"].tolist() == [[1] * nv + [0] * (7 - nv)]
    if name.endswith("_c4"):   # message compression (naive_compress.py): 3 x (7 tensors) more in the state_dict
        assert sum(k.startswith("naive_compressor.") for k in sd) == 21 and sd["naive_compressor.encoder.0.weight"].shape == (64, 256, 3, 3)


def test_partition_roundtrip_and_index():
    x = torch.randn(1, 3, 5, 8, 12)
    for grid in (False, True):
        t = cob._partition(x, 4, grid)
        assert t.shape == (6, 48, 5)
        assert torch.equal(cob._unpartition(t, 1, 3, 5, 8, 12, 4, grid), x)
    # window: token (l, w1, w2) of window (x, y) is pixel (4x+w1, 4y+w2); grid: (w1*X + x, w2*Y + y)
    t = cob._partition(x, 4, False)
    assert torch.equal(t[1 * 3 + 2, 2 * 16 + 1 * 4 + 3], x[0, 2, :, 4 * 1 + 1, 4 * 2 + 3])
    t = cob._partition(x, 4, True)
    assert torch.equal(t[1 * 3 + 2, 2 * 16 + 1 * 4 + 3], x[0, 2, :, 1 * 2 + 1, 3 * 3 + 2])
    idx = cob.relative_position_index(7, 4)
    assert idx.shape == (112, 112) and int(idx.max()) == 13 * 49 - 1 and int(idx.min()) == 0
    assert torch.equal(idx, torch.from_numpy(synth._relative_position_index(7, 4)))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cobevt_small_n3", "cobevt_small_n2_c4", "cobevt_full_n4", "cobevt_full_n8"])   # n8: BASELINE configs[2], L = 8
def test_gpu_forward_matches_golden_and_oracle(name):
    from airv2x_perception_amd.opencood_iface import Airv2xCoBEVT
    fx = load_fixture(name)
    hy, args, sd, dd = _case(fx)
    model = Airv2xCoBEVT(args)
    assert list(model.state_dict().keys()) == [str(k) for k in fx["spec_keys"]]
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda").eval()
    tr = {}
    out = model.engine().forward(dd, trace=tr)
    torch.cuda.synchronize()
    bs = int(fx["big_stride"])
    for i in range(3):
        assert_close(sample(tr[f"fax_block{i}"], bs), fx[f"fax_block{i}"], 3e-4, 3e-4, f"fax_block{i}")
    assert_close(sample(tr["fused"], int(fx["fused_stride"]) if "fused_stride" in fx else 2), fx["fused"], 3e-4, 3e-4, "fused")
    hs = int(fx["head_stride"]) if "head_stride" in fx else 1
    for k in ("psm", "rm", "obj"):
        assert_close(sample(out[k], hs), fx[k], 3e-4, 3e-4, k)
        if k + "_sum" in fx:   # a checksum over ALL cells of the map (full-grid fixture)
            tot, ref = float(out[k].double().sum()), float(fx[k + "_sum"])
            assert abs(tot - ref) <= 1e-5 * max(1.0, float(out[k].double().abs().sum())), (k, tot, ref)
    assert set(out.keys()) == {"psm", "rm", "obj"}
    o2 = model(dd)
    assert torch.equal(o2["psm"], out["psm"])


@pytest.mark.gpu
@pytest.mark.parametrize("grid,L,nv", [(0, 5, 3), (1, 5, 3), (2, 5, 3), (3, 5, 3), (4, 5, 3), (5, 5, 3), (0, 7, 7), (1, 7, 4), (4, 7, 4),
                                       (0, 8, 8), (1, 8, 1), (0, 8, 5), (1, 7, 2), (0, 3, 3), (1, 9, 6),
                                       (8, 5, 3), (9, 7, 4), (8, 8, 8), (9, 8, 5), (16, 8, 8), (17, 8, 5), (16, 7, 6),
                                       (0, 12, 12), (1, 15, 10), (0, 15, 15),
                                       (32, 8, 8), (33, 8, 5), (32, 7, 6), (33, 8, 7), (32, 5, 5), (33, 6, 6), (32, 8, 3)])
def test_gpu_fax_attention_kernel(grid, L, nv):
    """grid bit 5 (round 5): the split-3 kernel on the bf16 matrix cores (more than 4 valid agents; x3 mode of the engine).
    grid bit 0: partition (window / grid); default kernel for ws = 4: one wave per (window, head), everything in
    registers (up to 4 valid agents; bit 4 forces it beyond); bit 3: the workgroup-per-window transposed-score kernel; bit 2: the generic-window MFMA kernel; bit 1: the
    VALU reference kernel (T = 16 L > 128 tokens always takes the VALU kernel; with more than 8 valid agents it keeps the K / V of
    2 or 1 heads in LDS at a time instead of 4)."""
    from ctypes import c_void_p
    from airv2x_perception_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(grid + 10 * L)
    H, W, ws, heads = 8, 12, 4, 8
    C = heads * 32
    tok = torch.randn(1, L, C, H, W, generator=g)
    sd = {"a.to_qkv.weight": torch.randn(3 * C, C, generator=g) / 16, "a.to_out.0.weight": torch.eye(C),
          "a.relative_position_index": cob.relative_position_index(L, ws),
          "a.relative_position_bias_table.weight": torch.randn((2 * L - 1) * 49, heads, generator=g)}
    part = cob._partition(tok, ws, bool(grid & 1))
    km = torch.tensor([1] * nv + [0] * (L - nv)).view(1, L, 1).expand(part.shape[0], L, ws * ws).reshape(-1, L * ws * ws)
    ref = cob._unpartition(cob.attention(part, km, sd, "a", heads, L, ws), 1, L, C, H, W, ws, bool(grid & 1))
    qkv = torch.nn.functional.linear(tok.permute(0, 1, 3, 4, 2), sd["a.to_qkv.weight"])[0].contiguous().cuda()  # (L,H,W,3C)
    out = torch.empty((L, H, W, C), device="cuda")
    table = sd["a.relative_position_bias_table.weight"].cuda()
    P = lambda t: c_void_p(t.data_ptr())
    _lib.check(lib.av2x_fax_attention(P(qkv), P(table), P(out), L, nv, H, W, ws, heads, 32, grid,
                                      c_void_p(torch.cuda.current_stream().cuda_stream)), "fax")
    assert_close(out.permute(0, 3, 1, 2).cpu(), ref[0], 1e-4, 1e-5, "fax attention")


@pytest.mark.gpu
@pytest.mark.parametrize("L,nv,grid", [(8, 8, 0), (8, 6, 1), (7, 5, 0)])
def test_gpu_fax_attention_split3_error_against_fp64_not_above_the_fp32_mfma_kernel(L, nv, grid):
    """fax_attention_x3_kernel (K, V, scaled Q and P as hi + mid + lo bf16 terms on v_mfma_f32_32x32x16_bf16) against an fp64 evaluation of
    Attention.forward (swap_fusion_modules.py:78-127) on operands with a wide dynamic range: max and rms error not above those of
    fax_attention_mfma4_kernel (fp32-input MFMA) on the same operands -- the criterion of the split-3 convolutions (tests/test_gpu_wino_x3.py)."""
    from ctypes import c_void_p
    from airv2x_perception_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(31 * L + nv)
    H, W, ws, heads = 16, 24, 4, 8
    C = heads * 32
    qkv = torch.randn(L, H, W, 3 * C, generator=g) * torch.exp(0.7 * torch.randn(L, H, W, 1, generator=g))
    table = torch.randn((2 * L - 1) * 49, heads, generator=g)
    idx = cob.relative_position_index(L, ws)
    # fp64 reference on partitioned tokens
    def part(t):
        return cob._partition(t.permute(0, 3, 1, 2).unsqueeze(0).double(), ws, bool(grid))
    q, k, v = (part(qkv[..., i * C:(i + 1) * C]) for i in range(3))
    Nw, T, _ = q.shape
    q, k, v = (t.view(Nw, T, heads, 32).permute(0, 2, 1, 3) for t in (q, k, v))
    sim = (q * (32 ** -0.5)) @ k.transpose(-1, -2) + table.double()[idx].permute(2, 0, 1).unsqueeze(0)
    km = torch.tensor([1] * nv + [0] * (L - nv)).view(L, 1).expand(L, ws * ws).reshape(-1).bool()
    sim = sim.masked_fill(~km.view(1, 1, 1, T), float("-inf"))
    o = (sim.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(Nw, T, C)
    ref = cob._unpartition(o, 1, L, C, H, W, ws, bool(grid))[0].permute(0, 2, 3, 1)         # (L, H, W, C)
    qd, td = qkv.cuda().contiguous(), table.cuda()
    errs = {}
    P = lambda t: c_void_p(t.data_ptr())
    for name, flag in (("f32", 8), ("x3", 32)):
        out = torch.full((L, H, W, C), float("nan"), device="cuda")
        _lib.check(lib.av2x_fax_attention(P(qd), P(td), P(out), L, nv, H, W, ws, heads, 32, grid | flag,
                                          c_void_p(torch.cuda.current_stream().cuda_stream)), "fax")
        e = (out.cpu().double() - ref).abs()
        assert not torch.isnan(e).any(), name
        errs[name] = (float(e.max()), float(e.pow(2).mean().sqrt()))
    floor = 2.0 ** -23 * max(1.0, float(ref.abs().max()))
    assert errs["x3"][1] <= errs["f32"][1] * 1.05 + 0.02 * floor, errs
    assert errs["x3"][0] <= errs["f32"][0] * 1.25 + floor, errs


@pytest.mark.gpu
def test_gpu_layernorm_and_mean():
    from ctypes import c_void_p
    from airv2x_perception_amd import _lib
    lib = _lib.load()
    x = torch.randn(1000, 256) * 3 + 1
    g, b = torch.rand(256) + 0.5, torch.randn(256)
    xd, gd, bd = x.cuda(), g.cuda(), b.cuda()
    y = torch.empty_like(xd)
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: c_void_p(t.data_ptr())
    _lib.check(lib.av2x_layernorm(P(xd), P(gd), P(bd), P(y), 1000, 256, 1e-5, st), "ln")
    assert_close(y.cpu(), torch.nn.functional.layer_norm(x, (256,), g, b, 1e-5), 1e-5, 1e-5, "layernorm")
    z = torch.randn(7, 40, 64)
    zd = z.cuda()
    m = torch.empty((40, 64), device="cuda")
    _lib.check(lib.av2x_agent_mean(P(zd), P(m), 7, 40 * 64, st), "mean")
    assert_close(m.cpu(), z.mean(0), 1e-6, 1e-6, "agent mean")


@pytest.mark.gpu
def test_gpu_batch_of_two_frames_equals_two_single_frames():
    """B = 2 in the reference's collate layout: the trunk is batched over all agents, regroup + fusion run per sample."""
    from airv2x_perception_amd.opencood_iface import Airv2xCoBEVT
    fx = load_fixture("cobevt_small_n3")
    hy, args, sd, dd3 = _case(fx)
    rng = [float(v) for v in fx["lidar_range"]]
    voxd = [vox.points_to_voxels(vox.mask_points_by_range(synth.synthetic_cloud(i, int(fx["n_points"]), rng), rng), rng,
                                 hy["preprocess"]["args"]["voxel_size"]) for i in range(3)]
    dd2 = synth.build_data_dict([voxd[0], voxd[2]], ["vehicle", "drone"], max_cav_num=args["max_cav_num"])
    model = Airv2xCoBEVT(args)
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda").eval()
    eng = model.engine()
    eng.stream_k = False
    o3 = {k: v.clone() for k, v in eng.forward(dd3).items()}
    o2 = {k: v.clone() for k, v in eng.forward(dd2).items()}
    ob = eng.forward(synth.merge_frames([dd3, dd2]))
    for k in ("psm", "rm", "obj"):
        assert ob[k].shape[0] == 2 and torch.equal(ob[k][0:1], o3[k]) and torch.equal(ob[k][1:2], o2[k]), k


@pytest.mark.gpu
def test_gpu_full_house_and_empty_cloud_against_oracle():
    """Edge cases: (a) n = L = 7 agents (no padded agent: the fused q|k|v GEMM and the 8-key-tile attention path);
    (b) an agent whose cloud is empty after the crop -> the reference's two dummy points (sp_voxel_preprocessor.py:80-90)."""
    from airv2x_perception_amd.opencood_iface import Airv2xCoBEVT
    rng = [-25.6, -12.8, -3.0, 25.6, 12.8, 1.0]
    hy = synth.default_hypes_cobevt(rng)
    args, pp = hy["model"]["args"], hy["preprocess"]
    sd = synth.synthetic_state_dict(synth.cobevt_param_spec(args), seed=11)
    model = Airv2xCoBEVT(args)
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda").eval()
    dummy = np.array([[0, 0, 0, 0], [-0.218277, -11.13425732, -80.05884552, 1.230595649e-38]], dtype=np.float32)
    for types, empty in ((["vehicle"] * 3 + ["rsu"] * 2 + ["drone"] * 2, None), (["vehicle", "rsu", "drone"], 1)):
        voxd = []
        for i in range(len(types)):
            pts = vox.mask_points_by_range(synth.synthetic_cloud(20 + i, 500, rng), rng)
            if empty == i:
                pts = dummy
            voxd.append(vox.points_to_voxels(pts, rng, pp["args"]["voxel_size"]))
        dd = synth.build_data_dict(voxd, types, max_cav_num=args["max_cav_num"])
        out = model(dd)
        with torch.no_grad():
            ref = cob.cobevt_forward(dd, sd, args)
        for k in ("psm", "rm", "obj"):
            assert_close(out[k].cpu(), ref[k], 3e-4, 3e-4, f"{len(types)} agents, empty={empty}: {k}")
