"""GPU: the pipelined split-3 implicit GEMM (csrc/conv_x3p.hip, tile flag 0x0400 | 0x1000) through the C-ABI: bit-identical to
conv_igemm_bf16x3 (same hi / mid / lo terms, same six products in the same order, same K order) on 1x1 / 3x3 / strided / transposed
shapes, ragged M, channel slices and the fused epilogues; error against fp64 not above the fp32-MFMA kernel's.  Replaces the same
reference layers as av2x_conv2d (downsample_conv.py:8-54, base_bev_backbone.py deblocks, the token Linears of the fusion heads)."""
from ctypes import byref, c_void_p

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # n, h, w, cin, cout, ks, stride, relu
    (2, 23, 31, 64, 64, 3, 1, 1),
    (1, 20, 36, 256, 256, 1, 1, 0),
    (3, 17, 19, 128, 256, 3, 2, 1),
    (1, 50, 176, 384, 256, 1, 1, 1),
    (5, 7, 9, 48, 128, 1, 1, 2),        # cin % 32 != 0 (16-channel steps); GELU epilogue
    (1, 1, 3, 16, 64, 3, 1, 0),
    (2, 9, 11, 1024, 192, 1, 1, 0),
]


def _p(t):
    return c_void_p(t.data_ptr()) if t is not None else None


def _st():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("case", CASES)
def test_bit_identical_to_conv_igemm_bf16x3_and_fp32_accurate(case):
    from airv2x_perception_amd import _lib
    from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight, to_bf16x3_koct
    lib = _lib.load()
    n, h, w, cin, cout, ks, stride, relu = case
    pad = 1 if ks == 3 else 0
    g = torch.Generator().manual_seed(11 + cin + cout + ks)
    x = torch.randn(n, cin, h, w, generator=g) * torch.exp(torch.randn(n, cin, 1, 1, generator=g))
    wt = torch.randn(cout, cin, ks, ks, generator=g) / np.sqrt(cin * ks * ks)
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x.double(), wt.double(), None, stride=stride, padding=pad) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    ref = {0: ref, 1: F.relu(ref), 2: F.gelu(ref)}[relu].permute(0, 2, 3, 1)
    wp, coutp = pack_conv_weight(wt)
    w32, w3 = wp.cuda(), to_bf16x3_koct(wp).cuda()
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
    outs, errs = {}, {}
    scale_d, shift_d = scale.cuda(), shift.cuda()
    tiles = [("f32", (64 << 16) | 64, w32)] if cin % 32 == 0 else []
    tiles += [("x3", (128 << 16) | 64 | 0x0400, w3)] if cin % 32 == 0 else []
    tiles += [("x3p64", (128 << 16) | 64 | 0x1400, w3)] + ([("x3p128", (128 << 16) | 128 | 0x1400, w3)] if coutp % 128 == 0 else [])
    for name, tile, wgt in tiles:
        out = torch.full((n, ho, wo, cout), float("nan"), device="cuda")
        d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, in_ctot=cin, in_coff=0, ho=ho, wo=wo, cout=cout, coutp=coutp, out_ctot=cout, out_coff=0,
                          ks=ks, stride=stride, pad=pad, relu=relu, mode=0, up=1, tile=tile, sk_wgs=0)
        _lib.check(lib.av2x_conv2d(byref(d), _p(xd), _p(wgt), _p(scale_d), _p(shift_d), _p(out), _st()), name)
        o = out.cpu()
        assert not torch.isnan(o).any(), name
        e = (o.double() - ref).abs()
        outs[name], errs[name] = o, (float(e.max()), float(e.pow(2).mean().sqrt()))
    if "x3" in outs:
        assert torch.equal(outs["x3"], outs["x3p64"]), "x3p must give the bits of conv_igemm_bf16x3"
    if "x3p128" in outs:
        assert torch.equal(outs["x3p64"], outs["x3p128"])
    if "f32" in errs:
        assert errs["x3p64"][1] <= 1.05 * errs["f32"][1] and errs["x3p64"][0] <= 1.5 * errs["f32"][0], errs
    assert errs["x3p64"][0] <= 3e-5 * max(1.0, float(ref.abs().max())), errs


def test_deconv_nchw_slices_and_residual():
    from airv2x_perception_amd import _lib
    from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight, pack_deconv_weight, to_bf16x3_koct
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    # transposed convolution (kernel = stride = up), output into a channel slice of a concat buffer
    n, h, w, cin, cout, up = 2, 9, 7, 128, 64, 2
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cin, cout, up, up, generator=g) / np.sqrt(cin)
    shift = torch.randn(cout, generator=g) * 0.1
    ref = F.relu(F.conv_transpose2d(x.double(), wt.double(), shift.double(), stride=up)).permute(0, 2, 3, 1)
    wp, coutp = pack_deconv_weight(wt)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    outs = []
    shift_d = shift.cuda()
    for tile, wgt in (((128 << 16) | 64 | 0x0400, to_bf16x3_koct(wp).cuda()), ((128 << 16) | 64 | 0x1400, to_bf16x3_koct(wp).cuda()),
                      ((128 << 16) | 128 | 0x1400, to_bf16x3_koct(wp).cuda())):
        out = torch.full((n, h * up, w * up, cout + 32), 5.0, device="cuda")
        d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, in_ctot=cin, in_coff=0, ho=h, wo=w, cout=cout, coutp=coutp, out_ctot=cout + 32, out_coff=32,
                          ks=1, stride=1, pad=0, relu=1, mode=_lib.AV2X_DECONV, up=up, tile=tile, sk_wgs=0)
        _lib.check(lib.av2x_conv2d(byref(d), _p(xd), _p(wgt), None, _p(shift_d), _p(out), _st()), "deconv")
        o = out.cpu()
        assert torch.all(o[..., :32] == 5.0)
        assert float((o[..., 32:].double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
        outs.append(o)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    # residual operand + input slice
    n, h, w, cin, cout = 1, 12, 10, 64, 128
    xw = torch.randn(n, h, w, cin + 16, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) / np.sqrt(cin)
    res = torch.randn(n, h, w, cout, generator=g)
    shift = torch.randn(cout, generator=g)
    ref = F.conv2d(xw[..., 16:].permute(0, 3, 1, 2).double(), wt.double(), shift.double()).permute(0, 2, 3, 1) + res.double()
    wp, coutp = pack_conv_weight(wt)
    out = torch.empty(n, h, w, cout, device="cuda")
    xw_d, w3_d, shift_d, res_d = xw.cuda(), to_bf16x3_koct(wp).cuda(), shift.cuda(), res.cuda()
    d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, in_ctot=cin + 16, in_coff=16, ho=h, wo=w, cout=cout, coutp=coutp, out_ctot=cout, out_coff=0,
                      ks=1, stride=1, pad=0, relu=0, mode=0, up=1, tile=(128 << 16) | 128 | 0x1400, sk_wgs=0)
    _lib.check(lib.av2x_conv2d_res(byref(d), _p(xw_d), _p(w3_d), None, _p(shift_d), _p(res_d), _p(out), _st()), "res")
    assert float((out.cpu().double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    d.cin = d.in_ctot = 40
    assert lib.av2x_conv2d(byref(d), _p(out), _p(out), None, _p(out), _p(out), _st()) != 0


@pytest.mark.parametrize("which,name", [("w2c", "w2c_full_n4"), ("cobevt", "cobevt_full_n4"), ("v2xvit", "v2xvit_full_n4"), ("when2com", "when2com_full_n2")])
def test_goldens_in_x3_mode_at_unchanged_tolerances(which, name):
    """engine.wino_x3 + engine.x3p (bench.py --gemm x3): every model's full-grid golden at the tolerance of its fp32-MFMA test."""
    from tests.helpers import assert_close, load_fixture
    fx = load_fixture(name)
    if which == "w2c":
        from airv2x_perception_amd.opencood_iface import Airv2xWhere2com as M
        from tests.helpers import case_from_fixture
        hy, args, sd, dd, _, _ = case_from_fixture(fx)
        rtol, atol_of, hs = 2e-4, lambda ref: 2e-4, int(fx["sample_stride"])
    elif which == "cobevt":
        import tests.test_cobevt as tc
        from airv2x_perception_amd.opencood_iface import Airv2xCoBEVT as M
        hy, args, sd, dd = tc._case(fx)
        rtol, atol_of, hs = 3e-4, lambda ref: 3e-4, int(fx["head_stride"])
    elif which == "when2com":
        import tests.test_when2com as tw
        from airv2x_perception_amd.opencood_iface import Airv2xWhen2com as M
        hy, args, sd, dd = tw._case(fx)
        rtol, atol_of, hs = 3e-4, lambda ref: 3e-4, int(fx["head_stride"])
    else:
        import tests.test_v2xvit as tv
        from airv2x_perception_amd.opencood_iface import Airv2xV2XVit as M
        hy, args, sd, dd = tv._case(fx)
        rtol, atol_of, hs = 1e-3, lambda ref: 1e-4 * max(10.0, float(np.abs(ref).max())), int(fx["head_stride"])
    model = M(args)
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda").eval()
    eng = model.engine()
    eng.wino_x3 = True
    eng.x3p = True
    out = model(dd)
    for k in ("psm", "rm", "obj"):
        got = out[k].cpu().numpy()
        assert_close(got[..., ::hs, ::hs] if hs > 1 else got, fx[k], rtol, atol_of(fx[k]), f"{name} {k} (x3)")
    if which == "w2c":
        assert int(out["comm_rate"]) == int(fx["comm_rate"])


@pytest.mark.parametrize("m_tokens,cout,relu,res,bn", [(1000, 768, 0, False, 128), (333, 256, 2, True, 64), (4096 + 17, 1024, 2, False, 128)])
def test_layernorm_in_the_operand_load_equals_layernorm_then_linear(m_tokens, cout, relu, res, bn):
    """av2x_layernorm_stats + av2x_conv2d_ln (the token Linear normalises its rows while it loads them: PreNorm of base_transformer.py:9-20)
    against av2x_layernorm followed by av2x_conv2d_res on the materialised tensor: the same bits, ragged token counts, bias / GELU / residual
    epilogues, both pipelined split-3 tiles.  The statistics equal those of F.layer_norm's definition in float64."""
    from airv2x_perception_amd import _lib
    from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight, to_bf16x3_koct
    lib = _lib.load()
    C = 256
    g = torch.Generator().manual_seed(7 + cout)
    x = (torch.randn(m_tokens, C, generator=g) * torch.exp(torch.randn(m_tokens, 1, generator=g)) + torch.randn(m_tokens, 1, generator=g)).cuda()
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.2).cuda()
    wt = torch.randn(cout, C, 1, 1, generator=g) / 16
    wp, coutp = pack_conv_weight(wt)
    w3 = to_bf16x3_koct(wp).cuda()
    shift = (torch.randn(cout, generator=g) * 0.1).cuda()
    resid = torch.randn(m_tokens, cout, generator=g).cuda() if res else None
    d = _lib.ConvDesc(n=1, h=1, w=m_tokens, cin=C, in_ctot=C, in_coff=0, ho=1, wo=m_tokens, cout=cout, coutp=coutp, out_ctot=cout, out_coff=0,
                      ks=1, stride=1, pad=0, relu=relu, mode=0, up=1, tile=(128 << 16) | bn | 0x1400, sk_wgs=0)
    xn = torch.empty_like(x)
    _lib.check(lib.av2x_layernorm(_p(x), _p(gamma), _p(beta), _p(xn), m_tokens, C, 1e-5, _st()), "ln")
    want = torch.full((m_tokens, cout), float("nan"), device="cuda")
    _lib.check(lib.av2x_conv2d_res(byref(d), _p(xn), _p(w3), None, _p(shift), _p(resid), _p(want), _st()), "linear")
    stats = torch.empty((m_tokens, 2), device="cuda")
    _lib.check(lib.av2x_layernorm_stats(_p(x), _p(stats), m_tokens, C, 1e-5, _st()), "stats")
    got = torch.full((m_tokens, cout), float("nan"), device="cuda")
    _lib.check(lib.av2x_conv2d_ln(byref(d), _p(x), _p(stats), _p(gamma), _p(beta), _p(w3), None, _p(shift), _p(resid), _p(got), _st()), "ln+linear")
    assert torch.equal(got, want)
    xd = x.double().cpu()
    assert torch.allclose(stats[:, 0].double().cpu(), xd.mean(1), rtol=1e-5, atol=1e-6)
    assert torch.allclose(stats[:, 1].double().cpu(), 1.0 / torch.sqrt(xd.var(1, unbiased=False) + 1e-5), rtol=1e-5)
    # refused: a 3x3 layer, a channel slice, a non-pipelined tile
    d2 = _lib.ConvDesc(n=1, h=1, w=m_tokens, cin=C, in_ctot=C, in_coff=0, ho=1, wo=m_tokens, cout=cout, coutp=coutp, out_ctot=cout, out_coff=0,
                       ks=1, stride=1, pad=0, relu=relu, mode=0, up=1, tile=(128 << 16) | 64 | 0x0400, sk_wgs=0)
    assert lib.av2x_conv2d_ln(byref(d2), _p(x), _p(stats), _p(gamma), _p(beta), _p(w3), None, _p(shift), _p(resid), _p(got), _st()) != 0
    d3 = _lib.ConvDesc(n=1, h=1, w=m_tokens, cin=C // 2, in_ctot=C, in_coff=0, ho=1, wo=m_tokens, cout=cout, coutp=coutp, out_ctot=cout, out_coff=0,
                       ks=1, stride=1, pad=0, relu=relu, mode=0, up=1, tile=(128 << 16) | bn | 0x1400, sk_wgs=0)
    assert lib.av2x_conv2d_ln(byref(d3), _p(x), _p(stats), _p(gamma), _p(beta), _p(w3), None, _p(shift), _p(resid), _p(got), _st()) != 0
