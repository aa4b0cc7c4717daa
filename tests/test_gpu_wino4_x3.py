"""GPU: the split-3 Winograd F(4x4,3x3) kernel (csrc/conv_wino4_x3.hip; tile flag 0x60000000 | 0x0400) through the C-ABI.
Replaces the layers of common_modules/downsample_conv.py:8-54 / base_bev_backbone.py:6-154 that engine.wino4_rule selects, in x3 mode.
(a) per kernel: max and rms error against an fp64 convolution of the same fp32 operands are not above the fp32-MFMA F(4x4) kernel's
    (small slack where both sit at the rounding floor); edge shapes (H, W not multiples of 4, 1-pixel maps, workgroups that wrap rows
    and images, tiles beyond the last one), channel slices, every fused epilogue;
(b) per model: the goldens that reach the F(4x4) class (full grid) at their UNCHANGED tolerances with the kernel in the engine."""
from ctypes import byref, c_void_p

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import assert_close, load_fixture

pytestmark = pytest.mark.gpu

W4_X3 = 0x60000400 | (32 << 16) | 64
W4_F32 = 0x60000000 | (32 << 16) | 64
CASES = [
    # n, h, w, cin, cout, relu
    (1, 100, 352, 256, 256, 1),
    (2, 25, 88, 256, 256, 1),
    (4, 50, 176, 128, 128, 1),
    (3, 9, 13, 128, 128, 0),
    (1, 1, 7, 64, 64, 1),
    (2, 6, 1, 32, 64, 1),
    (5, 3, 5, 96, 192, 0),
    (1, 25, 87, 384, 256, 1),
    (1, 4, 4, 32, 64, 0),
]


@pytest.fixture(scope="module")
def lib():
    from airv2x_perception_amd import _lib
    return _lib.load()


def _p(t):
    return c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _pack(lib, wt):
    from airv2x_perception_amd import _lib
    from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight
    wp, coutp = pack_conv_weight(wt)
    wp = wp.cuda()
    cin = wt.shape[1]
    u4 = torch.empty(lib.av2x_wino4_weight_bytes(cin, coutp) // 4, device="cuda")
    _lib.check(lib.av2x_wino4_pack_weights(_p(wp), cin, coutp, _p(u4), _stream()), "av2x_wino4_pack_weights")
    u43 = torch.empty(lib.av2x_wino4_x3_weight_bytes(cin, coutp) // 2, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.av2x_wino4_x3_pack_weights(_p(wp), cin, coutp, _p(u43), _stream()), "av2x_wino4_x3_pack_weights")
    return u4, u43, coutp


def _run(lib, x, wgt, scale, shift, res, out, tile, relu, cin, cout, coutp, in_ctot=None, in_coff=0, out_ctot=None, out_coff=0):
    from airv2x_perception_amd import _lib
    n, h, w = x.shape[:3]
    d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, in_ctot=in_ctot or cin, in_coff=in_coff, ho=h, wo=w, cout=cout, coutp=coutp,
                      out_ctot=out_ctot or cout, out_coff=out_coff, ks=3, stride=1, pad=1, relu=relu, mode=0, up=1, tile=tile, sk_wgs=0)
    _lib.check(lib.av2x_conv2d_res(byref(d), _p(x), _p(wgt), _p(scale), _p(shift), _p(res), _p(out), _stream()), "av2x_conv2d")


@pytest.mark.parametrize("case", CASES)
def test_error_against_fp64_not_above_the_fp32_f4x4_kernel(lib, case):
    n, h, w, cin, cout, relu = case
    g = torch.Generator().manual_seed(177 + cin + cout + h)
    # wide dynamic range in the activations (what the hi / mid / lo split has to carry)
    x = torch.randn(n, cin, h, w, generator=g) * torch.exp(torch.randn(n, cin, h, w, generator=g))
    wt = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x.double(), wt.double(), None, padding=1) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    ref = (F.relu(ref) if relu else ref).permute(0, 2, 3, 1)
    u4, u43, coutp = _pack(lib, wt)
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    errs = {}
    scale_d, shift_d = scale.cuda(), shift.cuda()
    for name, tile, wgt in (("f32", W4_F32, u4), ("x3", W4_X3, u43)):
        out = torch.full((n, h, w, cout), float("nan"), device="cuda")
        _run(lib, xn, wgt, scale_d, shift_d, None, out, tile, relu, cin, cout, coutp)
        o = out.cpu()
        assert not torch.isnan(o).any(), name
        e = (o.double() - ref).abs()
        errs[name] = (float(e.max()), float(e.pow(2).mean().sqrt()))
    floor = 2.0 ** -23 * max(1.0, float(ref.abs().max()))          # one fp32 ulp of the largest output
    assert errs["x3"][1] <= errs["f32"][1] * 1.02 + 0.02 * floor, errs     # rms: not above the fp32-MFMA kernel's
    # max: ONE element out of up to 9 M; both kernels' error is dominated by the same fp32 transforms (coefficients up to 8), whose
    # summation order differs between them, so the single worst element fluctuates by tens of per cent either way
    assert errs["x3"][0] <= errs["f32"][0] * 1.30 + floor, errs
    assert errs["x3"][0] <= 1e-4 * max(1.0, float(ref.abs().max())), errs   # and the absolute bound of the fp32 F(4x4) test


def test_results_are_run_to_run_identical_and_independent_of_the_launch_size(lib):
    """One agent alone (what a shard rank launches) equals the same agent inside a four-agent launch, bit for bit."""
    n, h, w, cin, cout = 4, 25, 88, 256, 256
    g = torch.Generator().manual_seed(31)
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wt = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
    shift = torch.randn(cout, generator=g).cuda()
    u4, u43, coutp = _pack(lib, wt)
    full = torch.empty(n, h, w, cout, device="cuda")
    _run(lib, x, u43, None, shift, None, full, W4_X3, 1, cin, cout, coutp)
    again = torch.empty_like(full)
    _run(lib, x, u43, None, shift, None, again, W4_X3, 1, cin, cout, coutp)
    assert torch.equal(full, again)
    for i in range(n):
        one = torch.empty(1, h, w, cout, device="cuda")
        _run(lib, x[i:i + 1].contiguous(), u43, None, shift, None, one, W4_X3, 1, cin, cout, coutp)
        assert torch.equal(one[0], full[i]), i


@pytest.mark.parametrize("case", CASES + [(3, 50, 176, 128, 128, 1), (4, 25, 88, 256, 256, 1), (1, 7, 6, 128, 64, 0)])
@pytest.mark.parametrize("act,with_res", [(None, False), (5, True), (4, True)])
def test_pingpong_form_equals_the_four_wave_form_bit_for_bit(lib, case, act, with_res, monkeypatch):
    """conv_wino4_x3_pp (round 6: eight waves, two per SIMD, M / T roles in opposite phases -- the library's default) against
    conv_wino4_x3 (four waves; AV2X_W4X3_PP=0): the same six products per accumulator and chunk in the same order, the same transforms,
    the same finalising expressions -> torch.equal, on every shape class (full blocks, ragged edges, tiles beyond the last one, one- and
    four-cout-block layers, slices) and in the plain and the GENERAL epilogue."""
    n, h, w, cin, cout, relu = case
    relu = relu if act is None else act
    g = torch.Generator().manual_seed(4242 + cin + 3 * cout + h * w + relu)
    x = (torch.randn(n, h, w, cin + 32, generator=g) * torch.exp(torch.randn(n, h, w, cin + 32, generator=g))).cuda()
    wt = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
    scale, shift = (torch.rand(cout, generator=g) + 0.5).cuda(), (torch.randn(cout, generator=g) * 0.1).cuda()
    res = torch.rand(n, h, w, cout, generator=g).cuda() if with_res else None     # relu 4: the gate operand is a dense (pixels, cout) tensor
    _, u43, coutp = _pack(lib, wt)
    outs = []
    for pp in ("1", "0"):
        monkeypatch.setenv("AV2X_W4X3_PP", pp)
        out = torch.full((n, h, w, cout + 64), float("nan"), device="cuda")
        if relu == 4:       # tanh(z) * gate: the gate is read as (pixels, cout) whatever the output slice
            out = torch.full((n, h, w, cout), float("nan"), device="cuda")
            _run(lib, x, u43, scale, shift, res, out, W4_X3, relu, cin, cout, coutp, in_ctot=cin + 32, in_coff=32)
            outs.append(out)
        else:
            if res is not None:
                rs = torch.zeros(n, h, w, cout + 64, device="cuda")
                rs[..., 64:] = res
            _run(lib, x, u43, scale, shift, rs if res is not None else None, out, W4_X3, relu, cin, cout, coutp, in_ctot=cin + 32, in_coff=32,
                 out_ctot=cout + 64, out_coff=64)
            assert torch.isnan(out[..., :64]).all()                       # the other channels of the concat buffer are not touched
            outs.append(out[..., 64:].contiguous())
        torch.cuda.synchronize()
    assert not torch.isnan(outs[0]).any()
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("act,with_res", [(1, True), (3, True), (4, True), (5, True), (0, False)])
def test_epilogues_and_residual(lib, act, with_res):
    n, h, w, cin, cout = 1, 13, 18, 256, 128
    g = torch.Generator().manual_seed(900 + act)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
    bias = torch.randn(cout, generator=g) * 0.1
    res = torch.rand(n, h, w, cout, generator=g)
    z = F.conv2d(x.double(), wt.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
    ref = {0: z, 1: torch.relu(z) + res.double(), 3: torch.sigmoid(z) + res.double(), 4: torch.tanh(z) * res.double(),
           5: torch.relu(z + res.double())}[act]
    u4, u43, coutp = _pack(lib, wt)
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    out = torch.full((n, h, w, cout), float("nan"), device="cuda")
    _run(lib, xn, u43, None, bias.cuda(), res.cuda() if with_res else None, out, W4_X3, act, cin, cout, coutp)
    assert float((out.cpu().double() - ref).abs().max()) <= 5e-5 * max(1.0, float(ref.abs().max()))


def test_channel_slices_and_argument_checks(lib):
    """Input read from a channel slice of a wider tensor, output written into a slice of a concat buffer (how the backbone uses it)."""
    from airv2x_perception_amd import _lib
    n, h, w, cin, cout = 2, 11, 9, 64, 64
    g = torch.Generator().manual_seed(5)
    xw = torch.randn(n, h, w, cin + 32, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
    shift = torch.randn(cout, generator=g)
    u4, u43, coutp = _pack(lib, wt)
    ref = F.relu(F.conv2d(xw[..., 32:].permute(0, 3, 1, 2).double(), wt.double(), shift.double(), padding=1)).permute(0, 2, 3, 1)
    out = torch.full((n, h, w, cout + 64), 7.0, device="cuda")
    _run(lib, xw.cuda(), u43, None, shift.cuda(), None, out, W4_X3, 1, cin, cout, coutp, in_ctot=cin + 32, in_coff=32,
         out_ctot=cout + 64, out_coff=64)
    o = out.cpu()
    assert torch.all(o[..., :64] == 7.0)
    assert float((o[..., 64:].double() - ref).abs().max()) <= 5e-5 * max(1.0, float(ref.abs().max()))
    d = _lib.ConvDesc(n=1, h=4, w=4, cin=48, in_ctot=48, in_coff=0, ho=4, wo=4, cout=64, coutp=64, out_ctot=64, out_coff=0,
                      ks=3, stride=1, pad=1, relu=1, mode=0, up=1, tile=W4_X3, sk_wgs=0)
    t = torch.zeros(8192, device="cuda")
    assert lib.av2x_conv2d_res(byref(d), _p(t), _p(t), None, _p(t), None, _p(t), _stream()) != 0
    assert b"cin" in lib.av2x_last_error()
    d.cin = d.in_ctot = 64
    d.stride = 2
    assert lib.av2x_conv2d_res(byref(d), _p(t), _p(t), None, _p(t), None, _p(t), _stream()) != 0


def test_engine_takes_the_split3_f4x4_tile_for_the_wino4_class_in_x3_mode():
    from airv2x_perception_amd.opencood_iface import Airv2xWhere2com
    from tests.helpers import case_from_fixture
    fx = load_fixture("w2c_full_n4")
    hy, args, sd, dd, _, _ = case_from_fixture(fx)
    model = Airv2xWhere2com(args)
    model.load_state_dict(sd)
    model = model.to("cuda").eval()
    eng = model.engine()
    eng.wino_x3 = eng.x3p = eng.wino4_x3 = True
    out = model(dd)
    assert any(L._wu43 is not None for L in eng.shrink), "the shrink header's 256 -> 256 layer must run on conv_wino4_x3"
    hs = int(fx["sample_stride"])
    for k in ("psm", "rm", "obj"):
        got = out[k].cpu().numpy()
        assert_close(got[..., ::hs, ::hs] if hs > 1 else got, fx[k], 2e-4, 2e-4, f"w2c_full_n4 {k} (wino4_x3)")
    assert int(out["comm_rate"]) == int(fx["comm_rate"])


@pytest.mark.parametrize("which,name", [("cobevt", "cobevt_full_n4"), ("v2xvit", "v2xvit_full_n4")])
def test_other_models_goldens_with_the_split3_f4x4_tile_at_unchanged_tolerances(which, name):
    fx = load_fixture(name)
    if which == "cobevt":
        import tests.test_cobevt as tc
        from airv2x_perception_amd.opencood_iface import Airv2xCoBEVT as M
        hy, args, sd, dd = tc._case(fx)
        rtol, atol_of = 3e-4, lambda ref: 3e-4
    else:
        import tests.test_v2xvit as tv
        from airv2x_perception_amd.opencood_iface import Airv2xV2XVit as M
        hy, args, sd, dd = tv._case(fx)
        rtol, atol_of = 1e-3, lambda ref: 1e-4 * max(10.0, float(np.abs(ref).max()))
    model = M(args)
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda").eval()
    eng = model.engine()
    eng.wino_x3 = eng.x3p = eng.wino4_x3 = True
    out = model(dd)
    assert any(L._wu43 is not None for L in eng.shrink)
    hs = int(fx["head_stride"])
    for k in ("psm", "rm", "obj"):
        assert_close(out[k].cpu().numpy()[..., ::hs, ::hs], fx[k], rtol, atol_of(fx[k]), f"{name} {k} (wino4_x3)")
