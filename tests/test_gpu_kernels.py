"""GPU parity of every C-ABI kernel against the CPU oracle (run with -m gpu on an MI355X).

Tolerances (fp32 everywhere; the kernels and ATen's CPU kernels only differ in summation
order and in folding BatchNorm into one multiply-add):
    |hip - oracle| <= 1e-4 * |oracle| + 1e-4 * rms-ish scale of the tensor
Integer / mask outputs must be bit-exact, except mask cells whose smoothed confidence lies
within 1e-6 of the 0.01 threshold (a discontinuity: both sides are "right").
"""
import ctypes
from ctypes import byref, c_void_p

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import where2comm_oracle as orc
from tests.helpers import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from airv2x_perception_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return _lib.load()


def _p(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _conv_call(lib, x_nhwc, wp, scale, shift, out, *, cin, cout, coutp, ks, stride, pad, relu, mode=0, up=1,
               in_coff=0, out_ctot=None, out_coff=0, tile=0):
    from airv2x_perception_amd import _lib
    n, h, w, ctot = x_nhwc.shape
    d = _lib.ConvDesc()
    d.n, d.h, d.w, d.cin, d.in_ctot, d.in_coff = n, h, w, cin, ctot, in_coff
    if mode == 1:
        d.ho, d.wo = h, w
    else:
        d.ho, d.wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
    d.cout, d.coutp, d.out_ctot, d.out_coff = cout, coutp, out_ctot or cout, out_coff
    d.ks, d.stride, d.pad, d.relu, d.mode, d.up, d.tile = ks, stride, pad, relu, mode, up, tile
    _lib.check(lib.av2x_conv2d(byref(d), _p(x_nhwc), _p(wp), _p(scale), _p(shift), _p(out), _stream()), "conv")


CONV_CASES = [
    # n, h, w, cin, cout, ks, stride, pad, relu, tile
    (2, 20, 36, 64, 64, 3, 2, 1, 1, 0),
    (1, 17, 23, 64, 64, 3, 1, 1, 1, (128 << 16) | 64),
    (3, 16, 40, 64, 128, 3, 2, 1, 1, (128 << 16) | 128),
    (2, 9, 13, 128, 128, 3, 1, 1, 0, (64 << 16) | 64),
    (1, 12, 20, 256, 256, 3, 1, 1, 1, (64 << 16) | 128),
    (2, 10, 30, 384, 256, 1, 1, 0, 1, 0),
    (1, 25, 88, 256, 256, 3, 1, 1, 1, 0),
    (2, 7, 9, 256, 30, 1, 1, 0, 0, (128 << 16) | 32),
    (1, 5, 5, 32, 14, 1, 1, 0, 0, 0),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_matches_torch_cpu(lib, case):
    from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight
    n, h, w, cin, cout, ks, stride, pad, relu, tile = case
    g = torch.Generator().manual_seed(1234 + cin + cout + ks)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, ks, ks, generator=g) / np.sqrt(cin * ks * ks)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x, wt, None, stride=stride, padding=pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if relu:
        ref = F.relu(ref)
    wp, coutp = pack_conv_weight(wt)
    dev = "cuda"
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    ho, wo = ref.shape[2], ref.shape[3]
    out = torch.full((n, ho, wo, cout), float("nan"), device=dev)
    _conv_call(lib, xd, wp.to(dev), scale.to(dev), shift.to(dev), out, cin=cin, cout=cout, coutp=coutp, ks=ks,
               stride=stride, pad=pad, relu=relu, tile=tile)
    got = out.permute(0, 3, 1, 2).cpu()
    assert_close(got, ref, 1e-4, 1e-4, f"conv {case}")


def test_conv_channel_slices_and_nchw(lib):
    """input channel offset inside a wider tensor, output into a slice of a concat buffer, NCHW store."""
    from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight
    g = torch.Generator().manual_seed(7)
    n, h, w = 2, 11, 19
    x = torch.randn(n, 96, h, w, generator=g)
    wt = torch.randn(30, 64, 1, 1, generator=g) / 8
    b = torch.randn(30, generator=g)
    ref = F.conv2d(x[:, 32:96], wt, b)
    wp, coutp = pack_conv_weight(wt)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    out = torch.full((n, 30, h, w), float("nan"), device="cuda")
    _conv_call(lib, xd, wp.cuda(), None, b.cuda(), out, cin=64, cout=30, coutp=coutp, ks=1, stride=1, pad=0, relu=0,
               mode=2, in_coff=32)
    assert_close(out.cpu(), ref, 1e-4, 1e-4, "nchw head")
    cat = torch.zeros((n, h, w, 50), device="cuda")
    _conv_call(lib, xd, wp.cuda(), None, b.cuda(), cat, cin=64, cout=30, coutp=coutp, ks=1, stride=1, pad=0, relu=0,
               in_coff=32, out_ctot=50, out_coff=20)
    got = cat.permute(0, 3, 1, 2).cpu()
    assert_close(got[:, 20:50], ref, 1e-4, 1e-4, "slice")
    assert float(got[:, :20].abs().max()) == 0.0


@pytest.mark.parametrize("up,cin", [(1, 64), (2, 128), (4, 256)])
def test_deconv_matches_torch_cpu(lib, up, cin):
    from airv2x_perception_amd.opencood_iface.packing import pack_deconv_weight
    g = torch.Generator().manual_seed(up)
    n, h, w, cout = 2, 7, 11, 128
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cin, cout, up, up, generator=g) / np.sqrt(cin)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    ref = F.relu(F.conv_transpose2d(x, wt, None, stride=up) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    wp, ncol = pack_deconv_weight(wt)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    cat = torch.full((n, h * up, w * up, 384), float("nan"), device="cuda")
    _conv_call(lib, xd, wp.cuda(), scale.cuda(), shift.cuda(), cat, cin=cin, cout=cout, coutp=ncol, ks=1, stride=1,
               pad=0, relu=1, mode=1, up=up, out_ctot=384, out_coff=128)
    got = cat.permute(0, 3, 1, 2).cpu()[:, 128:256]
    assert_close(got, ref, 1e-4, 1e-4, f"deconv up={up}")


SK_CASES = [
    # n, h, w, cin, cout, ks, stride, mode/up, tile, sk_wgs
    (2, 25, 44, 256, 256, 3, 1, 0, (128 << 16) | 64 | 0xe000, 96),      # every tile cut in ~2
    (2, 25, 44, 256, 256, 3, 1, 0, (128 << 16) | 128 | 0xe000, 37),     # uneven ranges, tail + whole + head
    (1, 25, 44, 128, 128, 3, 1, 0, (64 << 16) | 64 | 0x6000, 1000),     # ranges shorter than a tile (middle segments)
    (3, 13, 21, 128, 256, 3, 2, 0, (64 << 16) | 128 | 0x6000, 64),      # stride 2, ragged M
    (1, 9, 14, 64, 64, 3, 1, 0, (128 << 16) | 64 | 0x6000, 100000),     # more workgroups than iterations
    (2, 12, 20, 256, 256, 3, 1, 0, (128 << 16) | 128 | 0x6000, 4),      # ranges = whole tiles only (no fix-up)
    (2, 7, 11, 128, 128, 1, 1, 2, (128 << 16) | 64 | 0xe000, 24),       # deconv up=2 scatter epilogue
    (2, 25, 44, 256, 256, 3, 1, 0, (128 << 16) | 64 | 0xe000, 20),      # 72 tiles on 20 workgroups: 60 whole + 12 split
    (2, 25, 44, 256, 256, 3, 1, 0, (128 << 16) | 64 | 0xe000, 50),      # 50 whole + 22 remainder tiles split over 50
    (2, 25, 44, 256, 256, 3, 1, 0, (128 << 16) | 64 | 0xe000, 24),      # 72 = 3 x 24: whole tiles only, nothing is cut
    (2, 25, 44, 256, 256, 3, 1, 0, (128 << 16) | 128 | 0xe200, 25),     # LDS-DMA kernel, 36 tiles: 25 whole + 11 split
]


@pytest.mark.parametrize("case", SK_CASES)
def test_conv_stream_k_matches_torch_and_data_parallel(lib, case):
    """av2x_conv2d_sk: stream-K schedule + fix-up == torch fp32 (1e-4) and == the data-parallel schedule up to
    summation order; residual + GELU go through the fix-up epilogue as well."""
    from airv2x_perception_amd import _lib
    from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight, pack_deconv_weight
    n, h, w, cin, cout, ks, stride, up, tile, wgs = case
    g = torch.Generator().manual_seed(99 + cin + wgs % 97)
    x = torch.randn(n, cin, h, w, generator=g)
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    pad = 1 if ks == 3 else 0
    if up:
        wt = torch.randn(cin, cout, up, up, generator=g) / np.sqrt(cin)
        ref = F.conv_transpose2d(x, wt, None, stride=up) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
        wp, coutp = pack_deconv_weight(wt)
        mode = 1
    else:
        wt = torch.randn(cout, cin, ks, ks, generator=g) / np.sqrt(cin * ks * ks)
        ref = F.conv2d(x, wt, None, stride=stride, padding=pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
        wp, coutp = pack_conv_weight(wt)
        mode = 0
    ho, wo = ref.shape[2], ref.shape[3]
    res = torch.randn(n, ho, wo, cout, generator=g)
    ref = F.gelu(ref) + res.permute(0, 3, 1, 2)
    xd, wd, sc, sh, rd = x.permute(0, 2, 3, 1).contiguous().cuda(), wp.cuda(), scale.cuda(), shift.cuda(), res.cuda()
    d = _lib.ConvDesc()
    d.n, d.h, d.w, d.cin, d.in_ctot, d.in_coff = n, h, w, cin, cin, 0
    d.ho, d.wo = (h, w) if up else (ho, wo)
    d.cout, d.coutp, d.out_ctot, d.out_coff = cout, coutp, cout, 0
    d.ks, d.stride, d.pad, d.relu, d.mode, d.up = (1, 1, 0, 2, 1, up) if up else (ks, stride, pad, 2, 0, 1)
    outs = []
    for t, k in ((tile, wgs), (tile & ~0x2000, 0)):
        d.tile, d.sk_wgs = t, k
        nbytes = int(lib.av2x_conv2d_sk_workspace_bytes(t, min(k, 4096)))
        ws = torch.empty(max(nbytes // 4, 1), device="cuda")
        out = torch.full((n, ho, wo, cout), float("nan"), device="cuda")
        if mode == 1:   # residual is rejected for deconv: add it on the host
            _lib.check(lib.av2x_conv2d_sk(byref(d), _p(xd), _p(wd), _p(sc), _p(sh), _p(None), _p(out), _p(ws), nbytes,
                                          _stream()), "conv_sk")
            out = out + rd
        else:
            _lib.check(lib.av2x_conv2d_sk(byref(d), _p(xd), _p(wd), _p(sc), _p(sh), _p(rd), _p(out), _p(ws), nbytes,
                                          _stream()), "conv_sk")
        outs.append(out.permute(0, 3, 1, 2).cpu())
    assert_close(outs[0], ref, 1e-4, 1e-4, f"stream-K {case}")
    assert_close(outs[0], outs[1], 2e-5, 2e-5, f"stream-K vs data-parallel {case}")
    # a too-small workspace is rejected, not overrun
    d.tile, d.sk_wgs = tile, wgs
    rc = lib.av2x_conv2d_sk(byref(d), _p(xd), _p(wd), _p(sc), _p(sh), _p(None), _p(out), _p(ws), 16, _stream())
    assert rc != 0 and b"workspace" in lib.av2x_last_error()


@pytest.mark.parametrize("tile,wgs", [((128 << 16) | 64 | 0xd000, 7), ((128 << 16) | 128 | 0xd000, 3), ((64 << 16) | 64 | 0x5000, 1000),
                                      ((128 << 16) | 64 | 0x5000, 5), ((128 << 16) | 128 | 0x5000, 2), ((64 << 16) | 128 | 0x5000, 11)])
@pytest.mark.parametrize("shape", [(2, 23, 31, 256, 256, 1, 1), (1, 19, 27, 64, 128, 3, 2), (3, 9, 14, 128, 128, 3, 1)])
def test_conv_persistent_schedule_is_bit_identical(lib, tile, wgs, shape):
    """tile flag 0x1000: persistent workgroups over whole tiles, cross-tile software pipeline; no K split, so the
    result must equal the data-parallel schedule bit for bit (uneven tile ranges, more workgroups than tiles,
    ragged M, 3x3 taps crossing tile boundaries, odd K-step counts)."""
    from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight
    n, h, w, cin, cout, ks, stride = shape
    pad = 1 if ks == 3 else 0
    g = torch.Generator().manual_seed(cin + cout + ks + wgs)
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wp, coutp = pack_conv_weight(torch.randn(cout, cin, ks, ks, generator=g) / np.sqrt(cin * ks * ks))
    sc, sh = (torch.rand(cout, generator=g) + 0.5).cuda(), torch.randn(cout, generator=g).cuda()
    ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
    res = torch.randn(n, ho, wo, cout, generator=g).cuda()
    from airv2x_perception_amd import _lib
    outs = []
    for t, k in ((tile & ~0x1000, 0), (tile, wgs)):
        d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, in_ctot=cin, in_coff=0, ho=ho, wo=wo, cout=cout, coutp=coutp, out_ctot=cout,
                          out_coff=0, ks=ks, stride=stride, pad=pad, relu=2, mode=0, up=1, tile=t, sk_wgs=k)
        out = torch.full((n, ho, wo, cout), float("nan"), device="cuda")
        _lib.check(lib.av2x_conv2d_res(byref(d), _p(x), _p(wp.cuda()), _p(sc), _p(sh), _p(res), _p(out), _stream()), "conv")
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    assert not torch.isnan(outs[1]).any()


GLDS_TILES = [(128 << 16) | 128 | 0x8200, (128 << 16) | 64 | 0x8200, (128 << 16) | 128 | 0x0200, (128 << 16) | 64 | 0x0200,
              (64 << 16) | 64 | 0x0200, (128 << 16) | 128 | 0xc200, (128 << 16) | 64 | 0xc200, (128 << 16) | 128 | 0x4200,
              (128 << 16) | 64 | 0x4200, (64 << 16) | 64 | 0x4200]


@pytest.mark.parametrize("tile", GLDS_TILES)
@pytest.mark.parametrize("shape", [(2, 23, 31, 256, 256, 1, 1, 0), (1, 19, 27, 64, 128, 3, 2, 0), (3, 9, 14, 128, 128, 3, 1, 0),
                                   (1, 25, 88, 256, 256, 3, 1, 0), (2, 10, 12, 128, 512, 1, 1, 2)])
def test_conv_lds_dma_tiles_are_bit_identical(lib, tile, shape):
    """tile flag 0x0200: operands go L2 -> LDS by buffer_load ... lds (XOR-swizzled lane-linear image, zero fill of the
    halo by the descriptor's range check) with 2 or 3 (0x4000) LDS stages.  Same K order per output element as the
    register-staged kernel, so every tile must reproduce it bit for bit -- 1x1 and 3x3, stride 2, ragged M, deconv
    scatter (mode 1, up = 2), residual + GELU epilogue."""
    from airv2x_perception_amd import _lib
    from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight, pack_deconv_weight
    n, h, w, cin, cout, ks, stride, up = shape
    mode = 1 if up > 0 else (2 if up < 0 else 0)
    pad = 1 if ks == 3 else 0
    g = torch.Generator().manual_seed(cin + cout + ks + (tile & 0xffff))
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    if mode == 1:
        wp, coutp = pack_deconv_weight(torch.randn(cin, cout // (up * up), up, up, generator=g) / np.sqrt(cin))
        cout_d = cout // (up * up)
    else:
        wp, coutp = pack_conv_weight(torch.randn(cout, cin, ks, ks, generator=g) / np.sqrt(cin * ks * ks))
        cout_d = cout
    if coutp % (tile & 0x1ff):
        pytest.skip("tile does not divide the padded output width")
    sc, sh = (torch.rand(cout_d, generator=g) + 0.5).cuda(), torch.randn(cout_d, generator=g).cuda()
    ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
    oshape = (n, ho * max(up, 1), wo * max(up, 1), cout_d) if mode != 2 else (n, cout_d, ho, wo)
    res = torch.randn(oshape, generator=g).cuda() if mode == 0 else None
    outs = []
    for t in (tile & ~0x0200, tile):
        d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, in_ctot=cin, in_coff=0, ho=ho, wo=wo, cout=cout_d, coutp=coutp, out_ctot=cout_d,
                          out_coff=0, ks=ks, stride=stride, pad=pad, relu=2 if mode == 0 else 1, mode=mode, up=max(up, 1), tile=t, sk_wgs=0)
        out = torch.full(oshape, float("nan"), device="cuda")
        _lib.check(lib.av2x_conv2d_res(byref(d), _p(x), _p(wp.cuda()), _p(sc), _p(sh), _p(res), _p(out), _stream()), "conv")
        outs.append(out)
    assert not torch.isnan(outs[1]).any()
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("tile,wgs", [((128 << 16) | 64 | 0xa200, 96), ((128 << 16) | 128 | 0xe200, 37), ((64 << 16) | 64 | 0x6200, 1000),
                                      ((128 << 16) | 64 | 0x2200, 50), ((128 << 16) | 64 | 0xa200, 20), ((128 << 16) | 128 | 0xa200, 7)])
def test_conv_lds_dma_stream_k_equals_register_staged_stream_k(lib, tile, wgs):
    """stream-K on the LDS-DMA kernel: same iteration ranges, same partial layout, same fix-up -> bit-identical to the
    register-staged stream-K schedule of the same tile / workgroup count."""
    from airv2x_perception_amd import _lib
    from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight
    n, h, w, cin, cout, ks = 2, 25, 44, 256, 256, 3
    g = torch.Generator().manual_seed(wgs)
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wp, coutp = pack_conv_weight(torch.randn(cout, cin, ks, ks, generator=g) / np.sqrt(cin * ks * ks))
    sc, sh = (torch.rand(cout, generator=g) + 0.5).cuda(), torch.randn(cout, generator=g).cuda()
    ws = torch.empty(int(lib.av2x_conv2d_sk_workspace_bytes(tile, wgs)) // 4 + 16, device="cuda")
    outs = []
    for t in ((tile & ~0x0200) | 0x4000, tile):   # the register-staged stream-K kernel always runs the prefetch-2 pipeline
        d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, in_ctot=cin, in_coff=0, ho=h, wo=w, cout=cout, coutp=coutp, out_ctot=cout,
                          out_coff=0, ks=ks, stride=1, pad=1, relu=1, mode=0, up=1, tile=t, sk_wgs=wgs)
        out = torch.full((n, h, w, cout), float("nan"), device="cuda")
        _lib.check(lib.av2x_conv2d_sk(byref(d), _p(x), _p(wp.cuda()), _p(sc), _p(sh), None, _p(out), _p(ws), ws.numel() * 4,
                                      _stream()), "conv sk")
        outs.append(out)
    assert not torch.isnan(outs[1]).any()
    assert torch.equal(outs[0], outs[1])


def test_conv_rejects_bad_arguments(lib):
    from airv2x_perception_amd import _lib
    d = _lib.ConvDesc()
    d.n, d.h, d.w, d.cin, d.in_ctot = 1, 4, 4, 30, 30  # cin not a multiple of 32
    d.ho, d.wo, d.cout, d.coutp, d.out_ctot, d.ks, d.stride, d.pad = 4, 4, 32, 32, 32, 1, 1, 0
    t = torch.zeros(16, device="cuda")
    rc = lib.av2x_conv2d(byref(d), _p(t), _p(t), _p(None), _p(t), _p(t), _stream())
    assert rc != 0 and b"cin" in lib.av2x_last_error()
    with pytest.raises(RuntimeError):
        _lib.check(rc, "conv")


@pytest.mark.parametrize("agent_type", ["vehicle", "rsu", "drone"])
def test_pillar_vfe_scatter(lib, agent_type):
    from airv2x_perception_amd import synth
    from airv2x_perception_amd.opencood_iface.packing import fold_bn
    from oracle import voxelize_oracle as vox
    rng = [-12.8, -6.4, -3.0, 12.8, 6.4, 1.0]
    hy = synth.default_hypes(rng)
    args = hy["model"]["args"]
    sd = synth.synthetic_state_dict(synth.where2com_param_spec(args), 3)
    cfg = args[agent_type]["lidar"]
    clouds = [synth.clustered_cloud(i, 6000, rng) for i in range(2)]
    for cl in clouds:  # a hot spot that overflows the 32-point cap
        cl[:80, 0] = 3.3 + 0.001 * np.arange(80)
        cl[:80, 1] = -2.1
    voxs = [vox.points_to_voxels(cl, rng, [0.4, 0.4, 4.0]) for cl in clouds]
    assert max(int(v[2].max()) for v in voxs) == 32 and min(int(v[2].min()) for v in voxs) == 1
    vf = torch.from_numpy(np.concatenate([v[0] for v in voxs]))
    vc = torch.from_numpy(np.concatenate([np.concatenate([np.full((v[1].shape[0], 1), k, np.int32), v[1]], 1)
                                          for k, v in enumerate(voxs)]))
    vn = torch.from_numpy(np.concatenate([v[2] for v in voxs]))
    prefix = synth.TYPE_PREFIX[agent_type] + ".0.0"
    pf = orc.pillar_vfe(vf, vn, vc, sd, prefix, cfg["voxel_size"], cfg["lidar_range"])
    nx, ny = 64, 32
    ref = orc.pillar_scatter(pf, vc, 2, nx, ny)
    sc, sh = fold_bn(sd, prefix + ".pfn_layers.0.norm")
    vs, r = cfg["voxel_size"], cfg["lidar_range"]
    geom = (ctypes.c_float * 6)(vs[0], vs[1], vs[2], vs[0] / 2 + r[0], vs[1] / 2 + r[1], vs[2] / 2 + r[2])
    canvas = torch.zeros((3, ny, nx, 64), device="cuda")
    from airv2x_perception_amd import _lib
    smap = torch.tensor([2, 0], dtype=torch.int32, device="cuda")  # agent 0 -> slot 2, agent 1 -> slot 0
    # keep every device tensor alive in a variable: a temporary passed through _p() would be freed
    # (and its block re-used by the next .cuda()) before the asynchronous kernel reads it
    d_vf, d_vc, d_vn = vf.cuda(), vc.cuda(), vn.cuda()
    d_w, d_sc, d_sh = sd[prefix + ".pfn_layers.0.linear.weight"].cuda(), sc.cuda(), sh.cuda()
    _lib.check(lib.av2x_pillar_vfe_scatter(_p(d_vf), _p(d_vc), _p(d_vn), vf.shape[0], _p(d_w), _p(d_sc), _p(d_sh),
                                           ctypes.cast(geom, c_void_p), _p(canvas), 0, _p(smap), 2, ny, nx,
                                           _stream()), "pillar")
    got = canvas.permute(0, 3, 1, 2).cpu()
    assert_close(got[2], ref[0], 1e-4, 1e-5, "agent0->slot2")
    assert_close(got[0], ref[1], 1e-4, 1e-5, "agent1->slot0")
    assert float(got[1].abs().max()) == 0.0
    # occupancy pattern (integer-exact scatter indices)
    assert torch.equal(got[2].abs().sum(0) > 0, ref[0].abs().sum(0) > 0)
    nz = torch.zeros(1, dtype=torch.int64, device="cuda")
    _lib.check(lib.av2x_count_nonzero(_p(canvas), canvas.numel(), _p(nz), _stream()), "nz")
    assert int(nz.item()) == int(ref.count_nonzero().item())
    # the counting entry: same canvas, and the counter it adds to equals count_nonzero of what it wrote (here on top of a preset value)
    canvas2 = torch.zeros((3, ny, nx, 64), device="cuda")
    nz2 = torch.zeros(32 * 16, dtype=torch.int64, device="cuda")       # AV2X_NZ_SLOTS counters, AV2X_NZ_STRIDE apart
    occ2 = torch.zeros((3, ny, nx), dtype=torch.uint8, device="cuda")
    nz2[16] = 7
    _lib.check(lib.av2x_pillar_vfe_scatter_count(_p(d_vf), _p(d_vc), _p(d_vn), vf.shape[0], _p(d_w), _p(d_sc), _p(d_sh),
                                                 ctypes.cast(geom, c_void_p), _p(canvas2), 0, _p(smap), 2, ny, nx, _p(nz2), _p(occ2),
                                                 _stream()), "pillar count")
    assert torch.equal(canvas2, canvas)
    want_occ = torch.zeros((3, ny, nx), dtype=torch.uint8)
    inside = (vc[:, 2] >= 0) & (vc[:, 2] < ny) & (vc[:, 3] >= 0) & (vc[:, 3] < nx)
    want_occ[torch.tensor([2, 0])[vc[inside, 0].long()], vc[inside, 2].long(), vc[inside, 3].long()] = 1
    assert torch.equal(occ2.cpu(), want_occ)                       # the occupancy bytes: exactly the cells a pillar was written to
    assert int(nz2.sum().item()) == 7 + int(nz.item())
    assert int(nz2.view(32, 16)[:, 1:].abs().sum().item()) == 0      # only the strided slots are touched
    tot = torch.zeros(1, dtype=torch.int64, device="cuda")
    _lib.check(lib.av2x_nonzero_slots_sum(_p(nz2), _p(tot), _stream()), "slots sum")
    assert int(tot.item()) == 7 + int(nz.item())


@pytest.mark.parametrize("n,h,w,relu,with_scale", [(2, 18, 26, 1, True), (1, 8, 130, 0, False), (3, 31, 17, 1, True)])
def test_sparse_first_conv_equals_the_dense_convolution(lib, n, h, w, relu, with_scale):
    """av2x_conv3x3s2_sparse (the first backbone convolution as a gather over the occupied taps) against F.conv2d in float64 on the same
    scattered canvas: ZeroPad2d(1) + Conv2d(64, 64, 3, stride 2) + folded BatchNorm (+ ReLU), base_bev_backbone.py:30-48.  Occupied cells
    include one whose 64 values are all zero; odd sizes and a pixel count that is not a multiple of the 64-pixel batches."""
    from airv2x_perception_amd import _lib
    from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight
    g = torch.Generator().manual_seed(5 + n)
    occ = (torch.rand(n, h, w, generator=g) < 0.12)
    canvas = torch.relu(torch.randn(n, h, w, 64, generator=g)) * occ[..., None]
    iy, ix = [int(v[0]) for v in torch.nonzero(occ[0], as_tuple=True)]
    canvas[0, iy, ix] = 0.0                                        # a pillar whose features are all zero stays "occupied"
    wt = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    scale = torch.rand(64, generator=g) + 0.5 if with_scale else None
    shift = torch.randn(64, generator=g) * 0.3
    ref = F.conv2d(canvas.permute(0, 3, 1, 2).double(), wt.double(), None, stride=2, padding=1)
    ref = ref * (scale.double().view(1, -1, 1, 1) if with_scale else 1.0) + shift.double().view(1, -1, 1, 1)
    if relu:
        ref = torch.relu(ref)
    ho, wo = ref.shape[2], ref.shape[3]
    wp, coutp = pack_conv_weight(wt)
    assert coutp == 64
    d_c, d_o, d_w = canvas.cuda(), occ.to(torch.uint8).cuda(), wp.cuda()
    d_sc, d_sh = (scale.cuda() if with_scale else None), shift.cuda()
    out = torch.full((n, ho, wo, 64), float("nan"), device="cuda")
    _lib.check(lib.av2x_conv3x3s2_sparse(_p(d_c), _p(d_o), _p(d_w), _p(d_sc) if with_scale else None, _p(d_sh), relu, _p(out), n, h, w, 64, 64,
                                         _stream()), "sparse conv")
    got = out.permute(0, 3, 1, 2).cpu().double()
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))
    assert lib.av2x_conv3x3s2_sparse(_p(d_c), _p(d_o), _p(d_w), None, _p(d_sh), relu, _p(out), n, h, w, 32, 64, _stream()) != 0


def test_comm_mask(lib):
    from airv2x_perception_amd import _lib, synth
    g = torch.Generator().manual_seed(11)
    n, c, h, w = 5, 14, 20, 44
    psm = torch.randn(n, c, h, w, generator=g) * 1.2 - 6.7
    hy = synth.default_hypes()
    sd = synth.synthetic_state_dict(synth.where2com_param_spec(hy["model"]["args"]), 0)
    record_len = torch.tensor([3, 2])
    comm_cfg = hy["model"]["args"]["where2com_fusion"]["communication"]
    masks, rate, maps = orc.communication(orc._split(psm, record_len), sd, comm_cfg)
    assert 0.05 < float(masks.mean()) < 0.98
    pd = torch.zeros((n, h, w, 16), device="cuda")
    pd[..., :c] = psm.permute(0, 2, 3, 1).cuda()
    conf = torch.empty((n, h, w), device="cuda"); smooth = torch.empty_like(conf); mask = torch.empty_like(conf)
    count = torch.zeros(2, dtype=torch.int32, device="cuda")
    samp = torch.tensor([0, 0, 0, 1, 1], dtype=torch.int32, device="cuda")
    ego = torch.tensor([1, 0, 0, 1, 0], dtype=torch.int32, device="cuda")
    gk = "fusion_net.naive_communication.gaussian_filter"
    d_gw, d_gb = sd[gk + ".weight"].reshape(-1).cuda(), sd[gk + ".bias"].cuda()
    _lib.check(lib.av2x_comm_mask(_p(pd), n, h, w, 16, c, _p(d_gw), _p(d_gb), 5, 0.01, _p(samp), _p(ego), _p(conf),
                                  _p(smooth), _p(mask), _p(count), _stream()), "comm_mask")
    assert_close(smooth.cpu(), maps[:, 0], 1e-5, 1e-8, "smoothed map")
    near = (maps[:, 0] - 0.01).abs() < 1e-6
    diff = (mask.cpu() != masks[:, 0]) & ~near
    assert not diff.any()
    # rate: exact popcount before the ego override
    ones = (smooth.cpu() > 0.01)
    assert count.cpu().tolist() == [int(ones[:3].sum()), int(ones[3:].sum())]
    my_rate = (count.cpu().float() / (record_len.float() * h * w)).sum() / 2
    assert abs(float(my_rate) - float(rate)) < 1e-6 + float(near.sum()) / (h * w)


@pytest.mark.parametrize("c,n", [(64, 1), (64, 4), (128, 3), (256, 8), (64, 15)])
def test_pixel_attention(lib, c, n):
    from airv2x_perception_amd import _lib
    g = torch.Generator().manual_seed(c + n)
    h, w = 9, 21
    x = torch.randn(n, c, h, w, generator=g) * 1.5
    x[1:, :, ::3] = 0  # masked-out cells of collaborators stay in the softmax as zero vectors
    ref = orc.attention_fusion(x)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    out = torch.empty((h, w, c), device="cuda")
    ptrs = (c_void_p * n)(*[xd[j].data_ptr() for j in range(n)])
    _lib.check(lib.av2x_pixel_attn_fuse(ptrs, n, h * w, c, _p(out), _stream()), "attn")
    assert_close(out.permute(2, 0, 1).cpu(), ref, 1e-4, 1e-5, f"attn c={c} n={n}")


def test_apply_mask(lib):
    from airv2x_perception_amd import _lib
    x = torch.randn(3, 10, 12, 64)
    m = (torch.rand(3, 10, 12) > 0.5).float()
    xd, md = x.cuda(), m.cuda()
    _lib.check(lib.av2x_apply_mask(_p(xd), _p(md), 3, 120, 64, _stream()), "apply_mask")
    assert torch.equal(xd.cpu(), x * m.unsqueeze(-1))


# ---------------------------------------------------------------------------------------------- Winograd F(2x2,3x3)
WINO_TILES = {"32x128": 0x40000000 | (32 << 16) | 128, "64x64": 0x40000000 | (64 << 16) | 64, "32x64": 0x40000000 | (32 << 16) | 64,
              "32x64h": 0x40000000 | (32 << 16) | 64 | 0x8000,   # h: 8 positions per wave, two workgroups per CU
              "32x32q": 0x40000000 | (32 << 16) | 32 | 0x8000}   # q: 4 positions per wave, three or four workgroups per CU
WINO_CASES = [
    # n, h, w, cin, cout, relu          (odd H / W: half-empty tiles; 1-pixel-high maps; blocks that wrap rows and images)
    (2, 25, 88, 256, 256, 1),
    (3, 9, 13, 128, 128, 0),
    (1, 1, 7, 64, 64, 1),
    (2, 6, 1, 8, 64, 1),
    (5, 3, 5, 72, 192, 0),
]


def _wino_weights(lib, wp, cin, coutp):
    u = torch.empty(lib.av2x_wino_weight_bytes(cin, coutp) // 4, device="cuda")
    from airv2x_perception_amd import _lib
    _lib.check(lib.av2x_wino_pack_weights(_p(wp), cin, coutp, _p(u), _stream()), "av2x_wino_pack_weights")
    return u


@pytest.mark.parametrize("case", WINO_CASES)
def test_winograd_conv_matches_fp64_and_its_tilings_agree_bit_for_bit(lib, case):
    """The Winograd form against an fp64 convolution of the same fp32 operands: tolerance 2e-5 * max|ref| (the direct
    kernel's own error against fp64 is ~1e-5 at these sizes), and the three workgroup shapes give the SAME bits (each
    output is produced by one lane from the same chunk order)."""
    from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight
    n, h, w, cin, cout, relu = case
    g = torch.Generator().manual_seed(4321 + cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x.double(), wt.double(), None, padding=1) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    ref = (F.relu(ref) if relu else ref).permute(0, 2, 3, 1)
    wp, coutp = pack_conv_weight(wt)
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    u = _wino_weights(lib, wp.cuda(), cin, coutp)
    outs = {}
    for name, tile in WINO_TILES.items():
        if cout % (tile & 0x1ff):
            continue
        out = torch.full((n, h, w, cout), float("nan"), device="cuda")
        _conv_call(lib, xn, u, scale.cuda(), shift.cuda(), out, cin=cin, cout=cout, coutp=coutp, ks=3, stride=1, pad=1,
                   relu=relu, tile=tile)
        outs[name] = out.cpu()
        err = float((outs[name].double() - ref).abs().max())
        assert err <= 2e-5 * max(1.0, float(ref.abs().max())), (name, err)
    first = next(iter(outs.values()))
    assert all(torch.equal(first, o) for o in outs.values()), list(outs)


WINO4_TILE = 0x60000000 | (32 << 16) | 64
WINO4_CASES = [
    # n, h, w, cin, cout, relu, with residual      (H / W not multiples of 4: partly empty tiles; blocks that wrap rows and images)
    (2, 25, 88, 256, 256, 1, False),
    (4, 100, 352, 128, 64, 1, False),
    (3, 9, 13, 128, 128, 0, False),
    (1, 1, 7, 64, 64, 1, False),
    (2, 6, 1, 8, 64, 5, True),
    (1, 50, 176, 64, 128, 1, True),
    (1, 13, 18, 256, 128, 3, True),      # sigmoid + residual (ConvGRU update gate of V2VNet: the GENERAL instantiation)
    (1, 13, 18, 256, 128, 4, True),      # tanh GATED by the residual operand
]


@pytest.mark.parametrize("case", WINO4_CASES)
def test_winograd_f4x4_matches_fp64(lib, case):
    """F(4x4,3x3) (csrc/conv_wino4.inc) against an fp64 convolution of the same fp32 operands.  Its transform constants (up to 8)
    cost accuracy: the bound is 1e-4 * max|ref| (measured and printed next to F(2x2,3x3)'s error on the same case: ~4x), still
    fp32-rounding level -- the model-level goldens hold at their unchanged tolerances with the rule that selects it.  Run-to-run
    identical (every output is formed in one fixed order)."""
    from airv2x_perception_amd import _lib
    from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight
    n, h, w, cin, cout, relu, with_res = case
    g = torch.Generator().manual_seed(777 + cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    res = torch.randn(n, h, w, cout, generator=g) if with_res else None
    ref = (F.conv2d(x.double(), wt.double(), None, padding=1) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)).permute(0, 2, 3, 1)
    if relu == 1:
        ref = F.relu(ref)
    elif relu == 3:
        ref = torch.sigmoid(ref)
    elif relu == 4:
        ref = torch.tanh(ref)
    if with_res:
        ref = ref * res.double() if relu == 4 else ref + res.double()
    if relu == 5:
        ref = F.relu(ref)
    wp, coutp = pack_conv_weight(wt)
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    u4 = torch.empty(lib.av2x_wino4_weight_bytes(cin, coutp) // 4, device="cuda")
    _lib.check(lib.av2x_wino4_pack_weights(_p(wp.cuda()), cin, coutp, _p(u4), _stream()), "av2x_wino4_pack_weights")
    sc, sh, rd = scale.cuda(), shift.cuda(), (res.cuda() if with_res else None)

    def run(uw, tile):
        out = torch.full((n, h, w, cout), float("nan"), device="cuda")
        d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, in_ctot=cin, in_coff=0, ho=h, wo=w, cout=cout, coutp=coutp, out_ctot=cout, out_coff=0,
                          ks=3, stride=1, pad=1, relu=relu, mode=0, up=1, tile=tile, sk_wgs=0)
        _lib.check(lib.av2x_conv2d_res(byref(d), _p(xn), _p(uw), _p(sc), _p(sh), _p(rd), _p(out), _stream()), "conv")
        return out.cpu()
    o4 = run(u4, WINO4_TILE)
    e4 = float((o4.double() - ref).abs().max())
    o2 = run(_wino_weights(lib, wp.cuda(), cin, coutp), WINO_TILES["32x64h"])
    e2 = float((o2.double() - ref).abs().max())
    print(f"F(4x4,3x3) max err {e4:.2e}, F(2x2,3x3) {e2:.2e}, max|ref| {float(ref.abs().max()):.2f}")
    assert e4 <= 1e-4 * max(1.0, float(ref.abs().max())), e4
    assert torch.equal(run(u4, WINO4_TILE), o4)


def test_winograd_f4x4_rule_is_a_function_of_the_layer_and_map_size_only():
    from airv2x_perception_amd import _lib
    from airv2x_perception_amd.opencood_iface.engine import ConvLayer, Where2ComEngine
    mk = lambda cin, cout, ks=3, stride=1: ConvLayer(None, None, None, cin, cout, cout, ks, stride, 1 if ks == 3 else 0, 1, _lib.AV2X_CONV)
    lat = object.__new__(Where2ComEngine)           # the rule reads class constants and the two mode flags only
    rule = lat.wino4_rule
    assert rule(mk(256, 256), 4, 100, 352) and rule(mk(256, 256), 1, 100, 352) and rule(mk(384, 256), 2, 100, 352)
    # never a function of the number of agents in the launch (sharded frame == single frame, batch == single)
    assert all(rule(mk(256, 256), n, 100, 352) for n in range(1, 16)) and not any(rule(mk(128, 128), n, 50, 176) for n in range(1, 16))
    assert not rule(mk(256, 256), 8, 25, 88) and not rule(mk(64, 64), 4, 100, 352) and not rule(mk(256, 128), 4, 100, 352)
    assert not rule(mk(128, 256, stride=2), 4, 100, 352) and not rule(mk(256, 256, ks=1), 4, 100, 352)
    # throughput mode (whole frames in flight on one GPU): the 128 -> 128 layers at 50 x 176 and the 256 -> 256 layers at 25 x 88 join the
    # class -- again whatever n; the agent-sharded frame keeps the latency-mode classes even with frames in flight
    thr = object.__new__(Where2ComEngine)
    thr.throughput_mode = True
    rule_t = thr.wino4_rule
    assert all(rule_t(mk(128, 128), n, 50, 176) and rule_t(mk(256, 256), n, 25, 88) and rule_t(mk(256, 256), n, 100, 352) for n in range(1, 16))
    assert not rule_t(mk(64, 64), 4, 100, 352) and not rule_t(mk(256, 256), 4, 12, 44) and not rule_t(mk(128, 256, stride=2), 4, 100, 352)
    shd = object.__new__(Where2ComEngine)
    shd.throughput_mode, shd.sharded_frame = True, True
    assert not shd.wino4_rule(mk(128, 128), 1, 50, 176) and not shd.wino4_rule(mk(256, 256), 2, 25, 88) and shd.wino4_rule(mk(256, 256), 1, 100, 352)


def test_winograd_conv_channel_slices_and_argument_checks(lib):
    """Input read from a channel slice of a wider tensor, output written into a slice of a concatenated map (what the
    engine's fused buffers do); everything outside the slice stays untouched.  Unsupported uses fail loudly."""
    from airv2x_perception_amd import _lib
    from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight
    n, h, w, cin, cout = 2, 11, 14, 64, 128
    g = torch.Generator().manual_seed(99)
    xw = torch.randn(n, h, w, 96, generator=g)                  # the layer consumes channels 32..95
    wt = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
    shift = torch.randn(cout, generator=g) * 0.1
    ref = F.relu(F.conv2d(xw[..., 32:].permute(0, 3, 1, 2).double(), wt.double(), shift.double(), padding=1)).permute(0, 2, 3, 1)
    wp, coutp = pack_conv_weight(wt)
    u = _wino_weights(lib, wp.cuda(), cin, coutp)
    for tname in ("32x128", "32x64h", "32x32q"):
        out = torch.full((n, h, w, 160), -7.0, device="cuda")      # written at channels 16..143
        _conv_call(lib, xw.cuda(), u, None, shift.cuda(), out, cin=cin, cout=cout, coutp=coutp, ks=3, stride=1, pad=1, relu=1,
                   in_coff=32, out_ctot=160, out_coff=16, tile=WINO_TILES[tname])
        o = out.cpu()
        assert float((o[..., 16:144].double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), tname
        assert bool((o[..., :16] == -7.0).all()) and bool((o[..., 144:] == -7.0).all()), tname
    d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, in_ctot=96, in_coff=32, ho=h, wo=w, cout=cout, coutp=coutp, out_ctot=160,
                      out_coff=16, ks=3, stride=1, pad=1, relu=1, mode=0, up=1, tile=WINO_TILES["32x128"], sk_wgs=0)
    args = (_p(xw.cuda()), _p(u), None, _p(shift.cuda()), None, _p(out), _stream())
    for field, bad in (("stride", 2), ("ks", 1), ("relu", 2), ("mode", 2), ("tile", 0x40000000 | (48 << 16) | 128)):
        keep = getattr(d, field)
        setattr(d, field, bad)
        if field == "stride":
            d.ho, d.wo = (h + 1) // 2, (w + 1) // 2
        assert lib.av2x_conv2d_res(byref(d), *args) != 0, field
        assert b"Winograd" in lib.av2x_last_error() or b"winograd" in lib.av2x_last_error().lower(), field
        setattr(d, field, keep)
        d.ho, d.wo = h, w


def test_winograd_rule_is_a_function_of_the_layer_only():
    """Which layers run as Winograd must not depend on timings or on the number of agents in the launch."""
    from airv2x_perception_amd import _lib
    from airv2x_perception_amd.opencood_iface.engine import ConvLayer, Where2ComEngine
    mk = lambda cin, cout, ks=3, stride=1, pad=1, relu=1, mode=_lib.AV2X_CONV: ConvLayer(None, None, None, cin, cout, cout, ks, stride, pad, relu, mode)
    rule = Where2ComEngine.wino_rule
    assert rule(mk(256, 256)) and rule(mk(128, 128)) and rule(mk(384, 256)) and rule(mk(128, 128, relu=0))
    assert rule(mk(64, 64)) and rule(mk(256, 64)) and not rule(mk(128, 256, stride=2)) and not rule(mk(256, 256, ks=1, pad=0))
    assert not rule(mk(32, 64)) and not rule(mk(256, 96)) and not rule(mk(256, 256, relu=2)) and not rule(mk(256, 256, mode=_lib.AV2X_DECONV))
    assert rule(mk(512, 256, relu=3)) and rule(mk(512, 256, relu=4))


@pytest.mark.parametrize("act", [3, 4, 1])
def test_winograd_conv_gru_epilogues(lib, act):
    """sigmoid / tanh epilogues and the residual operand (added, or for tanh multiplied as a gate: the zero-hidden ConvGRU
    of V2VNet) against fp64."""
    from airv2x_perception_amd import _lib
    from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight
    n, h, w, cin, cout = 1, 13, 18, 256, 128
    g = torch.Generator().manual_seed(500 + act)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
    bias = torch.randn(cout, generator=g) * 0.1
    res = torch.rand(n, h, w, cout, generator=g)
    z = F.conv2d(x.double(), wt.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
    ref = {3: torch.sigmoid(z) + res.double(), 4: torch.tanh(z) * res.double(), 1: torch.relu(z) + res.double()}[act]
    wp, coutp = pack_conv_weight(wt)
    u = _wino_weights(lib, wp.cuda(), cin, coutp)
    out = torch.empty(n, h, w, cout, device="cuda")
    xn, rg, bg = x.permute(0, 2, 3, 1).contiguous().cuda(), res.cuda(), bias.cuda()
    outs = []
    for tname in ("32x128", "32x64h", "32x32q"):
        d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, in_ctot=cin, in_coff=0, ho=h, wo=w, cout=cout, coutp=coutp, out_ctot=cout,
                          out_coff=0, ks=3, stride=1, pad=1, relu=act, mode=0, up=1, tile=WINO_TILES[tname], sk_wgs=0)
        out.fill_(float("nan"))
        _lib.check(lib.av2x_conv2d_res(byref(d), _p(xn), _p(u), None, _p(bg), _p(rg), _p(out), _stream()), "conv")
        assert float((out.cpu().double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max())), tname
        outs.append(out.cpu().clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_winograd_random_shapes_against_the_direct_kernel(lib):
    """Seeded sweep over odd shapes (1..9 images, maps from 1x1 to 23x37, Cin a multiple of 8 from 8 to 136, Cout a multiple of
    32 or 64): every Winograd tiling that the channel count admits must agree bit for bit with the others and with the
    direct implicit-GEMM kernel (where Cin % 32 == 0 lets that kernel run) to 3e-5 * max|y| -- two fp32 algorithms with
    different summation orders, each ~1e-5 from fp64."""
    from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight
    rng = np.random.default_rng(20240607)
    for trial in range(24):
        n = int(rng.integers(1, 10))
        h, w = int(rng.integers(1, 24)), int(rng.integers(1, 38))
        cin = 8 * int(rng.integers(1, 18))
        cout = int(rng.choice([32, 64, 96, 128, 192]))
        relu = int(rng.integers(0, 2))
        g = torch.Generator().manual_seed(1000 + trial)
        x = torch.randn(n, h, w, cin, generator=g).cuda()
        wt = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
        scale, shift = (torch.rand(cout, generator=g) + 0.5).cuda(), (torch.randn(cout, generator=g) * 0.1).cuda()
        wp, coutp = pack_conv_weight(wt)
        wp = wp.cuda()
        u = _wino_weights(lib, wp, cin, coutp)
        outs = {}
        for name, tile in WINO_TILES.items():
            if cout % (tile & 0x1ff):
                continue
            out = torch.full((n, h, w, cout), float("nan"), device="cuda")
            _conv_call(lib, x, u, scale, shift, out, cin=cin, cout=cout, coutp=coutp, ks=3, stride=1, pad=1, relu=relu, tile=tile)
            outs[name] = out
        assert outs, (cin, cout)
        first = next(iter(outs.values()))
        assert not torch.isnan(first).any(), (trial, n, h, w, cin, cout)
        for name, o in outs.items():
            assert torch.equal(first, o), (trial, name, n, h, w, cin, cout)
        if cin % 32 == 0:
            ref = torch.empty_like(first)
            _conv_call(lib, x, wp, scale, shift, ref, cin=cin, cout=cout, coutp=coutp, ks=3, stride=1, pad=1, relu=relu, tile=0)
            err = float((first - ref).abs().max())
            assert err <= 3e-5 * max(1.0, float(ref.abs().max())), (trial, err, n, h, w, cin, cout)
