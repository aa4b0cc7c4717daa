"""GPU: the bf16-ACTIVATION entry points of the V2X-ViT fusion in AMP mode (csrc/linear_bf16.hip, the bf16 instantiations of
csrc/v2xvit.hip; BASELINE configs[3] = V2X-ViT under autocast).

torch.autocast stores the outputs of nn.Linear / matmul as 16-bit tensors and keeps LayerNorm, softmax and `x + fn(x)` in fp32
(reference tools/train.py:118).  Here:
* av2x_linear_bf16 == an fp32 GEMM of the SAME bf16 operands (exact products, fp32 sums), bias / GELU / residual in fp32, result
  rounded once to bf16 (or kept fp32 with the fp32 residual);
* av2x_layernorm_bf16 == F.layer_norm in fp32, rounded once;
* the attention / split-attention kernels are the fp32 kernels instantiated on bf16 storage: on bf16-representable inputs their
  outputs equal the fp32 kernels' outputs rounded once (same arithmetic, same order) -- asserted BIT FOR BIT;
* the V2X-ViT encoder in bf16-activation mode against the fp32-activation AMP mode and the fp32 reference golden (drift bound of
  tests/test_amp.py).
"""
import ctypes
from ctypes import c_int32, c_void_p

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import assert_close, load_fixture

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _p(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def _st():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _pack(wt):
    from airv2x_perception_amd.opencood_iface.packing import interleave2_columns, pack_conv_weight, to_bf16_koct
    wp, _ = pack_conv_weight(wt.view(wt.shape[0], wt.shape[1], 1, 1))
    return interleave2_columns(to_bf16_koct(wp))


@pytest.mark.parametrize("m,cout,act,res,out16,ctot,coff", [
    (300, 1280, 0, False, True, 1280, 0),      # HGT projection; m not a multiple of the 128-token panel
    (517, 768, 0, False, True, 1280, 512),     # proj_kv: the k | v' columns of the last layer, written into the 1280-wide rows
    (1000, 2304, 0, False, True, 2304, 0),     # three window-attention QKVs
    (129, 256, 2, False, True, 256, 0),        # FFN first layer: GELU
    (640, 256, 0, True, False, 256, 0),        # a_linears / FFN second layer: fp32 out + fp32 residual (in place)
    (77, 256, 1, False, False, 256, 0),
    (256, 104, 0, False, True, 104, 0),        # cout not a multiple of 128: zero-padded columns are never stored
])
def test_linear_bf16_equals_fp32_gemm_of_the_same_operands(m, cout, act, res, out16, ctot, coff):
    from airv2x_perception_amd import _lib
    lib = _lib.load()
    g = _g(m + cout)
    a = (torch.randn(m, 256, generator=g) * 1.5).to(BF)
    wt = (torch.randn(cout, 256, generator=g) / 16).to(BF).float()
    b = torch.randn(cout, generator=g) * 0.2
    r = torch.randn(m, cout, generator=g) if res else None
    ref = a.double() @ wt.double().t() + b.double()
    if act == 1:
        ref = ref.clamp_min(0)
    elif act == 2:
        ref = F.gelu(ref)
    if res:
        ref = ref + r.double()
    w16, coutp = _pack(wt)
    assert coutp % 256 == 0
    ad, wd, bd = a.cuda(), w16.cuda(), b.cuda()
    if res:
        out = r.cuda().clone()            # in place: out == residual buffer, as the engine calls it
        rd = out
    else:
        out = torch.full((m, ctot), 7.0, device="cuda").to(BF if out16 else torch.float32)
        rd = None
    _lib.check(lib.av2x_linear_bf16(_p(ad), _p(wd), _p(bd), _p(rd), _p(out), m, 256, cout, coutp, 1 if out16 else 0, ctot, coff,
                                    cout if res else 0, 0, act, _st()), "av2x_linear_bf16")
    got = out.float().cpu()
    if out16:
        # one rounding to bf16 of a value that agrees with the float64 result to fp32 accumulation noise: within one bf16 ulp
        want = ref.float()
        err = (got[:, coff:coff + cout] - want).abs()
        tol = want.abs() * 2.0 ** -8 + 1e-6
        assert bool((err <= tol).all()), f"max err {float(err.max()):.3e}"
        exact = (got[:, coff:coff + cout] == want.to(BF).float()).float().mean()
        assert float(exact) > 0.98, f"only {float(exact):.4f} of the outputs are the correctly rounded value"
        if ctot > cout:      # columns outside the slice are untouched
            mask = torch.ones(ctot, dtype=torch.bool)
            mask[coff:coff + cout] = False
            assert bool((got[:, mask] == 7.0).all())
    else:
        assert_close(got.numpy(), ref.float().numpy(), 2e-5, 2e-5, "linear fp32 out")


def test_layernorm_bf16():
    from airv2x_perception_amd import _lib
    lib = _lib.load()
    g = _g(3)
    x = torch.randn(1001, 256, generator=g) * 3 + 0.5
    gm, bt = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g) * 0.1
    xd, gd, bd = x.cuda(), gm.cuda(), bt.cuda()
    y = torch.zeros(1001, 256, device="cuda", dtype=BF)
    _lib.check(lib.av2x_layernorm_bf16(_p(xd), _p(gd), _p(bd), _p(y), 1001, 256, 1e-5, _st()), "ln")
    y32 = torch.empty(1001, 256, device="cuda")
    _lib.check(lib.av2x_layernorm(_p(xd), _p(gd), _p(bd), _p(y32), 1001, 256, 1e-5, _st()), "ln")
    assert torch.equal(y, y32.to(BF))                       # the fp32 kernel's value, rounded once
    assert_close(y32.cpu().numpy(), F.layer_norm(x, (256,), gm, bt, 1e-5).numpy(), 1e-5, 1e-5, "layernorm")


def test_add_layernorm_bf16():
    """x += delta (bf16 Linear output), x written back, y = LayerNorm(x); delta NULL / y NULL forms."""
    from airv2x_perception_amd import _lib
    lib = _lib.load()
    g = _g(4)
    n = 777
    x = torch.randn(n, 256, generator=g) * 2
    d = torch.randn(n, 256, generator=g).to(BF)
    gm, bt = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g) * 0.1
    xd, dd, gd, bd = x.cuda(), d.cuda(), gm.cuda(), bt.cuda()
    y = torch.zeros(n, 256, device="cuda", dtype=BF)
    _lib.check(lib.av2x_add_layernorm_bf16(_p(xd), _p(dd), _p(gd), _p(bd), _p(y), n, 256, 1e-5, _st()), "add_ln")
    xs = x + d.float()
    assert torch.equal(xd.cpu(), xs)                         # one fp32 add per element
    y32 = torch.empty(n, 256, device="cuda")
    xs_d = xs.cuda()
    _lib.check(lib.av2x_layernorm(_p(xs_d), _p(gd), _p(bd), _p(y32), n, 256, 1e-5, _st()), "ln")
    assert torch.equal(y, y32.to(BF))
    x2 = x.cuda()
    _lib.check(lib.av2x_add_layernorm_bf16(_p(x2), _p(dd), None, None, None, n, 256, 1e-5, _st()), "add only")
    assert torch.equal(x2.cpu(), xs)


def test_hgt_attention_bf16_is_the_fp32_kernel_on_bf16_storage():
    from airv2x_perception_amd import _lib
    lib = _lib.load()
    n, hw = 5, 333
    g = _g(11)
    proj = (torch.randn(n, hw, 1280, generator=g)).to(BF).cuda()
    mask = (torch.rand(n, hw, generator=g) > 0.3).float()
    mask[0] = 1.0
    mask = mask.cuda()
    types = (c_int32 * n)(0, 1, 0, 1, 1)
    for nq in (n, 1):
        o16 = torch.zeros(n, hw, 256, device="cuda", dtype=BF)
        o32 = torch.zeros(n, hw, 256, device="cuda")
        p32 = proj.float()
        _lib.check(lib.av2x_hgt_attention_bf16(_p(proj), _p(mask), ctypes.cast(types, c_void_p), _p(o16), n, nq, hw, 8, 32, _st()), "hgt16")
        _lib.check(lib.av2x_hgt_attention_q(_p(p32), _p(mask), ctypes.cast(types, c_void_p), _p(o32), n, nq, hw, 8, 32, _st()), "hgt32")
        assert torch.equal(o16[:nq], o32[:nq].to(BF))


@pytest.mark.parametrize("heads,dh,ws,coff", [(16, 16, 2, 0), (8, 32, 4, 768), (4, 64, 4, 1536), (8, 32, 4 | 0x100, 768)])
def test_window_attention_bf16_is_the_fp32_kernel_on_bf16_storage(heads, dh, ws, coff):
    from airv2x_perception_amd import _lib
    lib = _lib.load()
    n, H, W = 3, 8, 12
    g = _g(heads + coff)
    qkv = torch.randn(n, H, W, 2304, generator=g).to(BF).cuda()
    w = ws & 0xff
    pos = torch.randn(2 * w - 1, 2 * w - 1, generator=g).cuda()
    o16 = torch.zeros(n, H, W, 256, device="cuda", dtype=BF)
    o32 = torch.zeros(n, H, W, 256, device="cuda")
    q32 = qkv.float()
    _lib.check(lib.av2x_window_attention_bf16(_p(qkv), 2304, coff, _p(pos), _p(o16), n, H, W, heads, dh, ws, _st()), "win16")
    _lib.check(lib.av2x_window_attention(_p(q32), 2304, coff, _p(pos), _p(o32), n, H, W, heads, dh, ws, _st()), "win32")
    assert torch.equal(o16, o32.to(BF))


@pytest.mark.parametrize("heads,dh,ws,coff,octot,ocoff", [(16, 16, 2, 0, 256, 0), (8, 32, 4, 768, 256, 0), (4, 64, 4, 1536, 768, 256)])
def test_window_attention_with_its_output_projection_equals_the_two_launches(heads, dh, ws, coff, octot, ocoff):
    """av2x_window_attention_linear_bf16 (the attention output stays in LDS as the A panel of to_out, mswin.py:52-96) against
    av2x_window_attention_bf16 + av2x_linear_bf16: same bits; the map must split into 4 x 16-pixel blocks."""
    from airv2x_perception_amd import _lib
    lib = _lib.load()
    n, H, W = 3, 12, 48
    g = _g(heads * 3 + coff)
    qkv = torch.randn(n, H, W, 2304, generator=g).to(BF).cuda()
    pos = torch.randn(2 * ws - 1, 2 * ws - 1, generator=g).cuda()
    wt = (torch.randn(256, 256, generator=g) / 16).to(BF).float()
    b = torch.randn(256, generator=g) * 0.2
    w16, coutp = _pack(wt)
    wd, bd = w16.cuda(), b.cuda()
    wat = torch.zeros(n, H, W, 256, device="cuda", dtype=BF)
    want = torch.full((n, H, W, octot), 7.0, device="cuda").to(BF)
    _lib.check(lib.av2x_window_attention_bf16(_p(qkv), 2304, coff, _p(pos), _p(wat), n, H, W, heads, dh, ws, _st()), "win16")
    _lib.check(lib.av2x_linear_bf16(_p(wat), _p(wd), _p(bd), None, _p(want), n * H * W, 256, 256, coutp, 1, octot, ocoff, 0, 0, 0, _st()), "lin")
    got = torch.full((n, H, W, octot), 7.0, device="cuda").to(BF)
    _lib.check(lib.av2x_window_attention_linear_bf16(_p(qkv), 2304, coff, _p(pos), _p(wd), _p(bd), _p(got), octot, ocoff, n, H, W, heads, dh, ws,
                                                     _st()), "win+lin")
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    # maps that do not split into 4 x 16 blocks, an unsupported (dim_head, window) pair
    assert lib.av2x_window_attention_linear_bf16(_p(qkv), 2304, coff, _p(pos), _p(wd), _p(bd), _p(got), octot, ocoff, n, H, 40, heads, dh, ws,
                                                 _st()) != 0
    assert lib.av2x_window_attention_linear_bf16(_p(qkv), 2304, coff, _p(pos), _p(wd), _p(bd), _p(got), octot, ocoff, n, H, W, 8, 32, 2,
                                                 _st()) != 0


@pytest.mark.parametrize("with_delta,order", [(True, (0, 1, 2)), (False, (2, 0, 1))])
def test_ln_qkv_window_attention_in_one_launch_equals_the_separate_launches(with_delta, order):
    """av2x_ln_qkv_window_attention_bf16 (LayerNorm -> QKV -> window attention -> to_out of a 4 x 16-pixel block in one workgroup, the
    2304-wide tensor never in HBM) against av2x_ln_linear_bf16 + 3 x av2x_window_attention_linear_bf16: same bits in the three branch maps."""
    from airv2x_perception_amd import _lib
    lib = _lib.load()
    n, H, W = 2, 8, 32
    cfg = [[(16, 16, 2), (8, 32, 4), (4, 64, 4)][i] for i in order]
    g = _g(17 + order[0])
    m = n * H * W
    x = torch.randn(m, 256, generator=g) * 2 + 0.3
    dl = torch.randn(m, 256, generator=g).to(BF)
    gm, bt = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g) * 0.1
    wq = (torch.randn(2304, 256, generator=g) / 16).to(BF).float()
    bq = torch.randn(2304, generator=g) * 0.1
    wo = [(torch.randn(256, 256, generator=g) / 16).to(BF).float() for _ in range(3)]
    bo = [torch.randn(256, generator=g) * 0.2 for _ in range(3)]
    pos = [torch.randn(2 * ws - 1, 2 * ws - 1, generator=g).cuda() for _, _, ws in cfg]
    (wq16, cqp) = _pack(wq)
    wo16 = [_pack(w_)[0] for w_ in wo]
    xd, dd, gd, bd, wqd, bqd = x.cuda(), dl.cuda(), gm.cuda(), bt.cuda(), wq16.cuda(), bq.cuda()
    wod, bod = [w_.cuda() for w_ in wo16], [b_.cuda() for b_ in bo]
    # --- separate launches
    qkv = torch.zeros(m, 2304, device="cuda", dtype=BF)
    x1 = xd.clone()
    _lib.check(lib.av2x_ln_linear_bf16(_p(x1), _p(dd) if with_delta else None, m if with_delta else 0, 0, _p(gd), _p(bd), 1e-5, _p(wqd), _p(bqd),
                                       0, 2304, cqp, None, None, 0, _p(qkv), 2304, 0, m, _st()), "ln+qkv")
    want = [torch.zeros(n, H, W, 256, device="cuda", dtype=BF) for _ in range(3)]
    for i, (h, dh, ws) in enumerate(cfg):
        _lib.check(lib.av2x_window_attention_linear_bf16(_p(qkv), 2304, 768 * i, _p(pos[i]), _p(wod[i]), _p(bod[i]), _p(want[i]), 256, 0, n, H, W,
                                                         h, dh, ws, _st()), "win+out")
    # --- one launch
    got = [torch.zeros(n, H, W, 256, device="cuda", dtype=BF) for _ in range(3)]
    w3 = torch.cat(wod, -2).contiguous()
    b3 = torch.cat(bod).contiguous()
    posv = (c_void_p * 3)(*[t.data_ptr() for t in pos])
    outv = (c_void_p * 3)(*[t.data_ptr() for t in got])
    hv, dv, wv = ((c_int32 * 3)(*[c[k] for c in cfg]) for k in range(3))
    x2 = xd.clone()
    args = [_p(x2), _p(dd) if with_delta else None, _p(gd), _p(bd), 1e-5, _p(wqd), _p(bqd), _p(w3), _p(b3), ctypes.cast(posv, c_void_p),
            ctypes.cast(outv, c_void_p), ctypes.cast(hv, c_void_p), ctypes.cast(dv, c_void_p), ctypes.cast(wv, c_void_p)]
    _lib.check(lib.av2x_ln_qkv_window_attention_bf16(*args, n, H, W, _st()), "mega")
    assert torch.equal(x2, xd)                                   # the stream is read only
    for i in range(3):
        assert torch.equal(got[i].view(torch.int16), want[i].view(torch.int16)), f"branch {i}"
    assert lib.av2x_ln_qkv_window_attention_bf16(*args, n, H, 40, _st()) != 0          # not 4 x 16 blocks
    bad = (c_int32 * 3)(8, 8, 8)
    assert lib.av2x_ln_qkv_window_attention_bf16(*args[:11], ctypes.cast(bad, c_void_p), *args[12:], n, H, W, _st()) != 0


def test_split_attention_bf16_is_the_fp32_kernel_on_bf16_storage():
    from airv2x_perception_amd import _lib
    lib = _lib.load()
    n, hw, C = 3, 700, 256
    g = _g(5)
    s = [torch.randn(n, hw, C, generator=g).to(BF).cuda() for _ in range(3)]
    s32 = [t.float() for t in s]
    gap16, gap32 = torch.zeros(n, C, device="cuda"), torch.zeros(n, C, device="cuda")
    scratch = torch.zeros(n, 128, C, device="cuda")
    _lib.check(lib.av2x_split_attn_gap_bf16(_p(s[0]), _p(s[1]), _p(s[2]), _p(gap16), _p(scratch), n, hw, C, _st()), "gap16")
    _lib.check(lib.av2x_split_attn_gap(_p(s32[0]), _p(s32[1]), _p(s32[2]), _p(gap32), _p(scratch), n, hw, C, _st()), "gap32")
    assert torch.equal(gap16, gap32)
    logits = torch.randn(n, 3 * C, generator=g).cuda()
    x = torch.randn(n, hw, C, generator=g).cuda()
    o16, o32 = x.clone(), x.clone()
    _lib.check(lib.av2x_split_attn_combine_bf16(_p(s[0]), _p(s[1]), _p(s[2]), _p(logits), _p(o16), _p(o16), n, hw, C, _st()), "c16")
    _lib.check(lib.av2x_split_attn_combine(_p(s32[0]), _p(s32[1]), _p(s32[2]), _p(logits), _p(o32), _p(o32), n, hw, C, _st()), "c32")
    assert torch.equal(o16, o32)
    # a pending residual add applied by the combine kernel == the add first (one fp32 add per element), then the plain combine
    dl = torch.randn(n, hw, C, generator=g).to(BF).cuda()
    od, oa = x.clone(), x + dl.float()
    _lib.check(lib.av2x_split_attn_combine_delta_bf16(_p(s[0]), _p(s[1]), _p(s[2]), _p(logits), _p(od), _p(dl), _p(od), n, hw, C, _st()), "cd")
    _lib.check(lib.av2x_split_attn_combine_bf16(_p(s[0]), _p(s[1]), _p(s[2]), _p(logits), _p(oa), _p(oa), n, hw, C, _st()), "c16")
    assert torch.equal(od, oa)


HALO_TILE = 0x10000000 | (128 << 16) | 128 | 0x0800


@pytest.mark.parametrize("n,h,w,cin,cout,ks,relu,out16,ctot,coff", [
    (2, 25, 88, 256, 256, 3, 1, True, 256, 0),       # block 2 of the trunk
    (1, 50, 176, 128, 128, 3, 1, True, 128, 0),
    (3, 9, 13, 64, 128, 3, 0, False, 128, 0),        # partly empty 8 x 16 blocks, fp32 out (the last shrink layer)
    (2, 20, 36, 384, 256, 1, 1, True, 256, 0),       # 1x1 over the concatenated map
    (1, 8, 16, 128, 256, 3, 1, True, 384, 128),      # output written into a channel slice
    (1, 1, 5, 64, 96, 3, 1, True, 96, 0),            # one-row map; cout < coutp (zero-padded weight columns are never stored)
])
def test_halo_conv_on_bf16_activations_equals_fp32_conv_of_the_same_operands(n, h, w, cin, cout, ks, relu, out16, ctot, coff):
    """csrc/conv_halo_bf16.inc: bf16 input activations and weights (products exact in fp32), fp32 accumulation, folded BN + ReLU,
    one rounding to bf16 (or fp32 out) -- against a float64 convolution of the same bf16 operands."""
    from ctypes import byref
    from airv2x_perception_amd import _lib
    from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight, to_bf16_koct
    lib = _lib.load()
    g = _g(cin + cout + h + ks)
    x = torch.randn(n, cin, h, w, generator=g).to(BF)
    wt = (torch.randn(cout, cin, ks, ks, generator=g) / np.sqrt(cin * ks * ks)).to(BF)
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x.double(), wt.double(), None, padding=ks // 2) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    ref = (F.relu(ref) if relu else ref).permute(0, 2, 3, 1)
    wp, coutp = pack_conv_weight(wt.float())
    coutp128 = (coutp + 127) // 128 * 128
    if coutp128 != coutp:       # the halo tile walks 128 columns at a time: pad the packed weights with zero columns
        wp = torch.cat([wp, torch.zeros(wp.shape[0], wp.shape[1], coutp128 - coutp, 4)], 2)
    wh = to_bf16_koct(wp).cuda()
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    out = torch.full((n, h, w, ctot), 7.0, device="cuda").to(BF if out16 else torch.float32)
    sc, sh = scale.cuda(), shift.cuda()
    d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, in_ctot=cin, in_coff=0, ho=h, wo=w, cout=cout, coutp=coutp128, out_ctot=ctot, out_coff=coff,
                      ks=ks, stride=1, pad=ks // 2, relu=relu, mode=0, up=1, tile=HALO_TILE, sk_wgs=0, act16=1 | (2 if out16 else 0))
    _lib.check(lib.av2x_conv2d_res(byref(d), _p(xd), _p(wh), _p(sc), _p(sh), None, _p(out), _st()), "halo conv")
    got = out.float().cpu()[..., coff:coff + cout].double()
    if out16:
        err = (got - ref).abs()
        assert bool((err <= ref.abs() * 2.0 ** -8 + 1e-5).all()), float(err.max())
        assert float((got == ref.float().to(BF).double()).float().mean()) > 0.97
    else:
        assert_close(got.numpy(), ref.numpy(), 2e-5, 2e-5, "halo conv fp32 out")
    if ctot > cout:
        o = out.float().cpu()
        assert bool((o[..., :coff] == 7.0).all()) and bool((o[..., coff + cout:] == 7.0).all())
    # argument checks: fp32 input, stride 2, a residual
    d.act16 = 2
    assert lib.av2x_conv2d_res(byref(d), _p(xd), _p(wh), _p(sc), _p(sh), None, _p(out), _st()) != 0


@pytest.mark.parametrize("name", ["v2xvit_small_n3", "v2xvit_full_n8"])
def test_v2xvit_bf16_activations_against_fp32_activation_amp_and_the_reference(name):
    """The same AMP frame with bf16 and with fp32 activation storage: both within the drift bound of tests/test_amp.py against the
    reference's fp32 outputs, and close to each other (the extra roundings are of the stored Linear / attention outputs)."""
    from airv2x_perception_amd.opencood_iface import Airv2xV2XVit as M
    import tests.test_v2xvit as tv
    from tests.test_amp import _drift
    fx = load_fixture(name)
    hy, args, sd, dd = tv._case(fx)
    model = M(args)
    model.load_state_dict(sd)
    model = model.to("cuda").eval()
    eng = model.engine()
    model.amp = True
    assert eng.bf16_activations is True
    a16 = {k: v.clone() for k, v in model(dd).items() if torch.is_tensor(v)}
    eng.bf16_activations = False
    a32 = {k: v.clone() for k, v in model(dd).items() if torch.is_tensor(v)}
    eng.bf16_activations = True
    again = model(dd)
    assert torch.equal(again["psm"], a16["psm"])            # run-to-run identical
    assert not torch.equal(a16["psm"], a32["psm"])
    d16, d32 = _drift(a16, fx), _drift(a32, fx)
    print(f"[{name}] drift vs the fp32 reference: bf16 activations {d16}, fp32 activations {d32}")
    assert all(v < 6e-2 for v in d16.values()), d16
    for k in ("psm", "rm", "obj"):
        scale = float(a32[k].abs().max())
        assert float((a16[k] - a32[k]).abs().max()) < 6e-2 * scale


@pytest.mark.parametrize("m,add_rows,cout,act,ctot,coff,ffn", [
    (300, 300, 1280, 0, 1280, 0, False),       # ln1 -> HGT projection, every row with a pending residual; m not a multiple of 64
    (517, 200, 768, 0, 1280, 512, False),      # proj_kv slice; the pending residual ends inside a panel (200 = 3 x 64 + 8)
    (1000, 0, 2304, 0, 2304, 0, False),        # ln2 -> the three window QKVs, nothing pending
    (129, 64, 256, 2, 256, 0, True),           # FeedForward: LayerNorm -> Linear + GELU -> Linear, hidden panel in LDS
    (640, 640, 256, 2, 256, 0, True),
    (64, 0, 256, 2, 256, 0, True),
])
def test_ln_linear_bf16_equals_the_separate_launches_bit_for_bit(m, add_rows, cout, act, ctot, coff, ffn):
    """av2x_ln_linear_bf16 (LayerNorm and the pending residual add folded into the Linear's panel load, FeedForward's second Linear on
    the LDS-resident hidden panel) against av2x_add_layernorm_bf16 + av2x_linear_bf16 (+ av2x_linear_bf16): same bits in `out` and in
    the updated x (reference base_transformer.py:12-37)."""
    from airv2x_perception_amd import _lib
    lib = _lib.load()
    g = _g(m * 7 + cout + add_rows)
    x = torch.randn(m, 256, generator=g) * 2 + 0.3
    dl = torch.randn(m, 256, generator=g).to(BF)
    gm, bt = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g) * 0.1
    wt = (torch.randn(cout, 256, generator=g) / 16).to(BF).float()
    b = torch.randn(cout, generator=g) * 0.2
    wt2 = (torch.randn(256, 256, generator=g) / 16).to(BF).float()
    b2 = torch.randn(256, generator=g) * 0.2
    (w16, coutp), (w216, _) = _pack(wt), _pack(wt2)
    gd, bd, wd, bbd, w2d, b2d, dd = gm.cuda(), bt.cuda(), w16.cuda(), b.cuda(), w216.cuda(), b2.cuda(), dl.cuda()
    # --- separate launches
    xs = x.cuda()
    xn = torch.zeros(m, 256, device="cuda", dtype=BF)
    if add_rows:
        _lib.check(lib.av2x_add_layernorm_bf16(_p(xs), _p(dd), _p(gd), _p(bd), _p(xn), add_rows, 256, 1e-5, _st()), "add_ln")
    if m > add_rows:
        _lib.check(lib.av2x_add_layernorm_bf16(_p(xs[add_rows:]), None, _p(gd), _p(bd), _p(xn[add_rows:]), m - add_rows, 256, 1e-5, _st()), "ln")
    want = torch.full((m, ctot), 7.0, device="cuda").to(BF)
    if ffn:
        hid = torch.zeros(m, 256, device="cuda", dtype=BF)
        _lib.check(lib.av2x_linear_bf16(_p(xn), _p(wd), _p(bbd), None, _p(hid), m, 256, 256, coutp, 1, 256, 0, 0, 0, act, _st()), "ff1")
        _lib.check(lib.av2x_linear_bf16(_p(hid), _p(w2d), _p(b2d), None, _p(want), m, 256, 256, 256, 1, ctot, coff, 0, 0, 0, _st()), "ff2")
    else:
        _lib.check(lib.av2x_linear_bf16(_p(xn), _p(wd), _p(bbd), None, _p(want), m, 256, cout, coutp, 1, ctot, coff, 0, 0, act, _st()), "lin")
    # --- one launch
    xf = x.cuda()
    got = torch.full((m, ctot), 7.0, device="cuda").to(BF)
    _lib.check(lib.av2x_ln_linear_bf16(_p(xf), _p(dd) if add_rows else None, add_rows, 1, _p(gd), _p(bd), 1e-5, _p(wd), _p(bbd), act, cout, coutp,
                                       _p(w2d) if ffn else None, _p(b2d) if ffn else None, 0, _p(got), ctot, coff, m, _st()), "ln_linear")
    assert torch.equal(xf, xs)
    assert torch.equal(xf[add_rows:].cpu(), x[add_rows:])           # rows without a pending residual are not rewritten
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    # argument checks
    # write_back_x = 0: the same output, x untouched (the add stays pending: av2x_split_attn_combine_delta_bf16 applies it)
    x0 = x.cuda()
    got0 = torch.full((m, ctot), 7.0, device="cuda").to(BF)
    _lib.check(lib.av2x_ln_linear_bf16(_p(x0), _p(dd) if add_rows else None, add_rows, 0, _p(gd), _p(bd), 1e-5, _p(wd), _p(bbd), act, cout, coutp,
                                       _p(w2d) if ffn else None, _p(b2d) if ffn else None, 0, _p(got0), ctot, coff, m, _st()), "ln_linear")
    assert torch.equal(x0.cpu(), x) and torch.equal(got0.view(torch.int16), want.view(torch.int16))
    assert lib.av2x_ln_linear_bf16(_p(xf), None, 5, 1, _p(gd), _p(bd), 1e-5, _p(wd), _p(bbd), act, cout, coutp, None, None, 0, _p(got), ctot, coff,
                                   m, _st()) != 0                  # pending rows without delta
    if not ffn and cout != 256:
        assert lib.av2x_ln_linear_bf16(_p(xf), None, 0, 1, _p(gd), _p(bd), 1e-5, _p(wd), _p(bbd), act, cout, coutp, _p(w2d), _p(b2d), 0, _p(got),
                                       ctot, coff, m, _st()) != 0  # the fused second Linear needs a hidden width of 256


@pytest.mark.parametrize("n,hw,add_agents", [(3, 128, 3), (2, 320, 0), (1, 64, 1)])
def test_combine_in_the_feedforward_launch_equals_the_separate_launches(n, hw, add_agents):
    """av2x_combine_ln_linear_bf16 (SplitAttn's combine computed in the FeedForward launch's panel load) against
    av2x_split_attn_combine(_delta)_bf16 + av2x_ln_linear_bf16: same bits in the updated x and in the FeedForward output."""
    from airv2x_perception_amd import _lib
    lib = _lib.load()
    C, m = 256, n * hw
    g = _g(n * 11 + hw)
    x = torch.randn(m, C, generator=g) * 2
    dl = torch.randn(m, C, generator=g).to(BF)
    s = [torch.randn(m, C, generator=g).to(BF).cuda() for _ in range(3)]
    logits = torch.randn(n, 3 * C, generator=g).cuda()
    gm, bt = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    w1 = (torch.randn(C, C, generator=g) / 16).to(BF).float()
    w2 = (torch.randn(C, C, generator=g) / 16).to(BF).float()
    b1, b2 = torch.randn(C, generator=g) * 0.2, torch.randn(C, generator=g) * 0.2
    (w1p, cp), (w2p, _) = _pack(w1), _pack(w2)
    gd, bd, w1d, w2d, b1d, b2d, dd = gm.cuda(), bt.cuda(), w1p.cuda(), w2p.cuda(), b1.cuda(), b2.cuda(), dl.cuda()
    add = add_agents * hw
    # --- separate launches
    xs = x.cuda()
    if add:
        _lib.check(lib.av2x_split_attn_combine_delta_bf16(_p(s[0]), _p(s[1]), _p(s[2]), _p(logits), _p(xs), _p(dd), _p(xs), n, hw, C, _st()), "cd")
    else:
        _lib.check(lib.av2x_split_attn_combine_bf16(_p(s[0]), _p(s[1]), _p(s[2]), _p(logits), _p(xs), _p(xs), n, hw, C, _st()), "c")
    want = torch.zeros(m, C, device="cuda", dtype=BF)
    _lib.check(lib.av2x_ln_linear_bf16(_p(xs), None, 0, 1, _p(gd), _p(bd), 1e-5, _p(w1d), _p(b1d), 2, C, cp, _p(w2d), _p(b2d), 0, _p(want), C, 0, m,
                                       _st()), "ffn")
    # --- one launch (the output overwrites the pending delta rows in place, as the engine calls it)
    xf = x.cuda()
    out = dd.clone()
    _lib.check(lib.av2x_combine_ln_linear_bf16(_p(xf), _p(out) if add else None, add, _p(s[0]), _p(s[1]), _p(s[2]), _p(logits), hw, _p(gd), _p(bd),
                                               1e-5, _p(w1d), _p(b1d), 2, C, cp, _p(w2d), _p(b2d), 0, _p(out), C, 0, m, _st()), "combine+ffn")
    assert torch.equal(xf, xs)
    assert torch.equal(out.view(torch.int16), want.view(torch.int16))
    # hw must be a multiple of 64 (a 64-token panel belongs to one agent) and divide m
    assert lib.av2x_combine_ln_linear_bf16(_p(xf), None, 0, _p(s[0]), _p(s[1]), _p(s[2]), _p(logits), 96, _p(gd), _p(bd), 1e-5, _p(w1d), _p(b1d), 2,
                                           C, cp, _p(w2d), _p(b2d), 0, _p(out), C, 0, m, _st()) != 0


@pytest.mark.parametrize("name", ["v2xvit_small_n3", "v2xvit_full_n8"])
def test_v2xvit_fused_layernorm_linear_frame_equals_the_unfused_frame(name):
    """The bf16-activation frame with LayerNorm folded into the Linears and the window attention into its output projection (default)
    and with all of them as separate launches: same bits."""
    from airv2x_perception_amd.opencood_iface import Airv2xV2XVit as M
    import tests.test_v2xvit as tv
    fx = load_fixture(name)
    hy, args, sd, dd = tv._case(fx)
    model = M(args)
    model.load_state_dict(sd)
    model = model.to("cuda").eval()
    eng = model.engine()
    model.amp = True
    assert eng.fuse_ln is True and eng.fuse_window_out is True and eng.fuse_qkv_window is True and eng.fuse_combine_ffn is True
    fused = {k: v.clone() for k, v in model(dd).items() if torch.is_tensor(v)}
    try:
        eng.fuse_qkv_window = eng.fuse_combine_ffn = False   # LayerNorm -> QKV and attention -> to_out as separate fused launches; own combine
        mid = {k: v.clone() for k, v in model(dd).items() if torch.is_tensor(v)}
        eng.fuse_ln = eng.fuse_window_out = False         # every LayerNorm, Linear and attention its own launch
        plain = {k: v.clone() for k, v in model(dd).items() if torch.is_tensor(v)}
    finally:
        eng.fuse_ln = eng.fuse_window_out = eng.fuse_qkv_window = eng.fuse_combine_ffn = True
    for k in ("psm", "rm", "obj"):
        assert torch.equal(fused[k], plain[k]), k
        assert torch.equal(mid[k], plain[k]), k
