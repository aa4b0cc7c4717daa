#!/usr/bin/env python3
"""conv_igemm_bf16x3 (split-3, tile flag 0x0400) vs the fp32-MFMA kernel: speed and error against an fp64 convolution."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctypes import byref, c_void_p
import torch, torch.nn.functional as F
from airv2x_perception_amd import _lib
from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight, to_bf16x3_koct
lib = _lib.load()
st = c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: c_void_p(t.data_ptr())
for (n, h, w, cin, cout, ks, stride) in ((2, 23, 31, 64, 64, 3, 1), (1, 20, 36, 256, 256, 1, 1), (3, 17, 19, 128, 256, 3, 2), (1, 50, 176, 256, 256, 3, 1),
                                         (4, 100, 352, 256, 256, 3, 1), (4, 100, 352, 64, 64, 3, 1), (4, 50, 176, 128, 128, 3, 1), (4, 25, 88, 256, 256, 3, 1), (7, 100, 352, 256, 256, 1, 1),
                                         (4, 100, 352, 384, 256, 1, 1), (8, 100, 352, 256, 1024, 1, 1), (8, 100, 352, 1024, 256, 1, 1), (4, 200, 704, 64, 64, 3, 2), (4, 25, 88, 256, 2048, 1, 1), (4, 100, 352, 64, 128, 1, 1)):
    if os.environ.get("AV2X_S3_ONLY") and os.environ["AV2X_S3_ONLY"] != f"{n},{h},{w},{cin},{cout}":
        continue
    pad = 1 if ks == 3 else 0
    g = torch.Generator().manual_seed(cin + ks)
    x = torch.randn(n, cin, h, w, generator=g) * torch.exp(torch.randn(n, cin, 1, 1, generator=g))
    wt = torch.randn(cout, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5
    small = n * h * w < 20000
    ref = F.conv2d(x.double(), wt.double(), None, stride=stride, padding=pad) if small else None
    wp, coutp = pack_conv_weight(wt)
    w32, w3 = wp.cuda(), to_bf16x3_koct(wp).cuda()
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
    sc, sh = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
    fl = 2.0 * n * ho * wo * cout * ks * ks * cin
    line = f"n={n} {h}x{w} {cin}->{cout} k{ks} s{stride} ideal_f32 {fl/157.3e6:7.1f}us |"
    y32 = None
    for tn, tile, wgt in (("f32 128x128w8d", (128 << 16) | 128 | 0xc000, w32), ("f32 64x64d", (64 << 16) | 64 | 0x4000, w32), ("f32 g128x64w8", (128 << 16) | 64 | 0x8200, w32), ("f32 g128x128w8", (128 << 16) | 128 | 0x8200, w32),
                          ("x3 128x128w8", (128 << 16) | 128 | 0x8400, w3), ("x3 128x64w8", (128 << 16) | 64 | 0x8400, w3),
                          ("x3 128x128", (128 << 16) | 128 | 0x0400, w3), ("x3 128x64", (128 << 16) | 64 | 0x0400, w3), ("x3 64x64", (64 << 16) | 64 | 0x0400, w3),
                          ("x3p 128x128", (128 << 16) | 128 | 0x1400, w3), ("x3p 128x64", (128 << 16) | 64 | 0x1400, w3),
                          ("x3db 128x128w8", (128 << 16) | 128 | 0xc400, w3), ("x3db 128x64w8", (128 << 16) | 64 | 0xc400, w3),
                          ("x3db 128x128", (128 << 16) | 128 | 0x4400, w3), ("x3db 128x64", (128 << 16) | 64 | 0x4400, w3), ("x3db 64x64", (64 << 16) | 64 | 0x4400, w3)):
        if coutp % (tile & 0x1ff): continue
        if len(sys.argv) > 1 and not any(k in tn for k in sys.argv[1].split(",")): continue
        y = torch.full((n, ho, wo, cout), float("nan"), device="cuda")
        d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, in_ctot=cin, in_coff=0, ho=ho, wo=wo, cout=cout, coutp=coutp, out_ctot=cout, out_coff=0, ks=ks, stride=stride, pad=pad, relu=0, mode=0, up=1, tile=tile, sk_wgs=0)
        call = lambda: _lib.check(lib.av2x_conv2d(byref(d), P(xd), P(wgt), P(sc), P(sh), P(y), st), "c")
        call(); torch.cuda.synchronize()
        if y32 is None: y32 = y.clone()
        if small:
            e = (y.permute(0, 3, 1, 2).cpu().double() - ref).abs()
            err = f"max {float(e.max()):.1e} rms {float(e.pow(2).mean().sqrt()):.1e}"
        else:
            err = f"vs f32 max {float((y - y32).abs().max()):.1e}"
        for _ in range(3): call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): call()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        line += f"\n    {tn:15s}: {err:28s} {us:7.1f}us {fl/us/1e6:6.1f}TF"
    print(line, flush=True)
