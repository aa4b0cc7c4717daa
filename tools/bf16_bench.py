import sys
sys.path.insert(0, "/root/repo")
from ctypes import byref, c_void_p
import torch, torch.nn.functional as F
from airv2x_perception_amd import _lib
from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight, to_bf16_koct
lib = _lib.load()
st = c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: c_void_p(t.data_ptr())
for (n, h, w, cin, cout, ks, stride) in ((2, 23, 31, 64, 64, 3, 1), (1, 20, 36, 256, 256, 1, 1), (3, 17, 19, 128, 256, 3, 2), (1, 100, 352, 256, 256, 3, 1), (7, 100, 352, 256, 256, 1, 1), (7, 100, 352, 256, 768, 1, 1)):
    pad = 1 if ks == 3 else 0
    g = torch.Generator().manual_seed(cin + ks)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5
    xb, wb = x.bfloat16().float(), wt.bfloat16().float()
    ref = F.relu(F.conv2d(xb, wb, None, stride=stride, padding=pad)) if n * h * w < 50000 else None
    wp, coutp = pack_conv_weight(wt)
    wh = to_bf16_koct(wp).cuda()
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
    sc, sh = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
    fl = 2.0 * n * ho * wo * cout * ks * ks * cin
    line = f"n={n} {h}x{w} {cin}->{cout} k{ks} s{stride} |"
    for tn, tile in (("128x128w8", (128 << 16) | 128 | 0x8800), ("128x64w8", (128 << 16) | 64 | 0x8800), ("128x128", (128 << 16) | 128 | 0x0800), ("128x64", (128 << 16) | 64 | 0x0800), ("64x64", (64 << 16) | 64 | 0x0800)):
        if coutp % (tile & 0x7ff): continue
        y = torch.full((n, ho, wo, cout), float("nan"), device="cuda")
        d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, in_ctot=cin, in_coff=0, ho=ho, wo=wo, cout=cout, coutp=coutp, out_ctot=cout, out_coff=0, ks=ks, stride=stride, pad=pad, relu=1, mode=0, up=1, tile=tile, sk_wgs=0)
        call = lambda: _lib.check(lib.av2x_conv2d(byref(d), P(xd), P(wh), P(sc), P(sh), P(y), st), "c")
        call(); torch.cuda.synchronize()
        err = float((y.permute(0, 3, 1, 2).cpu() - ref).abs().max()) if ref is not None else -1
        for _ in range(3): call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): call()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        line += f" {tn}: err={err:.1e} {us:7.1f}us {fl/us/1e6:6.1f}TF |"
    print(line, flush=True)
