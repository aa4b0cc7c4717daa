import sys
sys.path.insert(0, "/root/repo")
from ctypes import byref, c_void_p
import torch
from airv2x_perception_amd import _lib
from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight
lib = _lib.load()
st = c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: c_void_p(t.data_ptr())
for (M, cin, cout) in ((246400, 256, 256), (246400, 256, 768), (140800, 64, 128), (140800, 384, 256)):
    n, h, w = 1, 100, M // 100
    x = torch.randn(n, h, w, cin, device="cuda")
    wp, coutp = pack_conv_weight(torch.randn(cout, cin, 1, 1) / cin ** 0.5)
    wp = wp.cuda(); sc = torch.ones(cout, device="cuda"); sh = torch.zeros(cout, device="cuda")
    y = torch.empty(n, h, w, cout, device="cuda")
    fl = 2.0 * M * cin * cout
    line = f"M={M} K={cin} N={cout} ideal {fl/157.3e6:6.1f}us |"
    y0 = None
    for tn, tile, g in (("128x64w8d", (128 << 16) | 64 | 0xc000, 0), ("128x128w8d", (128 << 16) | 128 | 0xc000, 0),
                        ("128x64w8dP768", (128 << 16) | 64 | 0xd000, 768), ("128x64w8dP512", (128 << 16) | 64 | 0xd000, 512),
                        ("128x128w8dP512", (128 << 16) | 128 | 0xd000, 512), ("128x128w8dP256", (128 << 16) | 128 | 0xd000, 256),
                        ("64x64dP1024", (64 << 16) | 64 | 0x5000, 1024), ("128x128dP512", (128 << 16) | 128 | 0x5000, 512),
                        ("128x64dP768", (128 << 16) | 64 | 0x5000, 768)):
        d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, in_ctot=cin, in_coff=0, ho=h, wo=w, cout=cout, coutp=coutp, out_ctot=cout, out_coff=0, ks=1, stride=1, pad=0, relu=0, mode=0, up=1, tile=tile, sk_wgs=g)
        y.fill_(float("nan"))
        call = lambda: _lib.check(lib.av2x_conv2d(byref(d), P(x), P(wp), P(sc), P(sh), P(y), st), "c")
        for _ in range(3): call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        if y0 is None:
            y0 = y.clone()
        line += f" {tn}:{us:6.1f}us {fl/us/1e6:5.1f}TF{'' if torch.equal(y, y0) else ' MISMATCH'} |"
    print(line, flush=True)
