import sys
sys.path.insert(0, "/root/repo")
from ctypes import byref, c_void_p
import torch
from airv2x_perception_amd import _lib
from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight
lib = _lib.load()
st = c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: c_void_p(t.data_ptr())
for (M, cin, cout, ks) in ((246400, 256, 256, 1), (140800, 256, 256, 3)):
    n, h, w = 1, 100, M // 100
    x = torch.randn(n, h, w, cin, device="cuda")
    wp, coutp = pack_conv_weight(torch.randn(cout, cin, ks, ks) / (cin * ks * ks) ** 0.5)
    wp = wp.cuda(); sc = torch.ones(cout, device="cuda"); sh = torch.zeros(cout, device="cuda")
    y = torch.empty(n, h, w, cout, device="cuda")
    for tile in ((128 << 16) | 64 | 0xc000, (128 << 16) | 128 | 0xc000):
        d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, in_ctot=cin, in_coff=0, ho=h, wo=w, cout=cout, coutp=coutp, out_ctot=cout, out_coff=0, ks=ks, stride=1, pad=ks // 2, relu=0, mode=0, up=1, tile=tile, sk_wgs=0)
        for _ in range(4):
            _lib.check(lib.av2x_conv2d(byref(d), P(x), P(wp), P(sc), P(sh), P(y), st), "c")
torch.cuda.synchronize()
