"""PMC target: the K = 256, N = 256 GEMM (M = 246 400) with the data-parallel and the persistent 128x128 8-wave tile."""
import sys
sys.path.insert(0, "/root/repo")
from ctypes import byref, c_void_p
import torch
from airv2x_perception_amd import _lib
from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight
lib = _lib.load()
st = c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: c_void_p(t.data_ptr())
M, cin, cout = 246400, 256, 256
n, h, w = 1, 100, M // 100
x = torch.randn(n, h, w, cin, device="cuda")
wp, coutp = pack_conv_weight(torch.randn(cout, cin, 1, 1) / cin ** 0.5)
wp = wp.cuda(); sc = torch.ones(cout, device="cuda"); sh = torch.zeros(cout, device="cuda")
y = torch.empty(n, h, w, cout, device="cuda")
for tile, g in (((128 << 16) | 128 | 0xc000, 0), ((128 << 16) | 128 | 0xd000, 512)):
    d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, in_ctot=cin, in_coff=0, ho=h, wo=w, cout=cout, coutp=coutp, out_ctot=cout, out_coff=0, ks=1, stride=1, pad=0, relu=0, mode=0, up=1, tile=tile, sk_wgs=g)
    for _ in range(4):
        _lib.check(lib.av2x_conv2d(byref(d), P(x), P(wp), P(sc), P(sh), P(y), st), "c")
torch.cuda.synchronize()
