"""F(4x4,3x3) vs F(2x2,3x3) on the 3x3 / stride-1 layers of the headline frame (tile candidates timed alone)."""
import os
import sys
from ctypes import byref, c_void_p

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from airv2x_perception_amd import _lib  # noqa: E402
from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight  # noqa: E402

lib = _lib.load()
P = lambda t: c_void_p(t.data_ptr()) if t is not None else c_void_p(0)
st = c_void_p(torch.cuda.current_stream().cuda_stream)
TILES = {"F2 32x64h": 0x40000000 | (32 << 16) | 64 | 0x8000, "F2 32x32q": 0x40000000 | (32 << 16) | 32 | 0x8000, "F4 32x64": 0x60000000 | (32 << 16) | 64}
SHAPES = ((4, 100, 352, 8, 64), (4, 100, 352, 64, 64), (4, 100, 352, 128, 64), (4, 100, 352, 256, 64)) if os.environ.get("W4_SWEEP") else ((4, 100, 352, 256, 256), (1, 100, 352, 256, 256), (4, 100, 352, 64, 64), (4, 50, 176, 128, 128), (3, 50, 176, 128, 128),
                           (4, 25, 88, 256, 256), (8, 100, 352, 256, 256), (8, 50, 176, 128, 128), (8, 25, 88, 256, 256))
for n, h, w, cin, cout in SHAPES:
    x = torch.randn(n, h, w, cin, device="cuda")
    wt = torch.randn(cout, cin, 3, 3) / np.sqrt(cin * 9)
    wp, coutp = pack_conv_weight(wt)
    wp = wp.cuda()
    u2 = torch.empty(lib.av2x_wino_weight_bytes(cin, coutp) // 4, device="cuda")
    u4 = torch.empty(lib.av2x_wino4_weight_bytes(cin, coutp) // 4, device="cuda")
    _lib.check(lib.av2x_wino_pack_weights(P(wp), cin, coutp, P(u2), st), "p2")
    _lib.check(lib.av2x_wino4_pack_weights(P(wp), cin, coutp, P(u4), st), "p4")
    shift = torch.zeros(cout, device="cuda")
    out = torch.empty(n, h, w, cout, device="cuda")
    row = []
    for name, tile in TILES.items():
        d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, in_ctot=cin, in_coff=0, ho=h, wo=w, cout=cout, coutp=coutp, out_ctot=cout, out_coff=0,
                          ks=3, stride=1, pad=1, relu=1, mode=0, up=1, tile=tile, sk_wgs=0)
        uw = u4 if name.startswith("F4") else u2
        call = lambda: _lib.check(lib.av2x_conv2d_res(byref(d), P(x), P(uw), None, P(shift), None, P(out), st), "conv")
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        fl = 2.0 * n * h * w * cout * 9 * cin
        row.append(f"{name} {us:7.1f} us ({fl / us / 1e6:6.1f} TF eff)")
    print(f"[{n}x{h}x{w} {cin}->{cout}]  " + "   ".join(row), flush=True)
