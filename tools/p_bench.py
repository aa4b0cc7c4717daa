#!/usr/bin/env python3
"""Persistent (0x1000) vs data-parallel schedule on the 3x3 layers of the frame.  Usage: python tools/p_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctypes import byref, c_void_p
import torch
from airv2x_perception_amd import _lib
from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight
lib = _lib.load()
st = c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: c_void_p(t.data_ptr())
for (n, h, w, cin, cout, stride) in ((4, 100, 352, 256, 256, 1), (4, 100, 352, 64, 64, 1), (4, 50, 176, 128, 128, 1), (4, 25, 88, 256, 256, 1), (1, 100, 352, 256, 256, 1)):
    ho, wo = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
    M = n * ho * wo
    x = torch.randn(n, h, w, cin, device="cuda")
    wp, coutp = pack_conv_weight(torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5)
    wp = wp.cuda(); sc = torch.ones(cout, device="cuda"); sh = torch.zeros(cout, device="cuda")
    y = torch.empty(n, ho, wo, cout, device="cuda")
    fl = 2.0 * M * cin * cout * 9
    line = f"M={M} K={9*cin} N={cout} ideal {fl/157.3e6:6.1f}us |"
    y0 = None
    cfgs = [("128x128w8d", (128 << 16) | 128 | 0xc000, 0), ("128x64w8d", (128 << 16) | 64 | 0xc000, 0), ("64x64d", (64 << 16) | 64 | 0x4000, 0)]
    for g in (256, 512):
        cfgs.append((f"128x128w8dP{g}", (128 << 16) | 128 | 0xd000, g))
    for g in (512, 768):
        cfgs.append((f"128x64w8dP{g}", (128 << 16) | 64 | 0xd000, g))
    cfgs.append(("64x64dP1024", (64 << 16) | 64 | 0x5000, 1024))
    for tn, tile, g in cfgs:
        if coutp % (tile & 0xfff):
            continue
        d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, in_ctot=cin, in_coff=0, ho=ho, wo=wo, cout=cout, coutp=coutp, out_ctot=cout,
                          out_coff=0, ks=3, stride=stride, pad=1, relu=1, mode=0, up=1, tile=tile, sk_wgs=g)
        call = lambda: _lib.check(lib.av2x_conv2d(byref(d), P(x), P(wp), P(sc), P(sh), P(y), st), "c")
        y.fill_(float("nan"))
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
