#!/usr/bin/env python3
"""Micro-benchmark of av2x_conv2d on the layer shapes of the Where2Comm frame (N agents), per tile
configuration.  Usage: python tools/conv_bench.py [--agents 4] [--iters 20] [--layers L8,L6] [--tiles 128x128,64x64]"""
import argparse
import os
import sys
from ctypes import byref, c_void_p

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from airv2x_perception_amd import _lib  # noqa: E402
from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight  # noqa: E402

# name: (h, w, cin, cout, ks, stride)  input dims per agent
LAYERS = {
    "L1_b0_first": (200, 704, 64, 64, 3, 2),
    "L2_b0_rest": (100, 352, 64, 64, 3, 1),
    "L3_b1_first": (100, 352, 64, 128, 3, 2),
    "L4_b1_rest": (50, 176, 128, 128, 3, 1),
    "L5_b2_first": (50, 176, 128, 256, 3, 2),
    "L6_b2_rest": (25, 88, 256, 256, 3, 1),
    "L7_shrink1x1": (100, 352, 384, 256, 1, 1),
    "L8_shrink3x3": (100, 352, 256, 256, 3, 1),
}
TILES = [(128, 128), (128, 64), (64, 64), (64, 128), (128, 128 | 0x8000), (128, 64 | 0x8000),
         (128, 128 | 0x4000), (128, 64 | 0x4000), (64, 64 | 0x4000), (64, 128 | 0x4000), (128, 128 | 0xc000), (128, 64 | 0xc000)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=4)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--layers", default="")
    ap.add_argument("--tiles", default="")
    a = ap.parse_args()
    lib = _lib.load()
    def parse_tile(t):
        bm, rest = t.split("x")
        bn = int(rest.rstrip("wd"))
        return int(bm), bn | (0x8000 if "w" in rest else 0) | (0x4000 if "d" in rest else 0)
    tiles = [parse_tile(t) for t in a.tiles.split(",")] if a.tiles else TILES
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    for name, (h, w, cin, cout, ks, stride) in LAYERS.items():
        if a.layers and not any(name.startswith(p) for p in a.layers.split(",")):
            continue
        n = a.agents
        pad = 1 if ks == 3 else 0
        ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
        x = torch.randn(n, h, w, cin, device="cuda")
        wt = torch.randn(cout, cin, ks, ks) / (cin * ks * ks) ** 0.5
        wp, coutp = pack_conv_weight(wt)
        wp = wp.cuda()
        sc, sh = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
        y = torch.empty(n, ho, wo, cout, device="cuda")
        flops = 2.0 * n * ho * wo * cout * ks * ks * cin
        line = f"{name:14s} M={n*ho*wo:7d} K={ks*ks*cin:5d} N={cout:4d} {flops/1e9:7.1f} GF |"
        for bm, bn in tiles:
            if coutp % (bn & 0x3fff):
                continue
            d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, in_ctot=cin, in_coff=0, ho=ho, wo=wo, cout=cout, coutp=coutp,
                              out_ctot=cout, out_coff=0, ks=ks, stride=stride, pad=pad, relu=1, mode=0, up=1,
                              tile=(bm << 16) | bn)
            call = lambda: _lib.check(lib.av2x_conv2d(byref(d), c_void_p(x.data_ptr()), c_void_p(wp.data_ptr()),
                                                      c_void_p(sc.data_ptr()), c_void_p(sh.data_ptr()),
                                                      c_void_p(y.data_ptr()), st), "conv")
            for _ in range(3):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                call()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.iters
            line += f" {bm}x{bn & 0x3fff}{'w8' if bn & 0x8000 else ''}{'d' if bn & 0x4000 else ''}:{us:6.1f}us {flops/us/1e6:5.1f}TF|"
        print(line, flush=True)


if __name__ == "__main__":
    main()
