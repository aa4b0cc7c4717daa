#!/usr/bin/env python3
"""Stream-K sweep of av2x_conv2d_sk on the small-M layers of the frame: for every (tile, sk_wgs) prints
time / TFLOP/s and the max |diff| against the data-parallel schedule of the same tile.
Usage: python tools/sk_bench.py [--iters 20]"""
import argparse, os, sys
from ctypes import byref, c_void_p
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from airv2x_perception_amd import _lib  # noqa: E402
from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight  # noqa: E402

# name: (n, h, w, cin, cout, ks, stride)
LAYERS = {
    "b2_rest_n4": (4, 25, 88, 256, 256, 3, 1),
    "b2_rest_n3": (3, 25, 88, 256, 256, 3, 1),
    "b1_rest_n4": (4, 50, 176, 128, 128, 3, 1),
    "b1_rest_n3": (3, 50, 176, 128, 128, 3, 1),
    "b0_rest_n4": (4, 100, 352, 64, 64, 3, 1),
    "b2_first_n4": (4, 50, 176, 128, 256, 3, 2),
    "shrink3_n1": (1, 100, 352, 256, 256, 3, 1),
    "shrink3_n4": (4, 100, 352, 256, 256, 3, 1),
}
TILES = {"128x64w8d": (128, 64 | 0xc000), "g128x64w8": (128, 64 | 0x8200), "g128x64w8s3": (128, 64 | 0xc200),
         "128x64d": (128, 64 | 0x4000), "g128x64": (128, 64 | 0x0200),
         "128x128w8d": (128, 128 | 0xc000), "g128x128w8": (128, 128 | 0x8200), "64x64d": (64, 64 | 0x4000), "g64x64": (64, 64 | 0x0200)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--layers", default="")
    ap.add_argument("--wgs", default="256,512,768,1024")
    ap.add_argument("--tiles", default="")
    a = ap.parse_args()
    lib = _lib.load()
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    ws = torch.empty(64 << 20, dtype=torch.float32, device="cuda")  # 256 MiB
    P = lambda t: c_void_p(t.data_ptr())
    for name, (n, h, w, cin, cout, ks, stride) in LAYERS.items():
        if a.layers and not any(name.startswith(p) for p in a.layers.split(",")):
            continue
        pad = 1 if ks == 3 else 0
        ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
        x = torch.randn(n, h, w, cin, device="cuda")
        wp, coutp = pack_conv_weight(torch.randn(cout, cin, ks, ks) / (cin * ks * ks) ** 0.5)
        wp = wp.cuda()
        sc, sh = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
        y0 = torch.empty(n, ho, wo, cout, device="cuda")
        y = torch.empty_like(y0)
        flops = 2.0 * n * ho * wo * cout * ks * ks * cin
        print(f"{name:12s} M={n*ho*wo:7d} K={ks*ks*cin:5d} N={cout:4d} {flops/1e9:6.1f} GF  ideal {flops/157.3e6:6.1f} us", flush=True)

        def run(tile, wgs, out):
            d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, in_ctot=cin, in_coff=0, ho=ho, wo=wo, cout=cout, coutp=coutp,
                              out_ctot=cout, out_coff=0, ks=ks, stride=stride, pad=pad, relu=1, mode=0, up=1,
                              tile=tile, sk_wgs=wgs)
            call = lambda: _lib.check(lib.av2x_conv2d_sk(byref(d), P(x), P(wp), P(sc), P(sh), None, P(out), P(ws),
                                                         ws.numel() * 4, st), "conv")
            for _ in range(3):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                call()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / a.iters

        for tn, (bm, bn) in TILES.items():
            if a.tiles and tn not in a.tiles.split(","):
                continue
            if coutp % (bn & 0x01ff):
                continue
            base = run((bm << 16) | bn, 0, y0)
            line = f"   {tn:11s} dp:{base:6.1f}us {flops/base/1e6:5.1f}TF |"
            for g in (int(v) for v in a.wgs.split(",")):
                y.zero_()
                us = run((bm << 16) | bn | 0x2000, g, y)
                err = float((y - y0).abs().max())
                line += f" sk{g}:{us:6.1f}us {flops/us/1e6:5.1f}TF e={err:.1e} |"
            print(line, flush=True)


if __name__ == "__main__":
    main()
