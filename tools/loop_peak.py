#!/usr/bin/env python3
"""In-loop ceiling of conv_igemm_f32: shapes whose tile count is an exact multiple of the 256 CUs (no tail) with long K
(prologue / epilogue amortised).  Prints TFLOP/s per tile configuration and per tiles-per-CU; the gap to the 157.3 TF
peak that remains here is what the main loop itself loses (barrier bubbles, LDS latency, load stalls)."""
import os
import sys
from ctypes import byref, c_void_p

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from airv2x_perception_amd import _lib  # noqa: E402
from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight  # noqa: E402

TILES = {"g128x128w8": (128, 128 | 0x8200), "g128x128w8s3": (128, 128 | 0xc200), "g128x64w8": (128, 64 | 0x8200),
         "g128x64w8s3": (128, 64 | 0xc200), "g128x128": (128, 128 | 0x0200), "g128x64": (128, 64 | 0x0200),
         "128x64w8d": (128, 64 | 0xc000), "128x128w8d": (128, 128 | 0xc000), "128x64d": (128, 64 | 0x4000),
         "128x128d": (128, 128 | 0x4000), "128x64w8": (128, 64 | 0x8000), "128x128w8": (128, 128 | 0x8000), "64x64d": (64, 64 | 0x4000)}


def run(lib, h, w, cin, cout, ks, tile, iters=10):
    pad = 1 if ks == 3 else 0
    zero = os.environ.get("AV2X_ZERO_DATA") == "1"   # DVFS probe: zero operands draw less power -> higher sustained clock
    x = torch.zeros(1, h, w, cin, device="cuda") if zero else torch.randn(1, h, w, cin, device="cuda")
    wt = torch.zeros(cout, cin, ks, ks) if zero else torch.randn(cout, cin, ks, ks) / (cin * ks * ks) ** 0.5
    wp, coutp = pack_conv_weight(wt)
    wp = wp.cuda()
    sc, sh = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
    y = torch.empty(1, h, w, cout, device="cuda")
    d = _lib.ConvDesc(n=1, h=h, w=w, cin=cin, in_ctot=cin, in_coff=0, ho=h, wo=w, cout=cout, coutp=coutp, out_ctot=cout,
                      out_coff=0, ks=ks, stride=1, pad=pad, relu=1, mode=0, up=1, tile=(tile[0] << 16) | tile[1])
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    def call():
        rc = lib.av2x_conv2d(byref(d), c_void_p(x.data_ptr()), c_void_p(wp.data_ptr()), c_void_p(sc.data_ptr()),
                             c_void_p(sh.data_ptr()), c_void_p(y.data_ptr()), st)
        if rc:
            raise RuntimeError(lib.av2x_last_error())
    for _ in range(2):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    return 2.0 * h * w * cout * ks * ks * cin / us / 1e6, us


def main():
    lib = _lib.load()
    abl = os.environ.get("AV2X_ABLATE_LIB")      # tools/micro/libablate_<bits>.so (tools/micro/ablate.sh): timing-only variants
    if abl:
        import ctypes
        lib = ctypes.CDLL(abl)
        lib.av2x_conv2d.restype = ctypes.c_int32
        lib.av2x_conv2d.argtypes = [ctypes.POINTER(_lib.ConvDesc)] + [c_void_p] * 6
        lib.av2x_last_error.restype = ctypes.c_char_p
    sel = sys.argv[1].split(",") if len(sys.argv) > 1 else list(TILES)
    for ks, cin in ((3, 256),):
        for name in sel:
            bm, bn = TILES[name][0], TILES[name][1] & 0x1ff
            line = f"ks={ks} cin={cin:5d} cout=256 {name:11s}|"
            only = os.environ.get("AV2X_PEAK_ONLY")
            for per_cu in ((int(only),) if only else (1, 2, 3, 4, 6)):
                tiles = 256 * per_cu
                m = tiles * bm // (256 // bn)          # tiles = (m / bm) * (256 / bn)
                w = 128
                h = m // w
                tf, us = run(lib, h, w, cin, 256, ks, TILES[name])
                line += f" {per_cu}/CU: {tf:6.1f} TF ({us:7.1f} us)|"
            print(line, flush=True)


if __name__ == "__main__":
    main()
