#!/usr/bin/env python3
"""Winograd F(2x2,3x3) conv (tile flag 0x40000000) against the direct implicit-GEMM kernel on the frame's 3x3 / stride-1
layers: time, effective TFLOP/s (direct-conv FLOPs / time) and the error of both against an fp64 convolution.
Usage: python tools/wino_bench.py [--iters 20] [--layers name,...] [--check]"""
import argparse, os, sys
from ctypes import byref, c_void_p
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from airv2x_perception_amd import _lib  # noqa: E402
from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight  # noqa: E402

# name: (n, h, w, cin, cout)
LAYERS = {
    "shrink3_n4": (4, 100, 352, 256, 256),
    "shrink3_n1": (1, 100, 352, 256, 256),
    "b2_n4": (4, 25, 88, 256, 256),
    "b2_n3": (3, 25, 88, 256, 256),
    "b1_n4": (4, 50, 176, 128, 128),
    "b1_n3": (3, 50, 176, 128, 128),
    "b0_n4": (4, 100, 352, 64, 64),
    "b0_n1": (1, 100, 352, 64, 64),
    "b0_n8": (8, 100, 352, 64, 64),
    "odd_n2": (2, 25, 87, 64, 64),
    # fixed cost per workgroup: the b2_n4 geometry (144 workgroups = one round) at 1 .. 64 chunks of 8 input channels
    "k8": (4, 25, 88, 8, 256), "k32": (4, 25, 88, 32, 256), "k128": (4, 25, 88, 128, 256), "k512": (4, 25, 88, 512, 256),
}
WINO = {"w32x128": (32, 128), "w64x64": (64, 64), "w32x64": (32, 64), "w32x64h": (32, 64 | 0x8000), "w32x32q": (32, 32 | 0x8000)}
DIRECT = {"g64x64": (64, 64 | 0x0200), "g128x64w8": (128, 64 | 0x8200)}
# split-3 Winograd (csrc/conv_wino_x3.hip): three bf16 terms per fp32 operand on v_mfma_f32_32x32x16_bf16
X3 = {"x3_64x64": (64, 64 | 0x0400), "x3_32x64": (32, 64 | 0x0400), "x3_32x128": (32, 128 | 0x0400)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--layers", default="")
    ap.add_argument("--only-x3", action="store_true", help="run only the split-3 tiles (profiling)")
    ap.add_argument("--tiles", default="", help="restrict the split-3 tiles (x3_64x64,x3_32x64)")
    ap.add_argument("--check", action="store_true", help="compare against an fp64 conv (slow for the big layers)")
    a = ap.parse_args()
    lib = _lib.load()
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: c_void_p(t.data_ptr())
    for name, (n, h, w, cin, cout) in LAYERS.items():
        if a.layers and name not in a.layers.split(","):
            continue
        torch.manual_seed(0)
        x = torch.randn(n, h, w, cin, device="cuda")
        wt = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
        wp, coutp = pack_conv_weight(wt)
        wp = wp.cuda()
        u = torch.empty(lib.av2x_wino_weight_bytes(cin, coutp) // 4, device="cuda")
        _lib.check(lib.av2x_wino_pack_weights(P(wp), cin, coutp, P(u), st), "pack")
        u3 = None
        if cin % 16 == 0 and coutp % 64 == 0:
            u3 = torch.empty(lib.av2x_wino_x3_weight_bytes(cin, coutp) // 2, dtype=torch.bfloat16, device="cuda")
            _lib.check(lib.av2x_wino_x3_pack_weights(P(wp), cin, coutp, P(u3), st), "pack x3")
        sc, sh = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda") * 0.1
        flops = 2.0 * n * h * w * cout * 9 * cin
        print(f"{name:11s} M={n*h*w:7d} cin={cin} cout={cout} {flops/1e9:6.1f} GF  direct ideal {flops/157.3e6:6.1f} us", flush=True)
        ref = None
        if a.check:
            ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), wt.cuda().double(), padding=1)
            ref = torch.relu(ref * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)).permute(0, 2, 3, 1)

        def run(tile, wgt):
            out = torch.full((n, h, w, cout), float("nan"), device="cuda")
            d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, in_ctot=cin, in_coff=0, ho=h, wo=w, cout=cout, coutp=coutp,
                              out_ctot=cout, out_coff=0, ks=3, stride=1, pad=1, relu=1, mode=0, up=1, tile=tile, sk_wgs=0)
            call = lambda: _lib.check(lib.av2x_conv2d_res(byref(d), P(x), P(wgt), P(sc), P(sh), None, P(out), st), "conv")
            for _ in range(3):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                call()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / a.iters, out

        base = None
        for tn, (bm, bn) in DIRECT.items():
            if a.only_x3:
                break
            if coutp % (bn & 0x1ff) or cin % 32:
                continue
            us, y = run((bm << 16) | bn, wp)
            base = y if base is None else base
            err = f" err64={float((y.double() - ref).abs().max()):.2e}" if ref is not None else ""
            print(f"   {tn:10s} {us:7.1f} us {flops/us/1e6:6.1f} TF{err}", flush=True)
        for tn, (tb, cb) in WINO.items():
            if a.only_x3:
                break
            if cout % (cb & 0x1ff):
                continue
            us, y = run(0x40000000 | (tb << 16) | cb, u)
            err = f" err64={float((y.double() - ref).abs().max()):.2e} rms={float((y.double() - ref).pow(2).mean().sqrt()):.2e}" if ref is not None else ""
            dv = f" vs direct {float((y - base).abs().max()):.2e} (max|y| {float(base.abs().max()):.2f})" if base is not None else ""
            print(f"   {tn:10s} {us:7.1f} us {flops/us/1e6:6.1f} TF(eff){err}{dv} nan={int(torch.isnan(y).sum())}", flush=True)
        if cin % 32 == 0 and coutp % 64 == 0 and cout == coutp and not (a.tiles and "w4" not in a.tiles):
            # F(4x4,3x3): fp32-input MFMA (conv_wino4_f32) and split-3 operands on the bf16 MFMA (conv_wino4_x3)
            u4 = torch.empty(lib.av2x_wino4_weight_bytes(cin, coutp) // 4, device="cuda")
            _lib.check(lib.av2x_wino4_pack_weights(P(wp), cin, coutp, P(u4), st), "pack w4")
            u43 = torch.empty(lib.av2x_wino4_x3_weight_bytes(cin, coutp) // 2, dtype=torch.bfloat16, device="cuda")
            _lib.check(lib.av2x_wino4_x3_pack_weights(P(wp), cin, coutp, P(u43), st), "pack w4 x3")
            for tn, tile, wgt in (("w4_f32", 0x60000000 | (32 << 16) | 64, u4), ("w4_x3", 0x60000400 | (32 << 16) | 64, u43)):
                if a.only_x3 and tn == "w4_f32":
                    continue
                us, y = run(tile, wgt)
                err = f" err64={float((y.double() - ref).abs().max()):.2e} rms={float((y.double() - ref).pow(2).mean().sqrt()):.2e}" if ref is not None else ""
                dv = f" vs direct {float((y - base).abs().max()):.2e} (max|y| {float(base.abs().max()):.2f})" if base is not None else ""
                ex = f" {flops*36/144*6/us/1e6:7.1f} TF bf16 executed" if tn == "w4_x3" else f" {flops*36/144/us/1e6:7.1f} TF executed"
                print(f"   {tn:10s} {us:7.1f} us {flops/us/1e6:6.1f} TF(eff){ex}{err}{dv} nan={int(torch.isnan(y).sum())}", flush=True)
        for tn, (tb, cb) in X3.items():
            if u3 is None or (a.tiles and tn not in a.tiles.split(",")) or cout % (cb & 0x1ff):
                continue
            us, y = run(0x40000000 | (tb << 16) | cb, u3)
            err = f" err64={float((y.double() - ref).abs().max()):.2e} rms={float((y.double() - ref).pow(2).mean().sqrt()):.2e}" if ref is not None else ""
            dv = f" vs direct {float((y - base).abs().max()):.2e} (max|y| {float(base.abs().max()):.2f})" if base is not None else ""
            print(f"   {tn:10s} {us:7.1f} us {flops/us/1e6:6.1f} TF(eff) {flops*16/36*6/us/1e6:7.1f} TF bf16 executed{err}{dv} nan={int(torch.isnan(y).sum())}", flush=True)


if __name__ == "__main__":
    main()
