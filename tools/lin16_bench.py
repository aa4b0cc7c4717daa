"""Microbenchmark of av2x_linear_bf16 (csrc/linear_bf16.hip) on the V2X-ViT 8-agent shapes: GB/s of algorithmic HBM bytes."""
import sys
from ctypes import c_void_p

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from airv2x_perception_amd import _lib  # noqa: E402
from airv2x_perception_amd.opencood_iface.packing import interleave2_columns, pack_conv_weight, to_bf16_koct  # noqa: E402

lib = _lib.load()
import os
if os.environ.get("AV2X_LIN_LIB"):          # experiments: an alternative build of the same entry point
    import ctypes
    alt = ctypes.CDLL(os.environ["AV2X_LIN_LIB"])
    alt.av2x_linear_bf16.restype, alt.av2x_linear_bf16.argtypes = _lib.SIGNATURES["av2x_linear_bf16"]
    lib = alt
P = lambda t: c_void_p(t.data_ptr()) if t is not None else c_void_p(0)
st = c_void_p(torch.cuda.current_stream().cuda_stream)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 281600
for name, cout, out16, res, act in (("proj 256->1280 bf16", 1280, 1, 0, 0), ("qkv3 256->2304 bf16", 2304, 1, 0, 0), ("wout 256->256 bf16", 256, 1, 0, 0),
                                    ("ff1 256->256 gelu bf16", 256, 1, 0, 2), ("aout 256->256 fp32+res", 256, 0, 1, 0)):
    a = torch.randn(M, 256, device="cuda").to(torch.bfloat16)
    w = torch.randn(cout, 256) / 16
    wp, _ = pack_conv_weight(w.view(cout, 256, 1, 1))
    w16, coutp = interleave2_columns(to_bf16_koct(wp))
    w16 = w16.cuda()
    b = torch.randn(cout, device="cuda")
    out = torch.zeros(M, cout, device="cuda", dtype=torch.bfloat16 if out16 else torch.float32)
    r = out if res else None
    call = lambda: _lib.check(lib.av2x_linear_bf16(P(a), P(w16), P(b), P(r), P(out), M, 256, cout, coutp, out16, cout, 0, cout if res else 0, 0, act, st), "lin")
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    byt = M * 256 * 2 + M * cout * (2 if out16 else 4) + (M * cout * 4 if res else 0)
    fl = 2.0 * M * 256 * cout
    print(f"{name:26s} {us:8.1f} us   {byt / us / 1e6:6.2f} TB/s   {fl / us / 1e6:7.1f} TFLOP/s   ({byt / 1e6:.0f} MB)")
