#!/usr/bin/env python3
"""CoBEVT fused-axial attention (av2x_fax_attention) at the BASELINE grid: the fp32-input MFMA kernel (test-hook bit 3) against the split-3
kernel on the bf16 matrix cores (bit 5).  Usage: python tools/fax_bench.py [L] [n_valid] [iters]"""
import os, sys
from ctypes import c_void_p
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from airv2x_perception_amd import _lib  # noqa: E402

L, nv, iters = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 8), (2, 8), (3, 20)))
H, W, heads = 100, 352, 8
C = heads * 32
lib = _lib.load()
g = torch.Generator().manual_seed(0)
qkv = torch.randn(L, H, W, 3 * C, generator=g).cuda()
table = torch.randn((2 * L - 1) * 49, heads, generator=g).cuda()
out = torch.empty(L, H, W, C, device="cuda")
st = c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: c_void_p(t.data_ptr())
flop = 2.0 * (H * W // 16) * heads * (L * 16) * (nv * 16) * 32 * 2          # QK^T + PV, valid keys
res = {}
for name, flag in (("fp32-input MFMA (fax_attention_mfma4_kernel)", 8), ("split-3 bf16 MFMA (fax_attention_x3_kernel)", 32)):
    for grid in (0, 1):
        for _ in range(3):
            _lib.check(lib.av2x_fax_attention(P(qkv), P(table), P(out), L, nv, H, W, 4, heads, 32, grid | flag, st), "fax")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            _lib.check(lib.av2x_fax_attention(P(qkv), P(table), P(out), L, nv, H, W, 4, heads, 32, grid | flag, st), "fax")
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        res[(flag, grid)] = out.clone()
        print(f"L={L} valid={nv} {'grid' if grid else 'window'} partition  {name}: {us:8.1f} us  {flop / us / 1e6:7.1f} TFLOP/s fp32-equivalent", flush=True)
for grid in (0, 1):
    print(f"{'grid' if grid else 'window'}: max |x3 - f32| = {float((res[(32, grid)] - res[(8, grid)]).abs().max()):.3e}  (max |out| {float(res[(8, grid)].abs().max()):.2f})")
