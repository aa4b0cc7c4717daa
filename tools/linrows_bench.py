#!/usr/bin/env python3
"""av2x_linear_rows on the When2com first MLP layer (N = 256, K = 256*25*88): GB/s of weight streaming per row count."""
import sys, os, time
from ctypes import c_void_p
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from airv2x_perception_amd import _lib

lib = _lib.load()
N, K = 256, 256 * 25 * 88
w = torch.randn(N, K, device="cuda")
b = torch.randn(N, device="cuda")
P = lambda t: c_void_p(t.data_ptr())
st = c_void_p(torch.cuda.current_stream().cuda_stream)
for m in (1, 2, 4, 8):
    x = torch.randn(m, K, device="cuda")
    y = torch.empty(m, N, device="cuda")
    ws = torch.empty(max(lib.av2x_linear_rows_workspace_bytes(m, N, K) // 4, 1), device="cuda")
    for _ in range(3):
        _lib.check(lib.av2x_linear_rows(P(x), P(w), P(b), m, N, K, 1, P(y), P(ws), ws.numel() * 4, st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        _lib.check(lib.av2x_linear_rows(P(x), P(w), P(b), m, N, K, 1, P(y), P(ws), ws.numel() * 4, st))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    ref = torch.relu(x.double() @ w.double().t() + b.double())
    err = float((y.double() - ref).abs().max())
    print(f"wgs={os.environ.get('AV2X_LINROWS_WGS', '2048')} m={m}: {us:.1f} us  {N * K * 4 / us / 1e6:.2f} TB/s of weights  max err {err:.2e}")
