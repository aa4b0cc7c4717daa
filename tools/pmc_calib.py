#!/usr/bin/env python3
"""Known-byte-count launches for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
(MI355X_MICROARCH.md, HBM section: WRITE_SIZE and non-16B/lane reads are uncalibrated).

  av2x_fill_zero      writes   BYTES           reads 0
  torch copy_         writes   BYTES           reads BYTES   (16 B/lane)
  torch sum           writes   ~0              reads BYTES
  strided copy        writes   BYTES           reads BYTES   (4 B/lane, lanes 32 B apart: the dword-gather pattern)

BYTES = 576 MiB (> the 256 MiB Infinity Cache).  Run under `rocprofv3 --pmc FETCH_SIZE` and
`--pmc WRITE_SIZE` in separate passes; tools/pmc_traffic.py turns the two csv files into scale factors."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from airv2x_perception_amd import _lib

BYTES = 576 << 20
lib = _lib.load()
x = torch.empty(BYTES // 4, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    _lib.check(lib.av2x_fill_zero(x.data_ptr(), BYTES, st), "fill")
    y.copy_(x)
    s = x.sum()
    # dword-per-lane reads in 32-byte runs, every byte of every line used once per launch (the Winograd kernels' input gather pattern:
    # lane -> one fp32 of an 8-channel run, neighbouring lanes 32 B apart): a (N, 8) -> (8, N) transposing copy
    y.view(8, -1).copy_(x.view(-1, 8).t())
torch.cuda.synchronize()
print("calib done", float(s))
