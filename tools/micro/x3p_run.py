"""Launches conv_igemm_x3p on a CoBEVT token Linear (8 x 100 x 352 tokens, cin -> cout) REPS times: the target of tools/micro/pmc_x3p.sh.
python tools/micro/x3p_run.py [cin] [cout] [bn] [reps]"""
import os
import sys
from ctypes import byref, c_void_p

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from airv2x_perception_amd import _lib                                                                   # noqa: E402
from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight, to_bf16x3_koct                # noqa: E402

cin, cout, bn, reps = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 256), (2, 256), (3, 128), (4, 20)))
bm = int(os.environ.get("BM", "128"))
lib = _lib.load()
P = lambda t: c_void_p(t.data_ptr()) if t is not None else None
g = torch.Generator().manual_seed(1)
x = torch.randn(8, 100, 352, cin, generator=g).cuda()
wp, coutp = pack_conv_weight(torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5)
w3 = to_bf16x3_koct(wp.cuda())
out = torch.empty(8, 100, 352, cout, device="cuda")
one, zero = torch.ones(cout).cuda(), torch.zeros(cout).cuda()
d = _lib.ConvDesc(n=8, h=100, w=352, cin=cin, in_ctot=cin, in_coff=0, ho=100, wo=352, cout=cout, coutp=coutp, out_ctot=cout, out_coff=0,
                  ks=1, stride=1, pad=0, relu=0, mode=0, up=1, tile=(bm << 16) | bn | 0x1400, sk_wgs=0)
st = c_void_p(torch.cuda.current_stream().cuda_stream)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for i in range(reps + 3):
    if i == 3:
        ev[0].record()
    _lib.check(lib.av2x_conv2d_res(byref(d), P(x), P(w3), P(one), P(zero), None, P(out), st), "x3p")
ev[1].record()
torch.cuda.synchronize()
us = ev[0].elapsed_time(ev[1]) * 1e3 / reps
fl = 2.0 * 8 * 100 * 352 * cin * cout
print(f"x3p {bm}x{bn} {cin}->{cout}: {us:.1f} us per launch, {fl / us / 1e6:.1f} TFLOP/s fp32-equivalent, {6 * fl / us / 1e6:.0f} executed bf16 TFLOP/s "
      f"({6 * fl / us / 1e6 / 2500:.3f} of 2.5 PF)")
