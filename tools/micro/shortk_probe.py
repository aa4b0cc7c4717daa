"""Where does the time of the short-K layers go?  deblock 0 (M = 140 800, K = 64, N = 128, plain 1x1) with a contiguous and a
channel-sliced (384-wide concat) output, every direct tile; plus an elementwise copy of the same bytes as the HBM yardstick."""
import sys, os
sys.path.insert(0, os.getcwd())
from ctypes import byref, c_void_p
import torch
from airv2x_perception_amd import _lib
from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight
from airv2x_perception_amd.opencood_iface.engine import Where2ComEngine
lib = _lib.load()
st = c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: c_void_p(t.data_ptr())
def t_us(call, reps=20):
    for _ in range(3): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
ROT = 10      # rotate over this many buffer sets (> 1 GB in total): the 256 MB infinity cache cannot hold the working set
for (M, cin, cout, ctot) in ((140800, 64, 128, 128), (140800, 64, 128, 384), (140800, 256, 32, 32)):
    n, h, w = 4, 100, M // 400
    xs = [torch.randn(n, h, w, cin, device="cuda") for _ in range(ROT)]
    wp, coutp = pack_conv_weight(torch.randn(cout, cin, 1, 1) / cin ** 0.5)
    wp = wp.cuda(); sh = torch.zeros(cout, device="cuda")
    ys = [torch.empty(n, h, w, ctot, device="cuda") for _ in range(ROT)]
    srcs = [torch.empty(n, h, w, cout, device="cuda") for _ in range(ROT)]
    it = [0]
    def copy():
        k = it[0] % ROT; it[0] += 1
        ys[k][..., :cout].copy_(srcs[k])
    line = f"M={M} K={cin} N={cout} out_ctot={ctot}: copy-yardstick {t_us(copy, 40):5.1f}us |"
    for bm, bn in Where2ComEngine.TILE_CANDIDATES:
        if (bn & 0x1ff) > coutp: continue
        d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, in_ctot=cin, in_coff=0, ho=h, wo=w, cout=cout, coutp=coutp, out_ctot=ctot, out_coff=0, ks=1, stride=1, pad=0, relu=1, mode=0, up=1, tile=(bm << 16) | bn, sk_wgs=0)
        def call():
            k = it[0] % ROT; it[0] += 1
            _lib.check(lib.av2x_conv2d(byref(d), P(xs[k]), P(wp), None, P(sh), P(ys[k]), st), "c")
        try:
            us = t_us(call, 40)
        except Exception as e:
            continue
        line += f" {bm}x{bn & 0x1ff}{'w8' if bn & 0x8000 else ''}{'p' if bn & 0x4000 else ''}{'g' if bn & 0x200 else ''}:{us:5.1f}"
    print(line, flush=True)
