"""Time av2x_ln_qkv_window_attention_bf16 alone at the 8-agent V2X-ViT shape (281 600 tokens)."""
import ctypes, os, sys
from ctypes import c_void_p, c_int32
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from airv2x_perception_amd import _lib
from airv2x_perception_amd.opencood_iface.packing import interleave2_columns, pack_conv_weight, to_bf16_koct

BF = torch.bfloat16
if os.environ.get("AV2X_QW_LIB"):      # an ablation build of linear_bf16.hip alone (tools/micro/qw_ablate.sh): timing only
    lib = ctypes.CDLL(os.environ["AV2X_QW_LIB"])
    res, args = _lib.SIGNATURES["av2x_ln_qkv_window_attention_bf16"]
    lib.av2x_ln_qkv_window_attention_bf16.restype, lib.av2x_ln_qkv_window_attention_bf16.argtypes = res, args
else:
    lib = _lib.load()
p = lambda t: c_void_p(t.data_ptr()) if t is not None else c_void_p(0)
st = lambda: c_void_p(torch.cuda.current_stream().cuda_stream)


def pack(wt):
    wp, _ = pack_conv_weight(wt.view(wt.shape[0], wt.shape[1], 1, 1))
    return interleave2_columns(to_bf16_koct(wp))[0].cuda()


n, H, W = 8, 100, 352
m = n * H * W
g = torch.Generator().manual_seed(1)
x = (torch.randn(m, 256, generator=g) * 2).cuda()
dl = torch.randn(m, 256, generator=g).to(BF).cuda()
gm, bt = (torch.rand(256, generator=g) + 0.5).cuda(), (torch.randn(256, generator=g) * 0.1).cuda()
wq = pack((torch.randn(2304, 256, generator=g) / 16).to(BF).float())
bq = torch.zeros(2304).cuda()
w3 = torch.cat([pack((torch.randn(256, 256, generator=g) / 16).to(BF).float()) for _ in range(3)], -2).contiguous()
b3 = torch.randn(768, generator=g).cuda()
cfg = [(16, 16, 2), (8, 32, 4), (4, 64, 4)]
pos = [torch.randn(2 * ws - 1, 2 * ws - 1, generator=g).cuda() for _, _, ws in cfg]
out = [torch.zeros(n, H, W, 256, device="cuda", dtype=BF) for _ in range(3)]
posv = (c_void_p * 3)(*[t.data_ptr() for t in pos]); outv = (c_void_p * 3)(*[t.data_ptr() for t in out])
hv, dv, wv = ((c_int32 * 3)(*[c[k] for c in cfg]) for k in range(3))


def launch():
    _lib.check(lib.av2x_ln_qkv_window_attention_bf16(p(x), p(dl), p(gm), p(bt), 1e-5, p(wq), p(bq), p(w3), p(b3), ctypes.cast(posv, c_void_p),
               ctypes.cast(outv, c_void_p), ctypes.cast(hv, c_void_p), ctypes.cast(dv, c_void_p), ctypes.cast(wv, c_void_p), n, H, W, st()), "mega")


for _ in range(3):
    launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    launch()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
fl = 2.0 * m * 256 * (2304 + 768)
print(f"{us:.1f} us per launch, {fl / us / 1e6:.0f} TFLOP/s of GEMM, {m / 64 / 256:.1f} workgroups per CU -> {us / (m / 64 / 256):.2f} us per workgroup")
