"""Diagnostic: how many ReLU outputs of a conv+BN(batch)+ReLU chain have another sign than the float64 evaluation -- device vs torch fp32."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, torch.nn.functional as F
from airv2x_perception_amd.opencood_iface import train_ops as T
L, c, n, h, w = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 64, 2, 32, 48
g = torch.Generator().manual_seed(0)
x = torch.randn(n, c, h, w, generator=g)
ws = [torch.randn(c, c, 3, 3, generator=g) * (2.0 / (9 * c)) ** 0.5 for _ in range(L)]
gs = [torch.rand(c, generator=g) + 0.5 for _ in range(L)]
bs = [torch.randn(c, generator=g) * 0.2 for _ in range(L)]
def cpu(dtype):
    y, pre = x.to(dtype), []
    for i in range(L):
        a = F.batch_norm(F.conv2d(y, ws[i].to(dtype), None, padding=1), None, None, gs[i].to(dtype), bs[i].to(dtype), True, 0.01, 1e-3)
        pre.append(a); y = F.relu(a)
    return pre
p64, p32 = cpu(torch.float64), cpu(torch.float32)
y = x.permute(0, 2, 3, 1).contiguous().cuda()
for i in range(L):
    with torch.no_grad():
        y = T.conv_bn_act(y, ws[i].cuda(), gs[i].cuda(), bs[i].cuda(), 1, 1, act=True)
    yd = y.permute(0, 3, 1, 2).cpu()
    m64 = p64[i] > 0
    fd = (yd > 0) != m64
    f32 = (p32[i] > 0) != m64
    err_d = float((yd.double() - F.relu(p64[i])).abs().max()); err_32 = float((F.relu(p32[i]).double() - F.relu(p64[i])).abs().max())
    print(f"layer {i}: flips device {int(fd.sum())} (|a64| at flips <= {float(p64[i][fd].abs().max()) if fd.any() else 0:.1e}), torch32 {int(f32.sum())}; "
          f"max|y - y64| device {err_d:.2e} torch32 {err_32:.2e}")
