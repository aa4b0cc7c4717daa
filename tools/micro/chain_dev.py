"""Diagnostic: gradient deviation from float64 of a chain of L conv3x3 + BatchNorm(batch stats) [+ ReLU] layers -- the device
ops (train_ops.conv_bn_act) vs torch fp32 on the CPU.  With ReLU the deviation is dominated by activations that fall on the
other side of the kink; without it, by plain rounding."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, torch.nn.functional as F
from airv2x_perception_amd.opencood_iface import train_ops as T

def run(L, act, c=64, n=2, h=32, w=48, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, c, h, w, generator=g)
    ws = [torch.randn(c, c, 3, 3, generator=g) * (2.0 / (9 * c)) ** 0.5 for _ in range(L)]
    gs = [torch.rand(c, generator=g) + 0.5 for _ in range(L)]
    bs = [torch.randn(c, generator=g) * 0.2 for _ in range(L)]
    gy = torch.randn(n, c, h, w, generator=g)
    def cpu(dtype):
        xr = x.clone().to(dtype).requires_grad_(True)
        W = [t.clone().to(dtype).requires_grad_(True) for t in ws]
        y = xr
        for i in range(L):
            y = F.batch_norm(F.conv2d(y, W[i], None, padding=1), None, None, gs[i].to(dtype), bs[i].to(dtype), True, 0.01, 1e-3)
            if act: y = F.relu(y)
        y.backward(gy.to(dtype))
        return [xr.grad] + [t.grad for t in W]
    g64, g32 = cpu(torch.float64), cpu(torch.float32)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
    W = [t.cuda().requires_grad_(True) for t in ws]
    y = xd
    for i in range(L):
        y = T.conv_bn_act(y, W[i], gs[i].cuda(), bs[i].cuda(), 1, 1, act=act)
    y.backward(gy.permute(0, 2, 3, 1).contiguous().cuda())
    gd = [xd.grad.permute(0, 3, 1, 2).cpu()] + [t.grad.cpu() for t in W]
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
    d = [rel(a, b) for a, b in zip(gd, g64)]; r = [rel(a, b) for a, b in zip(g32, g64)]
    print("   per tensor [dx, dW1..dWL] device:", ["%.1e" % v for v in d])
    print(f"L={L} relu={act}: device median {np.median(d):.2e} worst {max(d):.2e} | torch fp32 median {np.median(r):.2e} worst {max(r):.2e}")

for L in (1, 4, 10):
    for act in (False, True):
        run(L, act)
