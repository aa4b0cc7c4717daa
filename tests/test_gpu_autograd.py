"""GPU: forward + backward of the differentiable convolution block (opencood_iface/autograd.py) against torch
autograd of the same fp32 expression on the CPU (the oracle form of BaseBEVBackbone / DownsampleConv layers), on the BEV
backbone's layer shapes incl. the stride-2 first layers.  Tolerance: fp32 sums over up to 140 800 pixels (wgrad) /
2 304 taps (dgrad) in a different order than the CPU's."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import assert_close

pytestmark = pytest.mark.gpu

CASES = [
    # n, h, w, cin, cout, ks, stride, affine, act
    (2, 20, 36, 64, 64, 3, 1, True, True),
    (2, 20, 36, 64, 64, 3, 2, True, True),       # block first layer: ZeroPad2d(1) + stride 2
    (1, 26, 44, 64, 128, 3, 2, True, True),
    (3, 13, 22, 128, 128, 3, 1, True, True),
    (1, 25, 88, 256, 256, 3, 1, True, True),
    (2, 10, 18, 384, 256, 1, 1, False, True),    # shrink 1x1 (bias, no BN)
    (2, 12, 20, 256, 256, 3, 1, False, False),
    (4, 50, 88, 128, 128, 3, 1, True, True),     # several pixel chunks in the weight gradient (17 600 pixels)
]


@pytest.mark.parametrize("case", CASES)
def test_conv_block_forward_and_backward_match_torch_autograd(case):
    from airv2x_perception_amd.opencood_iface.autograd import conv_affine_act
    n, h, w, cin, cout, ks, stride, affine, act = case
    g = torch.Generator().manual_seed(sum(case[:7]))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, ks, ks, generator=g) / np.sqrt(cin * ks * ks)
    scale = (torch.rand(cout, generator=g) + 0.5) if affine else None
    shift = torch.randn(cout, generator=g) * 0.2
    pad = 1 if ks == 3 else 0
    # ---- CPU reference graph
    xr, wr, sr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), shift.clone().requires_grad_(True)
    z = F.conv2d(xr, wr, None, stride=stride, padding=pad)
    z = z * (scale.view(1, -1, 1, 1) if affine else 1.0) + sr.view(1, -1, 1, 1)
    yr = F.relu(z) if act else z
    gy = torch.randn(yr.shape, generator=g)
    # ReLU'(0) is a step: an output whose pre-activation is within fp32 rounding of zero may be masked on one side and not
    # on the other (its gradient then differs on a whole 3x3xCin footprint).  Such outputs get no upstream gradient.
    gy[z.detach().abs() < 1e-4] = 0
    yr.backward(gy)
    # ---- device graph (NHWC)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
    wd = wt.cuda().requires_grad_(True)
    sd = shift.cuda().requires_grad_(True)
    yd = conv_affine_act(xd, wd, scale.cuda() if affine else None, sd, stride=stride, pad=pad, act=act)
    yd.backward(gy.permute(0, 2, 3, 1).contiguous().cuda())
    torch.cuda.synchronize()
    assert_close(yd.detach().permute(0, 3, 1, 2).cpu(), yr.detach(), 1e-4, 1e-4, "forward")
    sx = float(xr.grad.abs().max())
    assert_close(xd.grad.permute(0, 3, 1, 2).cpu(), xr.grad, 2e-4, 2e-5 * sx + 1e-6, "dx")
    sw = float(wr.grad.abs().max())
    assert_close(wd.grad.cpu(), wr.grad, 2e-4, 2e-5 * sw + 1e-6, "dw")
    assert_close(sd.grad.cpu(), sr.grad, 2e-4, 2e-5 * float(sr.grad.abs().max()) + 1e-6, "dshift")
    # deterministic: a second backward gives identical bits
    xd2 = x.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
    wd2 = wt.cuda().requires_grad_(True)
    y2 = conv_affine_act(xd2, wd2, scale.cuda() if affine else None, shift.cuda(), stride=stride, pad=pad, act=act)
    y2.backward(gy.permute(0, 2, 3, 1).contiguous().cuda())
    assert torch.equal(wd2.grad, wd.grad) and torch.equal(xd2.grad, xd.grad)


def test_two_layer_chain_trains_a_step():
    """Two stacked blocks + an MSE-like loss: one SGD step on the device lowers the loss, and the gradients of the first
    layer (through the second's data gradient) match torch autograd."""
    from airv2x_perception_amd.opencood_iface.autograd import conv_affine_act
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 64, 16, 24, generator=g)
    w1 = torch.randn(64, 64, 3, 3, generator=g) / 24
    w2 = torch.randn(128, 64, 3, 3, generator=g) / 24
    b1, b2 = torch.zeros(64), torch.zeros(128)
    tgt = torch.randn(2, 128, 8, 12, generator=g)

    b1 = b1 + 0.05   # keep the pre-activations of this small net away from the ReLU step
    b2 = b2 + 0.05

    def cpu_loss(w1, w2):
        y = F.relu(F.conv2d(F.relu(F.conv2d(x, w1, b1, padding=1)), w2, b2, stride=2, padding=1))
        return ((y - tgt) ** 2).mean()

    a, b = w1.clone().requires_grad_(True), w2.clone().requires_grad_(True)
    l0 = cpu_loss(a, b)
    l0.backward()
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    w1d, w2d = w1.cuda().requires_grad_(True), w2.cuda().requires_grad_(True)
    td = tgt.permute(0, 2, 3, 1).contiguous().cuda()

    def dev_loss():
        y = conv_affine_act(conv_affine_act(xd, w1d, None, b1.cuda()), w2d, None, b2.cuda(), stride=2)
        return ((y - td) ** 2).mean()

    l = dev_loss()
    l.backward()
    assert abs(float(l.detach()) - float(l0)) < 1e-5 * max(1.0, float(l0))
    assert_close(w1d.grad.cpu(), a.grad, 2e-4, 2e-5 * float(a.grad.abs().max()), "dw1 through two layers")
    assert_close(w2d.grad.cpu(), b.grad, 2e-4, 2e-5 * float(b.grad.abs().max()), "dw2")
    with torch.no_grad():
        w1d -= 0.5 * w1d.grad
        w2d -= 0.5 * w2d.grad
    assert float(dev_loss()) < float(l)
