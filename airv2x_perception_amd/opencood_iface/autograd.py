"""First slice of training on the MI355X path: a differentiable Conv2d (+ per-channel affine + ReLU) whose
forward AND backward run in libairv2x_hip.so -- the building block of BaseBEVBackbone / DownsampleConv / the heads
(models/common_modules/base_bev_backbone.py:41-105, downsample_conv.py:17-31), which the reference trains through torch
autograd (tools/train.py:220-247).

    y = act(scale[c] * conv2d(x, w, stride, pad) + shift[c])        x, y: NHWC fp32 on the device

backward (all fp32, deterministic):
    dz      = dy * act'(y)                       av2x_act_backward (ReLU mask from the stored output)
    d shift = sum over pixels of dz              av2x_channel_sum
    d w     = scale[co] * correlate(x, dz)       av2x_conv2d_wgrad (fp32 MFMA GEMM over the pixel axis)
    d x     = conv2d(dz * scale, rot180(w)^T)    the forward kernel itself on re-packed weights; stride 2 = the same on the
                                                 zero-upsampled dz (exact: the inserted zeros contribute nothing)
``scale`` is treated as a constant (a folded, frozen BatchNorm factor, or None): the reference's `backbone_fix` fine-tuning
regime.  Batch-statistics BatchNorm, the transposed convolutions, the fusion and the pillar encoders are in ``train_ops.py``;
``train_where2com.py`` assembles them into ``Airv2xWhere2com``'s ``.train()`` forward.
"""
from __future__ import annotations

import os
from ctypes import byref, c_void_p

import torch

from .. import _lib
from .packing import pack_conv_weight


def _runner(device):
    from .submodules import _Runner
    r = _RUNNERS.get(device)
    if r is None:
        r = _RUNNERS[device] = _Runner(device)
        r.stream_k = False      # gradients are compared term by term: keep the data-parallel (order-independent) schedules
        # the agent count changes per frame, so nearly every training step meets new conv shapes: timing ~20 candidates each (with
        # synchronises, next to the weight-gradient side stream) would stall the first epoch and could persist a noisy pick.
        # Shipped / cached table hits are still used; a miss falls back to the pick_tile rule.  All candidates are bit-identical.
        r.tune_on_miss = False
        # the training step is pinned to the reference's step gradient by gradient.  1x1 / strided / transposed convolutions and the token
        # Linears of the transformer fusions: the pipelined split-3 GEMM (conv_igemm_x3p: products as exact as fp32 products, 1.3-1.5x the
        # fp32-input MFMA's rate); AV2X_TRAIN_X3P=0 restores the fp32-input GEMM
        r.x3p = os.environ.get("AV2X_TRAIN_X3P", "1") != "0"
        # the F(4x4,3x3) class (the 256 -> 256 layers at 100 x 352) likewise on its split-3 kernel (conv_wino4_x3: 1.3x conv_wino4_f32), and
        # since round 5 the F(2x2,3x3) class with >= 64 channels on conv_wino_x3, forward and data gradient (round 4's kernel gained nothing
        # on these single-stream launches; round 5's: Where2Comm step 11.99 -> 11.51 ms, When2com 31.0 -> 30.2, V2VNet unchanged,
        # profiles/r05o_train_wino2_x3_ab.txt); AV2X_TRAIN_WINO2_X3=0 restores the fp32-input Winograd kernel
        r.wino_x3 = r.wino4_x3 = r.x3p
        r.wino2_x3 = r.x3p and os.environ.get("AV2X_TRAIN_WINO2_X3", "1") != "0"
    return r


_RUNNERS = {}
_P = lambda t: c_void_p(t.data_ptr()) if t is not None else None


def _desc(n, h, w, cin, cout, coutp, ks, stride, pad, relu, out_ctot=None):
    ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
    return _lib.ConvDesc(n=n, h=h, w=w, cin=cin, in_ctot=cin, in_coff=0, ho=ho, wo=wo, cout=cout, coutp=coutp,
                         out_ctot=out_ctot or cout, out_coff=0, ks=ks, stride=stride, pad=pad, relu=relu, mode=0, up=1, tile=0, sk_wgs=0), ho, wo


class ConvAffineAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, scale, shift, stride, pad, act):
        if x.device.type != "cuda" or x.dtype != torch.float32:
            raise RuntimeError("ConvAffineAct runs on fp32 HIP tensors only (no CPU path exists)")
        from .engine import ConvLayer
        r = _runner(x.device)
        x = x.contiguous()
        n, h, w, cin = x.shape
        cout, _, ks, _ = weight.shape
        wp, coutp = pack_conv_weight(weight)
        L = ConvLayer(wp.to(x.device), scale, shift if shift is not None else torch.zeros(cout, device=x.device), cin, cout, coutp,
                      ks, stride, pad, 1 if act else 0)
        ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
        y = torch.empty((n, ho, wo, cout), dtype=torch.float32, device=x.device)
        r.conv(L, x, n, h, w, y)
        ctx.save_for_backward(x, weight, scale, y)
        ctx.cfg = (stride, pad, act, shift is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, scale, y = ctx.saved_tensors
        stride, pad, act, has_shift = ctx.cfg
        from .engine import ConvLayer
        r = _runner(x.device)
        lib = r.lib
        st = r.stream()
        dy = dy.contiguous()
        n, h, w, cin = x.shape
        cout, _, ks, _ = weight.shape
        _, ho, wo, _ = y.shape
        rows = n * ho * wo
        dz = torch.empty_like(dy)
        _lib.check(lib.av2x_act_backward(_P(y), _P(dy), None, rows, cout, 1 if act else 0, _P(dz), st), "av2x_act_backward")
        dshift = None
        if has_shift and ctx.needs_input_grad[3]:
            ws = torch.empty(int(lib.av2x_channel_sum_workspace_bytes(rows, cout)) // 4 + 4, device=x.device)
            dshift = torch.empty(cout, device=x.device)
            _lib.check(lib.av2x_channel_sum(_P(dz), rows, cout, _P(ws), _P(dshift), st), "av2x_channel_sum")
        if scale is not None:   # everything upstream of the affine sees dz * scale
            _lib.check(lib.av2x_act_backward(None, _P(dz), _P(scale), rows, cout, 0, _P(dz), st), "av2x_act_backward")
        dw = None
        if ctx.needs_input_grad[1]:
            d, _, _ = _desc(n, h, w, cin, cout, cout, ks, stride, pad, 0)
            ws = torch.empty(int(lib.av2x_conv2d_wgrad_workspace_bytes(byref(d))) // 4 + 4, device=x.device)
            dw = torch.empty_like(weight)
            _lib.check(lib.av2x_conv2d_wgrad(byref(d), _P(x), _P(dz), _P(ws), _P(dw), st), "av2x_conv2d_wgrad")
        dx = None
        if ctx.needs_input_grad[0]:
            if cout % 32:
                raise NotImplementedError("data gradient needs cout % 32 == 0 (the head convolutions are leaves of the backbone graph)")
            wt = weight.detach().flip(2, 3).transpose(0, 1).contiguous()          # (cin, cout, k, k): rot180, channels swapped
            wp, cp = pack_conv_weight(wt)
            Lb = ConvLayer(wp.to(x.device), None, torch.zeros(cin, device=x.device), cout, cin, cp, ks, 1, ks // 2 if ks == 3 else 0, 0)
            if stride == 1:
                if pad != ks // 2:
                    raise NotImplementedError("'same' padding only")
                src, hs, wsz = dz, ho, wo
            elif stride == 2 and ks == 3 and pad == 1 and h == 2 * ho and w == 2 * wo:
                src = torch.zeros((n, h, w, cout), dtype=torch.float32, device=x.device)   # zero-upsampled dz
                src[:, ::2, ::2] = dz
                hs, wsz = h, w
            else:
                raise NotImplementedError("data gradient: stride 1, or 3x3 stride 2 pad 1 on even sizes (every layer of the BEV backbone)")
            dx = torch.empty((n, h, w, cin), dtype=torch.float32, device=x.device)
            r.conv(Lb, src, n, hs, wsz, dx)
        return dx, dw, None, dshift, None, None, None


def conv_affine_act(x, weight, scale=None, shift=None, stride=1, pad=1, act=True):
    """y = act(scale * conv2d(x, weight) + shift) on NHWC fp32 HIP tensors, differentiable in x, weight and shift."""
    return ConvAffineAct.apply(x, weight, scale, shift, stride, pad, act)
