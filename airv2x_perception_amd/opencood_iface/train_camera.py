"""Train-mode forward of the camera branch (``LiftSplatShootEncoder.forward``, common_modules/airv2x_encoder.py:309-336, with
``self.training``): CamEncode on the EfficientNet-B0 trunk (sub_modules/lss_submodule.py:50-189), the ground-truth-depth lift + voxel
pooling (airv2x_encoder.py:133-167, 208-275) and BevEncode (:312-350) as a graph of HIP forward / backward ops under torch autograd
(tools/train.py:220-247 is what the reference does).

    stem / MBConv expand / project / Up / BevEncode convolutions + BatchNorm (batch statistics)     train_ops.conv_bn_act
        (activations NHWC, channel counts zero-padded to multiples of 32: padded channels stay exactly zero, their parameters get none)
    swish, sigmoid                                                                                UnaryFn          (csrc/train_camera.hip)
    MBConv depthwise conv (k 3 / 5, stride 1 / 2, TF "same" padding) + BatchNorm                    DwConvFn + BatchNormFn
        data gradient = the same depthwise kernel on the flipped taps (stride 2: on the zero-upsampled gradient);
        weight gradient = one fixed-order per-(image, channel) reduction per tap (av2x_gap with a second operand)
    squeeze-and-excite: mean over pixels, two row GEMMs, channel scale                            GapFn, LinearRowsFn, ChannelScaleFn
    skip connections (MBConv: add, with stochastic depth in training; BasicBlock: ReLU after the add)    AddActFn
    nn.Upsample(bilinear, align_corners=True)                                                     ResizeFn         (fixed-point adjoint)
    depth one-hot (x) features -> voxel pooling                                                   LiftGtFn         (adjoint: a gather)
    mean over the modality maps (Airv2xBase.fuse_bev, airv2x_base_model.py:167-177)                Mean2Fn

Ground-truth depth (``use_depth_gt: true``, the shipped camera YAML) only: the predicted-depth head trains nothing on this path either
way -- ``LiftSplatShootEncoder.forward`` drops the ``depth_items`` CamEncode returns (:296-307, 327-331), so no depth loss sees them.
The trunk is the restated EfficientNet-B0 of oracle/camera_oracle.py (efficientnet_pytorch is absent from this image: trunk parity is
unpinned, as for the eval path).
"""
from __future__ import annotations

import ctypes
from ctypes import c_void_p

import torch
import torch.nn.functional as Fn

from .. import _lib
from . import train_fusion_ops as F
from . import train_ops as T
from .autograd import _runner
from .camera import EFF_EPS, TV_EPS, effnet_b0_blocks
from .train_when2com import linear_rows

_P = T._P
EFF_MOM, TV_MOM = 0.01, 0.1          # efficientnet_pytorch builds its BatchNorms with momentum 1 - 0.99; torchvision / nn defaults: 0.1
DROP_CONNECT = 0.2                   # efficientnet-b0's drop_connect_rate (stochastic depth, training only; lss_submodule.py:127-133)


def _p32(c):
    return (c + 31) // 32 * 32


def _padw(w, cout_p, cin_p):
    """Conv weight (cout, cin, k, k) zero-padded to (cout_p, cin_p, k, k) (a differentiable data-movement op)."""
    cout, cin = w.shape[:2]
    if cout == cout_p and cin == cin_p:
        return w
    return Fn.pad(w, (0, 0, 0, 0, 0, cin_p - cin, 0, cout_p - cout))


def _padv(v, c_p):
    return v if v.shape[0] == c_p else Fn.pad(v, (0, c_p - v.shape[0]))


def _update(sd, bn, st, c, momentum):
    """nn.BatchNorm's running-statistics update from the batch statistics of the REAL channels."""
    mean, var, count = st
    T.update_running_stats(sd[bn + ".running_mean"], sd[bn + ".running_var"], sd.get(bn + ".num_batches_tracked"), (mean[:c], var[:c], count), 1,
                           momentum=momentum)


# ------------------------------------------------------------------------------------------------ elementwise
class UnaryFn(torch.autograd.Function):
    """y = act(x): 3 sigmoid, 6 swish."""

    @staticmethod
    def forward(ctx, x, act):
        T._check_dev(x)
        r = _runner(x.device)
        x = x.contiguous()
        y = torch.empty_like(x)
        _lib.check(r.lib.av2x_unary_forward(_P(x), x.numel(), act, _P(y), r.stream()), "av2x_unary_forward")
        ctx.save_for_backward(x)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        r = _runner(x.device)
        dx = torch.empty_like(x)
        _lib.check(r.lib.av2x_unary_backward(_P(x), _P(dy.contiguous()), x.numel(), ctx.act, _P(dx), r.stream()), "av2x_unary_backward")
        return dx, None


def swish(x):
    return UnaryFn.apply(x, 6)


def sigmoid(x):
    return UnaryFn.apply(x, 3)


class AddActFn(torch.autograd.Function):
    """y = a + b, or relu(a + b)."""

    @staticmethod
    def forward(ctx, a, b, relu):
        T._check_dev(a)
        r = _runner(a.device)
        a, b = a.contiguous(), b.contiguous()
        y = torch.empty_like(a)
        _lib.check(r.lib.av2x_add_act(_P(a), _P(b), a.numel(), 1 if relu else 0, _P(y), r.stream()), "av2x_add_act")
        ctx.relu = relu
        if relu:
            ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        if not ctx.relu:
            return dy, dy, None
        (y,) = ctx.saved_tensors
        r = _runner(y.device)
        c = y.shape[-1]
        dz = torch.empty_like(dy)
        _lib.check(r.lib.av2x_act_backward(_P(y), _P(dy), None, y.numel() // c, c, 1, _P(dz), r.stream()), "av2x_act_backward")
        return dz, dz, None


def add_act(a, b, relu=False):
    return AddActFn.apply(a, b, relu)


class Mean2Fn(torch.autograd.Function):
    """(a + b) / 2: the mean over two modality maps."""

    @staticmethod
    def forward(ctx, a, b):
        T._check_dev(a)
        r = _runner(a.device)
        a, b = a.contiguous(), b.contiguous()
        y = torch.empty_like(a)
        _lib.check(r.lib.av2x_mean2(_P(a), _P(b), _P(y), a.numel(), r.stream()), "av2x_mean2")
        return y

    @staticmethod
    def backward(ctx, dy):
        r = _runner(dy.device)
        dy = dy.contiguous()
        d2 = torch.empty((2,) + tuple(dy.shape), dtype=torch.float32, device=dy.device)
        _lib.check(r.lib.av2x_scale_broadcast(_P(dy), _P(d2), 2, dy.numel(), 0.5, r.stream()), "av2x_scale_broadcast")
        return d2[0], d2[1]


# ------------------------------------------------------------------------------------------------ BatchNorm on a given map
class BatchNormFn(torch.autograd.Function):
    """Train-mode BatchNorm of an NHWC map (the BatchNorm after the depthwise convolution); stats_out receives (mean, var, count)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, eps, stats_out):
        T._check_dev(z)
        z = z.contiguous()
        y, mean, var, rstd, scale, shift, count = T.bn_train_forward(z, gamma, beta, eps, False, None)
        if stats_out is not None:
            stats_out.append((mean, var, count))
        ctx.save_for_backward(z, mean, rstd, scale, shift)
        return y

    @staticmethod
    def backward(ctx, dy):
        z, mean, rstd, scale, shift = ctx.saved_tensors
        dz, dgamma, dbeta = T.bn_backward(dy.contiguous(), z, mean, rstd, scale, shift, False)
        return dz, dgamma, dbeta, None, None


# ------------------------------------------------------------------------------------------------ depthwise convolution
_ONES = {}


def _dw_launch(r, x, w, k, s, pad_t, pad_l, ho, wo):
    n, h, wd, c = x.shape
    out = torch.empty((n, ho, wo, c), dtype=torch.float32, device=x.device)
    one = _ONES.get((c, x.device))
    if one is None:
        one = _ONES[(c, x.device)] = torch.ones(c, dtype=torch.float32, device=x.device)
    zero = T._zeros(c, x.device)
    _lib.check(r.lib.av2x_dwconv2d(_P(x), n, h, wd, c, _P(w), _P(one), _P(zero), k, s, pad_t, pad_l, ho, wo, 0, _P(out), r.stream()), "av2x_dwconv2d")
    return out


class DwConvFn(torch.autograd.Function):
    """Depthwise k x k convolution (k 3 / 5, stride 1 / 2) with explicit (before, after) zero padding; w (k*k, c), x NHWC."""

    @staticmethod
    def forward(ctx, x, w, k, s, pad):
        T._check_dev(x)
        r = _runner(x.device)
        x, w = x.contiguous(), w.contiguous()
        n, h, wd, c = x.shape
        pa, pb = pad
        ho, wo = (h + pa + pb - k) // s + 1, (wd + pa + pb - k) // s + 1
        y = _dw_launch(r, x, w, k, s, pa, pa, ho, wo)
        ctx.save_for_backward(x, w)
        ctx.cfg = (k, s, pa, pb, ho, wo)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        k, s, pa, pb, ho, wo = ctx.cfg
        r = _runner(x.device)
        dy = dy.contiguous()
        n, h, wd, c = x.shape
        dx = dw = None
        if ctx.needs_input_grad[0]:
            g = dy
            if s == 2:      # the zero-upsampled gradient: entry (2 i, 2 j) = dy (i, j)
                g = torch.zeros((n, 2 * ho - 1, 2 * wo - 1, c), dtype=torch.float32, device=x.device)
                g[:, ::2, ::2] = dy
            wf = w.view(k, k, c).flip(0, 1).reshape(k * k, c).contiguous()
            dx = _dw_launch(r, g, wf, k, 1, k - 1 - pa, k - 1 - pa, h, wd)
        if ctx.needs_input_grad[1]:
            # dw[ky, kx, c] = sum_{n, i, j} dy[n, i, j, c] * xpad[n, s i + ky, s j + kx, c]: all taps in one launch, fixed summation order
            ws = torch.empty(int(r.lib.av2x_dwconv2d_wgrad_workspace_bytes(n, ho, wo, c, k)) // 4 + 1, dtype=torch.float32, device=x.device)
            dw = torch.empty((k * k, c), dtype=torch.float32, device=x.device)
            _lib.check(r.lib.av2x_dwconv2d_wgrad(_P(x), _P(dy), n, h, wd, c, k, s, pa, ho, wo, _P(ws), _P(dw), r.stream()), "av2x_dwconv2d_wgrad")
        return dx, dw, None, None, None


# ------------------------------------------------------------------------------------------------ squeeze-and-excite pieces
class GapFn(torch.autograd.Function):
    """(n, h, w, c) -> (n, c): the mean over the pixels (F.adaptive_avg_pool2d(x, 1))."""

    @staticmethod
    def forward(ctx, x):
        T._check_dev(x)
        r = _runner(x.device)
        x = x.contiguous()
        n, h, w, c = x.shape
        out = torch.empty((n, c), dtype=torch.float32, device=x.device)
        ws = torch.empty(int(r.lib.av2x_gap_workspace_bytes(n, h * w, c)) // 4 + 1, dtype=torch.float32, device=x.device)
        _lib.check(r.lib.av2x_gap(_P(x), None, n, h * w, c, 1.0 / (h * w), _P(ws), _P(out), r.stream()), "av2x_gap")
        ctx.shape = (n, h, w, c)
        return out

    @staticmethod
    def backward(ctx, dg):
        n, h, w, c = ctx.shape
        r = _runner(dg.device)
        dx = torch.empty((n, h, w, c), dtype=torch.float32, device=dg.device)
        _lib.check(r.lib.av2x_channel_broadcast(_P(dg.contiguous()), None, n, h * w, c, 1.0 / (h * w), _P(dx), r.stream()), "av2x_channel_broadcast")
        return dx


class ChannelScaleFn(torch.autograd.Function):
    """y[n, p, c] = x[n, p, c] * g[n, c]."""

    @staticmethod
    def forward(ctx, x, g):
        T._check_dev(x)
        r = _runner(x.device)
        x, g = x.contiguous(), g.contiguous()
        n, h, w, c = x.shape
        y = torch.empty_like(x)
        _lib.check(r.lib.av2x_channel_broadcast(_P(g), _P(x), n, h * w, c, 1.0, _P(y), r.stream()), "av2x_channel_broadcast")
        ctx.save_for_backward(x, g)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g = ctx.saved_tensors
        r = _runner(x.device)
        dy = dy.contiguous()
        n, h, w, c = x.shape
        dx = torch.empty_like(x)
        _lib.check(r.lib.av2x_channel_broadcast(_P(g), _P(dy), n, h * w, c, 1.0, _P(dx), r.stream()), "av2x_channel_broadcast")
        dg = torch.empty_like(g)
        ws = torch.empty(int(r.lib.av2x_gap_workspace_bytes(n, h * w, c)) // 4 + 1, dtype=torch.float32, device=x.device)
        _lib.check(r.lib.av2x_gap(_P(x), _P(dy), n, h * w, c, 1.0, _P(ws), _P(dg), r.stream()), "av2x_gap")
        return dx, dg


# ------------------------------------------------------------------------------------------------ upsampling, lift
class ResizeFn(torch.autograd.Function):
    """nn.Upsample(scale_factor, mode="bilinear", align_corners=True) of an NHWC map."""

    @staticmethod
    def forward(ctx, x, scale):
        T._check_dev(x)
        r = _runner(x.device)
        x = x.contiguous()
        n, h, w, c = x.shape
        h2, w2 = h * scale, w * scale
        y = torch.empty((n, h2, w2, c), dtype=torch.float32, device=x.device)
        _lib.check(r.lib.av2x_resize_bilinear(_P(x), n, h, w, c, c, 0, h2, w2, 0, 0, h2, w2, _P(y), c, 0, r.stream()), "av2x_resize_bilinear")
        ctx.cfg = (n, h, w, c, h2, w2)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, h, w, c, h2, w2 = ctx.cfg
        r = _runner(dy.device)
        dx = torch.empty((n, h, w, c), dtype=torch.float32, device=dy.device)
        _lib.check(r.lib.av2x_resize_bilinear_backward(_P(dy.contiguous()), n, h, w, c, h2, w2, None, _P(dx), r.stream()), "av2x_resize_bilinear_backward")
        return dx, None


class LiftGtFn(torch.autograd.Function):
    """One-hot of the binned ground-truth depth (x) image features, lifted along the camera rays and summed into the BEV grid
    (av2x_lss_lift_pool with ``target`` = training: out-of-range depths are clipped, not masked -- camera_utils.py:278-288)."""

    @staticmethod
    def forward(ctx, feat, enc, flat, params, B, N, target):
        T._check_dev(feat)
        r = _runner(feat.device)
        feat = feat.contiguous()
        planes, H, W = flat.shape[1:]
        ny, nx = int(enc.nx[1]), int(enc.nx[0])
        pooled = torch.empty((B, ny, nx, enc.C), dtype=torch.float32, device=feat.device)
        ws = torch.empty(int(r.lib.av2x_lss_pool_workspace_bytes(B, nx, ny, 1, enc.C)), dtype=torch.uint8, device=feat.device)
        _lib.check(r.lib.av2x_lss_lift_pool(_P(feat), None, _P(flat), planes, H, W, enc.ds, ctypes.cast(enc._depth3, c_void_p), enc.nbins,
                                            enc.depth_mode, 1 if target else 0, _P(enc.frustum), _P(params), B, N, enc.fH, enc.fW, enc.C,
                                            ctypes.cast(enc._lo, c_void_p), ctypes.cast(enc._dx, c_void_p), ctypes.cast(enc._nx, c_void_p),
                                            _P(ws), _P(pooled), r.stream()), "av2x_lss_lift_pool")
        ctx.save_for_backward(flat, params)
        ctx.cfg = (enc, B, N, target, tuple(feat.shape))
        return pooled

    @staticmethod
    def backward(ctx, dout):
        flat, params = ctx.saved_tensors
        enc, B, N, target, shape = ctx.cfg
        r = _runner(dout.device)
        planes, H, W = flat.shape[1:]
        dfeat = torch.empty(shape, dtype=torch.float32, device=dout.device)
        _lib.check(r.lib.av2x_lss_lift_pool_backward(_P(dout.contiguous()), _P(flat), planes, H, W, enc.ds, ctypes.cast(enc._depth3, c_void_p),
                                                     enc.nbins, enc.depth_mode, 1 if target else 0, _P(enc.frustum), _P(params), B, N, enc.fH,
                                                     enc.fW, enc.C, ctypes.cast(enc._lo, c_void_p), ctypes.cast(enc._dx, c_void_p),
                                                     ctypes.cast(enc._nx, c_void_p), _P(dfeat), r.stream()), "av2x_lss_lift_pool_backward")
        return dfeat, None, None, None, None, None, None


class MaxPoolFn(torch.autograd.Function):
    """nn.MaxPool2d(ks, stride, pad) on NHWC maps (the stem pool of CamEncode_Resnet101); the gradient goes to the first maximum of a window."""

    @staticmethod
    def forward(ctx, x, ks, stride, pad):
        T._check_dev(x)
        r = _runner(x.device)
        x = x.contiguous()
        n, h, w, c = x.shape
        ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
        y = torch.empty((n, ho, wo, c), dtype=torch.float32, device=x.device)
        _lib.check(r.lib.av2x_maxpool2d(_P(x), n, h, w, c, ks, stride, pad, ho, wo, _P(y), r.stream()), "av2x_maxpool2d")
        ctx.save_for_backward(x)
        ctx.cfg = (ks, stride, pad, ho, wo)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        ks, stride, pad, ho, wo = ctx.cfg
        r = _runner(x.device)
        n, h, w, c = x.shape
        dx = torch.empty_like(x)
        _lib.check(r.lib.av2x_maxpool2d_backward(_P(x), _P(dy.contiguous()), n, h, w, c, ks, stride, pad, ho, wo, _P(dx), r.stream()),
                   "av2x_maxpool2d_backward")
        return dx, None, None, None


class SoftmaxChFn(torch.autograd.Function):
    """CamEncode.get_depth_dist (:89-92): softmax over the first ``d`` channels of (n, h, w, stride) logits -> (n, h, w, d)."""

    @staticmethod
    def forward(ctx, logit, d):
        T._check_dev(logit)
        r = _runner(logit.device)
        logit = logit.contiguous()
        n, h, w, stride = logit.shape
        prob = torch.empty((n, h, w, d), dtype=torch.float32, device=logit.device)
        _lib.check(r.lib.av2x_softmax_channels(_P(logit), n * h * w, d, stride, _P(prob), r.stream()), "av2x_softmax_channels")
        ctx.save_for_backward(prob)
        ctx.stride = stride
        return prob

    @staticmethod
    def backward(ctx, dprob):
        (prob,) = ctx.saved_tensors
        r = _runner(dprob.device)
        n, h, w, d = prob.shape
        dlogit = torch.empty((n, h, w, ctx.stride), dtype=torch.float32, device=dprob.device)
        _lib.check(r.lib.av2x_softmax_channels_backward(_P(prob), _P(dprob.contiguous()), n * h * w, d, ctx.stride, _P(dlogit), r.stream()),
                   "av2x_softmax_channels_backward")
        return dlogit, None


class LiftProbFn(torch.autograd.Function):
    """Predicted depth distribution (x) image features, lifted along the camera rays and summed into the BEV grid (CamEncode.forward
    :176-186 + voxel_pooling; av2x_lss_lift_pool with ``prob``): the (B, N, D, fH, fW, C) volume exists in neither direction."""

    @staticmethod
    def forward(ctx, feat, prob, enc, params, B, N):
        T._check_dev(feat)
        r = _runner(feat.device)
        feat, prob = feat.contiguous(), prob.contiguous()
        ny, nx = int(enc.nx[1]), int(enc.nx[0])
        pooled = torch.empty((B, ny, nx, enc.C), dtype=torch.float32, device=feat.device)
        ws = torch.empty(int(r.lib.av2x_lss_pool_workspace_bytes(B, nx, ny, 1, enc.C)), dtype=torch.uint8, device=feat.device)
        _lib.check(r.lib.av2x_lss_lift_pool(_P(feat), _P(prob), None, 0, 0, 0, enc.ds, ctypes.cast(enc._depth3, c_void_p), enc.nbins,
                                            enc.depth_mode, 0, _P(enc.frustum), _P(params), B, N, enc.fH, enc.fW, enc.C,
                                            ctypes.cast(enc._lo, c_void_p), ctypes.cast(enc._dx, c_void_p), ctypes.cast(enc._nx, c_void_p),
                                            _P(ws), _P(pooled), r.stream()), "av2x_lss_lift_pool")
        ctx.save_for_backward(feat, prob, params)
        ctx.cfg = (enc, B, N)
        return pooled

    @staticmethod
    def backward(ctx, dout):
        feat, prob, params = ctx.saved_tensors
        enc, B, N = ctx.cfg
        r = _runner(dout.device)
        dfeat, dprob = torch.empty_like(feat), torch.empty_like(prob)
        _lib.check(r.lib.av2x_lss_lift_pool_prob_backward(_P(dout.contiguous()), _P(feat), _P(prob), enc.nbins, _P(enc.frustum), _P(params), B, N,
                                                          enc.fH, enc.fW, enc.C, ctypes.cast(enc._lo, c_void_p), ctypes.cast(enc._dx, c_void_p),
                                                          ctypes.cast(enc._nx, c_void_p), _P(dfeat), _P(dprob), r.stream()),
                   "av2x_lss_lift_pool_prob_backward")
        return dfeat, dprob, None, None, None, None


# ------------------------------------------------------------------------------------------------ the modules
def _conv_bn(P, sd, x, wkey, bn, stride, pad, eps, mom, act, cin_p=None, weight=None):
    """Conv2d (no bias) + BatchNorm (batch statistics) [+ ReLU] on channel-padded NHWC maps; the BatchNorm's running statistics
    are updated from the real channels."""
    w = P[wkey] if weight is None else weight
    cout = w.shape[0]
    cout_p = _p32(cout)
    wp = _padw(w, cout_p, cin_p if cin_p is not None else x.shape[-1])
    st = []
    y = T.conv_bn_act(x, wp, _padv(P[bn + ".weight"], cout_p), _padv(P[bn + ".bias"], cout_p), stride, pad, eps=eps, act=act, stats_out=st)
    _update(sd, bn, st[0], cout, mom)
    return y


def mbconv(P, sd, q, row, x, drop_rate, training):
    """MBConvBlock.forward of efficientnet_pytorch (restated: oracle/camera_oracle.py:82-113) on a channel-padded NHWC map."""
    cin, cout, k, s, e, se, pad = row
    mid = cin * e
    mid_p = _p32(mid)
    inp = x
    if e != 1:
        x = swish(_conv_bn(P, sd, x, q + "_expand_conv.weight", q + "_bn0", 1, 0, EFF_EPS, EFF_MOM, False))
    wdw = _padv_cols(P[q + "_depthwise_conv.weight"].reshape(mid, k * k).t(), mid_p)
    z = DwConvFn.apply(x, wdw, k, s, tuple(pad))
    st = []
    x = swish(BatchNormFn.apply(z, _padv(P[q + "_bn1.weight"], mid_p), _padv(P[q + "_bn1.bias"], mid_p), EFF_EPS, st))
    _update(sd, q + "_bn1", st[0], mid, EFF_MOM)
    # squeeze-and-excite (:101-103): two 1x1 convolutions on the pooled vector = row GEMMs
    se_p = (se + 3) // 4 * 4
    g = GapFn.apply(x)
    wr = Fn.pad(P[q + "_se_reduce.weight"].reshape(se, mid), (0, mid_p - mid, 0, se_p - se))
    we = Fn.pad(P[q + "_se_expand.weight"].reshape(mid, se), (0, se_p - se, 0, mid_p - mid))
    g = swish(linear_rows(g, wr, _padv(P[q + "_se_reduce.bias"], se_p), 0))
    g = sigmoid(linear_rows(g, we, _padv(P[q + "_se_expand.bias"], mid_p), 0))
    x = ChannelScaleFn.apply(x, g)
    x = _conv_bn(P, sd, x, q + "_project_conv.weight", q + "_bn2", 1, 0, EFF_EPS, EFF_MOM, False)
    if s == 1 and cin == cout:
        if drop_rate and training:      # stochastic depth: x / keep * Bernoulli(keep) per image
            keep = 1.0 - drop_rate
            n, h, w, _ = x.shape
            m = torch.floor(keep + torch.rand(n, device=x.device)) / keep
            x = T.MaskMul.apply(x, m.view(n, 1, 1).expand(n, h, w).contiguous())
        x = add_act(x, inp)
    return x


def _padv_cols(m, c_p):
    """(k*k, c) tap-major depthwise weights, columns zero-padded to c_p."""
    return m if m.shape[1] == c_p else Fn.pad(m, (0, c_p - m.shape[1]))


def up_block(P, sd, p, x1, x2, c_skip, scale):
    """Up.forward (:39-47): upsample x1, pad it to x2's size, concat [x2 | x1] (x2 channel-padded), two Conv3x3 + BN + ReLU."""
    x1 = ResizeFn.apply(x1, scale)
    dy, dx = x2.shape[1] - x1.shape[1], x2.shape[2] - x1.shape[2]
    if dy < 0 or dx < 0:
        raise NotImplementedError("Up: the upsampled map is larger than the skip map (negative F.pad)")
    if dy or dx:
        x1 = Fn.pad(x1, (0, 0, dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))
    x = torch.cat([x2, x1], -1)
    cs_p = x2.shape[-1]
    w0 = P[p + "conv.0.weight"]
    if cs_p != c_skip:      # the skip map's padded channels get zero weight columns
        w0 = torch.cat([w0[:, :c_skip], w0.new_zeros((w0.shape[0], cs_p - c_skip, 3, 3)), w0[:, c_skip:]], 1)
    x = _conv_bn(P, sd, x, None, p + "conv.1", 1, 1, TV_EPS, TV_MOM, True, weight=w0)
    return _conv_bn(P, sd, x, p + "conv.3.weight", p + "conv.4", 1, 1, TV_EPS, TV_MOM, True)


def cam_features(P, sd, p, flat, training, drop_connect=None, depth_bins=0):
    """CamEncode.get_eff_features + image_head (:118-165): flat (BN, 4, H, W) device images -> (BN, fH, fW, C) NHWC.
    ``drop_connect``: the trunk's stochastic-depth rate (None: DROP_CONNECT, efficientnet-b0's 0.2; 0 switches it off).
    ``depth_bins`` D > 0 (use_depth_gt false): also the depth distribution softmax(depth_head(features)) (:180-181) -> (feat, prob)."""
    if drop_connect is None:
        drop_connect = DROP_CONNECT
    t = p + "trunk."
    x = Fn.pad(flat[:, :3].permute(0, 2, 3, 1), (0, 29, 0, 1, 0, 1)).contiguous()       # NHWC, 32 channel slots, static "same" pad (0 before, 1 after)
    x = swish(_conv_bn(P, sd, x, t + "_conv_stem.weight", t + "_bn0", 2, 0, EFF_EPS, EFF_MOM, False))
    ends, prev = [], x
    rows = effnet_b0_blocks()
    for i, row in enumerate(rows):
        rate = drop_connect * float(i) / len(rows) if drop_connect else 0.0
        x = mbconv(P, sd, f"{t}_blocks.{i}.", row, x, rate, training)
        if prev.shape[1] > x.shape[1]:
            ends.append(prev)
        prev = x
    ends.append(x)
    r3, r4, r5 = ends[2], ends[3], ends[4]
    u1 = up_block(P, sd, p + "up1.", r5, r4, 112, 2)
    f = up_block(P, sd, p + "up2.", u1, r3, 40, 2) if (p + "up2.conv.0.weight") in P else u1     # img_downsample 16: no up2 (lss_submodule.py:74-75)
    feat = T.conv_bias_act(f, P[p + "image_head.weight"], P[p + "image_head.bias"], 1, 0, False)
    if not depth_bins:
        return feat
    dp = _p32(depth_bins)                               # the head's output channels padded to a multiple of 32; softmax over the real ones
    wd = Fn.pad(P[p + "depth_head.weight"], (0, 0, 0, 0, 0, 0, 0, dp - depth_bins))
    logit = T.conv_bias_act(f, wd, _padv(P[p + "depth_head.bias"], dp), 1, 0, False)
    return feat, SoftmaxChFn.apply(logit, depth_bins)


def bottleneck(P, sd, q, x, stride):
    """torchvision Bottleneck (1x1 -> 3x3 / stride -> 1x1, expansion 4) in train mode."""
    idt = x
    if (q + "downsample.0.weight") in P:
        idt = _conv_bn(P, sd, x, q + "downsample.0.weight", q + "downsample.1", stride, 0, TV_EPS, TV_MOM, False)
    y = _conv_bn(P, sd, x, q + "conv1.weight", q + "bn1", 1, 0, TV_EPS, TV_MOM, True)
    y = _conv_bn(P, sd, y, q + "conv2.weight", q + "bn2", stride, 1, TV_EPS, TV_MOM, True)
    y = _conv_bn(P, sd, y, q + "conv3.weight", q + "bn3", 1, 0, TV_EPS, TV_MOM, False)
    return add_act(y, idt, True)


def cam_features_resnet101(P, sd, p, flat, depth_bins=0):
    """CamEncode_Resnet101.resnet101_forward + heads (lss_submodule.py:262-310) in train mode: flat (BN, 4, H, W) -> feat (BN, fH, fW, C)
    [, softmax depth distribution]."""
    from ..synth import RESNET101_LAYERS
    x = Fn.pad(flat[:, :3].permute(0, 2, 3, 1), (0, 29)).contiguous()              # NHWC, the colour planes in 32 channel slots
    x = _conv_bn(P, sd, x, p + "conv1.weight", p + "bn1", 2, 3, TV_EPS, TV_MOM, True)
    x = MaxPoolFn.apply(x, 3, 2, 1)
    for li, (planes, nb, stride) in enumerate(RESNET101_LAYERS, 1):
        for bi in range(nb):
            x = bottleneck(P, sd, f"{p}layer{li}.{bi}.", x, stride if bi == 0 else 1)
    feat = T.conv_bias_act(x, P[p + "image_head.weight"], P[p + "image_head.bias"], 1, 0, False)
    if not depth_bins:
        return feat
    dp = _p32(depth_bins)
    wd = Fn.pad(P[p + "depth_head.weight"], (0, 0, 0, 0, 0, 0, 0, dp - depth_bins))
    logit = T.conv_bias_act(x, wd, _padv(P[p + "depth_head.bias"], dp), 1, 0, False)
    return feat, SoftmaxChFn.apply(logit, depth_bins)


def basic_block(P, sd, q, x, stride):
    idt = x
    if (q + "downsample.0.weight") in P:
        idt = _conv_bn(P, sd, x, q + "downsample.0.weight", q + "downsample.1", stride, 0, TV_EPS, TV_MOM, False)
    y = _conv_bn(P, sd, x, q + "conv1.weight", q + "bn1", stride, 1, TV_EPS, TV_MOM, True)
    y = _conv_bn(P, sd, y, q + "conv2.weight", q + "bn2", 1, 1, TV_EPS, TV_MOM, False)
    return add_act(y, idt, True)


def bev_encode(P, sd, b, x):
    """BevEncode.forward (:335-350) on x (B, ny, nx, C) NHWC."""
    x = _conv_bn(P, sd, x, b + "conv1.weight", b + "bn1", 2, 3, TV_EPS, TV_MOM, True)
    x1 = basic_block(P, sd, b + "layer1.1.", basic_block(P, sd, b + "layer1.0.", x, 1), 1)
    x = basic_block(P, sd, b + "layer2.1.", basic_block(P, sd, b + "layer2.0.", x1, 2), 1)
    x = basic_block(P, sd, b + "layer3.1.", basic_block(P, sd, b + "layer3.0.", x, 2), 1)
    x = up_block(P, sd, b + "up1.", x, x1, 64, 4)
    x = ResizeFn.apply(x, 2)
    x = _conv_bn(P, sd, x, b + "up2.1.weight", b + "up2.2", 1, 1, TV_EPS, TV_MOM, True)
    return T.conv_bias_act(x, P[b + "up2.4.weight"], P[b + "up2.4.bias"], 1, 0, False)


def lss_encoder_train(P, sd, prefix, enc, cam_inputs, training=True):
    """One agent type's LiftSplatShootEncoder in train mode -> spatial_features (B, ny, nx, bevout) NHWC with its autograd graph.
    ``enc``: the type's packed ``camera.CameraEncoder`` (geometry only: frustum, grid, depth bins -- no weights are read from it)."""
    dev = next(iter(P.values())).device
    imgs = cam_inputs["imgs"]
    if imgs.device != dev or imgs.dtype != torch.float32 or not imgs.is_contiguous():
        imgs = imgs.to(dev, torch.float32).contiguous()
    B, N, planes, H, W = imgs.shape
    if enc.use_gt and planes < 4:
        raise ValueError("use_depth_gt: the images need a 4th (depth) plane")
    if (H // enc.ds, W // enc.ds) != (enc.fH, enc.fW):
        raise ValueError(f"camera images are {H}x{W}; data_aug_conf.final_dim says {enc.fH * enc.ds}x{enc.fW * enc.ds}")
    flat = imgs.view(B * N, planes, H, W)
    if enc.cfg["camera_encoder"] == "Resnet101":
        out = cam_features_resnet101(P, sd, prefix + "camencode.", flat, depth_bins=0 if enc.use_gt else enc.nbins)
    else:
        out = cam_features(P, sd, prefix + "camencode.", flat, training, depth_bins=0 if enc.use_gt else enc.nbins)
    feat = out if enc.use_gt else out[0]
    if tuple(feat.shape[1:3]) != (enc.fH, enc.fW):
        raise ValueError(f"camera image {H}x{W}: the stride-8 feature map is {tuple(feat.shape[1:3])}, the frustum expects {(enc.fH, enc.fW)}")
    if enc.use_gt:
        pooled = LiftGtFn.apply(feat, enc, flat, enc._cam_params(cam_inputs), B, N, training)
    else:
        pooled = LiftProbFn.apply(feat, out[1], enc, enc._cam_params(cam_inputs), B, N)
    return bev_encode(P, sd, prefix + "bevencode.", pooled)
