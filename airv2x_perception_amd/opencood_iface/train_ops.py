"""Differentiable building blocks of the train-mode Where2Comm path: every forward AND backward runs in
libairv2x_hip.so; torch only owns the tensors, the autograd graph bookkeeping and a handful of per-channel vector updates.

The reference trains through torch autograd (tools/train.py:220-247) over

    PFNLayer (Linear 10->64, BatchNorm1d batch statistics, ReLU, max over 32)      airv2x_pillar_vfe.py:27-49
    PointPillarScatter                                                            point_pillar_scatter.py:39-80
    Conv3x3 / ConvTranspose(k = s) + BatchNorm2d (batch statistics) + ReLU        base_bev_backbone.py:41-105
    x * communication_mask, AttentionFusion per pixel                             where2comm_fuse.py:152-164,228-249
    Conv + bias + ReLU (DownsampleConv), 1x1 heads                                downsample_conv.py:17-31

Each is a ``torch.autograd.Function`` here.  Layout: NHWC fp32 on the device throughout.  Batch statistics are returned
to the caller through ``stats_out`` (a python list that receives ``(mean, biased var, count)``) so that it can apply the
running-statistics update as often as the reference's schedule does.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import byref, c_void_p

import torch

from .. import _lib
from .autograd import _P, _desc, _runner

BN_EPS = 1e-3       # every BatchNorm on the path (airv2x_pillar_vfe.py:21, base_bev_backbone.py:52,65,83)
BN_MOMENTUM = 0.01


def _check_dev(x):
    if x.device.type != "cuda" or x.dtype != torch.float32:
        raise RuntimeError("the training ops run on fp32 HIP tensors only (no CPU path exists)")
    if x.device.index is not None and x.device.index != torch.cuda.current_device():
        # every launch goes to the CURRENT device's stream (engine.stream()); a tensor of another device would be handed to it
        raise RuntimeError(f"the training ops launch on the current HIP device (cuda:{torch.cuda.current_device()}) but the tensor lives on "
                           f"{x.device}: call torch.cuda.set_device(...) / wrap the step in `with torch.cuda.device(...)`")


# ------------------------------------------------------------------------------------------------ weight packing (device side)
def pack_conv_weight_dev(w, flipped=False):
    """(Cout, Cin, k, k) -> (k*k, Cin/4, CoutP, 4), CoutP = Cout rounded up to 32 -- packing.pack_conv_weight on the device, one launch
    (the weights change every optimiser step).  ``flipped``: the packing of ``w.flip(2, 3).transpose(0, 1)`` -- the data gradient's
    weights -- read straight from the parameter."""
    w = w.detach()
    if w.dtype != torch.float32 or w.device.type != "cuda":
        raise RuntimeError("pack_conv_weight_dev: fp32 HIP tensors only (no CPU path exists)")
    if not w.is_contiguous():
        w = w.contiguous()
    cout, cin, kh, kw = (w.shape[1], w.shape[0], w.shape[2], w.shape[3]) if flipped else w.shape
    if cin % 4 or kh != kw:
        raise ValueError("square kernels, cin a multiple of 4")
    coutp = (cout + 31) // 32 * 32
    r = _runner(w.device)
    out = torch.empty(kh * kw, cin // 4, coutp, 4, dtype=torch.float32, device=w.device)
    _lib.check(r.lib.av2x_pack_conv_weight(_P(w), cout, cin, kh, 1 if flipped else 0, _P(out), r.stream()), "av2x_pack_conv_weight")
    return out, coutp


def split3_dev(wp):
    """packing.to_bf16x3_koct of a packed weight (taps, cin/4, coutp, 4) on the device, one launch: (3, taps, cin/8, coutp, 8) bf16."""
    taps, q4, coutp, four = wp.shape
    if four != 4 or q4 % 2 or wp.dtype != torch.float32 or wp.device.type != "cuda" or not wp.is_contiguous():
        raise RuntimeError("split3_dev: a contiguous fp32 HIP packing with cin % 8 == 0")
    r = _runner(wp.device)
    out = torch.empty((3, taps, q4 // 2, coutp, 8), dtype=torch.bfloat16, device=wp.device)
    _lib.check(r.lib.av2x_split3_koct(_P(wp), taps, 4 * q4, coutp, _P(out), r.stream()), "av2x_split3_koct")
    return out


def seed_x3p(r, L, ent=None):
    """The step's 1x1 / strided / transposed convolutions and token Linears run on the pipelined split-3 GEMM (engine.conv's x3p class:
    fp32 operands as three bf16 terms, six products, fp32 accumulation -- products as exact as fp32 products): hand the launcher the
    split planes, made by one launch and cached with the parameter's packing, instead of its lazy torch chain."""
    if not r.x3p or AMP_STEP[0] or L.cin % 16 or L.coutp % 64:
        return
    if ent is not None and ent[6] is not None:
        L._w3 = ent[6]
        return
    L._w3 = split3_dev(L.w)
    if ent is not None:
        ent[6] = L._w3


def pack_deconv_weight_dev(w):
    """ConvTranspose2d weight (Cin, Cout, s, s), kernel == stride -> (1, Cin/4, s*s*Cout, 4); column = (i*s + j)*Cout + co."""
    w = w.detach()
    cin, cout, s, s2 = w.shape
    if s != s2 or cin % 4 or cout % 32:
        raise ValueError("deconv: kernel == stride, cin % 4 == 0, cout % 32 == 0")
    ncol = s * s * cout
    t = w.permute(0, 2, 3, 1).reshape(cin // 4, 4, ncol).permute(0, 2, 1)
    return t.reshape(1, cin // 4, ncol, 4).contiguous(), ncol


_ZEROS = {}


def _zeros(c, device):
    """Cached all-zero shift vector (read-only)."""
    z = _ZEROS.get((c, device))
    if z is None:
        z = _ZEROS[(c, device)] = torch.zeros(c, device=device)
    return z


# ------------------------------------------------------------------------------------------------ raw launches
_PACKED = {}       # id(parameter) -> (version, weakref, packed, coutp, winograd-transformed | None): one packing per optimiser step


def _packed(r, weight, flipped=False, owner=None):
    """The kernel packing of a convolution weight (``flipped``: of its 180-degree-rotated, channel-transposed form -- the data
    gradient's weights), cached per parameter and version counter: a training step uses every weight in up to three passes."""
    import weakref
    # ``owner``: the nn.Parameter a reshaped view was taken from (nn.Linear's (cout, cin) weight seen as a 1x1 convolution): the cache
    # is keyed on it -- the view is a new object on every call (it shares the parameter's storage and version counter)
    own = owner if owner is not None else weight
    key = (id(own), flipped, tuple(weight.shape))
    ent = _PACKED.get(key)
    # ``_version`` does not move for writes through ``.data`` nor for ``module.to(device)`` (which re-points ``.data``): the storage
    # address and the device are part of the validity check, like the eval engine's _version()
    stamp = (weight._version, weight.data_ptr(), weight.device)
    if ent is not None and ent[0] == stamp and ent[1]() is own:
        return ent[2], ent[3], ent[4], ent
    wp, coutp = pack_conv_weight_dev(weight, flipped)
    cout, cin, ks = (weight.shape[1], weight.shape[0], weight.shape[2]) if flipped else (weight.shape[0], weight.shape[1], weight.shape[2])
    # engine.wino_rule's weight side: the class's transformed weights are made by conv_raw, for the kernel it picks, on first need
    u = bool(r.winograd and ks == 3 and cin >= 64 and cin % 8 == 0 and cout % 64 == 0 and cout == coutp)
    ent = [stamp, weakref.ref(own) if (own.is_leaf and isinstance(own, torch.nn.Parameter)) else None, wp, coutp, u, None, None, None, None, None]
    if ent[1] is not None:
        if len(_PACKED) > 4096:
            _PACKED.clear()
        _PACKED[key] = ent
    # ent[4]: Winograd-eligible weight (bool); ent[5]: the F(4x4,3x3)-transformed weights; ent[6]: the split-3 planes (seed_x3p); ent[7]: the
    # F(4x4) class as split-3 planes; ent[8]: the F(2x2) class as split-3 planes; ent[9]: the F(2x2) class's fp32 transformed weights
    return wp, coutp, u, ent


# AMP training (tools/train.py:50,107-130: the forward runs under ``amp.autocast`` and the loss goes through a GradScaler): while
# set, every Conv2d / ConvTranspose2d product of the step -- forward AND the data gradients, which autograd runs in the forward's
# dtype -- uses bf16 matrix-core operands with fp32 accumulation (conv_igemm_bf16); BatchNorm statistics, the pillar encoder,
# the attention, the loss and the weight gradients stay fp32, activations stay fp32 in HBM.  forward_train() sets it per step.
AMP_STEP = [False]


def set_amp_step(on):
    AMP_STEP[0] = bool(on)


class amp_scope:
    """``with amp_scope(flag):`` -- AMP_STEP inside the block, the previous value afterwards.  forward_train() wraps the forward in it
    (the flag does NOT outlive the call: a later fp32 use of the autograd nodes -- a stand-alone sub-module, an op-level call -- must not
    inherit the precision of somebody's last step); every node whose backward launches convolutions records the flag of ITS forward
    in ``ctx.amp`` and re-enters the scope in backward, which autograd runs after forward_train() has returned."""

    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        self.prev = AMP_STEP[0]
        AMP_STEP[0] = self.on
        # torch.autocast itself is switched OFF inside: AMP here means "the step's Conv2d / Linear products take bf16 matrix-core
        # operands" (AMP_STEP, read by the kernels' launchers); the few ATen ops of the step (folded HGT projections, the RTE embedding,
        # SplitAttn's squeeze path) feed fp32 kernels and must stay fp32 -- under autocast they returned bf16 tensors that the kernels
        # then read as fp32 (half the bytes: garbage weights, and a memory fault at the BASELINE grid)
        self._noautocast = torch.autocast("cuda", enabled=False)
        self._noautocast.__enter__()
        return self

    def __exit__(self, *exc):
        self._noautocast.__exit__(*exc)
        AMP_STEP[0] = self.prev
        return False


def conv_raw(x, weight, stride, pad, scale=None, shift=None, act=0, flipped=False, owner=None):
    """act(scale * conv2d(x, w) + shift) through the engine's launcher (direct or Winograd kernels, autotuned).  ``flipped``:
    convolve with the 180-degree-rotated, channel-transposed weights (the data gradient)."""
    from .engine import ConvLayer
    r = _runner(x.device)
    n, h, w, cin = x.shape
    wp, coutp, u, ent = _packed(r, weight, flipped, owner)
    cout, ks = (weight.shape[1], weight.shape[2]) if flipped else (weight.shape[0], weight.shape[2])
    sh = shift if shift is not None else _zeros(cout, x.device)
    L = ConvLayer(wp, scale, sh, cin, cout, coutp, ks, stride, pad, act)
    ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
    if u and r.wino_rule(L):   # transformed weights made on this stream, without engine._wu's cross-stream synchronise
        # the F(4x4,3x3) class (engine.wino4_rule: the 256 -> 256 layers at 100 x 352, forward and data gradient): its transformed
        # weights are cached with the packing, per parameter and version, built on the launch stream -- engine._wu4 would re-run the
        # transform on every call of this throw-away layer object and synchronise the host each time
        if AMP_STEP[0] and cin % 8 == 0:
            pass                                    # r.amp below: the bf16-operand GEMM takes the layer, no Winograd weights are read
        elif r.wino4 and r.wino4_rule(L, n, ho, wo):
            if r.wino_x3 and r.wino4_x3 and cin % 32 == 0 and not AMP_STEP[0]:      # engine.conv's choice: the split-3 form of the class
                if ent[7] is None:
                    u43 = torch.empty(r.lib.av2x_wino4_x3_weight_bytes(cin, coutp) // 2, dtype=torch.bfloat16, device=x.device)
                    _lib.check(r.lib.av2x_wino4_x3_pack_weights(_P(wp), cin, coutp, _P(u43), r.stream()), "av2x_wino4_x3_pack_weights")
                    ent[7] = u43
                L._wu43 = ent[7]
            else:
                if ent[5] is None:
                    u4 = torch.empty(r.lib.av2x_wino4_weight_bytes(cin, coutp) // 4, dtype=torch.float32, device=x.device)
                    _lib.check(r.lib.av2x_wino4_pack_weights(_P(wp), cin, coutp, _P(u4), r.stream()), "av2x_wino4_pack_weights")
                    ent[5] = u4
                L._wu4 = ent[5]
        elif r.wino_x3 and r.wino2_x3 and r.wino_x3_rule(L) and not AMP_STEP[0]:      # engine.conv's next choice: conv_wino_x3
            if ent[8] is None:
                u3 = torch.empty(r.lib.av2x_wino_x3_weight_bytes(cin, coutp) // 2, dtype=torch.bfloat16, device=x.device)
                _lib.check(r.lib.av2x_wino_x3_pack_weights(_P(wp), cin, coutp, _P(u3), r.stream()), "av2x_wino_x3_pack_weights")
                ent[8] = u3
            L._wu3 = ent[8]
        else:
            if ent[9] is None:
                u2 = torch.empty(r.lib.av2x_wino_weight_bytes(cin, coutp) // 4, dtype=torch.float32, device=x.device)
                _lib.check(r.lib.av2x_wino_pack_weights(_P(wp), cin, coutp, _P(u2), r.stream()), "av2x_wino_pack_weights")
                ent[9] = u2
            L._wu = ent[9]
    else:
        seed_x3p(r, L, ent)
    y = torch.empty((n, ho, wo, cout), dtype=torch.float32, device=x.device)
    r.amp = AMP_STEP[0] and cin % 8 == 0
    try:
        r.conv(L, x, n, h, w, y)
    finally:
        r.amp = False
    return y


def conv_wgrad(x, dz, weight_shape, stride, pad):
    r = _runner(x.device)
    n, h, w, cin = x.shape
    cout, _, ks, _ = weight_shape
    d, _, _ = _desc(n, h, w, cin, cout, cout, ks, stride, pad, 0)
    ws = torch.empty(int(r.lib.av2x_conv2d_wgrad_workspace_bytes(byref(d))) // 4 + 4, device=x.device)
    dw = torch.empty(tuple(weight_shape), dtype=torch.float32, device=x.device)
    prof = getattr(r, "wgrad_profile", None)
    if prof is not None:      # roofline pass of tools/train_bench.py: an event pair around the launch, on the launch stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(r.lib.av2x_conv2d_wgrad(byref(d), _P(x), _P(dz), _P(ws), _P(dw), r.stream()), "av2x_conv2d_wgrad")
    if prof is not None:
        e1.record()
        prof.append((2.0 * n * d.ho * d.wo * cout * ks * ks * cin, e0, e1, (n * d.ho * d.wo, cin, cout, ks, stride)))
    return dw


def conv_dgrad(dz, weight, stride, pad, in_hw, owner=None):
    """Data gradient of conv2d: the forward kernel on the 180-degree-rotated, channel-transposed weights (stride 2: on the
    zero-upsampled dz -- exact, the inserted zeros contribute nothing)."""
    n, ho, wo, cout = dz.shape
    h, w = in_hw
    cin, ks = weight.shape[1], weight.shape[2]
    if cout % 4:
        raise NotImplementedError("data gradient needs cout % 4 == 0 (pad the head convolutions)")
    if stride == 1:
        if pad != ks // 2:
            raise NotImplementedError("'same' padding only")
        src = dz
    elif stride == 2 and ks % 2 == 1 and pad == ks // 2 and 2 * (ho - 1) < h and 2 * (wo - 1) < w:
        # y[i] = sum_k xpad[2 i + k] w[k]  =>  dx = the stride-1 'same' correlation of u (u[2 i] = dz[i], zeros elsewhere) with the flipped
        # taps: 3x3 (the BEV backbone), 1x1 (ResNet's downsample shortcut) and 7x7 (BevEncode's first convolution) alike, even or odd sizes
        src = torch.zeros((n, h, w, cout), dtype=torch.float32, device=dz.device)
        src[:, 0:2 * ho:2, 0:2 * wo:2] = dz
    else:
        raise NotImplementedError("data gradient: stride 1, or an odd kernel with stride 2 and pad = ks // 2")
    return conv_raw(src, weight, 1, ks // 2, flipped=True, owner=owner)


def bn_stats(z):
    r = _runner(z.device)
    c = z.shape[-1]
    rows = z.numel() // c
    ws = torch.empty(int(r.lib.av2x_bn_workspace_bytes(rows, c)) // 8 + 1, dtype=torch.float64, device=z.device)
    mean = torch.empty(c, device=z.device)
    var = torch.empty(c, device=z.device)
    _lib.check(r.lib.av2x_bn_stats(_P(z), rows, c, _P(ws), _P(mean), _P(var), r.stream()), "av2x_bn_stats")
    return mean, var, rows


def affine_act(z, scale, shift, act, out=None):
    r = _runner(z.device)
    c = z.shape[-1]
    rows = z.numel() // c
    y = out if out is not None else torch.empty_like(z)
    _lib.check(r.lib.av2x_affine_act(_P(z), rows, c, _P(scale), _P(shift), 1 if act else 0, _P(y), r.stream()), "av2x_affine_act")
    return y


def bn_backward(dy, z, mean, rstd, scale, shift, act):
    r = _runner(z.device)
    c = z.shape[-1]
    rows = z.numel() // c
    ws = torch.empty(int(r.lib.av2x_bn_workspace_bytes(rows, c)) // 8 + 1, dtype=torch.float64, device=z.device)
    dgamma, dbeta = torch.empty(c, device=z.device), torch.empty(c, device=z.device)
    dz = torch.empty_like(z)
    _lib.check(r.lib.av2x_bn_backward(_P(dy), _P(z), rows, c, _P(mean), _P(rstd), _P(scale), _P(shift), 1 if act else 0, _P(ws),
                                      _P(dgamma), _P(dbeta), _P(dz), r.stream()), "av2x_bn_backward")
    return dz, dgamma, dbeta


def _fold(mean, var, gamma, beta, eps):
    rstd = torch.rsqrt(var + eps)
    scale = gamma.detach() * rstd
    shift = beta.detach() - mean * scale
    return rstd, scale, shift


def bn_finalize(mean, var, count, gamma, beta, eps, running=None, momentum=BN_MOMENTUM):
    """One launch: rstd, scale = gamma * rstd, shift = beta - mean * scale and -- ``running`` = (running_mean, running_var,
    num_batches_tracked | None, times) -- nn.BatchNorm's train-mode update applied ``times`` times with these statistics."""
    r = _runner(mean.device)
    c = mean.numel()
    rstd, scale, shift = torch.empty_like(mean), torch.empty_like(mean), torch.empty_like(mean)
    rm = rv = nbt = None
    times = 0
    if running is not None:
        rm, rv, nbt, times = running
    _lib.check(r.lib.av2x_bn_finalize(_P(mean), _P(var), _P(gamma.detach()), _P(beta.detach()), c, float(eps), int(count), float(momentum),
                                      int(times), _P(rstd), _P(scale), _P(shift), _P(rm), _P(rv), _P(nbt), r.stream()), "av2x_bn_finalize")
    if running is not None:   # the kernel wrote the buffers behind torch's back: consumers key re-packing on the version counters
        for t in (rm, rv, nbt):
            if t is not None:
                torch.autograd.graph.increment_version(t)
    return rstd, scale, shift


def bn_train_forward(z, gamma, beta, eps, act, running=None, momentum=BN_MOMENTUM):
    """Train-mode BatchNorm (+ ReLU) of an NHWC map in one C call: -> (y, mean, var, rstd, scale, shift, count)."""
    r = _runner(z.device)
    c = z.shape[-1]
    rows = z.numel() // c
    ws = torch.empty(int(r.lib.av2x_bn_workspace_bytes(rows, c)) // 8 + 1, dtype=torch.float64, device=z.device)
    st5 = torch.empty((5, c), dtype=torch.float32, device=z.device)
    y = torch.empty_like(z)
    rm = rv = nbt = None
    times = 0
    if running is not None:
        rm, rv, nbt, times = running
    _lib.check(r.lib.av2x_bn_train_forward(_P(z), rows, c, _P(gamma.detach()), _P(beta.detach()), float(eps), float(momentum), int(times),
                                           1 if act else 0, _P(ws), _P(st5), _P(y), _P(rm), _P(rv), _P(nbt), r.stream()),
               "av2x_bn_train_forward")
    if running is not None:   # the kernel wrote the buffers behind torch's back: consumers key re-packing on the version counters
        for t in (rm, rv, nbt):
            if t is not None:
                torch.autograd.graph.increment_version(t)
    return y, st5[0], st5[1], st5[2], st5[3], st5[4], rows


def update_running_stats(running_mean, running_var, num_batches_tracked, stats, times=1, momentum=BN_MOMENTUM):
    """nn.BatchNorm's train-mode side effect, applied ``times`` times with the same batch statistics (the reference runs
    the backbone more than once per step on the same input: airv2x_where2com.py:119,124)."""
    mean, var, count = stats
    unbiased = var * (count / max(count - 1, 1))
    with torch.no_grad():
        for _ in range(times):
            running_mean.mul_(1 - momentum).add_(mean, alpha=momentum)
            running_var.mul_(1 - momentum).add_(unbiased, alpha=momentum)
        if num_batches_tracked is not None:
            num_batches_tracked.add_(times)


_SIDE = {}
OVERLAP_WGRAD = os.environ.get("AV2X_TRAIN_OVERLAP", "1") != "0"


def _wgrad_and_dgrad(x, dz, weight, stride, pad, need_w, need_x, owner=None):
    """The two gradients of a convolution are independent given dz: the weight gradient (one round of <= 256 workgroups, one per
    CU) goes to a side stream, the data gradient (small maps: partially filled waves of workgroups) stays on the current one --
    each fills what the other leaves idle.  Same kernels, same bits."""
    if not (need_w and need_x and OVERLAP_WGRAD):
        dw = conv_wgrad(x, dz, weight.shape, stride, pad) if need_w else None
        dx = conv_dgrad(dz, weight, stride, pad, x.shape[1:3], owner) if need_x else None
        return dw, dx
    main = torch.cuda.current_stream(x.device)
    side = _SIDE.get(x.device)
    if side is None:
        from .engine import pooled_streams      # the process-wide pool: see engine.pooled_streams (four hardware queues)
        side = _SIDE[x.device] = pooled_streams(x.device, 1)[0]
    side.wait_stream(main)
    with torch.cuda.stream(side):
        dw = conv_wgrad(x, dz, weight.shape, stride, pad)
    dx = conv_dgrad(dz, weight, stride, pad, x.shape[1:3], owner)
    main.wait_stream(side)
    dw.record_stream(main)      # allocated under the side stream, consumed by autograd on the current one
    x.record_stream(side)       # read by the side stream: the allocator must not recycle them before it is done
    dz.record_stream(side)
    return dw, dx


# ------------------------------------------------------------------------------------------------ Conv + BN(batch) + ReLU
class ConvBNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, gamma, beta, stride, pad, eps, act, stats_out, running):
        _check_dev(x)
        x = x.contiguous()
        z = conv_raw(x, weight, stride, pad)
        ctx.amp = AMP_STEP[0]
        y, mean, var, rstd, scale, shift, count = bn_train_forward(z, gamma, beta, eps, act, running)
        if stats_out is not None:
            stats_out.append((mean, var, count))
        ctx.save_for_backward(x, weight, z, mean, rstd, scale, shift)
        ctx.cfg = (stride, pad, act)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, z, mean, rstd, scale, shift = ctx.saved_tensors
        stride, pad, act = ctx.cfg
        dz, dgamma, dbeta = bn_backward(dy.contiguous(), z, mean, rstd, scale, shift, act)
        with amp_scope(ctx.amp):
            dw, dx = _wgrad_and_dgrad(x, dz, weight, stride, pad, ctx.needs_input_grad[1], ctx.needs_input_grad[0])
        return dx, dw, dgamma, dbeta, None, None, None, None, None, None


def conv_bn_act(x, weight, gamma, beta, stride=1, pad=1, eps=BN_EPS, act=True, stats_out=None, running=None):
    """``running`` = (running_mean, running_var, num_batches_tracked | None, times): updated in place, as nn.BatchNorm does."""
    return ConvBNAct.apply(x, weight, gamma, beta, stride, pad, eps, act, stats_out, running)


# ------------------------------------------------------------------------------------------------ ConvTranspose(k = s) + BN + ReLU
def _space_to_depth(t, s):
    n, hs, ws, c = t.shape
    if s == 1:
        return t
    return t.view(n, hs // s, s, ws // s, s, c).permute(0, 1, 3, 2, 4, 5).reshape(n, hs // s, ws // s, s * s * c)


class DeconvBNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, gamma, beta, eps, act, stats_out, running):
        from .engine import ConvLayer
        _check_dev(x)
        x = x.contiguous()
        r = _runner(x.device)
        n, h, w, cin = x.shape
        _, cout, s, _ = weight.shape
        wp, ncol = pack_deconv_weight_dev(weight)
        L = ConvLayer(wp, None, _zeros(cout, x.device), cin, cout, ncol, 1, 1, 0, 0, _lib.AV2X_DECONV, s)
        seed_x3p(r, L)
        z = torch.empty((n, h * s, w * s, cout), dtype=torch.float32, device=x.device)
        r.amp = AMP_STEP[0] and cin % 8 == 0
        try:
            r.conv(L, x, n, h, w, z)
        finally:
            r.amp = False
        y, mean, var, rstd, scale, shift, count = bn_train_forward(z, gamma, beta, eps, act, running)
        if stats_out is not None:
            stats_out.append((mean, var, count))
        ctx.save_for_backward(x, weight, z, mean, rstd, scale, shift)
        ctx.act = act
        ctx.amp = AMP_STEP[0]
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, z, mean, rstd, scale, shift = ctx.saved_tensors
        cin, cout, s, _ = weight.shape
        dz, dgamma, dbeta = bn_backward(dy.contiguous(), z, mean, rstd, scale, shift, ctx.act)
        d2 = _space_to_depth(dz, s).contiguous()                                   # (n, h, w, s*s*cout), column = (i, j, co)
        dw = dx = None
        if ctx.needs_input_grad[1]:
            g = conv_wgrad(x, d2, (s * s * cout, cin, 1, 1), 1, 0)                 # [(i, j, co)][ci]
            dw = g.view(s, s, cout, cin).permute(3, 2, 0, 1).contiguous()
        if ctx.needs_input_grad[0]:
            wb = weight.detach().permute(0, 2, 3, 1).reshape(cin, s * s * cout, 1, 1)
            with amp_scope(ctx.amp):
                dx = conv_raw(d2, wb, 1, 0)
        return dx, dw, dgamma, dbeta, None, None, None, None


def deconv_bn_act(x, weight, gamma, beta, eps=BN_EPS, act=True, stats_out=None, running=None):
    return DeconvBNAct.apply(x, weight, gamma, beta, eps, act, stats_out, running)


# ------------------------------------------------------------------------------------------------ Conv + bias (+ ReLU)
class ConvBiasAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, act):
        _check_dev(x)
        x = x.contiguous()
        y = conv_raw(x, weight, stride, pad, None, bias.detach(), 1 if act else 0)
        ctx.save_for_backward(x, weight, y)
        ctx.cfg = (stride, pad, act)
        ctx.amp = AMP_STEP[0]
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        stride, pad, act = ctx.cfg
        r = _runner(x.device)
        dy = dy.contiguous()
        cout = weight.shape[0]
        rows = y.numel() // cout
        if act:
            dz = torch.empty_like(dy)
            _lib.check(r.lib.av2x_act_backward(_P(y), _P(dy), None, rows, cout, 1, _P(dz), r.stream()), "av2x_act_backward")
        else:
            dz = dy
        db = None
        if ctx.needs_input_grad[2]:
            ws = torch.empty(int(r.lib.av2x_channel_sum_workspace_bytes(rows, cout)) // 4 + 4, device=x.device)
            db = torch.empty(cout, device=x.device)
            _lib.check(r.lib.av2x_channel_sum(_P(dz), rows, cout, _P(ws), _P(db), r.stream()), "av2x_channel_sum")
        with amp_scope(ctx.amp):
            dw, dx = _wgrad_and_dgrad(x, dz, weight, stride, pad, ctx.needs_input_grad[1], ctx.needs_input_grad[0])
        return dx, dw, db, None, None, None


def conv_bias_act(x, weight, bias, stride=1, pad=1, act=True):
    return ConvBiasAct.apply(x, weight, bias, stride, pad, act)


# ------------------------------------------------------------------------------------------------ mask, per-pixel attention
class MaskMul(torch.autograd.Function):
    """x (n, h, w, c) * mask (n, h, w): `x = x * communication_masks` (where2comm_fuse.py:236); the mask carries no gradient."""

    @staticmethod
    def forward(ctx, x, mask):
        _check_dev(x)
        r = _runner(x.device)
        y = x.contiguous().clone()
        n, h, w, c = y.shape
        _lib.check(r.lib.av2x_apply_mask(_P(y), _P(mask), n, h * w, c, r.stream()), "av2x_apply_mask")
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        r = _runner(dy.device)
        dx = dy.contiguous().clone()
        n, h, w, c = dx.shape
        _lib.check(r.lib.av2x_apply_mask(_P(dx), _P(mask), n, h * w, c, r.stream()), "av2x_apply_mask")
        return dx, None


class PixelAttn(torch.autograd.Function):
    """AttentionFusion of one sample (where2comm_fuse.py:152-164): x (k, h, w, c), agent 0 = ego -> (h, w, c)."""

    @staticmethod
    def forward(ctx, x):
        _check_dev(x)
        r = _runner(x.device)
        x = x.contiguous()
        k, h, w, c = x.shape
        out = torch.empty((h, w, c), dtype=torch.float32, device=x.device)
        arr = (c_void_p * k)(*[x[j].data_ptr() for j in range(k)])
        _lib.check(r.lib.av2x_pixel_attn_fuse(arr, k, h * w, c, _P(out), r.stream()), "av2x_pixel_attn_fuse")
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        r = _runner(x.device)
        k, h, w, c = x.shape
        dout = dout.contiguous()
        dx = torch.empty_like(x)
        arr = (c_void_p * k)(*[x[j].data_ptr() for j in range(k)])
        darr = (c_void_p * k)(*[dx[j].data_ptr() for j in range(k)])
        _lib.check(r.lib.av2x_pixel_attn_backward(arr, k, h * w, c, _P(dout), darr, r.stream()), "av2x_pixel_attn_backward")
        return dx


# ------------------------------------------------------------------------------------------------ pillar encoders + scatter
def _pillar_stats(lib, st, vf, vc, vn, geom, W, dev):
    M = int(vf.shape[0])
    ws = torch.empty(int(lib.av2x_pillar_train_workspace_bytes(M)) // 8 + 1, dtype=torch.float64, device=dev)
    mom = torch.empty(110, dtype=torch.float64, device=dev)
    _lib.check(lib.av2x_pillar_moments(_P(vf), _P(vc), _P(vn), M, geom, _P(ws), _P(mom), st), "av2x_pillar_moments")
    N = 32.0 * M
    S, F = mom[:10], mom[10:].view(10, 10)
    Wd = W.detach().double()
    mean64 = Wd @ S / N
    var64 = ((Wd @ F) * Wd).sum(1) / N - mean64 * mean64
    return S, F, mean64, mean64.float(), var64.clamp_min(0).float(), N


def _pillar_param_grads(out, S, F, mean64, rstd, scale, Wc, N):
    """(d linear.weight, d norm.weight, d norm.bias) from the kernel's G[c][k] = sum g feat_k, d beta, d gamma and the moments:
    d lin_r = scale (g_r - d beta / N - xhat_r d gamma / N), dW = sum_r d lin_r (x) feat_r, with
    sum_r xhat_r (x) feat_r = rstd (W F - mean S^T)."""
    G, dbeta, dgamma = out[:, :10], out[:, 10], out[:, 11]
    Wd = Wc.double()
    xf = rstd.double().unsqueeze(1) * (Wd @ F - mean64.unsqueeze(1) * S.unsqueeze(0))
    dW = scale.double().unsqueeze(1) * (G - (dbeta / N).unsqueeze(1) * S.unsqueeze(0) - (dgamma / N).unsqueeze(1) * xf)
    return dW.float(), dgamma.float(), dbeta.float()


class PillarEncode(torch.autograd.Function):
    """Every agent type's PillarVFE (train-mode BatchNorm1d) + PointPillarScatter into one canvas (n, ny, nx, 64).

    ``groups``: list of dicts {vf (M,32,4) f32, vc (M,4) i32, vn (M,) i32, slots [canvas slot of the type's agent i], geom
    (ctypes float[6])}; ``params`` = (linear.weight, norm.weight, norm.bias) per group, flattened.  ``stats_out`` receives one
    (mean, biased var, count) per group."""

    @staticmethod
    def forward(ctx, groups, n_total, ny, nx, eps, stats_out, *params):
        dev = groups[0]["vf"].device
        r = _runner(dev)
        lib, st = r.lib, r.stream()
        canvas = torch.zeros((n_total, ny, nx, 64), dtype=torch.float32, device=dev)
        saved = []
        for gi, g in enumerate(groups):
            W, gamma, beta = params[3 * gi:3 * gi + 3]
            vf, vc, vn = g["vf"], g["vc"], g["vn"]
            M = int(vf.shape[0])
            geom = ctypes.cast(g["geom"], c_void_p)
            S, F, mean64, mean, var, N = _pillar_stats(lib, st, vf, vc, vn, geom, W, dev)
            rstd, scale, shift = _fold(mean, var, gamma, beta, eps)
            Wc = W.detach().contiguous()
            sl = g["slots"]
            contiguous = sl == list(range(sl[0], sl[0] + len(sl)))
            smap = None if contiguous else torch.tensor(sl, dtype=torch.int32, device=dev)
            _lib.check(lib.av2x_pillar_vfe_scatter(_P(vf), _P(vc), _P(vn), M, _P(Wc), _P(scale), _P(shift), geom, _P(canvas), sl[0],
                                                   _P(smap), len(sl), ny, nx, st), "av2x_pillar_vfe_scatter")
            if stats_out is not None:
                stats_out.append((mean, var, int(N)))
            saved.append((S, F, mean64, rstd, scale, shift, smap, Wc))
        ctx.groups, ctx.saved, ctx.dims = groups, saved, (ny, nx)
        return canvas

    @staticmethod
    def backward(ctx, dcanvas):
        dev = dcanvas.device
        r = _runner(dev)
        lib, st = r.lib, r.stream()
        dcanvas = dcanvas.contiguous()
        ny, nx = ctx.dims
        grads = []
        for gi, g in enumerate(ctx.groups):
            S, F, mean64, rstd, scale, shift, smap, Wc = ctx.saved[gi]
            vf, vc, vn = g["vf"], g["vc"], g["vn"]
            M = int(vf.shape[0])
            N = 32.0 * M
            ws = torch.empty(int(lib.av2x_pillar_train_workspace_bytes(M)) // 8 + 1, dtype=torch.float64, device=dev)
            out = torch.empty((64, 12), dtype=torch.float64, device=dev)
            mean = mean64.float()
            sl = g["slots"]
            _lib.check(lib.av2x_pillar_vfe_backward(_P(vf), _P(vc), _P(vn), M, _P(Wc), _P(scale), _P(shift), _P(mean), _P(rstd),
                                                    ctypes.cast(g["geom"], c_void_p), _P(dcanvas), sl[0], _P(smap), len(sl), ny, nx,
                                                    _P(ws), _P(out), st), "av2x_pillar_vfe_backward")
            grads += list(_pillar_param_grads(out, S, F, mean64, rstd, scale, Wc, N))
        return (None, None, None, None, None, None) + tuple(grads)


def pillar_encode(groups, n_total, ny, nx, params, eps=BN_EPS, stats_out=None):
    return PillarEncode.apply(groups, n_total, ny, nx, eps, stats_out, *params)


# ------------------------------------------------------------------------------------------------ the stand-alone halves
class PillarFeatures(torch.autograd.Function):
    """PillarVFE alone in train mode (airv2x_pillar_vfe.py:105-160): (M, 32, 4) pillars -> (M, 64) features, BatchNorm1d batch
    statistics, differentiable in the Linear / BatchNorm1d parameters."""

    @staticmethod
    def forward(ctx, vf, vc, vn, geom, eps, stats_out, W, gamma, beta):
        dev = vf.device
        r = _runner(dev)
        lib, st = r.lib, r.stream()
        g = ctypes.cast(geom, c_void_p)
        S, F, mean64, mean, var, N = _pillar_stats(lib, st, vf, vc, vn, g, W, dev)
        rstd, scale, shift = _fold(mean, var, gamma, beta, eps)
        Wc = W.detach().contiguous()
        out = torch.empty((vf.shape[0], 64), dtype=torch.float32, device=dev)
        _lib.check(lib.av2x_pillar_vfe(_P(vf), _P(vc), _P(vn), int(vf.shape[0]), _P(Wc), _P(scale), _P(shift), g, _P(out), st), "av2x_pillar_vfe")
        if stats_out is not None:
            stats_out.append((mean, var, int(N)))
        ctx.saved = (vf, vc, vn, geom, S, F, mean64, rstd, scale, shift, Wc, N)
        return out

    @staticmethod
    def backward(ctx, dfeat):
        vf, vc, vn, geom, S, F, mean64, rstd, scale, shift, Wc, N = ctx.saved
        dev = vf.device
        r = _runner(dev)
        M = int(vf.shape[0])
        ws = torch.empty(int(r.lib.av2x_pillar_train_workspace_bytes(M)) // 8 + 1, dtype=torch.float64, device=dev)
        out = torch.empty((64, 12), dtype=torch.float64, device=dev)
        mean = mean64.float()
        _lib.check(r.lib.av2x_pillar_vfe_backward_rows(_P(vf), _P(vc), _P(vn), M, _P(Wc), _P(scale), _P(shift), _P(mean), _P(rstd),
                                                       ctypes.cast(geom, c_void_p), _P(dfeat.contiguous()), _P(ws), _P(out), r.stream()),
                   "av2x_pillar_vfe_backward_rows")
        dW, dgamma, dbeta = _pillar_param_grads(out, S, F, mean64, rstd, scale, Wc, N)
        return None, None, None, None, None, None, dW, dgamma, dbeta


class PillarScatter(torch.autograd.Function):
    """PointPillarScatter (point_pillar_scatter.py:39-80): (M, C) features + (M, 4) [agent, z, y, x] -> (n, ny, nx, C) canvas;
    backward gathers the canvas gradient at the pillars."""

    @staticmethod
    def forward(ctx, feats, coords, n, ny, nx):
        r = _runner(feats.device)
        feats = feats.contiguous()
        M, C = feats.shape
        canvas = torch.zeros((n, ny, nx, C), dtype=torch.float32, device=feats.device)
        _lib.check(r.lib.av2x_pillar_scatter(_P(feats), _P(coords), M, C, _P(canvas), n, ny, nx, r.stream()), "av2x_pillar_scatter")
        ctx.saved = (coords, M, C, n, ny, nx)
        return canvas

    @staticmethod
    def backward(ctx, dcanvas):
        coords, M, C, n, ny, nx = ctx.saved
        r = _runner(dcanvas.device)
        dcanvas = dcanvas.contiguous()
        dfeat = torch.empty((M, C), dtype=torch.float32, device=dcanvas.device)
        _lib.check(r.lib.av2x_pillar_gather(_P(dcanvas), _P(coords), M, C, _P(dfeat), n, ny, nx, r.stream()), "av2x_pillar_gather")
        return dfeat, None, None, None, None
