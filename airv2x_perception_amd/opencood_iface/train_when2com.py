"""Train-mode forward of ``Airv2xWhen2com`` (models/airv2x_when2com.py:112-151 with ``self.training``): the graph the reference hands
to torch autograd (tools/train.py:220-247), built from HIP forward / backward ops.

    encoders, BaseBEVBackbone, DownsampleConv                                     train_ops (shared with the other models)
    When2comFusion.forward (when2com_modules/when2com.py:60-134), per sample:
        warp_affine_simple of every agent into the ego frame (:118-120)           WarpAffineSimpleFn   (bilinear adjoint, fixed point)
        policy_net4: 5 x Conv3x3 (+ bias) + BatchNorm (batch statistics) + ReLU   train_ops.conv_bn_act (+ the bias in the running mean)
        km_generator x 2: Linear(256 h w -> 256) + ReLU, Linear -> 128 + ReLU,
                          Linear -> key / query size (:283-297)                  LinearRowsFn          (HBM-bound row GEMMs, w read once)
        attention_net.linear, softmax over the keys, weighted sum of the maps     When2comFuseFn
    cls / reg / obj heads                                                         one 32-column GEMM

``mode: softmax`` only (the shipped configuration; the reference's "activated" branch raises IndexError, when2com.py:58).
"""
from __future__ import annotations

from ctypes import c_void_p

import numpy as np
import torch

from .. import _lib
from . import train_ops as T
from .autograd import _runner
from .engine import frame_layout
from .train_where2com import _block, _deblock, _heads, _running, _shrink, encode_train
from .when2com_engine import normalized_pairwise

_P = T._P


class WarpAffineSimpleFn(torch.autograd.Function):
    """warp_affine_simple (torch_transformation_utils.py:327-334: affine_grid + grid_sample, bilinear, zeros, align_corners=False)
    with a constant theta (n, 2, 3)."""

    @staticmethod
    def forward(ctx, x, theta):
        T._check_dev(x)
        r = _runner(x.device)
        x = x.contiguous()
        n, H, W, C = x.shape
        y = torch.empty_like(x)
        _lib.check(r.lib.av2x_warp_affine_simple(_P(x), _P(theta), _P(y), n, H, W, C, r.stream()), "av2x_warp_affine_simple")
        ctx.save_for_backward(theta)
        return y

    @staticmethod
    def backward(ctx, dy):
        (theta,) = ctx.saved_tensors
        r = _runner(dy.device)
        dy = dy.contiguous()
        n, H, W, C = dy.shape
        dx = torch.empty_like(dy)
        ws = torch.empty(int(r.lib.av2x_warp_affine_backward_workspace_bytes(n, H, W, C)), dtype=torch.uint8, device=dy.device)
        _lib.check(r.lib.av2x_warp_affine_simple_backward(_P(dy), _P(theta), _P(dx), _P(ws), n, H, W, C, r.stream()),
                   "av2x_warp_affine_simple_backward")
        return dx, None


def warp_affine_simple(x, theta):
    return WarpAffineSimpleFn.apply(x, theta)


class LinearRowsFn(torch.autograd.Function):
    """y (m, n) = act(x (m, k) W^T + b) for a few rows (one per agent); W is nn.Linear's (n, k) as stored.  Forward and backward stream
    W once each (av2x_linear_rows, av2x_linear_rows_backward): the first km_generator layer is 577 MB at the default grid."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        T._check_dev(x)
        r = _runner(x.device)
        x = x.contiguous()
        m, k = x.shape
        n = weight.shape[0]
        w = weight.detach().contiguous()
        y = torch.empty((m, n), dtype=torch.float32, device=x.device)
        need = int(r.lib.av2x_linear_rows_workspace_bytes(m, n, k))
        ws = torch.empty(max(need // 4, 1), dtype=torch.float32, device=x.device)
        _lib.check(r.lib.av2x_linear_rows(_P(x), _P(w), _P(bias.detach()) if bias is not None else None, m, n, k, int(act), _P(y), _P(ws),
                                          ws.numel() * 4, r.stream()), "av2x_linear_rows")
        ctx.save_for_backward(x, weight, y)
        ctx.cfg = (int(act), bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        act, has_bias = ctx.cfg
        r = _runner(x.device)
        dy = dy.contiguous()
        m, k = x.shape
        n = weight.shape[0]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty((n, k), dtype=torch.float32, device=x.device) if ctx.needs_input_grad[1] else None
        db = torch.empty((n,), dtype=torch.float32, device=x.device) if (has_bias and ctx.needs_input_grad[2]) else None
        _lib.check(r.lib.av2x_linear_rows_backward(_P(x), _P(weight.detach().contiguous()), _P(y), _P(dy), m, n, k, act, _P(dx), _P(dw), _P(db),
                                                   r.stream()), "av2x_linear_rows_backward")
        return dx, dw, db, None


def linear_rows(x, weight, bias=None, act=0):
    return LinearRowsFn.apply(x, weight, bias, act)


class When2comFuseFn(torch.autograd.Function):
    """MIMOGeneralDotProductAttention.forward (:320-348) for one query (the ego): p = softmax_j(keys_j . q), out = sum_j p_j maps_j."""

    @staticmethod
    def forward(ctx, keys, q, maps):
        T._check_dev(maps)
        r = _runner(maps.device)
        keys, q, maps = keys.contiguous(), q.contiguous(), maps.contiguous()
        n = maps.shape[0]
        per = maps[0].numel()
        out = torch.empty((1,) + tuple(maps.shape[1:]), dtype=torch.float32, device=maps.device)
        coef = torch.empty((n,), dtype=torch.float32, device=maps.device)
        arr = (c_void_p * n)(*[maps[j].data_ptr() for j in range(n)])
        _lib.check(r.lib.av2x_when2com_fuse(_P(keys), _P(q), n, keys.shape[1], arr, per, _P(out), _P(coef), r.stream()), "av2x_when2com_fuse")
        ctx.save_for_backward(keys, q, maps, coef)
        return out

    @staticmethod
    def backward(ctx, dout):
        keys, q, maps, coef = ctx.saved_tensors
        r = _runner(maps.device)
        dout = dout.contiguous()
        n = maps.shape[0]
        per = maps[0].numel()
        dmaps = torch.empty_like(maps) if ctx.needs_input_grad[2] else None
        dkeys = torch.empty_like(keys) if ctx.needs_input_grad[0] else None
        dq = torch.empty_like(q) if ctx.needs_input_grad[1] else None
        arr = (c_void_p * n)(*[maps[j].data_ptr() for j in range(n)])
        darr = (c_void_p * n)(*[dmaps[j].data_ptr() for j in range(n)]) if dmaps is not None else None
        ws = torch.empty(int(r.lib.av2x_when2com_fuse_backward_workspace_bytes(n)) // 4 + 1, dtype=torch.float32, device=maps.device)
        _lib.check(r.lib.av2x_when2com_fuse_backward(_P(keys), _P(q), _P(coef), n, keys.shape[1], arr, per, _P(dout), darr, _P(dkeys), _P(dq),
                                                     _P(ws), r.stream()), "av2x_when2com_fuse_backward")
        return dkeys, dq, dmaps


def when2com_attention(keys, q, maps):
    return When2comFuseFn.apply(keys, q, maps)


POLICY_BN_EPS, POLICY_BN_MOMENTUM = 1e-5, 0.1       # nn.BatchNorm2d defaults (conv2DBatchNormRelu :160-163), not the backbone's 1e-3 / 0.01


def _policy(P, sd, x, prefix="fusion_net.query_key_net."):
    """policy_net4 (:300-317): conv2DBatchNormRelu x 5.  The convolutions carry a bias in front of their BatchNorm: it cancels in the
    normalised output (its gradient is exactly zero) but is part of the batch mean nn.BatchNorm folds into ``running_mean``."""
    for i, s in enumerate((1, 1, 2, 1, 2), 1):
        p = f"{prefix}conv{i}.cbr_unit"
        st = []
        x = T.conv_bn_act(x, P[p + ".0.weight"], P[p + ".1.weight"], P[p + ".1.bias"], s, 1, eps=POLICY_BN_EPS, stats_out=st)
        mean, var, count = st[0]
        T.update_running_stats(sd[p + ".1.running_mean"], sd[p + ".1.running_var"], sd.get(p + ".1.num_batches_tracked"),
                               (mean + P[p + ".0.bias"].detach(), var, count), 1, momentum=POLICY_BN_MOMENTUM)
    return x


def _km(P, x, prefix):
    """km_generator (:283-297) on NCHW-flattened policy maps (m, 256 h w)."""
    x = linear_rows(x, P[prefix + "fc.0.weight"], P[prefix + "fc.0.bias"], 1)
    x = linear_rows(x, P[prefix + "fc.2.weight"], P[prefix + "fc.2.bias"], 1)
    return linear_rows(x, P[prefix + "fc.4.weight"], P[prefix + "fc.4.bias"], 0)


def when2com_fusion(P, sd, s, record_len, theta, prefix="fusion_net."):
    """s (sum n, H, W, C) shrink-header maps -> (B, H, W, C); theta (B, L, L, 2, 3) numpy: the normalised pairwise matrices."""
    outs, a0 = [], 0
    dev = s.device
    for b, k in enumerate(record_len):
        th = torch.from_numpy(np.ascontiguousarray(theta[b, 0, :k], dtype=np.float32)).to(dev)
        warped = warp_affine_simple(s[a0:a0 + k], th)
        a0 += k
        qk = _policy(P, sd, warped, prefix + "query_key_net.")
        flat = qk.permute(0, 3, 1, 2).reshape(k, -1)                 # the reference flattens NCHW (c, y, x) (:296); data movement only
        keys = _km(P, flat, prefix + "key_net.")
        query = _km(P, flat[0:1], prefix + "query_net.")
        q = linear_rows(query, P[prefix + "attention_net.linear.weight"], P[prefix + "attention_net.linear.bias"], 0)
        outs.append(when2com_attention(keys, q, warped))
    return torch.cat(outs, 0) if len(outs) > 1 else outs[0]


def _forward_train(model, data_dict):
    args = model.args
    P = dict(model.named_parameters())
    sd = model.state_dict(keep_vars=True)
    dev = next(iter(P.values())).device
    if dev.type != "cuda":
        raise RuntimeError("Airv2xWhen2com (MI355X build) has no CPU path: move the module to the GPU (model.to('cuda'))")
    r = _runner(dev)
    mf = args["modality_fusion"]
    bb, w2 = mf["base_bev_backbone"], args["when2com_fusion"]
    if w2["mode"] != "softmax":
        raise NotImplementedError("When2com mode %r: only the shipped 'softmax' mode is built" % (w2["mode"],))
    from ..synth import model_compression
    compression = model_compression(args)          # NaiveCompressor(256, args["compression"]) behind the shrink header, as the reference reads it
    record_len, slots = frame_layout(args["collaborators"], data_dict)
    B, n = len(record_len), sum(record_len)
    if n == 0:
        raise ValueError("empty frame: no agent has lidar input")
    canvas, nz = encode_train(args, P, sd, data_dict, slots, n, dev, r)
    feats, x = [], canvas
    for i, (ln, st) in enumerate(zip(bb["layer_nums"], bb["layer_strides"])):
        x = _block(P, sd, i, x, ln, st, 1)
        feats.append(x)
    s = torch.cat([_deblock(P, sd, i, f, 1) for i, f in enumerate(feats)], -1)
    s = _shrink(P, mf["shrink_header"], s)
    if compression:
        from .train_cobevt import _compressor
        s = _compressor(P, sd, s)
    H, W = s.shape[1:3]
    # communication_rates (:118): non-zeros of the maps the agents share
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    sc = s.detach().contiguous()
    _lib.check(r.lib.av2x_count_nonzero(_P(sc), sc.numel(), _P(cnt), r.stream()), "av2x_count_nonzero")
    pair = data_dict["img_pairwise_t_matrix_collab"]
    pair = pair.detach().cpu().numpy() if isinstance(pair, torch.Tensor) else np.asarray(pair)
    if pair.shape[0] != B:
        raise ValueError("img_pairwise_t_matrix_collab batch size does not match record_len")
    theta = normalized_pairwise(pair, H, W, w2["voxel_size"][0], w2["downsample_rate"])
    fused = when2com_fusion(P, sd, s, record_len, theta)
    names = ["cls_head", "reg_head"] + (["obj_head"] if args["obj_head"] else [])
    outs = _heads(P, names, fused)
    out = {"psm": outs[0], "rm": outs[1]}
    if args["obj_head"]:
        out["obj"] = outs[2]
    out.update({"mask": 0, "comm_rate": (int(cnt[0].item()) / B) if model.sync_comm_rate else cnt[0].double() / B})
    return out


def forward_train(model, data_dict):
    """One train-mode forward.  torch.autocast around the call (tools/train.py:118) or ``model.amp = True`` selects AMP for THIS
    step's convolutions (train_ops.amp_scope); the row GEMMs and the attention stay fp32."""
    from .airv2x_where2com import _amp_requested
    with T.amp_scope(_amp_requested(model)):
        return _forward_train(model, data_dict)
