"""Differentiable building blocks of the transformer-style fusion heads: forward AND backward of every node run in
libairv2x_hip.so (csrc/train_fusion.hip + the convolution kernels); torch owns the tensors and the autograd bookkeeping.

The reference trains these modules through torch autograd (tools/train.py:220-247):

    nn.LayerNorm                                         base_transformer.py:9, swap_fusion_modules.py:272
    nn.Linear (+ residual of PreNormResidual :12)        1x1 convolutions over the NHWC token buffer
    nn.GELU                                              base_transformer.py:30
    Attention (windows / grids over agents x 4 x 4)      swap_fusion_modules.py:78-127
    Reduce("b m d h w -> b d h w", "mean")               swap_fusion_modules.py:270
    nn.Dropout                                           mask drawn with torch's generator (as the reference draws it), applied on the device

Token tensors are (L, H, W, C) fp32 NHWC, exactly the eval engine's layout (cobevt_engine.py).
"""
from __future__ import annotations

import torch

from .. import _lib
from . import train_ops as T
from .autograd import _P, _runner

LN_EPS = 1e-5


def _chan_sum(r, mat, rows, c):
    """Fixed-order column sums of a (rows, c) fp32 matrix."""
    ws = torch.empty(int(r.lib.av2x_channel_sum_workspace_bytes(rows, c)) // 4 + 4, device=mat.device)
    out = torch.empty(c, device=mat.device)
    _lib.check(r.lib.av2x_channel_sum(_P(mat), rows, c, _P(ws), _P(out), r.stream()), "av2x_channel_sum")
    return out


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta):
        T._check_dev(x)
        r = _runner(x.device)
        x = x.contiguous()
        c = x.shape[-1]
        y = torch.empty_like(x)
        _lib.check(r.lib.av2x_layernorm(_P(x), _P(gamma.detach()), _P(beta.detach()), _P(y), x.numel() // c, c, LN_EPS, r.stream()), "av2x_layernorm")
        ctx.save_for_backward(x, gamma)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        r = _runner(x.device)
        dy = dy.contiguous()
        c = x.shape[-1]
        n = x.numel() // c
        rows = int(r.lib.av2x_layernorm_backward_rows(n))
        partial = torch.empty((2, rows, c), device=x.device)
        dx = torch.empty_like(x)
        _lib.check(r.lib.av2x_layernorm_backward(_P(x), _P(gamma.detach()), _P(dy), n, c, LN_EPS, _P(dx), _P(partial), r.stream()),
                   "av2x_layernorm_backward")
        dg = _chan_sum(r, partial[0], rows, c) if ctx.needs_input_grad[1] else None
        db = _chan_sum(r, partial[1], rows, c) if ctx.needs_input_grad[2] else None
        return dx, dg, db


def layer_norm(x, gamma, beta):
    return LayerNormFn.apply(x, gamma, beta)


class LinearFn(torch.autograd.Function):
    """y = x W^T (+ b) (+ residual) over the last axis of an (n, h, w, cin) token tensor; W is nn.Linear's (cout, cin)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual):
        from .engine import ConvLayer
        T._check_dev(x)
        r = _runner(x.device)
        x = x.contiguous()
        n, h, w, cin = x.shape
        cout = weight.shape[0]
        w4 = weight.detach().view(cout, cin, 1, 1)
        wp, coutp, _, ent = T._packed(r, w4, owner=weight)    # keyed on the nn.Linear parameter: one packing per optimiser step
        L = ConvLayer(wp, None, bias.detach() if bias is not None else T._zeros(cout, x.device), cin, cout, coutp, 1, 1, 0, 0)
        T.seed_x3p(r, L, ent)
        y = torch.empty((n, h, w, cout), dtype=torch.float32, device=x.device)
        r.amp = T.AMP_STEP[0]
        try:
            r.conv(L, x, n, h, w, y, residual=residual.contiguous() if residual is not None else None)
        finally:
            r.amp = False
        ctx.save_for_backward(x, weight)
        ctx.has = (bias is not None, residual is not None)
        ctx.amp = T.AMP_STEP[0]
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        r = _runner(x.device)
        dy = dy.contiguous()
        cout, cin = weight.shape
        rows = dy.numel() // cout
        db = _chan_sum(r, dy, rows, cout) if (ctx.has[0] and ctx.needs_input_grad[2]) else None
        w4 = weight.detach().view(cout, cin, 1, 1)
        with T.amp_scope(ctx.amp):
            dw, dx = T._wgrad_and_dgrad(x, dy, w4, 1, 0, ctx.needs_input_grad[1], ctx.needs_input_grad[0], owner=weight)
        if dw is not None:
            dw = dw.view(cout, cin)
        return dx, dw, db, (dy if (ctx.has[1] and ctx.needs_input_grad[3]) else None)


def linear(x, weight, bias=None, residual=None):
    return LinearFn.apply(x, weight, bias, residual)


class GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z):
        T._check_dev(z)
        r = _runner(z.device)
        z = z.contiguous()
        y = torch.empty_like(z)
        _lib.check(r.lib.av2x_gelu(_P(z), None, _P(y), z.numel(), r.stream()), "av2x_gelu")
        ctx.save_for_backward(z)
        return y

    @staticmethod
    def backward(ctx, dy):
        (z,) = ctx.saved_tensors
        r = _runner(z.device)
        dz = torch.empty_like(z)
        _lib.check(r.lib.av2x_gelu(_P(z), _P(dy.contiguous()), _P(dz), z.numel(), r.stream()), "av2x_gelu")
        return dz


def gelu(z):
    return GeluFn.apply(z)


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, residual):
        T._check_dev(x)
        r = _runner(x.device)
        x = x.contiguous()
        if residual is not None:
            if residual.shape != x.shape:
                raise ValueError("dropout: the residual must have the input's shape")
            residual = residual.contiguous()
        # the Bernoulli draw happens IN the kernel (Philox keyed by a seed from torch's generator: torch.manual_seed governs it, as it does
        # the reference's nn.Dropout; the streams differ, as they do between any two dropout implementations).  The backward regenerates
        # the keep-mask from the seed: no mask tensor, no torch.rand / compare / cast launches
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        y = torch.empty_like(x)
        if x.numel() % 4:
            raise ValueError("dropout: the element count must be a multiple of 4")
        _lib.check(r.lib.av2x_dropout_seeded(_P(x), _P(residual), _P(y), x.numel(), p, seed, r.stream()), "av2x_dropout_seeded")
        ctx.seed, ctx.p, ctx.has_res = seed, p, residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        r = _runner(dy.device)
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        _lib.check(r.lib.av2x_dropout_seeded(_P(dy), None, _P(dx), dy.numel(), ctx.p, ctx.seed, r.stream()), "av2x_dropout_seeded")
        return dx, None, (dy if (ctx.has_res and ctx.needs_input_grad[2]) else None)


def dropout(x, p, training=True, residual=None):
    """nn.Dropout(p)(x) (+ residual: the skip connection added right after it, in the same launch)."""
    if not training or p <= 0.0:
        return x if residual is None else x + residual
    if p >= 1.0:
        raise ValueError("dropout probability must be < 1")
    return DropoutFn.apply(x, float(p), residual)


class FaxAttentionFn(torch.autograd.Function):
    """Attention.forward (swap_fusion_modules.py:78-127) on the to_qkv output of ALL windows of one sample: qkv (L, H, W, 3C),
    relative-position bias table ((2L-1)(2ws-1)^2, heads) -> (L, H, W, C) (heads merged, before to_out)."""

    @staticmethod
    def forward(ctx, qkv, table, n_valid, ws, heads, dim_head, grid):
        T._check_dev(qkv)
        r = _runner(qkv.device)
        qkv = qkv.contiguous()
        L, H, W, C3 = qkv.shape
        out = torch.empty((L, H, W, C3 // 3), dtype=torch.float32, device=qkv.device)
        _lib.check(r.lib.av2x_fax_attention(_P(qkv), _P(table.detach().contiguous()), _P(out), L, n_valid, H, W, ws, heads, dim_head, grid, r.stream()),
                   "av2x_fax_attention")
        ctx.save_for_backward(qkv, table, out)
        ctx.cfg = (n_valid, ws, heads, dim_head, grid)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, table, out = ctx.saved_tensors
        n_valid, ws, heads, dim_head, grid = ctx.cfg
        r = _runner(qkv.device)
        L, H, W, _ = qkv.shape
        dqkv = torch.empty_like(qkv)
        dtab = torch.empty_like(table)
        wsb = torch.empty(int(r.lib.av2x_fax_attention_backward_workspace_bytes(L, ws, heads)), dtype=torch.uint8, device=qkv.device)
        _lib.check(r.lib.av2x_fax_attention_backward(_P(qkv), _P(table.detach().contiguous()), _P(out), _P(dout.contiguous()), L, n_valid, H, W, ws,
                                                     heads, dim_head, grid, _P(dqkv), _P(dtab), _P(wsb), r.stream()), "av2x_fax_attention_backward")
        return dqkv, dtab, None, None, None, None, None


def fax_attention(qkv, table, n_valid, ws, heads, dim_head, grid):
    return FaxAttentionFn.apply(qkv, table, n_valid, ws, heads, dim_head, grid)


class AgentMeanFn(torch.autograd.Function):
    """(L, H, W, C) -> (1, H, W, C): the mean over the agent axis (swap_fusion_modules.py:270), padded agents included."""

    @staticmethod
    def forward(ctx, x):
        T._check_dev(x)
        r = _runner(x.device)
        x = x.contiguous()
        y = torch.empty((1,) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
        _lib.check(r.lib.av2x_agent_mean(_P(x), _P(y), x.shape[0], y.numel(), r.stream()), "av2x_agent_mean")
        ctx.L = x.shape[0]
        return y

    @staticmethod
    def backward(ctx, dy):
        r = _runner(dy.device)
        dy = dy.contiguous()
        dx = torch.empty((ctx.L,) + tuple(dy.shape[1:]), dtype=torch.float32, device=dy.device)
        _lib.check(r.lib.av2x_scale_broadcast(_P(dy), _P(dx), ctx.L, dy.numel(), 1.0 / ctx.L, r.stream()), "av2x_scale_broadcast")
        return dx


def agent_mean(x):
    return AgentMeanFn.apply(x)


# ======================================================================================================= V2X-ViT nodes
def _types_arr(types):
    import ctypes
    return (ctypes.c_int32 * len(types))(*[int(t) for t in types])


class HgtAttentionFn(torch.autograd.Function):
    """HGTCavAttention's attention core (hmsa.py:133-151) on the folded projections: proj (n, H, W, 1280) =
    [q'(->type 0) | q'(->type 1) | k | v'(type 0 <-) | v'(type 1 <-)], mask (n, H, W) = key agent visible at the pixel ->
    (n, H, W, 256) heads merged (before a_linears)."""

    @staticmethod
    def forward(ctx, proj, mask, types, heads, dim_head):
        import ctypes
        T._check_dev(proj)
        r = _runner(proj.device)
        proj = proj.contiguous()
        n, H, W, _ = proj.shape
        out = torch.empty((n, H, W, heads * dim_head), dtype=torch.float32, device=proj.device)
        ta = _types_arr(types)
        _lib.check(r.lib.av2x_hgt_attention(_P(proj), _P(mask), ctypes.cast(ta, ctypes.c_void_p), _P(out), n, H * W, heads, dim_head, r.stream()),
                   "av2x_hgt_attention")
        ctx.save_for_backward(proj, mask)
        ctx.cfg = (list(types), heads, dim_head)
        return out

    @staticmethod
    def backward(ctx, dout):
        import ctypes
        proj, mask = ctx.saved_tensors
        types, heads, dim_head = ctx.cfg
        r = _runner(proj.device)
        n, H, W, _ = proj.shape
        dproj = torch.empty_like(proj)
        ta = _types_arr(types)
        _lib.check(r.lib.av2x_hgt_attention_backward(_P(proj), _P(mask), ctypes.cast(ta, ctypes.c_void_p), _P(dout.contiguous()), _P(dproj), n, H * W,
                                                     heads, dim_head, r.stream()), "av2x_hgt_attention_backward")
        return dproj, None, None, None, None


def hgt_attention(proj, mask, types, heads, dim_head):
    return HgtAttentionFn.apply(proj, mask, types, heads, dim_head)


class PyramidWindowFn(torch.autograd.Function):
    """The three BaseWindowAttention cores (mswin.py:52-96) of a PyramidWindowAttention on ONE [q|k|v] x 3 buffer: qkv3 (n, H, W, sum 3 h_i d_i),
    pos_i (2 w_i - 1, 2 w_i - 1) -> three (n, H, W, h_i d_i) maps (before to_out)."""

    @staticmethod
    def forward(ctx, qkv3, pos0, pos1, pos2, cfg):
        T._check_dev(qkv3)
        r = _runner(qkv3.device)
        qkv3 = qkv3.contiguous()
        n, H, W, ctot = qkv3.shape
        outs, coff = [], 0
        for (h, dh, ws), pos in zip(cfg, (pos0, pos1, pos2)):
            o = torch.empty((n, H, W, h * dh), dtype=torch.float32, device=qkv3.device)
            _lib.check(r.lib.av2x_window_attention(_P(qkv3), ctot, coff, _P(pos.detach().contiguous()), _P(o), n, H, W, h, dh, ws, r.stream()),
                       "av2x_window_attention")
            outs.append(o)
            coff += 3 * h * dh
        ctx.save_for_backward(qkv3, pos0, pos1, pos2, *outs)
        ctx.cfg = cfg
        return tuple(outs)

    @staticmethod
    def backward(ctx, d0, d1, d2):
        qkv3, pos0, pos1, pos2, o0, o1, o2 = ctx.saved_tensors
        r = _runner(qkv3.device)
        n, H, W, ctot = qkv3.shape
        dqkv = torch.empty_like(qkv3)
        dpos, coff = [], 0
        for (h, dh, ws), pos, o, d in zip(ctx.cfg, (pos0, pos1, pos2), (o0, o1, o2), (d0, d1, d2)):
            dp = torch.empty_like(pos)
            wsb = torch.empty(int(r.lib.av2x_window_attention_backward_workspace_bytes(n, H, W, h, ws)), dtype=torch.uint8, device=qkv3.device)
            _lib.check(r.lib.av2x_window_attention_backward(_P(qkv3), ctot, coff, _P(pos.detach().contiguous()), _P(o), _P(d.contiguous()), _P(dqkv), _P(dp),
                                                            _P(wsb), n, H, W, h, dh, ws, r.stream()), "av2x_window_attention_backward")
            dpos.append(dp)
            coff += 3 * h * dh
        return dqkv, dpos[0], dpos[1], dpos[2], None


def pyramid_window_attention(qkv3, pos, cfg):
    return PyramidWindowFn.apply(qkv3, pos[0], pos[1], pos[2], tuple(tuple(int(v) for v in c) for c in cfg))


def _split_logits(gap, fc1_w, bn_w, bn_b, fc2_w):
    """SplitAttn's squeeze path on the (n, C) mean (split_attn.py:51-53): fc1 -> LayerNorm ("bn1") -> ReLU -> fc2.  A handful of
    C-vectors per agent: plain tensor algebra (differentiable for the backward below)."""
    import torch.nn.functional as TF
    # fp32 whatever autocast region the step runs in: the kernels on either side read / write fp32 (under torch.autocast F.linear would
    # return bf16 -- half the bytes av2x_split_attn_combine then reads: a memory fault at the BASELINE grid, garbage weights below it)
    with torch.autocast("cuda", enabled=False):
        f32 = lambda t: t.float()
        g = TF.relu(TF.layer_norm(TF.linear(f32(gap), f32(fc1_w)), (gap.shape[-1],), f32(bn_w), f32(bn_b), LN_EPS))
        return TF.linear(g, f32(fc2_w))


class SplitAttnFn(torch.autograd.Function):
    """SplitAttn.forward (split_attn.py:40-63) + the residual of the enclosing PreNorm block: three branch maps (n, H, W, C) -> out."""

    @staticmethod
    def forward(ctx, s0, s1, s2, res, fc1_w, bn_w, bn_b, fc2_w):
        T._check_dev(s0)
        r = _runner(s0.device)
        s0, s1, s2, res = s0.contiguous(), s1.contiguous(), s2.contiguous(), res.contiguous()
        n, H, W, C = s0.shape
        gap = torch.empty((n, C), dtype=torch.float32, device=s0.device)
        scratch = torch.empty((n, 128, C), dtype=torch.float32, device=s0.device)
        _lib.check(r.lib.av2x_split_attn_gap(_P(s0), _P(s1), _P(s2), _P(gap), _P(scratch), n, H * W, C, r.stream()), "av2x_split_attn_gap")
        with torch.no_grad():
            logits = _split_logits(gap, fc1_w, bn_w, bn_b, fc2_w).contiguous()
        out = torch.empty_like(s0)
        _lib.check(r.lib.av2x_split_attn_combine(_P(s0), _P(s1), _P(s2), _P(logits), _P(res), _P(out), n, H * W, C, r.stream()), "av2x_split_attn_combine")
        ctx.save_for_backward(s0, s1, s2, gap, fc1_w, bn_w, bn_b, fc2_w)
        return out

    @staticmethod
    def backward(ctx, dout):
        s0, s1, s2, gap, fc1_w, bn_w, bn_b, fc2_w = ctx.saved_tensors
        r = _runner(s0.device)
        dout = dout.contiguous()
        n, H, W, C = s0.shape
        da = torch.empty((n, 3, C), dtype=torch.float32, device=s0.device)
        ws = torch.empty(int(r.lib.av2x_split_attn_backward_workspace_bytes(n, C)) // 4, dtype=torch.float32, device=s0.device)
        _lib.check(r.lib.av2x_split_attn_sums(_P(s0), _P(s1), _P(s2), _P(dout), _P(da), _P(ws), n, H * W, C, r.stream()), "av2x_split_attn_sums")
        with torch.enable_grad():     # the squeeze path again, on (n, C) vectors, for its vector-Jacobian product
            g_ = gap.detach().requires_grad_(True)
            ps = [p.detach().requires_grad_(True) for p in (fc1_w, bn_w, bn_b, fc2_w)]
            a = torch.softmax(_split_logits(g_, *ps).view(n, 3, C), dim=1)        # RadixSoftmax(3, 1)
            grads = torch.autograd.grad((a * da).sum(), [g_] + ps)
        wts = a.detach().contiguous()
        dgap = grads[0].contiguous()
        d0, d1, d2 = torch.empty_like(s0), torch.empty_like(s0), torch.empty_like(s0)
        _lib.check(r.lib.av2x_split_attn_backward(_P(dout), _P(wts), _P(dgap), _P(d0), _P(d1), _P(d2), n, H * W, C, r.stream()), "av2x_split_attn_backward")
        return d0, d1, d2, dout, grads[1], grads[2], grads[3], grads[4]


def split_attn(s0, s1, s2, res, fc1_w, bn_w, bn_b, fc2_w):
    return SplitAttnFn.apply(s0, s1, s2, res, fc1_w, bn_w, bn_b, fc2_w)


class WarpAffineFn(torch.autograd.Function):
    """warp_affine (torch_transformation_utils.py:337-381: affine_grid + grid_sample, bilinear, zeros, align_corners) with a constant theta."""

    @staticmethod
    def forward(ctx, x, theta):
        T._check_dev(x)
        r = _runner(x.device)
        x = x.contiguous()
        n, H, W, C = x.shape
        y = torch.empty_like(x)
        _lib.check(r.lib.av2x_warp_affine(_P(x), _P(theta), _P(y), n, H, W, C, r.stream()), "av2x_warp_affine")
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
        _lib.check(r.lib.av2x_warp_affine_backward(_P(dy), _P(theta), _P(dx), _P(ws), n, H, W, C, r.stream()), "av2x_warp_affine_backward")
        return dx, None


def warp_affine(x, theta):
    return WarpAffineFn.apply(x, theta)


class AddAgentVectorFn(torch.autograd.Function):
    """x (n, H, W, C) + v (n, C) broadcast over the map (RTE, v2xvit_basic.py:58-80)."""

    @staticmethod
    def forward(ctx, x, v):
        T._check_dev(x)
        r = _runner(x.device)
        y = x.contiguous().clone()
        n, H, W, C = y.shape
        _lib.check(r.lib.av2x_add_agent_vector(_P(y), _P(v.detach().contiguous()), n, H * W * C, C, r.stream()), "av2x_add_agent_vector")
        return y

    @staticmethod
    def backward(ctx, dy):
        r = _runner(dy.device)
        dy = dy.contiguous()
        n, H, W, C = dy.shape
        dv = torch.stack([_chan_sum(r, dy[a], H * W, C) for a in range(n)]) if ctx.needs_input_grad[1] else None
        return dy, dv


def add_agent_vector(x, v):
    return AddAgentVectorFn.apply(x, v)
