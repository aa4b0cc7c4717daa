"""ORACLE (test infrastructure, not product code): AirV2X CoBEVT-LiDAR forward.

CPU fp32 restatement of models/airv2x_cobevt.py:112-156 and the fused-axial-attention fusion
(models/cobevt_modules/swap_fusion_modules.py: Attention :14-127, SwapFusionBlockMask :130-195,
SwapFusionEncoder :233-280; base_transformer.py: PreNormResidual :6-13, FeedForward :26-38;
fuse_utils.py regroup :13-64) as plain functions over a state_dict, without einops.
Parity: PINNED by tests/golden/cobevt_*.npz (tools/gen_golden.py runs the real reference).
The per-agent trunk (encoders, backbone, shrink) is shared with oracle/where2comm_oracle.py.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import where2comm_oracle as w2c

LN_EPS = 1e-5  # nn.LayerNorm default (base_transformer.py:9, swap_fusion_modules.py:272)


def regroup(x, record_len, max_len):
    """fuse_utils.py:13-64: split by record_len, zero-pad the agent axis to max_len.
    x (sumN,C,H,W) -> (B,L,C,H,W), mask (B,L) int64."""
    feats, masks = [], []
    for s in w2c._split(x, record_len):
        n = s.shape[0]
        pad = torch.zeros(max_len - n, *s.shape[1:], dtype=s.dtype)
        feats.append(torch.cat([s, pad], 0).unsqueeze(0))
        masks.append([1] * n + [0] * (max_len - n))
    return torch.cat(feats, 0), torch.tensor(masks, dtype=torch.int64)


def relative_position_index(L, ws):
    """swap_fusion_modules.py:53-76 for window [L, ws, ws]; tokens ordered (l, w1, w2)."""
    coords = torch.stack(torch.meshgrid(torch.arange(L), torch.arange(ws), torch.arange(ws), indexing="ij"))
    cf = torch.flatten(coords, 1)
    rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += L - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 2] += ws - 1
    rel[:, :, 0] *= (2 * ws - 1) * (2 * ws - 1)
    rel[:, :, 1] *= 2 * ws - 1
    return rel.sum(-1)


def _partition(x, ws, grid):
    """x (B,L,C,H,W) -> (B*X*Y, L*ws*ws, C) tokens ordered (l, w1, w2).
    window: 'b m d (x w1) (y w2) -> b m x y w1 w2 d' (:167-172); grid: 'b m d (w1 x) (w2 y) -> ...' (:185-190)."""
    B, L, C, H, W = x.shape
    X, Y = H // ws, W // ws
    if not grid:
        t = x.view(B, L, C, X, ws, Y, ws).permute(0, 3, 5, 1, 4, 6, 2)   # b x y l w1 w2 c
    else:
        t = x.view(B, L, C, ws, X, ws, Y).permute(0, 4, 6, 1, 3, 5, 2)   # b x y l w1 w2 c
    return t.reshape(B * X * Y, L * ws * ws, C)


def _unpartition(t, B, L, C, H, W, ws, grid):
    X, Y = H // ws, W // ws
    t = t.view(B, X, Y, L, ws, ws, C)
    if not grid:
        return t.permute(0, 3, 6, 1, 4, 2, 5).reshape(B, L, C, H, W)      # b l c (x w1) (y w2)
    return t.permute(0, 3, 6, 4, 1, 5, 2).reshape(B, L, C, H, W)          # b l c (w1 x) (w2 y)


def attention(tok, key_mask, sd, p, heads, L, ws):
    """Attention.forward :78-127 on partitioned tokens.  tok (Nw, T, C); key_mask (Nw, T) 1 = valid key."""
    Nw, T, C = tok.shape
    d = C // heads
    qkv = F.linear(tok, sd[p + ".to_qkv.weight"])
    q, k, v = qkv.chunk(3, dim=-1)
    sh = lambda t: t.view(Nw, T, heads, d).permute(0, 2, 1, 3)
    q, k, v = sh(q) * (d ** -0.5), sh(k), sh(v)
    sim = torch.matmul(q, k.transpose(-1, -2))
    idx = sd[p + ".relative_position_index"]
    bias = sd[p + ".relative_position_bias_table.weight"][idx]            # (T, T, heads)
    sim = sim + bias.permute(2, 0, 1)
    sim = sim.masked_fill(key_mask[:, None, None, :] == 0, -float("inf"))
    out = torch.matmul(F.softmax(sim, dim=-1), v)
    out = out.permute(0, 2, 1, 3).reshape(Nw, T, C)
    return F.linear(out, sd[p + ".to_out.0.weight"])


def _ln(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], LN_EPS)


def swap_block(x, mask, sd, p, heads, ws):
    """SwapFusionBlockMask.forward :154-195.  x (B,L,C,H,W), mask (B,L)."""
    B, L, C, H, W = x.shape
    for grid, name in ((False, "window"), (True, "grid")):
        tok = _partition(x, ws, grid)
        km = mask.view(B, 1, L, 1).expand(B, (H // ws) * (W // ws), L, ws * ws).reshape(-1, L * ws * ws)
        a = f"{p}.{name}_attention"
        tok = attention(_ln(tok, sd, a + ".norm"), km, sd, a + ".fn", heads, L, ws) + tok
        f = f"{p}.{name}_ffd"
        hdn = F.gelu(F.linear(_ln(tok, sd, f + ".norm"), sd[f + ".fn.net.0.weight"], sd[f + ".fn.net.0.bias"]))
        tok = F.linear(hdn, sd[f + ".fn.net.3.weight"], sd[f + ".fn.net.3.bias"]) + tok
        x = _unpartition(tok, B, L, C, H, W, ws, grid)
    return x


def swap_fusion_encoder(x, mask, sd, fax, trace=None):
    """SwapFusionEncoder.forward :277-280 (mask=True variant) -> (B,C,H,W)."""
    heads = fax["input_dim"] // fax["dim_head"]
    for i in range(fax["depth"]):
        x = swap_block(x, mask, sd, f"fusion_net.layers.{i}", heads, fax["window_size"])
        if trace is not None:
            trace[f"fax_block{i}"] = x
    m = x.mean(dim=1).permute(0, 2, 3, 1)                                  # b h w d
    m = F.layer_norm(m, (m.shape[-1],), sd["fusion_net.mlp_head.2.weight"], sd["fusion_net.mlp_head.2.bias"], LN_EPS)
    m = F.linear(m, sd["fusion_net.mlp_head.3.weight"], sd["fusion_net.mlp_head.3.bias"])
    return m.permute(0, 3, 1, 2)


def cobevt_forward(data_dict, sd, args, trace=None):
    """models/airv2x_cobevt.py:112-156 (det task; NaiveCompressor :121-123 when args["compression"] > 0)."""
    feats, record_len = w2c.extract_features(data_dict, sd, args)
    sf2d, _ = w2c.backbone_forward(feats, sd, args["base_bev_backbone"])
    s = w2c.shrink_conv(sf2d, sd, args["shrink_header"]) if args["shrink_header"]["use"] else sf2d
    if args.get("compression", 0) > 0:
        s = w2c.naive_compress(s, sd)
    L = sum(args["max_cav"].values())
    x, mask = regroup(s, record_len, L)
    fused = swap_fusion_encoder(x, mask, sd, args["fax_fusion"], trace)
    out = {"psm": w2c.head(fused, sd, "cls_head"), "rm": w2c.head(fused, sd, "reg_head")}
    if args["obj_head"]:
        out["obj"] = w2c.head(fused, sd, "obj_head")
    if trace is not None:
        trace.update({"shrink": s, "regroup": x, "mask": mask, "fused": fused})
    return out
