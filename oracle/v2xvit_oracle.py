"""ORACLE (test infrastructure, not product code): AirV2X V2X-ViT-LiDAR forward.

CPU fp32 restatement of models/airv2x_v2xvit.py:108-167 and the V2X-ViT fusion
(models/v2xvit_modules/v2xvit_basic.py: STTF :17-38, RTE :41-80, V2XFusionBlock :83-132,
V2XTEncoder :135-200; hmsa.py HGTCavAttention :6-158; mswin.py BaseWindowAttention :21-99,
PyramidWindowAttention :102-145; split_attn.py :6-63; base_transformer.py PreNorm/FeedForward;
common_modules/torch_transformation_utils.py warp chain :15-381) as plain functions over a
state_dict, without einops.  Parity: PINNED by tests/golden/v2xvit_*.npz (tools/gen_golden.py).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import where2comm_oracle as w2c
from .cobevt_oracle import regroup

LN_EPS = 1e-5


# ------------------------------------------------------------------ a16: warp chain
def discretized_matrix(m, discrete_ratio, downsample_rate):
    """torch_transformation_utils.py:116-143: rows/cols (0,1 | 0,1,3) of the 4x4, translation in pixels, fp32."""
    m = m[:, :, [0, 1], :][:, :, :, [0, 1, 3]].clone()
    m[:, :, :, -1] = m[:, :, :, -1] / (discrete_ratio * downsample_rate)
    return m.float()


def _eye(B, dtype):
    return torch.eye(3, dtype=dtype)[None].repeat(B, 1, 1)


def transformation_matrix(M, dsize):
    """get_transformation_matrix / get_rotation_matrix2d :265-308: rotate about the image centre, then translate."""
    H, W = dsize
    B = M.shape[0]
    center = torch.tensor([W / 2, H / 2], dtype=M.dtype).unsqueeze(0)
    sh, shi, rot = _eye(B, M.dtype), _eye(B, M.dtype), _eye(B, M.dtype)
    sh[:, :2, 2] = center
    shi[:, :2, 2] = -center
    rot[:, :2, :2] = M[:, :2, :2]
    T = (sh @ rot @ shi)[:, :2, :].clone()
    T[..., 2] += M[..., 2]
    return T


def _norm_pix(h, w, dtype):
    t = torch.tensor([[1.0, 0.0, -1.0], [0.0, 1.0, -1.0], [0.0, 0.0, 1.0]], dtype=dtype)
    t[0, 0] = t[0, 0] * 2.0 / (1e-14 if w == 1 else w - 1.0)
    t[1, 1] = t[1, 1] * 2.0 / (1e-14 if h == 1 else h - 1.0)
    return t.unsqueeze(0)


def affine_theta(M, src_hw, dsize):
    """The (B,2,3) theta handed to F.affine_grid by warp_affine (:337-381): homography, normalise, invert."""
    H3 = F.pad(M, [0, 0, 0, 1], "constant", value=0.0)
    H3[..., -1, -1] += 1.0
    sn = _norm_pix(src_hw[0], src_hw[1], M.dtype)
    dn = _norm_pix(dsize[0], dsize[1], M.dtype)
    dst_norm_trans_src_norm = dn @ (H3 @ torch.inverse(sn))
    return torch.inverse(dst_norm_trans_src_norm)[:, :2, :]


def warp_affine(src, M, dsize, mode="bilinear"):
    theta = affine_theta(M, src.shape[-2:], dsize)
    grid = F.affine_grid(theta, [src.shape[0], src.shape[1], dsize[0], dsize[1]], align_corners=True)
    return F.grid_sample(src, grid.to(src.dtype), align_corners=True, mode=mode, padding_mode="zeros")   # the cast is a no-op in fp32


def roi_and_cav_mask(shape, cav_mask, scm, discrete_ratio, downsample_rate):
    """get_roi_and_cav_mask :15-53 -> (B,H,W,1,L)."""
    B, L, H, W, _ = shape
    T = transformation_matrix(discretized_matrix(scm, discrete_ratio, downsample_rate).reshape(-1, 2, 3), (H, W))
    roi = warp_affine(torch.ones(B * L, 1, H, W, dtype=T.dtype), T, (H, W), mode="nearest").reshape(B, L, 1, H, W)
    com = roi * cav_mask.view(B, L, 1, 1, 1).to(roi.dtype)
    return com.permute(0, 3, 4, 2, 1)


# ------------------------------------------------------------------ encoder pieces
def rte_table(n_hid, max_len=100):
    """RelTemporalEncoding.__init__ v2xvit_basic.py:46-56 (the table is ALSO in the state_dict: rte.emb.emb.weight)."""
    position = torch.arange(0.0, max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, n_hid, 2) * -(math.log(10000.0) / n_hid))
    emb = torch.zeros(max_len, n_hid)
    emb[:, 0::2] = torch.sin(position * div_term) / math.sqrt(n_hid)
    emb[:, 1::2] = torch.cos(position * div_term) / math.sqrt(n_hid)
    return emb


def rte(x, dts, sd, p, ratio):
    """RTE.forward :71-80: x[b,i] += lin(emb[dt * ratio])."""
    out = x.clone()
    for b in range(x.shape[0]):
        for i in range(x.shape[1]):
            e = sd[p + ".emb.emb.weight"][int(dts[b, i]) * ratio]
            out[b, i] = x[b, i] + F.linear(e, sd[p + ".emb.lin.weight"], sd[p + ".emb.lin.bias"])
    return out


def sttf(x, scm, discrete_ratio, downsample_rate):
    """STTF.forward :23-38: warp agents 1..L-1 into the ego frame."""
    x = x.permute(0, 1, 4, 2, 3)
    B, L, C, H, W = x.shape
    d = discretized_matrix(scm, discrete_ratio, downsample_rate)
    T = transformation_matrix(d[:, 1:, :, :].reshape(-1, 2, 3), (H, W))
    cav = warp_affine(x[:, 1:].reshape(-1, C, H, W), T, (H, W)).reshape(B, -1, C, H, W)
    return torch.cat([x[:, 0].unsqueeze(1), cav], dim=1).permute(0, 1, 3, 4, 2)


def _ln(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], LN_EPS)


def hgt_attention(x, mask, types, sd, p, heads, dim_head):
    """HGTCavAttention.forward hmsa.py:115-158.  x (B,L,H,W,C), mask (B,H,W,1,L), types (B,L) int."""
    B, L, H, W, C = x.shape
    scale = dim_head ** -0.5
    xp = x.permute(0, 2, 3, 1, 4)                                            # b h w l c
    lin = lambda name, b, i: F.linear(xp[b, :, :, i, :], sd[f"{p}.{name}_linears.{int(types[b, i])}.weight"],
                                      sd[f"{p}.{name}_linears.{int(types[b, i])}.bias"])
    q = torch.stack([torch.stack([lin("q", b, i) for i in range(L)], 2) for b in range(B)])   # b h w l c
    k = torch.stack([torch.stack([lin("k", b, i) for i in range(L)], 2) for b in range(B)])
    v = torch.stack([torch.stack([lin("v", b, i) for i in range(L)], 2) for b in range(B)])
    sp = lambda t: t.view(B, H, W, L, heads, dim_head).permute(0, 4, 1, 2, 3, 5)                 # b m h w l c
    q, k, v = sp(q), sp(k), sp(v)
    e = types.long()[:, :, None] * 2 + types.long()[:, None, :]                                   # (B,L,L) relation index
    w_att = sd[p + ".relation_att"][e].permute(0, 3, 1, 2, 4, 5)                                  # b m i j p q
    w_msg = sd[p + ".relation_msg"][e].permute(0, 3, 1, 2, 4, 5)
    att = torch.einsum("bmhwip,bmijpq,bmhwjq->bmhwij", q, w_att, k) * scale
    att = att.masked_fill(mask.unsqueeze(1) == 0, -float("inf"))
    att = att.softmax(dim=-1)
    v_msg = torch.einsum("bmijpc,bmhwjp->bmhwijc", w_msg, v)
    out = torch.einsum("bmhwij,bmhwijc->bmhwic", att, v_msg)
    out = out.permute(0, 2, 3, 4, 1, 5).reshape(B, H, W, L, heads * dim_head)
    out = torch.stack([torch.stack([F.linear(out[b, :, :, i, :], sd[f"{p}.a_linears.{int(types[b, i])}.weight"],
                                             sd[f"{p}.a_linears.{int(types[b, i])}.bias"]) for i in range(L)], 2)
                       for b in range(B)])
    return out.permute(0, 3, 1, 2, 4)                                                             # b l h w c


def window_attention(x, sd, p, heads, dim_head, ws):
    """BaseWindowAttention.forward mswin.py:49-99 (relative_pos_embedding=True).  x (B,L,H,W,C)."""
    B, L, H, W, C = x.shape
    nh, nw = H // ws, W // ws
    qkv = F.linear(x, sd[p + ".to_qkv.weight"]).chunk(3, dim=-1)
    part = lambda t: t.view(B, L, nh, ws, nw, ws, heads, dim_head).permute(0, 1, 6, 2, 4, 3, 5, 7).reshape(
        B, L, heads, nh * nw, ws * ws, dim_head)
    q, k, v = part(qkv[0]), part(qkv[1]), part(qkv[2])
    dots = torch.einsum("blmhic,blmhjc->blmhij", q, k) * (dim_head ** -0.5)
    idx = torch.tensor([[a, b] for a in range(ws) for b in range(ws)])
    rel = idx[None, :, :] - idx[:, None, :] + ws - 1
    dots = dots + sd[p + ".pos_embedding"][rel[:, :, 0], rel[:, :, 1]]
    out = torch.einsum("blmhij,blmhjc->blmhic", dots.softmax(dim=-1), v)
    out = out.view(B, L, heads, nh, nw, ws, ws, dim_head).permute(0, 1, 3, 5, 4, 6, 2, 7).reshape(B, L, H, W, heads * dim_head)
    return F.linear(out, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])


def split_attn(windows, sd, p, gap_reduce=None):
    """SplitAttn.forward split_attn.py:40-63 (radix 3, cardinality 1).  ``gap_reduce`` (tests of the column-sharded
    fusion): maps the local mean over this strip of the map to the mean over the whole map."""
    sw, mw, bw = windows
    B, L, _, _, C = sw.shape
    gap = (sw + mw + bw).mean((2, 3), keepdim=True)
    if gap_reduce is not None:
        gap = gap_reduce(gap)
    g = F.relu(_ln(F.linear(gap, sd[p + ".fc1.weight"]), sd, p + ".bn1"))
    a = F.linear(g, sd[p + ".fc2.weight"])
    a = F.softmax(a.view(B, L, 1, 3, -1), dim=3).reshape(B, -1).view(B, L, 1, 1, -1)
    return sw * a[..., 0:C] + mw * a[..., C:2 * C] + bw * a[..., 2 * C:]


def pyramid_window_attention(x, sd, p, cfg, gap_reduce=None):
    outs = [window_attention(x, sd, f"{p}.pwmsa.{i}", h, d, w)
            for i, (h, d, w) in enumerate(zip(cfg["heads"], cfg["dim_head"], cfg["window_size"]))]
    if cfg["fusion_method"] == "split_attn":
        return split_attn(outs, sd, p + ".split_attn", gap_reduce)
    return sum(outs) / len(outs)


def encoder(x, mask, scm, sd, enc, trace=None, strip=None, gap_reduce=None):
    """V2XTEncoder.forward v2xvit_basic.py:174-200 + V2XTransformer :210-213.  x (B,L,H,W,C+3).
    ``strip`` = (first column, columns) + ``gap_reduce``: the blocks run on a column strip only (sharded-fusion tests)."""
    p = "fusion_net.encoder"
    cav, pw = enc["cav_att_config"], enc["pwindow_att_config"]
    prior = x[..., -3:]
    x = x[..., :-3]
    if cav["use_RTE"]:
        x = rte(x, prior[:, :, 0, 0, 1].to(torch.int), sd, p + ".rte", cav["RTE_ratio"])
    x = sttf(x, scm, enc["sttf"]["voxel_size"][0], enc["sttf"]["downsample_rate"])
    if trace is not None:
        trace["after_sttf"] = x
    B, L = x.shape[:2]
    com_mask = (roi_and_cav_mask(x.shape, mask, scm, enc["sttf"]["voxel_size"][0], enc["sttf"]["downsample_rate"])
                if enc["use_roi_mask"] else mask.view(B, 1, 1, 1, L))
    if trace is not None:
        trace["com_mask"] = com_mask
    types = prior[:, :, 0, 0, 2].to(torch.int)
    if strip is not None:
        c0, wc = strip
        x = x[:, :, :, c0:c0 + wc]
        if com_mask.dim() == 5 and com_mask.shape[2] > 1:
            com_mask = com_mask[:, :, c0:c0 + wc]
    for d in range(enc["depth"]):
        for nb in range(enc["num_blocks"]):
            q = f"{p}.layers.{d}.0.layers.{nb}"
            x = hgt_attention(_ln(x, sd, q + ".0.norm"), com_mask, types, sd, q + ".0.fn", cav["heads"], cav["dim_head"]) + x
            if trace is not None:
                trace[f"hgt{d}"] = x
            x = pyramid_window_attention(_ln(x, sd, q + ".1.norm"), sd, q + ".1.fn", pw, gap_reduce) + x
        f = f"{p}.layers.{d}.1"
        h = F.gelu(F.linear(_ln(x, sd, f + ".norm"), sd[f + ".fn.net.0.weight"], sd[f + ".fn.net.0.bias"]))
        x = F.linear(h, sd[f + ".fn.net.3.weight"], sd[f + ".fn.net.3.bias"]) + x
        if trace is not None:
            trace[f"layer{d}"] = x
    return x[:, 0]


def v2xvit_forward(data_dict, sd, args, trace=None):
    """models/airv2x_v2xvit.py:108-167 (det task)."""
    mf = args["modality_fusion"]
    feats, record_len = w2c.extract_features(data_dict, sd, args)
    comm_rate = int(feats.count_nonzero().item())
    sf2d, _ = w2c.backbone_forward(feats, sd, mf["base_bev_backbone"])
    s = w2c.shrink_conv(sf2d, sd, mf["shrink_header"]) if mf["shrink_header"]["use"] else sf2d
    if mf.get("compression", 0) > 0:      # NaiveCompressor(256, args["compression"]) (airv2x_v2xvit.py:42-44,122-123; airv2x_when2com.py:50-52,122-123)
        s = w2c.naive_compress(s, sd)
    L = args["max_cav_num"]
    x, mask = regroup(s, record_len, L)                                          # b l c h w
    prior = data_dict["prior_encoding"].unsqueeze(-1).unsqueeze(-1).repeat(1, 1, 1, x.shape[3], x.shape[4])
    x = torch.cat([x, prior], dim=2).permute(0, 1, 3, 4, 2).contiguous()
    fused = encoder(x, mask, data_dict["spatial_correction_matrix"], sd, args["transformer"]["encoder"], trace)
    fused = fused.permute(0, 3, 1, 2).contiguous()
    out = {"psm": w2c.head(fused, sd, "cls_head"), "rm": w2c.head(fused, sd, "reg_head"), "comm_rate": comm_rate}
    if args["obj_head"]:
        out["obj"] = w2c.head(fused, sd, "obj_head")
    if trace is not None:
        trace["fused"] = fused
    return out
