"""ORACLE (test infrastructure, not product code): AirV2X When2com-LiDAR forward.

CPU fp32 restatement of models/airv2x_when2com.py:112-151 and the fusion
(models/when2com_modules/when2com.py: When2comFusion.forward :60-134 in its shipped 'softmax' mode, policy_net4 :300-317,
km_generator :283-297, MIMOGeneralDotProductAttention :320-348, conv2DBatchNormRelu :137-168) as plain functions over a
state_dict.  warp_affine_simple = common_modules/torch_transformation_utils.py:327-334.
Parity: PINNED by tests/golden/when2com_*.npz (tools/gen_golden.py runs the real reference).
The per-agent trunk (encoders, backbone, shrink) is shared with oracle/where2comm_oracle.py.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import where2comm_oracle as w2c

BN_EPS = 1e-5   # nn.BatchNorm2d default (conv2DBatchNormRelu, when2com.py:160-163) -- NOT the backbone's 1e-3
BN_MOMENTUM = 0.1   # nn.BatchNorm2d default, likewise (the backbone's BatchNorms are built with 0.01)


def normalized_pairwise(pairwise_t_matrix, H, W, discrete_ratio, downsample_rate):
    """when2com.py:86-104: rows (0,1) x cols (0,1,3) of the 4x4, shear terms rescaled by the aspect ratio, translation
    in units of half the map.  (B,L,L,4,4) -> (B,L,L,2,3) fp32."""
    m = pairwise_t_matrix[:, :, :, [0, 1], :][:, :, :, :, [0, 1, 3]].clone().float()
    m[..., 0, 1] = m[..., 0, 1] * H / W
    m[..., 1, 0] = m[..., 1, 0] * W / H
    m[..., 0, 2] = m[..., 0, 2] / (downsample_rate * discrete_ratio * W) * 2
    m[..., 1, 2] = m[..., 1, 2] / (downsample_rate * discrete_ratio * H) * 2
    return m


def warp_affine_simple(src, M, dsize):
    """torch_transformation_utils.py:327-334 (bilinear, zeros, align_corners=False)."""
    B, C = src.shape[:2]
    grid = F.affine_grid(M, [B, C, dsize[0], dsize[1]], align_corners=False).to(src)
    return F.grid_sample(src, grid, align_corners=False)


def _cbr(x, sd, p, stride):
    """conv2DBatchNormRelu :137-170: Conv2d (with bias) + BatchNorm2d (default eps 1e-5, momentum 0.1) + ReLU; inside
    ``where2comm_oracle.train_mode()`` the BatchNorm uses batch statistics and updates its running statistics, as ``.train()`` does."""
    x = F.conv2d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], stride=stride, padding=1)
    if w2c._STATE["train"]:
        nbt = sd.get(p + ".1.num_batches_tracked")
        if nbt is not None:
            nbt += 1
        x = F.batch_norm(x, sd[p + ".1.running_mean"], sd[p + ".1.running_var"], sd[p + ".1.weight"], sd[p + ".1.bias"],
                         True, BN_MOMENTUM, BN_EPS)
    else:
        x = F.batch_norm(x, sd[p + ".1.running_mean"], sd[p + ".1.running_var"], sd[p + ".1.weight"], sd[p + ".1.bias"],
                         False, 0.0, BN_EPS)
    return F.relu(x)


def policy_net(x, sd, p):
    """policy_net4 :300-317: 256 -> 512 -> 256 -> (s2) 256 -> 256 -> (s2) 256."""
    for i, s in enumerate((1, 1, 2, 1, 2), 1):
        x = _cbr(x, sd, f"{p}.conv{i}.cbr_unit", s)
    return x


def km_generator(feat, sd, p):
    x = feat.reshape(feat.shape[0], -1)                                     # NCHW flatten (:296)
    x = F.relu(F.linear(x, sd[p + ".fc.0.weight"], sd[p + ".fc.0.bias"]))
    x = F.relu(F.linear(x, sd[p + ".fc.2.weight"], sd[p + ".fc.2.bias"]))
    return F.linear(x, sd[p + ".fc.4.weight"], sd[p + ".fc.4.bias"])


def when2com_fuse(x, record_len, pairwise_t_matrix, sd, cfg, prefix="fusion_net", trace=None):
    """When2comFusion.forward :60-134 -> ((B,C,H,W), comm rate as a python float)."""
    _, C, H, W = x.shape
    B = pairwise_t_matrix.shape[0]
    t = normalized_pairwise(pairwise_t_matrix, H, W, cfg["voxel_size"][0], cfg["downsample_rate"])
    outs, nz = [], []
    for b, xb in enumerate(w2c._split(x, record_len)):
        N = xb.shape[0]
        nz.append(int(xb.count_nonzero()))
        nb = warp_affine_simple(xb, t[b, 0, :N], (H, W))                   # every agent into the ego frame
        qk = policy_net(nb, sd, prefix + ".query_key_net")
        keys = km_generator(qk, sd, prefix + ".key_net")                    # (N, key_size)
        query = km_generator(qk[0:1], sd, prefix + ".query_net")            # (1, query_size): the ego asks
        q = F.linear(query, sd[prefix + ".attention_net.linear.weight"], sd[prefix + ".attention_net.linear.bias"])
        attn = torch.softmax(keys @ q.t(), dim=0)                           # (N, 1): softmax over the keys (:333-335)
        coef = attn
        if cfg["mode"] == "activated":
            # activated_select :45-58 indexes the (1, N, 1) coefficient tensor with a 2 x 2 diagonal and raises IndexError
            # in the reference for the single-query layout this forward builds: there is no behaviour to restate
            raise NotImplementedError("mode 'activated' cannot run in the reference (IndexError at when2com.py:58)")
        outs.append((coef.view(N, 1, 1, 1) * nb).sum(0, keepdim=True))
        if trace is not None:
            trace[f"warped{b}"], trace[f"keys{b}"], trace[f"query{b}"], trace[f"coef{b}"] = nb, keys, q, coef.view(-1)
            trace[f"policy{b}"] = qk
    return torch.cat(outs, 0), sum(nz) / B


def when2com_forward(data_dict, sd, args, trace=None):
    """models/airv2x_when2com.py:112-151 (det task)."""
    feats, record_len = w2c.extract_features(data_dict, sd, args)
    mf = args["modality_fusion"]
    sf2d, _ = w2c.backbone_forward(feats, sd, mf["base_bev_backbone"])
    s = w2c.shrink_conv(sf2d, sd, mf["shrink_header"]) if mf["shrink_header"]["use"] else sf2d
    if mf.get("compression", 0) > 0:      # NaiveCompressor(256, args["compression"]) (airv2x_v2xvit.py:42-44,122-123; airv2x_when2com.py:50-52,122-123)
        s = w2c.naive_compress(s, sd)
    fused, rate = when2com_fuse(s, record_len, data_dict["img_pairwise_t_matrix_collab"], sd, args["when2com_fusion"], trace=trace)
    out = {"psm": w2c.head(fused, sd, "cls_head"), "rm": w2c.head(fused, sd, "reg_head")}
    if args["obj_head"]:
        out["obj"] = w2c.head(fused, sd, "obj_head")
    out.update({"mask": 0, "comm_rate": rate})
    if trace is not None:
        trace.update({"shrink": s, "fused": fused})
    return out
