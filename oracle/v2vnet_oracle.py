"""ORACLE (test infrastructure, not product code): AirV2X V2VNet-LiDAR forward.

CPU fp32 restatement of models/airv2x_v2vnet.py:191-244 (det task) and its fusion
models/v2vnet_modules/v2v_fuse.py:54-180 (V2VNetFusion.forward, as written: every node is updated in every iteration,
ConvGRU with a zero initial hidden state and a one-step sequence, models/v2vnet_modules/convgru.py:52-73,141-190)
as plain functions over a state_dict.  warp_affine_simple = common_modules/torch_transformation_utils.py:327-334.
Parity: PINNED by tests/golden/v2vnet_*.npz (tools/gen_golden.py runs the real reference).
The per-agent trunk (encoders, backbone, shrink) is shared with oracle/where2comm_oracle.py.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import where2comm_oracle as w2c
from .when2com_oracle import normalized_pairwise, warp_affine_simple


def conv_gru_step(x, sd, p):
    """ConvGRU.forward on a (1,1,2C,H,W) sequence with hidden_state=None (convgru.py:141-190): one cell, h_cur = 0."""
    h_cur = torch.zeros(x.shape[0], sd[p + ".conv_can.weight"].shape[0], x.shape[2], x.shape[3])
    combined = torch.cat([x, h_cur], 1)
    cc = F.conv2d(combined, sd[p + ".conv_gates.weight"], sd[p + ".conv_gates.bias"], padding=1)
    gamma, beta = torch.split(cc, h_cur.shape[1], 1)
    reset, update = torch.sigmoid(gamma), torch.sigmoid(beta)
    combined = torch.cat([x, reset * h_cur], 1)
    cnm = torch.tanh(F.conv2d(combined, sd[p + ".conv_can.weight"], sd[p + ".conv_can.bias"], padding=1))
    return (1 - update) * h_cur + update * cnm


def v2vnet_fuse(x, record_len, pairwise_t_matrix, sd, cfg, prefix="fusion_net", trace=None):
    """V2VNetFusion.forward :54-180 -> ((B,C,H,W), comm rate as a python float)."""
    if cfg["conv_gru"]["num_layers"] != 1:
        raise NotImplementedError("one ConvGRU layer (every shipped v2vfusion block)")
    _, C, H, W = x.shape
    B, L = pairwise_t_matrix.shape[:2]
    split = list(torch.split(x, [int(v) for v in record_len]))
    t = normalized_pairwise(pairwise_t_matrix, H, W, cfg["voxel_size"][0], cfg["downsample_rate"])
    roi = torch.zeros(B, L, L, 1, H, W)
    for b in range(B):
        for i in range(int(record_len[b])):
            roi[b, i] = warp_affine_simple(torch.ones(L, 1, H, W), t[b][i], (H, W))
    nodes, comm = split, []
    for it in range(cfg["num_iteration"]):
        new_nodes = []
        for b in range(B):
            N = int(record_len[b])
            tm = t[b][:N, :N]
            upd = []
            for i in range(N):
                mask = roi[b, i, :N]
                comm.append(int(nodes[b].count_nonzero()))
                nb = warp_affine_simple(nodes[b], tm[i], (H, W))
                ego = nodes[b][i].unsqueeze(0).repeat(N, 1, 1, 1)
                msg = F.conv2d(torch.cat([nb, ego], 1), sd[prefix + ".msg_cnn.weight"], sd[prefix + ".msg_cnn.bias"], padding=1) * mask
                if cfg["agg_operator"] == "avg":
                    agg = msg.mean(0)
                elif cfg["agg_operator"] == "max":
                    agg = msg.max(0)[0]
                else:
                    raise ValueError("agg_operator has wrong value")
                cat = torch.cat([nodes[b][i], agg], 0)
                if cfg["gru_flag"]:
                    out = conv_gru_step(cat.unsqueeze(0), sd, prefix + ".conv_gru.cell_list.0")[0]
                else:
                    out = nodes[b][i] + agg
                if trace is not None and b == 0 and i == 0:
                    trace[f"agg_it{it}"], trace[f"node0_it{it}"] = agg.clone(), out.clone()
                upd.append(out.unsqueeze(0))
            new_nodes.append(torch.cat(upd, 0))
        nodes = new_nodes
    rate = float(sum(comm)) / B
    out = torch.cat([n[0:1] for n in nodes], 0)
    out = F.linear(out.permute(0, 2, 3, 1), sd[prefix + ".mlp.weight"], sd[prefix + ".mlp.bias"]).permute(0, 3, 1, 2)
    return out, rate


def v2vnet_forward(data_dict, sd, args, trace=None):
    """Airv2xV2VNet.forward :191-244 (det, LiDAR, compression 0); the two debug PNG writes (:203-205, :213-214) have no
    effect on the outputs and are dropped."""
    mf = args["modality_fusion"]
    feats, record_len = w2c.extract_features(data_dict, sd, args)
    sf2d, _ = w2c.backbone_forward(feats, sd, mf["base_bev_backbone"])
    s = w2c.shrink_conv(sf2d, sd, mf["shrink_header"]) if mf["shrink_header"]["use"] else sf2d
    if mf.get("compression", 0) > 0:      # NaiveCompressor(256, args["compression"]) (airv2x_v2vnet.py:42-44, 180-181)
        s = w2c.naive_compress(s, sd)
    fused, rate = v2vnet_fuse(s, record_len, data_dict["img_pairwise_t_matrix_collab"], sd, args["v2vfusion"], trace=trace)
    out = {"psm": w2c.head(fused, sd, "cls_head"), "rm": w2c.head(fused, sd, "reg_head")}
    if args["obj_head"]:
        out["obj"] = w2c.head(fused, sd, "obj_head")
    out.update({"mask": 0, "comm_rate": rate})
    if trace is not None:
        trace.update({"shrink": s, "fused": fused})
    return out
