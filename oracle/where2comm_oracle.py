"""ORACLE (test infrastructure, not product code).

CPU fp32 restatement of the reference's AirV2X Where2Comm-LiDAR forward pass,
written as plain functions over a ``state_dict``.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product path (``airv2x_perception_amd``) never does.

Parity status: PINNED.  ``tools/gen_golden.py`` imports the real reference
(``/root/reference``, this container only) and stores its outputs under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks every function here
against those vectors.

Every function cites the reference lines it restates (paths relative to
``/root/reference/opencood``).
"""
from __future__ import annotations


import numpy as np
import torch
import torch.nn.functional as F

AGENT_TYPES = ("vehicle", "rsu", "drone")
TYPE_PREFIX = {"vehicle": "veh_models", "rsu": "rsu_models", "drone": "drone_models"}
BN_EPS = 1e-3  # every BatchNorm on the path: airv2x_pillar_vfe.py:21, base_bev_backbone.py:52,65,83


BN_MOMENTUM = 0.01  # same lines
_STATE = {"train": False}


class train_mode:
    """``with train_mode():`` -- every BatchNorm below uses batch statistics and updates the running statistics of the
    ``state_dict`` it reads IN PLACE (momentum 0.01, ``num_batches_tracked`` += 1), as nn.BatchNorm does under ``.train()``;
    the functions are plain differentiable torch ops, so ``torch.autograd`` of them is the checker of the HIP backward."""

    def __enter__(self):
        self.prev = _STATE["train"]
        _STATE["train"] = True

    def __exit__(self, *a):
        _STATE["train"] = self.prev


def _bn(x, sd, prefix):
    """BatchNorm: eval mode (running statistics) unless inside ``train_mode()``."""
    if _STATE["train"]:
        nbt = sd.get(prefix + ".num_batches_tracked")
        if nbt is not None:
            nbt += 1
        return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                            sd[prefix + ".weight"], sd[prefix + ".bias"], True, BN_MOMENTUM, BN_EPS)
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], False, 0.0, BN_EPS)


# ---------------------------------------------------------------- a4: PillarVFE
def pillar_vfe(voxel_features, voxel_num_points, coords, sd, prefix, voxel_size, pc_range):
    """models/common_modules/airv2x_pillar_vfe.py:105-160 (+ PFNLayer :27-45).

    voxel_features (M,32,4) f32, voxel_num_points (M,) i32, coords (M,4) i32 [b,z,y,x]
    -> (M,64).  Always returns (M,64) (the reference's ``squeeze()`` :156 differs
    only for M == 1).
    """
    vx, vy, vz = voxel_size
    x_off = vx / 2 + pc_range[0]
    y_off = vy / 2 + pc_range[1]
    z_off = vz / 2 + pc_range[2]
    vf = voxel_features
    points_mean = vf[:, :, :3].sum(dim=1, keepdim=True) / voxel_num_points.type_as(vf).view(-1, 1, 1)
    f_cluster = vf[:, :, :3] - points_mean
    f_center = torch.zeros_like(vf[:, :, :3])
    f_center[:, :, 0] = vf[:, :, 0] - (coords[:, 3].to(vf.dtype).unsqueeze(1) * vx + x_off)
    f_center[:, :, 1] = vf[:, :, 1] - (coords[:, 2].to(vf.dtype).unsqueeze(1) * vy + y_off)
    f_center[:, :, 2] = vf[:, :, 2] - (coords[:, 1].to(vf.dtype).unsqueeze(1) * vz + z_off)
    feats = torch.cat([vf, f_cluster, f_center], dim=-1)  # use_absolute_xyz, no distance
    P = feats.shape[1]
    mask = voxel_num_points.int().unsqueeze(1) > torch.arange(P, dtype=torch.int32).view(1, -1)
    feats = feats * mask.unsqueeze(-1).type_as(vf)
    x = F.linear(feats, sd[prefix + ".pfn_layers.0.linear.weight"])
    x = _bn(x.permute(0, 2, 1), sd, prefix + ".pfn_layers.0.norm").permute(0, 2, 1)
    x = F.relu(x)
    return torch.max(x, dim=1)[0]


# ---------------------------------------------------------------- a5: scatter
def pillar_scatter(pillar_features, coords, n_agents, nx, ny):
    """models/common_modules/point_pillar_scatter.py:43-80: idx = z + y*nx + x,
    canvas[:, idx] = pillars^T per agent.  -> (n_agents, 64, ny, nx)."""
    C = pillar_features.shape[1]
    out = torch.zeros(n_agents, C, ny * nx, dtype=pillar_features.dtype)
    for b in range(n_agents):
        m = coords[:, 0] == b
        c = coords[m]
        idx = (c[:, 1] + c[:, 2] * nx + c[:, 3]).long()
        out[b][:, idx] = pillar_features[m].t()
    return out.view(n_agents, C, ny, nx)


# ---------------------------------------------------------------- a6: encoders + repack
def extract_features(data_dict, sd, args, trace=None):
    """models/common_modules/airv2x_base_model.py:101-248 for B=1 frames: run each type's encoders (one per entry of
    ``modalities``: Sequential(PillarVFE, PointPillarScatter) for "lidar", LiftSplatShootEncoder for "cam"), average the
    modality maps (``fuse_bev`` :167-177), then concatenate in the order vehicle, rsu, drone (``repack_batch`` iterates
    ``batch_dicts`` in that insertion order :127-150, :215-231)."""
    outs = []
    n_total = 0
    for t in AGENT_TYPES:
        if t not in args["collaborators"]:
            continue
        d = data_dict[t]
        if len(d["batch_idxs"]) == 0:
            continue
        maps = []
        for mi, m in enumerate(args[t]["modalities"]):
            if m == "lidar":
                lid = d["batch_merged_lidar_features_torch"]
                cfg = args[t]["lidar"]
                pf = pillar_vfe(lid["voxel_features"], lid["voxel_num_points"], lid["voxel_coords"], sd,
                                f"{TYPE_PREFIX[t]}.{mi}.0", cfg["voxel_size"], cfg["lidar_range"])
                n_t = int(lid["voxel_coords"][:, 0].max().item()) + 1  # point_pillar_scatter.py:43
                nx, ny, nz = [int(v) for v in cfg["point_pillar_scatter"]["grid_size"]]
                maps.append(pillar_scatter(pf, lid["voxel_coords"], n_t, nx, ny))
            elif m == "cam":
                from . import camera_oracle as cam
                tr = {} if trace is not None else None
                maps.append(cam.lss_encoder_forward(sd, f"{TYPE_PREFIX[t]}.{mi}.", d["batch_merged_cam_inputs"], args[t]["cam"], trace=tr))
                if trace is not None:
                    trace["cam_" + t] = tr
            else:
                raise NotImplementedError(m)
        outs.append(maps[0] if len(maps) == 1 else torch.mean(torch.stack(maps, dim=0), dim=0))
        n_total += outs[-1].shape[0]
    feats = torch.cat(outs, dim=0)
    record_len = torch.tensor([n_total], dtype=torch.int32)
    return feats, record_len


# ---------------------------------------------------------------- a7: backbone
def backbone_block(x, sd, i, layer_num):
    """base_bev_backbone.py:41-70: ZeroPad2d(1)+Conv3x3 stride 2 (pad 0)+BN+ReLU,
    then layer_num x [Conv3x3 pad 1 + BN + ReLU]."""
    p = f"backbone.blocks.{i}"
    x = F.conv2d(F.pad(x, (1, 1, 1, 1)), sd[f"{p}.1.weight"], None, stride=2, padding=0)
    x = F.relu(_bn(x, sd, f"{p}.2"))
    idx = 4
    for _ in range(layer_num):
        x = F.conv2d(x, sd[f"{p}.{idx}.weight"], None, stride=1, padding=1)
        x = F.relu(_bn(x, sd, f"{p}.{idx + 1}"))
        idx += 3
    return x


def backbone_deblock(x, sd, i, stride):
    """base_bev_backbone.py:71-105: ConvTranspose2d(k=s, stride=s, no bias)+BN+ReLU, or -- stride < 1, a "deblock" that down-samples --
    Conv2d(k, stride=k, no bias)+BN+ReLU with k = round(1 / stride)."""
    p = f"backbone.deblocks.{i}"
    if stride >= 1:
        x = F.conv_transpose2d(x, sd[f"{p}.0.weight"], None, stride=int(stride))
    else:
        x = F.conv2d(x, sd[f"{p}.0.weight"], None, stride=int(round(1.0 / stride)))
    return F.relu(_bn(x, sd, f"{p}.1"))


def backbone_final_deblock(x, sd, bb_cfg):
    """base_bev_backbone.py:107-121, 151-152: one more ConvTranspose2d + BN + ReLU on the concatenated map when ``upsample_strides`` has
    an entry beyond the levels."""
    nlev = len(bb_cfg["layer_nums"])
    if len(bb_cfg.get("upsample_strides", [])) <= nlev:
        return x
    return backbone_deblock(x, sd, nlev, bb_cfg["upsample_strides"][-1])


def backbone_forward(x, sd, bb_cfg):
    """base_bev_backbone.py:125-154 -> (spatial_features_2d, [block outputs])."""
    ups, blocks = [], []
    for i, n in enumerate(bb_cfg["layer_nums"]):
        x = backbone_block(x, sd, i, n)
        blocks.append(x)
        ups.append(backbone_deblock(x, sd, i, bb_cfg["upsample_strides"][i]))
    return backbone_final_deblock(torch.cat(ups, dim=1), sd, bb_cfg), blocks


# ---------------------------------------------------------------- a8: shrink header
def shrink_conv(x, sd, sh_cfg):
    """downsample_conv.py:8-54: per layer Conv(k,stride,pad)+ReLU, Conv3x3 pad1 +ReLU (biases, no BN)."""
    for li, (k, s, pd) in enumerate(zip(sh_cfg["kernal_size"], sh_cfg["stride"], sh_cfg["padding"])):
        p = f"shrink_conv.layers.{li}.double_conv"
        x = F.relu(F.conv2d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], stride=s, padding=pd))
        x = F.relu(F.conv2d(x, sd[p + ".2.weight"], sd[p + ".2.bias"], stride=1, padding=1))
    return x


# ---------------------------------------------------------------- a9: heads
def naive_compress(x, sd, prefix="naive_compressor"):
    """NaiveCompressor.forward (models/common_modules/naive_compress.py:38-42): encoder Conv3x3+BN(eps 1e-3)+ReLU,
    decoder 2 x [Conv3x3+BN+ReLU]; the convolutions carry biases."""
    for conv, bn in (("encoder.0", "encoder.1"), ("decoder.0", "decoder.1"), ("decoder.3", "decoder.4")):
        x = F.conv2d(x, sd[f"{prefix}.{conv}.weight"], sd[f"{prefix}.{conv}.bias"], padding=1)
        x = F.relu(_bn(x, sd, f"{prefix}.{bn}"))
    return x


def head(x, sd, name):
    """airv2x_where2com.py:60-69: 1x1 conv with bias."""
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"])


# ---------------------------------------------------------------- a11: communication mask
def communication(psm_split, sd, comm_cfg, topk=None):
    """where2comm_modules/where2comm_fuse.py:83-149: the eval branch, or -- ``topk`` = one K per sample, the value the
    reference draws as int(H * W * random.uniform(0, 1)) (:106) -- the training branch (:104-121).

    psm_split: list over samples of (L_b, A*C, H, W).  Returns (mask (sum L,1,H,W),
    rate 0-dim tensor, smoothed maps (for threshold-margin checks in tests)).
    """
    thr = comm_cfg["threshold"]
    smooth = "gaussian_smooth" in comm_cfg
    masks, rates, maps = [], [], []
    B = len(psm_split)
    for b in range(B):
        ori, _ = psm_split[b].sigmoid().max(dim=1, keepdim=True)
        if smooth:
            k = sd["fusion_net.naive_communication.gaussian_filter.weight"].shape[-1]
            cm = F.conv2d(ori, sd["fusion_net.naive_communication.gaussian_filter.weight"],
                          sd["fusion_net.naive_communication.gaussian_filter.bias"], padding=(k - 1) // 2)
        else:
            cm = ori
        L, _, H, W = cm.shape
        if topk is not None:
            flat = cm.reshape(L, H * W)
            _, idx = torch.topk(flat, k=int(topk[b]), sorted=False)
            m = torch.scatter(torch.zeros_like(flat), -1, idx, torch.ones(L, int(topk[b]), dtype=flat.dtype)).reshape(L, 1, H, W)
        elif thr:
            m = torch.where(cm > thr, torch.ones_like(cm), torch.zeros_like(cm))
        else:
            m = torch.ones_like(cm)
        rates.append(m.sum() / (L * H * W))  # measured before the ego override (:137)
        m[0] = 1  # ego always transmits to itself (:141)
        masks.append(m)
        maps.append(cm)
    return torch.cat(masks, 0), sum(rates) / B, torch.cat(maps, 0)


# ---------------------------------------------------------------- a12: per-pixel attention
def attention_fusion(x):
    """where2comm_fuse.py:152-164 + :41-45: per pixel softmax(X X^T / sqrt(C)) X, keep row 0.
    x (n,C,H,W) -> (C,H,W)."""
    n, C, H, W = x.shape
    q = x.view(n, C, -1).permute(2, 0, 1)  # (HW, n, C)
    score = torch.bmm(q, q.transpose(1, 2)) / np.sqrt(C)
    attn = F.softmax(score, -1)
    ctx = torch.bmm(attn, q)
    return ctx.permute(1, 2, 0).view(n, C, H, W)[0]


def _split(x, record_len):
    cs = torch.cumsum(record_len, dim=0)
    return torch.tensor_split(x, cs[:-1].cpu())


# ---------------------------------------------------------------- a10: Where2comm multi-scale fusion
def where2comm_fuse(x, psm_single, record_len, sd, args, trace=None, topk=None, comm_mask=None):
    """where2comm_fuse.py:198-263 (multi_scale, not fully connected).
    x (sumN,64,H,W) canvas features; returns (fused (B,384,H/2,W/2), rate)."""
    bb = args["modality_fusion"]["base_bev_backbone"]
    fcfg = args["where2com_fusion"]
    ups = []
    rate = None
    for i, n in enumerate(fcfg["layer_nums"]):
        x = backbone_block(x, sd, i, n)
        if i == 0:
            if fcfg["fully"]:
                rate = torch.tensor(1)
            else:
                masks, rate, maps = communication(_split(psm_single, record_len), sd, fcfg["communication"], topk)
                if comm_mask is not None:   # replay a recorded mask (the top-K / threshold cut is discontinuous in the logits)
                    masks = comm_mask.to(x.dtype).reshape(masks.shape)
                if trace is not None:
                    trace["comm_mask"] = masks          # Communication's output, before any resize
                    trace["comm_map"] = maps
                if x.shape[-1] != masks.shape[-1]:
                    masks = F.interpolate(masks, size=(x.shape[-2], x.shape[-1]), mode="bilinear",
                                          align_corners=False)
                    if trace is not None:
                        trace["comm_mask_resized"] = masks
                x = x * masks
        fused = torch.stack([attention_fusion(xb) for xb in _split(x, record_len)])
        if trace is not None:
            trace[f"masked_block{i}"] = x
            trace[f"fused{i}"] = fused
        ups.append(backbone_deblock(fused, sd, i, bb["upsample_strides"][i]))
    return backbone_final_deblock(torch.cat(ups, dim=1), sd, bb), rate      # where2comm_fuse.py:260-263


def where2comm_fuse_single(x, psm_single, record_len, sd, args, trace=None, topk=None, comm_mask=None):
    """where2comm_fuse.py:264-286 (multi_scale: false): mask the (compressed) shrunk map, one per-pixel attention per sample.
    x (sumN,256,H,W) -> (fused (B,256,H,W), rate)."""
    fcfg = args["where2com_fusion"]
    if fcfg["fully"]:
        rate = torch.tensor(1)
    else:
        masks, rate, maps = communication(_split(psm_single, record_len), sd, fcfg["communication"], topk)
        if comm_mask is not None:
            masks = comm_mask.to(x.dtype).reshape(masks.shape)
        x = x * masks
        if trace is not None:
            trace["comm_mask"] = masks
            trace["comm_map"] = maps
    fused = torch.stack([attention_fusion(xb) for xb in _split(x, record_len)])
    if trace is not None:
        trace["masked_single"] = x
    return fused, rate


# ---------------------------------------------------------------- full forward
def where2com_forward(data_dict, sd, args, trace=None, reference_schedule=False, topk=None, comm_mask=None):
    """models/airv2x_where2com.py:117-179 (det task; multi_scale true / false, compression 0 / r).

    The reference evaluates the backbone twice before the fusion (:119, :124); in
    eval mode both passes give identical tensors, so the oracle runs it once
    unless ``reference_schedule`` is set (to time the as-written cost; and REQUIRED inside ``train_mode()``, where every
    pass updates the running statistics).  ``topk``: the training branch of the communication mask, one K per sample.
    The debug PNG (:137-139) has no effect on outputs and is dropped.
    """
    mf = args["modality_fusion"]
    feats, record_len = extract_features(data_dict, sd, args, trace)
    if reference_schedule:
        backbone_forward(feats, sd, mf["base_bev_backbone"])
    sf2d, blocks = backbone_forward(feats, sd, mf["base_bev_backbone"])
    comm_rate = int(feats.count_nonzero().item())  # :122
    s = shrink_conv(sf2d, sd, mf["shrink_header"]) if mf["shrink_header"]["use"] else sf2d
    psm_single = head(s, sd, "cls_head")  # :145
    sc = s
    if mf.get("compression", 0) > 0:      # :147-150 (the ratio is args["compression"], read by the constructor :52)
        sc = naive_compress(s, sd)
    if args["where2com_fusion"]["multi_scale"]:
        fused, rate = where2comm_fuse(feats, psm_single, record_len, sd, args, trace, topk, comm_mask)  # :153-159 (sc is dead here)
        fs = shrink_conv(fused, sd, mf["shrink_header"]) if mf["shrink_header"]["use"] else fused      # :161-162
    else:
        fused, rate = where2comm_fuse_single(sc, psm_single, record_len, sd, args, trace, topk, comm_mask)  # :163-166
        fs = fused
    out = {"psm": head(fs, sd, "cls_head"), "rm": head(fs, sd, "reg_head")}
    if args["obj_head"]:
        out["obj"] = head(fs, sd, "obj_head")
    out.update({"mask": 0, "com": rate, "comm_rate": comm_rate})
    if trace is not None:
        trace.update({"spatial_features": feats, "block0": blocks[0], "block1": blocks[1], "block2": blocks[2],
                      "spatial_features_2d": sf2d, "shrink": s, "psm_single": psm_single,
                      "fused_2d": fused, "fused_shrink": fs})
    return out
