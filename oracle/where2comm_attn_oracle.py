"""ORACLE (test infrastructure, not product code): the OPV2V-style Where2comm fusion.

CPU fp32 restatement of models/where2comm_modules/where2comm_attn.py (Where2comm.forward :269-404, AttenFusion :55-67 with
ScaledDotProductAttention :46-52, MaxFusion :70-75) and models/where2comm_modules/where2comm.py (Communication.forward
:47-116) as plain functions over state_dicts; warp_affine_simple = common_modules/torch_transformation_utils.py:327-334.
Parity: PINNED by tests/golden/w2c_attn.npz (tools/gen_golden.py runs the real reference modules with the reference's
BaseBEVBackbone).

As-written behaviour that this file keeps (the GPU path must match the reference, not an idealised one):
  * the third return value of Communication.forward is the communication VOLUME (mean over the samples of
    count_nonzero(level-0 features x thresholded mask)), which Where2comm hands back under the name `communication_rates`;
  * `communication_mask_nodiag[::2] = 1` (where2comm.py:104-108): every agent with an EVEN index within its sample
    transmits everything, only the odd ones are thresholded;
  * single-scale branch: `node_features * communication_masks[b]` (:394) indexes the CONCATENATED (sum N,1,H,W) mask tensor
    with the sample index b, i.e. all agents of sample b are multiplied by the mask of global agent b;
  * the pairwise matrix is normalised once with the H, W of the tensor handed to forward() (:293-307) and the same
    normalised matrix is used at every level.
The 'Transformer' aggregation mode is not restated: EncodeLayer.forward calls nn.MultiheadAttention with a `quality_map`
keyword (:108-110) that torch's module does not have, so with_scm configurations cannot run in the reference either.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import where2comm_oracle as w2c
from .when2com_oracle import normalized_pairwise, warp_affine_simple


def communication(batch_x, batch_conf, sd, comm_cfg, prefix="naive_communication."):
    """where2comm.py:47-116.  batch_x / batch_conf: lists over the samples of (N_b, C, H, W) / (N_b, A, H, W).
    -> (masks (sum N,1,H,W) with the even agents forced to one, volume as np.float64, smoothed maps)."""
    thre = comm_cfg["thre"]
    masks, vols, maps = [], [], []
    for x, conf in zip(batch_x, batch_conf):
        ori = conf.sigmoid().max(dim=1)[0].unsqueeze(1)
        if "gaussian_smooth" in comm_cfg:
            wgt, bias = sd[prefix + "gaussian_filter.weight"], sd[prefix + "gaussian_filter.bias"]
            cm = F.conv2d(ori, wgt, bias, padding=(wgt.shape[-1] - 1) // 2)
        else:
            cm = ori
        m = torch.where(cm > thre, torch.ones_like(cm), torch.zeros_like(cm))
        vols.append((x * m).count_nonzero().item())
        m = m.clone()
        m[::2] = 1.0
        masks.append(m)
        maps.append(cm)
    return torch.cat(masks, 0), np.sum(vols) / len(batch_x), torch.cat(maps, 0)


def atten_fusion(x):
    """AttenFusion.forward :60-67: (N,C,H,W) -> (C,H,W); the same arithmetic as where2comm_fuse's AttentionFusion."""
    return w2c.attention_fusion(x)


def max_fusion(x):
    return torch.max(x, dim=0)[0]


def _fuser(mode):
    if mode == "ATTEN":
        return atten_fusion
    if mode == "MAX":
        return max_fusion
    raise NotImplementedError(f"agg_operator mode {mode!r}")


def _split(x, record_len):
    cs = np.cumsum(record_len)[:-1].tolist()
    return torch.tensor_split(x, cs)


def resnet_features(x, sd, bb_cfg, prefix="backbone.resnet."):
    """ResNetModified._forward_impl (coalign_modules/resblock.py:259-268, levels layer0, layer1, ...) with BasicBlock.forward
    (:72-88): the level maps."""
    def bn(t, p):
        return F.batch_norm(t, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)
    outs = []
    for li, (n, st) in enumerate(zip(bb_cfg["layer_nums"], bb_cfg["layer_strides"])):
        for j in range(n):
            q = f"{prefix}layer{li}.{j}."
            s_ = st if j == 0 else 1
            idt = x
            if (q + "downsample.0.weight") in sd:
                idt = bn(F.conv2d(x, sd[q + "downsample.0.weight"], None, s_), q + "downsample.1")
            y = F.relu(bn(F.conv2d(x, sd[q + "conv1.weight"], None, s_, 1), q + "bn1"))
            y = bn(F.conv2d(y, sd[q + "conv2.weight"], None, 1, 1), q + "bn2")
            x = F.relu(y + idt)
        outs.append(x)
    return outs


def where2comm_attn(x, rm, record_len, pairwise_t_matrix, sd, cfg, backbone_sd=None, bb_cfg=None, trace=None, with_resnet=False):
    """Where2comm.forward :269-404.  ``sd``: the fusion module's own state_dict (the gaussian filter, or empty);
    ``backbone_sd`` / ``bb_cfg``: BaseBEVBackbone's state_dict (keys "backbone.blocks..." as oracle/where2comm_oracle.py
    reads them) and config, for the multi-scale form.  -> (fused (B,C',H',W'), communication volume or tensor(0))."""
    record_len = [int(v) for v in record_len]
    _, C, H, W = x.shape
    B = pairwise_t_matrix.shape[0]
    t = normalized_pairwise(pairwise_t_matrix, H, W, cfg["voxel_size"][0], cfg["downsample_rate"])
    fuse = _fuser(cfg["agg_operator"]["mode"])
    has_comm = "communication" in cfg
    vol = torch.tensor(0)
    if cfg["multi_scale"]:
        ups = []
        n_levels = len(cfg["layer_nums"])
        feats = resnet_features(x, backbone_sd, bb_cfg) if with_resnet else None     # :312-314: all levels from the UNMASKED input
        for i in range(n_levels):
            x = feats[i] if with_resnet else w2c.backbone_block(x, backbone_sd, i, bb_cfg["layer_nums"][i])
            if i == 0 and has_comm:
                masks, vol, maps = communication(_split(x, record_len), _split(rm, record_len), sd, cfg["communication"])
                x = x * masks
                if trace is not None:
                    trace["mask"], trace["smooth"] = masks, maps
            fused = []
            for b, xb in enumerate(_split(x, record_len)):
                n = record_len[b]
                fused.append(fuse(warp_affine_simple(xb, t[b, 0, :n], xb.shape[-2:])))
            fused = torch.stack(fused)
            if trace is not None:
                trace[f"fused{i}"] = fused
            if len(bb_cfg.get("upsample_strides", [])) > 0:
                ups.append(w2c.backbone_deblock(fused, backbone_sd, i, bb_cfg["upsample_strides"][i]))
            else:
                ups.append(fused)
        out = torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]
        if len(bb_cfg.get("upsample_strides", [])) > n_levels:
            out = w2c.backbone_deblock(out, backbone_sd, n_levels, bb_cfg["upsample_strides"][n_levels])
        return out, vol
    feats = _split(x, record_len)
    if has_comm:
        masks, vol, maps = communication(feats, _split(rm, record_len), sd, cfg["communication"])
        if trace is not None:
            trace["mask"], trace["smooth"] = masks, maps
    fused = []
    for b, xb in enumerate(feats):
        n = record_len[b]
        if has_comm:
            xb = xb * masks[b]      # as written (:394): the mask of GLOBAL agent b, broadcast over the sample's agents
        fused.append(fuse(warp_affine_simple(xb, t[b, 0, :n], (H, W))))
    return torch.stack(fused), vol
