"""Train-mode forward of ``Airv2xWhere2com`` (models/airv2x_where2com.py:117-179 with ``self.training``): the graph the
reference hands to torch autograd (tools/train.py:220-247), built from the HIP forward / backward ops of ``train_ops``.

As-written schedule of the reference, kept:
  * the backbone runs twice on the scattered canvas (:119, :124) and its blocks a third time inside the fusion
    (where2comm_fuse.py:218).  Both full passes see the same input, so they are computed ONCE here and every BatchNorm's
    running statistics are updated twice with the same batch statistics; blocks[0] of the fusion pass sees that input a
    third time (third identical update), blocks[1..2] see the MASKED maps of all agents (ego included) and the deblocks the
    fused maps: fresh batch statistics, one more update each -- ``num_batches_tracked`` advances by 3 per step;
  * the single-agent ``psm`` only drives the communication mask (random top-K in train mode, where2comm_fuse.py:104-121:
    K = int(H * W * random.uniform(0, 1)) from python's ``random``, one draw per sample, as the reference draws it), so
    no gradient flows through the two full passes: they run under ``no_grad``;
  * the loss sees the heads on the fused map only.
"""
from __future__ import annotations

import json

import random
from ctypes import c_float

import torch

from .. import _lib
from . import train_ops as T
from .autograd import _P, _runner
from .engine import AGENT_TYPES, frame_layout

TYPE_PREFIX = {"vehicle": "veh_models", "rsu": "rsu_models", "drone": "drone_models"}


def _bn_update(sd, prefix, stats, times):
    T.update_running_stats(sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd.get(prefix + ".num_batches_tracked"), stats, times)


def _running(sd, prefix, times):
    return (sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd.get(prefix + ".num_batches_tracked"), times)


def _block(P, sd, i, x, n_layers, stride, times, prefix="backbone."):
    """backbone.blocks[i] (base_bev_backbone.py:41-70) in train mode; BatchNorm running statistics updated ``times`` times."""
    idx = 1
    for li in range(n_layers + 1):
        bn = f"{prefix}blocks.{i}.{idx + 1}"
        x = T.conv_bn_act(x, P[f"{prefix}blocks.{i}.{idx}.weight"], P[bn + ".weight"], P[bn + ".bias"], stride if li == 0 else 1, 1,
                          running=_running(sd, bn, times))
        idx += 3
    return x


def _deblock(P, sd, i, x, times, prefix="backbone."):
    bn = f"{prefix}deblocks.{i}.1"
    return T.deconv_bn_act(x, P[f"{prefix}deblocks.{i}.0.weight"], P[bn + ".weight"], P[bn + ".bias"], running=_running(sd, bn, times))


def _shrink(P, cfg, x, prefix="shrink_conv."):
    if not cfg.get("use", True):
        return x
    for li, (k, pd) in enumerate(zip(cfg["kernal_size"], cfg["padding"])):
        p = f"{prefix}layers.{li}.double_conv"
        x = T.conv_bias_act(x, P[p + ".0.weight"], P[p + ".0.bias"], 1, pd, True)
        x = T.conv_bias_act(x, P[p + ".2.weight"], P[p + ".2.bias"], 1, 1, True)
    return x


def _heads(P, names, x):
    """The 1x1 heads as ONE convolution padded to a multiple of 32 output channels; returns the NCHW maps."""
    w = torch.cat([P[n + ".weight"] for n in names], 0)
    b = torch.cat([P[n + ".bias"] for n in names], 0)
    tot = w.shape[0]
    padc = (tot + 31) // 32 * 32 - tot
    if padc:
        w = torch.cat([w, w.new_zeros((padc,) + tuple(w.shape[1:]))], 0)
        b = torch.cat([b, b.new_zeros(padc)], 0)
    y = T.conv_bias_act(x, w, b, 1, 0, False)                       # (B, H, W, 32)
    outs, o = [], 0
    for n in names:
        c = P[n + ".weight"].shape[0]
        outs.append(y[..., o:o + c].permute(0, 3, 1, 2).contiguous())
        o += c
    return outs


_CAM_GEOMETRY = {}


def encode_train(args, P, sd, data_dict, slots, n, dev, r, model=None):
    """The per-type encoders in train mode: PillarVFE (BatchNorm1d batch statistics, running statistics updated once) + scatter for
    every agent of the frame -> (canvas (n, ny, nx, 64) with its autograd graph, device counter of its non-zero elements).
    Agent types with a camera encoder (``model`` gives the packed geometry of its eval engine): LiftSplatShootEncoder in train mode
    (train_camera.py), and the mean over the modality maps where a type carries both (Airv2xBase.fuse_bev)."""
    groups, params, prefixes = [], [], []
    ny = nx = None
    cam_types = [t for t in AGENT_TYPES if t in slots and "cam" in args[t]["modalities"]]
    for t in AGENT_TYPES:
        if t not in slots or "lidar" not in args[t]["modalities"]:
            continue
        mi = args[t]["modalities"].index("lidar")
        lid = data_dict[t]["batch_merged_lidar_features_torch"]
        cfg = args[t]["lidar"]
        vs, rng = cfg["voxel_size"], cfg["lidar_range"]
        g = [int(v) for v in cfg["point_pillar_scatter"]["grid_size"]]
        nx, ny = g[0], g[1]
        geom = (c_float * 6)(vs[0], vs[1], vs[2], vs[0] / 2 + rng[0], vs[1] / 2 + rng[1], vs[2] / 2 + rng[2])
        groups.append({"vf": lid["voxel_features"].to(dev).contiguous().float(), "vc": lid["voxel_coords"].to(dev).contiguous().to(torch.int32),
                       "vn": lid["voxel_num_points"].to(dev).contiguous().to(torch.int32), "slots": slots[t], "geom": geom})
        p = f"{TYPE_PREFIX[t]}.{mi}.0.pfn_layers.0"
        params += [P[p + ".linear.weight"], P[p + ".norm.weight"], P[p + ".norm.bias"]]
        prefixes.append(p + ".norm")
    canvas = None
    if groups:
        st = []
        canvas = T.pillar_encode(groups, n, ny, nx, params, stats_out=st)
        for p, s in zip(prefixes, st):
            _bn_update(sd, p, s, 1)
    if cam_types:
        from . import train_camera as TC
        from .camera import CameraGeometry
        geo = _CAM_GEOMETRY      # keyed by the CONTENT of the type's camera block (an id(args) key can be re-used by a later model's dict)
        rows = [None] * n
        if canvas is not None:
            for t in AGENT_TYPES:
                if t in slots and "lidar" in args[t]["modalities"]:
                    for s_ in slots[t]:
                        rows[s_] = canvas[s_:s_ + 1]
        for t in cam_types:
            mi = args[t]["modalities"].index("cam")
            ci = data_dict[t].get("batch_merged_cam_inputs")
            if ci is None:
                raise ValueError(f"{t}: the model has a camera encoder but the frame carries no batch_merged_cam_inputs")
            if int(ci["imgs"].shape[0]) != len(slots[t]):
                raise ValueError(f"{t}: {int(ci['imgs'].shape[0])} camera rigs for {len(slots[t])} agents")
            gkey = (t, str(dev), json.dumps(args[t]["cam"], sort_keys=True, default=str))
            if gkey not in geo:            # frustum / grid / depth bins of the type: weight-free, built once per camera configuration
                geo[gkey] = CameraGeometry(args[t]["cam"], dev)
            bev = TC.lss_encoder_train(P, sd, f"{TYPE_PREFIX[t]}.{mi}.", geo[gkey], ci, True if model is None else model.training)
            for j, s_ in enumerate(slots[t]):
                rows[s_] = bev[j:j + 1] if rows[s_] is None else TC.Mean2Fn.apply(bev[j:j + 1], rows[s_])
        canvas = torch.cat(rows, 0)
    nz = torch.zeros(1, dtype=torch.int64, device=dev)
    _lib.check(r.lib.av2x_count_nonzero(_P(canvas), canvas.numel(), _P(nz), r.stream()), "av2x_count_nonzero")
    return canvas, nz


def _forward_train(model, data_dict, topk=None, mask=None, trace=None):
    """-> output dict of the reference (psm / rm / obj require grad).  ``topk``: one K per sample instead of the random draw;
    ``mask`` (n, H, W): replay a recorded communication mask instead of the one computed here (the top-K cut is discontinuous
    in the single-agent logits; parity tests replay the reference's); ``trace``: dict that receives the computed mask."""
    args = model.args
    P = dict(model.named_parameters())
    sd = model.state_dict(keep_vars=True)
    dev = next(iter(P.values())).device
    if dev.type != "cuda":
        raise RuntimeError("Airv2xWhere2com (MI355X build) has no CPU path: move the module to the GPU (model.to('cuda'))")
    r = _runner(dev)
    # torch.autocast around the forward (tools/train.py:118) or model.amp = True -> bf16 matrix-core operands for this step's
    # convolutions, forward and data gradients alike (train_ops.AMP_STEP); a GradScaler on top works unchanged (fp32 gradients)
    mf = args["modality_fusion"]
    bb = mf["base_bev_backbone"]
    fcfg = args["where2com_fusion"]
    record_len, slots = frame_layout(args["collaborators"], data_dict)
    B, n = len(record_len), sum(record_len)
    if n == 0:
        raise ValueError("empty frame: no agent has lidar input")

    canvas, nz = encode_train(args, P, sd, data_dict, slots, n, dev, r, model)

    layer_nums, strides, ups = bb["layer_nums"], bb["layer_strides"], bb["upsample_strides"]
    if len(ups) != len(layer_nums) or any(u < 1 for u in ups):
        raise NotImplementedError("training: BaseBEVBackbone with down-sampling deblocks / the extra final deblock (eval mode runs them)")
    from ..synth import model_compression
    compression = model_compression(args)          # NaiveCompressor(256, args["compression"]) (airv2x_where2com.py:50-52)
    if not model.multi_scale:
        return _forward_train_single_scale(model, args, P, sd, dev, r, data_dict, canvas, nz, record_len, compression, topk, mask, trace)
    # ---- blocks[0] once, with the graph (the fusion pass's blocks[0] sees the same canvas): three identical updates
    y0 = _block(P, sd, 0, canvas, layer_nums[0], strides[0], 3)
    # ---- the two full backbone passes + shrink + cls_head: mask only, no gradient
    with torch.no_grad():
        feats = [y0.detach()]
        for i in (1, 2):
            feats.append(_block(P, sd, i, feats[-1], layer_nums[i], strides[i], 2))
        cat = torch.cat([_deblock(P, sd, i, feats[i], 2) for i in range(3)], -1)
        s = _shrink(P, mf["shrink_header"], cat)
        psm_single = T.conv_raw(s, P["cls_head.weight"], 1, 0, None, P["cls_head.bias"].detach(), 0)      # (n, H, W, A*C)
        if compression:     # airv2x_where2com.py:147-150: the compressor runs on the shrunk map in this branch too; its output is dead for
            from .train_cobevt import _compressor           # the multi-scale fusion (no gradient reaches it) but its BatchNorms see the batch
            _compressor(P, sd, s)
        H, W = psm_single.shape[1:3]
        if fcfg["fully"]:
            mask, com = None, torch.tensor(1, device=dev)
        else:
            if tuple(y0.shape[1:3]) != (H, W):
                raise NotImplementedError("mask/feature size mismatch (where2comm_fuse.py:230) is never taken by AirV2X configs")
            if topk is None:
                topk = [int(H * W * random.uniform(0, 1)) for _ in range(B)]      # where2comm_fuse.py:106, one draw per sample
            r.A, r.C = args["anchor_number"], args["num_class"]
            comm = fcfg["communication"]
            if "gaussian_smooth" in comm:
                gw = P["fusion_net.naive_communication.gaussian_filter.weight"].detach()
                r.gauss_w, r.gauss_b, r.gauss_k = gw.reshape(-1).contiguous(), P["fusion_net.naive_communication.gaussian_filter.bias"].detach(), int(gw.shape[-1])
            else:
                r.gauss_w, r.gauss_b, r.gauss_k = torch.ones(1, device=dev), torch.zeros(1, device=dev), 1
            r.threshold = float(comm["threshold"] or 0.0)
            r._frame += 1
            cmask, count, _, rl = r.comm_mask(psm_single, n, H, W, record_len, topk=topk)
            if trace is not None:
                trace["comm_mask"] = cmask.clone()
                trace["psm_single"] = psm_single
            mask = cmask.clone() if mask is None else mask.to(dev, torch.float32).reshape(n, H, W).contiguous()
            com = r.comm_rate(count, rl, B, H * W)

    # ---- fusion pass with the graph
    def fuse(x):
        outs, a0 = [], 0
        for k in record_len:
            outs.append(T.PixelAttn.apply(x[a0:a0 + k]))
            a0 += k
        return torch.stack(outs)

    x = T.MaskMul.apply(y0, mask) if mask is not None else y0
    ups_out = [_deblock(P, sd, 0, fuse(x), 1)]
    for i in (1, 2):
        x = _block(P, sd, i, x, layer_nums[i], strides[i], 1)
        ups_out.append(_deblock(P, sd, i, fuse(x), 1))
    fused = torch.cat(ups_out, -1)
    fs = _shrink(P, mf["shrink_header"], fused)
    names = ["cls_head", "reg_head"] + (["obj_head"] if args["obj_head"] else [])
    outs = _heads(P, names, fs)
    out = {"psm": outs[0], "rm": outs[1]}
    if args["obj_head"]:
        out["obj"] = outs[2]
    out.update({"mask": 0, "com": com, "comm_rate": int(nz[0].item()) if model.sync_comm_rate else nz[0]})
    return out


def _comm_mask_train(r, args, P, dev, psm_single, n, record_len, topk, mask, trace):
    """The training branch of Communication (where2comm_fuse.py:104-121) on the single-agent scores -> (mask (n, H, W) float, com)."""
    fcfg = args["where2com_fusion"]
    B = len(record_len)
    H, W = psm_single.shape[1:3]
    if topk is None:
        topk = [int(H * W * random.uniform(0, 1)) for _ in range(B)]      # where2comm_fuse.py:106, one draw per sample
    r.A, r.C = args["anchor_number"], args["num_class"]
    comm = fcfg["communication"]
    if "gaussian_smooth" in comm:
        gw = P["fusion_net.naive_communication.gaussian_filter.weight"].detach()
        r.gauss_w, r.gauss_b, r.gauss_k = gw.reshape(-1).contiguous(), P["fusion_net.naive_communication.gaussian_filter.bias"].detach(), int(gw.shape[-1])
    else:
        r.gauss_w, r.gauss_b, r.gauss_k = torch.ones(1, device=dev), torch.zeros(1, device=dev), 1
    r.threshold = float(comm["threshold"] or 0.0)
    r._frame += 1
    cmask, count, _, rl = r.comm_mask(psm_single, n, H, W, record_len, topk=topk)
    if trace is not None:
        trace["comm_mask"] = cmask.clone()
        trace["psm_single"] = psm_single
    m = cmask.clone() if mask is None else mask.to(dev, torch.float32).reshape(n, H, W).contiguous()
    return m, r.comm_rate(count, rl, B, H * W)


def _forward_train_single_scale(model, args, P, sd, dev, r, data_dict, canvas, nz, record_len, compression, topk, mask, trace):
    """``multi_scale: false`` in train mode (airv2x_where2com.py:117-179 with :163-166; where2comm_fuse.py:264-286): the backbone runs twice
    on the canvas (:119, :124 -- identical batch statistics, two running-statistic updates), the second pass carries the graph; shrink
    header -> single-agent scores (mask only) -> NaiveCompressor (if any, with the graph: it is live here) -> x mask -> one per-pixel
    attention per sample -> heads on the fused map."""
    mf = args["modality_fusion"]
    bb = mf["base_bev_backbone"]
    fcfg = args["where2com_fusion"]
    layer_nums, strides = bb["layer_nums"], bb["layer_strides"]
    B, n = len(record_len), sum(record_len)
    x, ups_out = canvas, []
    for i in range(len(layer_nums)):
        x = _block(P, sd, i, x, layer_nums[i], strides[i], 2)
        ups_out.append(_deblock(P, sd, i, x, 2))
    s = _shrink(P, mf["shrink_header"], torch.cat(ups_out, -1))
    with torch.no_grad():
        psm_single = T.conv_raw(s.detach(), P["cls_head.weight"], 1, 0, None, P["cls_head.bias"].detach(), 0)
    if compression:
        from .train_cobevt import _compressor
        s = _compressor(P, sd, s)
    if fcfg["fully"]:
        m, com = None, torch.tensor(1, device=dev)
    else:
        with torch.no_grad():
            m, com = _comm_mask_train(r, args, P, dev, psm_single, n, record_len, topk, mask, trace)
    xm = T.MaskMul.apply(s, m) if m is not None else s
    outs, a0 = [], 0
    for k in record_len:
        outs.append(T.PixelAttn.apply(xm[a0:a0 + k]))
        a0 += k
    fused = torch.stack(outs)
    names = ["cls_head", "reg_head"] + (["obj_head"] if args["obj_head"] else [])
    hs = _heads(P, names, fused)
    out = {"psm": hs[0], "rm": hs[1]}
    if args["obj_head"]:
        out["obj"] = hs[2]
    out.update({"mask": 0, "com": com, "comm_rate": int(nz[0].item()) if model.sync_comm_rate else nz[0]})
    return out


def forward_train(model, data_dict, topk=None, mask=None, trace=None):
    """One train-mode forward.  torch.autocast around the call (tools/train.py:118) or ``model.amp = True`` selects AMP for THIS
    step only: the flag lives for the duration of the forward (train_ops.amp_scope) and every node carries it into its backward."""
    from .airv2x_where2com import _amp_requested
    with T.amp_scope(_amp_requested(model)):
        return _forward_train(model, data_dict, topk=topk, mask=mask, trace=trace)
