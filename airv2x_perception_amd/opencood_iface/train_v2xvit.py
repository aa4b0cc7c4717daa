"""Train-mode forward of ``Airv2xV2XVit`` (models/airv2x_v2xvit.py:108-167 with ``self.training``): the graph the reference hands to torch
autograd, built from HIP forward / backward ops.

    encoders, BaseBEVBackbone once, DownsampleConv                                 train_ops (as train_cobevt.py)
    regroup (only the real agents are kept: padded agents are masked as keys of every cross-agent attention and V2XTransformer returns
             agent 0, so they reach neither the output nor any gradient)
    RTE: x[i] += lin(emb[dt_i * ratio])                                            tensor algebra on C-vectors + av2x_add_agent_vector
    STTF: warp_affine of agents 1.. into the ego frame                            WarpAffineFn (adjoint scatter in the backward)
    depth x { num_blocks x { PreNorm(HGTCavAttention) + x, PreNorm(PyramidWindowAttention, SplitAttn) + x }, PreNorm(FeedForward) + x }
        HGT: per-type q / k / v Linears with relation_att / relation_msg FOLDED into the weights -- the fold itself (hmsa.py:84-104:
             33 k-element matrix products per relation) is differentiable tensor algebra on the weights, so the relation matrices and the
             Linears get their gradients through it; the per-pixel attention and everything that touches a map run in HIP kernels
    heads on the ego's map                                                          one 32-column GEMM
"""
from __future__ import annotations

import numpy as np
import torch

from . import train_fusion_ops as F
from . import train_ops as T
from . import warp as warp_host
from .autograd import _P, _runner
from .engine import frame_layout
from .train_where2com import _block, _deblock, _heads, _shrink, encode_train
from .. import _lib


def _groups(types):
    out, s = [], 0
    for i in range(1, len(types) + 1):
        if i == len(types) or types[i] != types[s]:
            out.append((s, i, types[s]))
            s = i
    return out


def folded_projection(P, h, t, heads, dh):
    """Weights / bias of the projection an agent of node type ``t`` needs: rows [q'(t->0) | q'(t->1) | k | v'(0<-t) | v'(1<-t)]
    (v2xvit_engine.py folds the same way for inference, in fp64; here in differentiable fp32 tensor algebra)."""
    ratt, rmsg = P[h + ".relation_att"], P[h + ".relation_msg"]
    Wq, bq = P[f"{h}.q_linears.{t}.weight"], P[f"{h}.q_linears.{t}.bias"]
    Wk, bk = P[f"{h}.k_linears.{t}.weight"], P[f"{h}.k_linears.{t}.bias"]
    Wv, bv = P[f"{h}.v_linears.{t}.weight"], P[f"{h}.v_linears.{t}.bias"]
    C = Wq.shape[1]
    ws, bs = [], []
    for tj in range(2):      # q' = w_att[e = t*2 + tj]^T q per head (einsum 'i p, p q, j q', hmsa.py:139-141)
        e = t * 2 + tj
        ws.append(torch.einsum("mpq,mpc->mqc", ratt[e], Wq.view(heads, dh, C)).reshape(heads * dh, C))
        bs.append(torch.einsum("mpq,mp->mq", ratt[e], bq.view(heads, dh)).reshape(-1))
    ws.append(Wk)
    bs.append(bk)
    for ti in range(2):      # v' = w_msg[e = ti*2 + t]^T v (einsum 'i j p c, j p', :150)
        e = ti * 2 + t
        ws.append(torch.einsum("mpc,mpk->mck", rmsg[e], Wv.view(heads, dh, C)).reshape(heads * dh, C))
        bs.append(torch.einsum("mpc,mp->mc", rmsg[e], bv.view(heads, dh)).reshape(-1))
    return torch.cat(ws, 0), torch.cat(bs, 0)


def encoder(P, x, prior, scm, enc, training=True, p="fusion_net.encoder"):
    """V2XTEncoder.forward (v2xvit_basic.py:174-200) + V2XTransformer's ``output[:, 0]`` on the n real agents: x (n, H, W, C) -> (1, H, W, C)."""
    n, H, W, C = x.shape
    dev = x.device
    cav, pw = enc["cav_att_config"], enc["pwindow_att_config"]
    if not cav["use_hetero"] or pw["fusion_method"] != "split_attn" or not pw["relative_pos_embedding"]:
        raise NotImplementedError("only the shipped V2X-ViT configuration (hetero attention, split_attn, relative pos)")
    types = [int(prior[i, 2]) for i in range(n)]
    dts = [int(prior[i, 1]) for i in range(n)]
    r = _runner(dev)
    if cav["use_RTE"]:
        rows = P[p + ".rte.emb.emb.weight"][[dt * cav["RTE_ratio"] for dt in dts]]
        vec = torch.nn.functional.linear(rows, P[p + ".rte.emb.lin.weight"], P[p + ".rte.emb.lin.bias"])      # (n, C) vectors
        x = F.add_agent_vector(x, vec)
    d = warp_host.discretized_matrix(np.asarray(scm)[:n], enc["sttf"]["voxel_size"][0], enc["sttf"]["downsample_rate"])
    theta = torch.from_numpy(warp_host.affine_theta(warp_host.transformation_matrix(d, (H, W)), (H, W), (H, W))).to(dev)
    if n > 1:
        x = torch.cat([x[0:1], F.warp_affine(x[1:], theta[1:].contiguous())], 0)
    mask = torch.empty((n, H, W), dtype=torch.float32, device=dev)
    if enc["use_roi_mask"]:
        ones = torch.ones(n, dtype=torch.int32, device=dev)
        _lib.check(r.lib.av2x_roi_mask(_P(theta), _P(ones), _P(mask), n, H, W, r.stream()), "av2x_roi_mask")
    else:
        mask.fill_(1.0)
    heads, dh = cav["heads"], cav["dim_head"]
    wcfg = list(zip(pw["heads"], pw["dim_head"], pw["window_size"]))
    groups = _groups(types)
    for di in range(enc["depth"]):
        for nb in range(enc["num_blocks"]):
            q = f"{p}.layers.{di}.0.layers.{nb}"
            h = q + ".0.fn"
            # ---- x = HGT(LN(x)) + x
            xn = F.layer_norm(x, P[q + ".0.norm.weight"], P[q + ".0.norm.bias"])
            fold = {t: folded_projection(P, h, t, heads, dh) for t in set(types)}
            proj = torch.cat([F.linear(xn[a:b], fold[t][0], fold[t][1]) for (a, b, t) in groups], 0)
            att = F.hgt_attention(proj, mask, types, heads, dh)
            outs = []
            for (a, b, t) in groups:
                if cav["dropout"] > 0 and training:     # out = drop_out(to_out(out)) (hmsa.py:153-154), residual added by PreNorm's caller
                    outs.append(F.dropout(F.linear(att[a:b], P[f"{h}.a_linears.{t}.weight"], P[f"{h}.a_linears.{t}.bias"]), cav["dropout"], True, x[a:b]))
                else:
                    outs.append(F.linear(att[a:b], P[f"{h}.a_linears.{t}.weight"], P[f"{h}.a_linears.{t}.bias"], x[a:b]))
            x = torch.cat(outs, 0) if len(outs) > 1 else outs[0]
            # ---- x = SplitAttn(window attentions(LN(x))) + x
            w = q + ".1.fn"
            xn = F.layer_norm(x, P[q + ".1.norm.weight"], P[q + ".1.norm.bias"])
            qkv3 = F.linear(xn, torch.cat([P[f"{w}.pwmsa.{i}.to_qkv.weight"] for i in range(3)], 0))
            wat = F.pyramid_window_attention(qkv3, [P[f"{w}.pwmsa.{i}.pos_embedding"] for i in range(3)], wcfg)
            br = []
            for i in range(3):
                o = F.linear(wat[i], P[f"{w}.pwmsa.{i}.to_out.0.weight"], P[f"{w}.pwmsa.{i}.to_out.0.bias"])
                br.append(F.dropout(o, pw["dropout"], training))      # to_out = Sequential(Linear, Dropout) (mswin.py:47)
            sa = w + ".split_attn"
            x = F.split_attn(br[0], br[1], br[2], x, P[sa + ".fc1.weight"], P[sa + ".bn1.weight"], P[sa + ".bn1.bias"], P[sa + ".fc2.weight"])
        f = f"{p}.layers.{di}.1"
        pdrop = enc["feed_forward"]["dropout"]
        xn = F.layer_norm(x, P[f + ".norm.weight"], P[f + ".norm.bias"])
        hdn = F.gelu(F.linear(xn, P[f + ".fn.net.0.weight"], P[f + ".fn.net.0.bias"]))
        if pdrop > 0 and training:
            x = F.dropout(F.linear(F.dropout(hdn, pdrop), P[f + ".fn.net.3.weight"], P[f + ".fn.net.3.bias"]), pdrop, True, x)
        else:
            x = F.linear(hdn, P[f + ".fn.net.3.weight"], P[f + ".fn.net.3.bias"], x)
    return x[0:1]


def _forward_train(model, data_dict):
    args = model.args
    P = dict(model.named_parameters())
    sd = model.state_dict(keep_vars=True)
    dev = next(iter(P.values())).device
    if dev.type != "cuda":
        raise RuntimeError("Airv2xV2XVit (MI355X build) has no CPU path: move the module to the GPU (model.to('cuda'))")
    r = _runner(dev)
    mf = args["modality_fusion"]
    bb = mf["base_bev_backbone"]
    from ..synth import model_compression
    compression = model_compression(args)          # NaiveCompressor(256, args["compression"]) behind the shrink header, as the reference reads it
    record_len, slots = frame_layout(args["collaborators"], data_dict)
    B, n = len(record_len), sum(record_len)
    if n == 0:
        raise ValueError("empty frame: no agent has lidar input")
    if max(record_len) > int(args["max_cav_num"]):
        raise ValueError(f"{max(record_len)} agents in a sample exceed max_cav_num = {args['max_cav_num']}")
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
    prior = data_dict["prior_encoding"].detach().cpu().numpy()
    scm = data_dict["spatial_correction_matrix"].detach().cpu().numpy()
    fused, a0 = [], 0
    for b, k in enumerate(record_len):
        fused.append(encoder(P, s[a0:a0 + k], prior[b], scm[b], args["transformer"]["encoder"], model.training))
        a0 += k
    fused = torch.cat(fused, 0) if B > 1 else fused[0]
    names = ["cls_head", "reg_head"] + (["obj_head"] if args["obj_head"] else [])
    outs = _heads(P, names, fused)
    out = {"psm": outs[0], "rm": outs[1]}
    if args["obj_head"]:
        out["obj"] = outs[2]
    out["comm_rate"] = int(nz[0].item()) if getattr(model, "sync_comm_rate", True) else nz[0]
    return out


def forward_train(model, data_dict):
    """One train-mode forward.  torch.autocast around the call (tools/train.py:118) or ``model.amp = True`` selects AMP for THIS
    step only: the flag lives for the duration of the forward (train_ops.amp_scope) and every node carries it into its backward."""
    from .airv2x_where2com import _amp_requested
    with T.amp_scope(_amp_requested(model)):
        return _forward_train(model, data_dict)
