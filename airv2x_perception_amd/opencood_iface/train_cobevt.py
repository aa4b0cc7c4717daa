"""Train-mode forward of ``Airv2xCoBEVT`` (models/airv2x_cobevt.py:112-156 with ``self.training``): the graph the reference hands to
torch autograd (tools/train.py:220-247), built from HIP forward / backward ops.

    encoders (PillarVFE with BatchNorm1d batch statistics + scatter)              train_ops.pillar_encode
    BaseBEVBackbone once (Conv / ConvTranspose + BatchNorm batch statistics)      train_ops.conv_bn_act / deconv_bn_act
    DownsampleConv                                                                train_ops.conv_bias_act
    [NaiveCompressor: Conv3x3 + bias + BatchNorm + ReLU x 3, naive_compress.py:12-36]
    regroup: zero-pad the agent axis to max_cav_num (fuse_utils.py:13-64)
    SwapFusionEncoder (swap_fusion_modules.py:233-280): depth x {window, grid} x
        {PreNormResidual(Attention), PreNormResidual(FeedForward)}                train_fusion_ops: LayerNorm, Linear(+residual), fused-axial
                                                                                  attention, GELU, Dropout -- each with its HIP backward
        mean over agents -> LayerNorm -> Linear (mlp_head)
    cls / reg / obj heads                                                         one 32-column GEMM

Every BatchNorm's running statistics are updated once per step (the reference runs the backbone once here).
"""
from __future__ import annotations

import torch

from . import train_fusion_ops as F
from . import train_ops as T
from .autograd import _runner
from .engine import frame_layout
from .train_where2com import _block, _deblock, _heads, _running, _shrink, encode_train


def _compressor(P, sd, x, prefix="naive_compressor."):
    """NaiveCompressor.forward (naive_compress.py:33-36): encoder, decoder = Conv3x3 (+ bias) + BatchNorm(eps 1e-3) + ReLU.  A bias
    in front of a batch-statistics BatchNorm cancels in the forward; its gradient is exactly zero."""
    for conv, bn in (("encoder.0", "encoder.1"), ("decoder.0", "decoder.1"), ("decoder.3", "decoder.4")):
        x = T.conv_bn_act(x, P[f"{prefix}{conv}.weight"], P[f"{prefix}{bn}.weight"], P[f"{prefix}{bn}.bias"], 1, 1,
                          running=_running(sd, prefix + bn, 1))
        # ... but the batch MEAN the BatchNorm tracks is that of conv(x) + bias: the running mean moves by momentum x bias on top
        with torch.no_grad():
            sd[f"{prefix}{bn}.running_mean"].add_(P[f"{prefix}{conv}.bias"].detach(), alpha=T.BN_MOMENTUM)
    return x


def swap_fusion_encoder(P, x, n_valid, fax, training=True, prefix="fusion_net."):
    """x (L, H, W, C) tokens of ONE sample (agents beyond n_valid are zero padding) -> (1, H, W, C)."""
    L, H, W, C = x.shape
    ws, dh, p_drop = fax["window_size"], fax["dim_head"], float(fax.get("drop_out", 0.0))
    heads = C // dh
    if H % ws or W % ws:
        raise ValueError(f"BEV map {H}x{W} is not divisible by the window size {ws}")
    for i in range(fax["depth"]):
        for gi, part in enumerate(("window", "grid")):
            a, f = f"{prefix}layers.{i}.{part}_attention", f"{prefix}layers.{i}.{part}_ffd"
            xn = F.layer_norm(x, P[a + ".norm.weight"], P[a + ".norm.bias"])
            qkv = F.linear(xn, P[a + ".fn.to_qkv.weight"])
            att = F.fax_attention(qkv, P[a + ".fn.relative_position_bias_table.weight"], n_valid, ws, heads, dh, gi)
            if p_drop > 0 and training:     # to_out = Sequential(Linear, Dropout) (:42-44): the residual is added after the dropout
                x = F.dropout(F.linear(att, P[a + ".fn.to_out.0.weight"]), p_drop, True, x)
            else:
                x = F.linear(att, P[a + ".fn.to_out.0.weight"], None, x)
            xn = F.layer_norm(x, P[f + ".norm.weight"], P[f + ".norm.bias"])
            hdn = F.gelu(F.linear(xn, P[f + ".fn.net.0.weight"], P[f + ".fn.net.0.bias"]))
            if p_drop > 0 and training:     # FeedForward = Linear, GELU, Dropout, Linear, Dropout (base_transformer.py:28-35)
                x = F.dropout(F.linear(F.dropout(hdn, p_drop), P[f + ".fn.net.3.weight"], P[f + ".fn.net.3.bias"]), p_drop, True, x)
            else:
                x = F.linear(hdn, P[f + ".fn.net.3.weight"], P[f + ".fn.net.3.bias"], x)
    m = F.agent_mean(x)
    m = F.layer_norm(m, P[prefix + "mlp_head.2.weight"], P[prefix + "mlp_head.2.bias"])
    return F.linear(m, P[prefix + "mlp_head.3.weight"], P[prefix + "mlp_head.3.bias"])


def _forward_train(model, data_dict):
    args = model.args
    P = dict(model.named_parameters())
    sd = model.state_dict(keep_vars=True)
    dev = next(iter(P.values())).device
    if dev.type != "cuda":
        raise RuntimeError("Airv2xCoBEVT (MI355X build) has no CPU path: move the module to the GPU (model.to('cuda'))")
    r = _runner(dev)
    bb, fax = args["base_bev_backbone"], args["fax_fusion"]
    record_len, slots = frame_layout(args["collaborators"], data_dict)
    B, n = len(record_len), sum(record_len)
    if n == 0:
        raise ValueError("empty frame: no agent has lidar input")
    L = int(sum(args["max_cav"].values()))
    if max(record_len) > L:
        raise ValueError(f"{max(record_len)} agents in a sample exceed max_cav_num = {L}")
    canvas, nz = encode_train(args, P, sd, data_dict, slots, n, dev, r)
    feats, x = [], canvas
    for i, (ln, st) in enumerate(zip(bb["layer_nums"], bb["layer_strides"])):
        x = _block(P, sd, i, x, ln, st, 1)
        feats.append(x)
    s = torch.cat([_deblock(P, sd, i, f, 1) for i, f in enumerate(feats)], -1)
    s = _shrink(P, args["shrink_header"], s)
    if int(args.get("compression", 0) or 0) > 0:
        # NaiveCompressor (naive_compress.py:10-44): Conv3x3 + BN + ReLU down to C / ratio, two back up.  Its convolutions carry a bias in
        # front of their BatchNorm: it cancels in the normalised output (its gradient is exactly zero -- the parameter keeps grad None)
        # but is part of the batch mean nn.BatchNorm folds into running_mean.
        for name in ("encoder.0", "decoder.0", "decoder.3"):
            cv = "naive_compressor." + name
            bn = cv[:-1] + str(int(cv[-1]) + 1)
            st = []
            s = T.conv_bn_act(s, P[cv + ".weight"], P[bn + ".weight"], P[bn + ".bias"], 1, 1, stats_out=st)
            mean, var, count = st[0]
            T.update_running_stats(sd[bn + ".running_mean"], sd[bn + ".running_var"], sd.get(bn + ".num_batches_tracked"),
                                   (mean + P[cv + ".bias"].detach(), var, count), 1)
    fused, a0 = [], 0
    for k in record_len:                    # regroup (fuse_utils.py:13-64): zero-pad every sample to L agents, fuse per sample
        xs = s[a0:a0 + k]
        a0 += k
        if k < L:
            xs = torch.cat([xs, xs.new_zeros((L - k,) + tuple(xs.shape[1:]))], 0)
        fused.append(swap_fusion_encoder(P, xs, k, fax, model.training))
    fused = torch.cat(fused, 0) if B > 1 else fused[0]
    names = ["cls_head", "reg_head"] + (["obj_head"] if args["obj_head"] else [])
    outs = _heads(P, names, fused)
    out = {"psm": outs[0], "rm": outs[1]}
    if args["obj_head"]:
        out["obj"] = outs[2]
    return out


def forward_train(model, data_dict):
    """One train-mode forward.  torch.autocast around the call (tools/train.py:118) or ``model.amp = True`` selects AMP for THIS
    step only: the flag lives for the duration of the forward (train_ops.amp_scope) and every node carries it into its backward."""
    from .airv2x_where2com import _amp_requested
    with T.amp_scope(_amp_requested(model)):
        return _forward_train(model, data_dict)
