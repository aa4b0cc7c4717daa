"""Train-mode forward of ``Airv2xV2VNet`` (models/airv2x_v2vnet.py:191-244 with ``self.training``): the graph the reference hands to
torch autograd (tools/train.py:220-247), built from HIP forward / backward ops.

    encoders, BaseBEVBackbone, DownsampleConv                                     train_ops (shared with the other models)
    V2VNetFusion.forward (v2vnet_modules/v2v_fuse.py:54-180), per sample, num_iteration times, for every node i:
        warp_affine_simple of every node into node i's frame (:142-146)           train_when2com.WarpAffineSimpleFn
        msg_cnn(cat[neighbour, node i]) * roi mask (:147-153)                     train_ops.conv_bias_act + MaskMul
        mean / max over the neighbours (:155-158)                                 AgentMeanFn / AgentMaxFn
        ConvGRU, one step, zero hidden state (convgru.py:52-73, :141-190):
            out = sigmoid(update gate) * tanh(candidate)                          two conv_bias_act + GruGateFn
            (the reset gate multiplies the zero state, and the state's input channels see zeros: their weights get the exact zero
             gradient the reference's autograd gives them -- the convolutions run on the 2C input channels that carry data)
        mlp on the ego node (:174-178)                                            train_fusion_ops.linear
    cls / reg / obj heads                                                         one 32-column GEMM
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from . import train_fusion_ops as F
from . import train_ops as T
from .autograd import _runner
from .engine import frame_layout
from .train_when2com import warp_affine_simple
from .train_where2com import _block, _deblock, _heads, _shrink, encode_train
from .when2com_engine import normalized_pairwise

_P = T._P


class GruGateFn(torch.autograd.Function):
    """out = sigmoid(beta) * tanh(cnm): h_next = (1 - update) * h_cur + update * cnm with h_cur = 0 (convgru.py:66-73)."""

    @staticmethod
    def forward(ctx, beta, cnm):
        T._check_dev(beta)
        r = _runner(beta.device)
        beta, cnm = beta.contiguous(), cnm.contiguous()
        out = torch.empty_like(cnm)
        _lib.check(r.lib.av2x_gru_gate(_P(beta), _P(cnm), beta.numel(), _P(out), r.stream()), "av2x_gru_gate")
        ctx.save_for_backward(beta, cnm)
        return out

    @staticmethod
    def backward(ctx, dout):
        beta, cnm = ctx.saved_tensors
        r = _runner(beta.device)
        dout = dout.contiguous()
        db, dc = torch.empty_like(beta), torch.empty_like(cnm)
        _lib.check(r.lib.av2x_gru_gate_backward(_P(beta), _P(cnm), _P(dout), beta.numel(), _P(db), _P(dc), r.stream()), "av2x_gru_gate_backward")
        return db, dc


class AgentMaxFn(torch.autograd.Function):
    """(n, H, W, C) -> (1, H, W, C): torch.max(message, dim=0)[0] (:157-158); the gradient goes to the first maximising agent."""

    @staticmethod
    def forward(ctx, x):
        T._check_dev(x)
        r = _runner(x.device)
        x = x.contiguous()
        y = torch.empty((1,) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
        idx = torch.empty(y.numel(), dtype=torch.uint8, device=x.device)
        _lib.check(r.lib.av2x_agent_argmax(_P(x), x.shape[0], y.numel(), _P(y), _P(idx), r.stream()), "av2x_agent_argmax")
        ctx.save_for_backward(idx)
        ctx.n = x.shape[0]
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        r = _runner(dy.device)
        dy = dy.contiguous()
        dx = torch.empty((ctx.n,) + tuple(dy.shape[1:]), dtype=torch.float32, device=dy.device)
        _lib.check(r.lib.av2x_agent_argmax_backward(_P(dy), _P(idx), ctx.n, dy.numel(), _P(dx), r.stream()), "av2x_agent_argmax_backward")
        return dx


def v2vnet_fusion(P, s, record_len, theta, cfg, prefix="fusion_net.", comm=None):
    """s (sum n, H, W, C) shrink-header maps -> (B, H, W, C); theta (B, L, L, 2, 3) numpy: the normalised pairwise matrices.
    ``comm``: a (1,) int64 device counter that receives the reference's comm_rates bookkeeping (v2v_fuse.py:138: the non-zeros of the
    sample's node features, appended once per node and iteration) -- one device reduction per (sample, iteration), no host read."""
    if cfg["conv_gru"]["num_layers"] != 1:
        raise NotImplementedError("one ConvGRU layer (every shipped v2vfusion block)")
    if cfg["agg_operator"] not in ("avg", "max"):
        raise NotImplementedError("agg_operator 'weight' needs the aggregation weights no AirV2X data loader provides")
    dev = s.device
    _, H, W, C = s.shape
    cell = prefix + "conv_gru.cell_list.0."
    # the data-carrying slices of the ConvGRU weights, taken ONCE per step (differentiable views: the rest of the tensors gets zeros)
    wg = P[cell + "conv_gates.weight"][C:2 * C, :2 * C].contiguous()       # update gate rows, (node, aggregate) input channels
    bg = P[cell + "conv_gates.bias"][C:2 * C].contiguous()
    wc = P[cell + "conv_can.weight"][:, :2 * C].contiguous()
    bc = P[cell + "conv_can.bias"]
    wm, bm = P[prefix + "msg_cnn.weight"], P[prefix + "msg_cnn.bias"]
    outs, a0 = [], 0
    for b, k in enumerate(record_len):
        nodes = s[a0:a0 + k]
        a0 += k
        ths = [torch.from_numpy(np.ascontiguousarray(theta[b, i, :k], dtype=np.float32)).to(dev) for i in range(k)]
        with torch.no_grad():       # roi_mask (:97-104): the warp of an all-ones map, per receiving node
            ones = torch.ones((k, H, W, 64), dtype=torch.float32, device=dev)
            rois = [warp_affine_simple(ones, th)[..., 0].contiguous() for th in ths]
        for _ in range(cfg["num_iteration"]):
            upd = []
            if comm is not None:
                nd = nodes.detach().contiguous()
                cnt = torch.zeros(1, dtype=torch.int64, device=dev)
                r_ = _runner(dev)
                _lib.check(r_.lib.av2x_count_nonzero(_P(nd), nd.numel(), _P(cnt), r_.stream()), "av2x_count_nonzero")
                comm.add_(cnt * k)
            for i in range(k):
                nb = warp_affine_simple(nodes, ths[i])
                ego = nodes[i:i + 1].expand(k, -1, -1, -1)
                msg = T.MaskMul.apply(T.conv_bias_act(torch.cat([nb, ego], -1), wm, bm, 1, 1, False), rois[i])
                agg = F.agent_mean(msg) if cfg["agg_operator"] == "avg" else AgentMaxFn.apply(msg)
                if cfg["gru_flag"]:
                    x = torch.cat([nodes[i:i + 1], agg], -1)
                    out = GruGateFn.apply(T.conv_bias_act(x, wg, bg, 1, 1, False), T.conv_bias_act(x, wc, bc, 1, 1, False))
                else:
                    out = nodes[i:i + 1] + agg
                upd.append(out)
            nodes = torch.cat(upd, 0) if k > 1 else upd[0]
        outs.append(nodes[0:1])
    fused = torch.cat(outs, 0) if len(outs) > 1 else outs[0]
    return F.linear(fused, P[prefix + "mlp.weight"], P[prefix + "mlp.bias"])


def _forward_train(model, data_dict):
    args = model.args
    P = dict(model.named_parameters())
    sd = model.state_dict(keep_vars=True)
    dev = next(iter(P.values())).device
    if dev.type != "cuda":
        raise RuntimeError("Airv2xV2VNet (MI355X build) has no CPU path: move the module to the GPU (model.to('cuda'))")
    r = _runner(dev)
    mf = args["modality_fusion"]
    bb, cfg = mf["base_bev_backbone"], args["v2vfusion"]
    from ..synth import model_compression
    compression = model_compression(args)          # NaiveCompressor(256, args["compression"]) behind the shrink header (airv2x_v2vnet.py:42-44)
    record_len, slots = frame_layout(args["collaborators"], data_dict)
    B, n = len(record_len), sum(record_len)
    if n == 0:
        raise ValueError("empty frame: no agent has lidar input")
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
    H, W = s.shape[1:3]
    pair = data_dict["img_pairwise_t_matrix_collab"]
    pair = pair.detach().cpu().numpy() if isinstance(pair, torch.Tensor) else np.asarray(pair)
    if pair.shape[0] != B:
        raise ValueError("img_pairwise_t_matrix_collab batch size does not match record_len")
    theta = normalized_pairwise(pair, H, W, cfg["voxel_size"][0], cfg["downsample_rate"])
    comm = torch.zeros(1, dtype=torch.int64, device=dev)
    fused = v2vnet_fusion(P, s, record_len, theta, cfg, comm=comm)
    names = ["cls_head", "reg_head"] + (["obj_head"] if args["obj_head"] else [])
    outs = _heads(P, names, fused)
    out = {"psm": outs[0], "rm": outs[1]}
    if args["obj_head"]:
        out["obj"] = outs[2]
    # comm_rates (:138): the non-zeros of every sample's node features, counted once per (iteration, node), / B (:172): a python float as
    # in the reference unless model.sync_comm_rate is False (then the device scalar: no host read-back in the step)
    out.update({"mask": 0, "comm_rate": (float(comm.item()) / B) if getattr(model, "sync_comm_rate", True) else comm[0].double() / B})
    return out


def forward_train(model, data_dict):
    """One train-mode forward.  torch.autocast around the call (tools/train.py:118) or ``model.amp = True`` selects AMP for THIS
    step's convolutions and the mlp (train_ops.amp_scope)."""
    from .airv2x_where2com import _amp_requested
    with T.amp_scope(_amp_requested(model)):
        return _forward_train(model, data_dict)
