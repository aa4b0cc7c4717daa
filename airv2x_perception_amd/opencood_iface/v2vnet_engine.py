"""Host-side driver of the V2VNet-LiDAR path (models/airv2x_v2vnet.py:191-244) on one MI355X.

Per-agent trunk = the Where2Comm engine's.  Fusion = V2VNetFusion (v2vnet_modules/v2v_fuse.py:54-180): for
``num_iteration`` rounds every node i receives the other nodes' maps warped into its frame (warp_affine_simple with the
normalised pairwise matrix), a 3x3 message convolution on [neighbour | self], the ROI-masked mean (or max) over the
neighbours and a ConvGRU update; node 0 of the last round goes through a Linear and the heads.

What the build does differently, with identical results up to fp32 summation order:
* msg_cnn is linear in its concatenated input: conv([warp_j | x_i]) = conv_a(warp_j) + conv_b(x_i) + bias, and the second
  term does not depend on j -> N + 1 convolutions of 256 -> 256 instead of N of 512 -> 256 per node;
* the ConvGRU is called with hidden_state=None on a one-step sequence (v2v_fuse.py:171-175; convgru.py:157-164), so its
  hidden state is exactly zero: the reset gate and the hidden-channel third of both convolutions multiply zeros.  Only
  update = sigmoid(conv_gates[C:2C, :2C]) and h = update * tanh(conv_can[:, :2C]) remain -- two 512 -> 256 convolutions
  (sigmoid / gated-tanh epilogues of conv_igemm) instead of 768 -> 512 and 768 -> 256;
* the output is node 0 of the last round (v2v_fuse.py:176-178), so the last round updates node 0 only (the other nodes'
  round-1 maps are still needed: they are node 0's neighbours).  ``comm_rate`` keeps the reference's bookkeeping (:139:
  the non-zero count of the CURRENT node maps, once per (round, node)).
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from .engine import ConvLayer, Where2ComEngine, _ptr
from .packing import pack_conv_weight
from .when2com_engine import normalized_pairwise


class V2VNetEngine(Where2ComEngine):
    # throughput mode keeps the latency-mode Winograd classes here: handing the 128- / 256-channel backbone layers to F(4x4,3x3) with frames in
    # flight measured SLOWER for this model (240 vs 249 / 134 vs 136 frames/s at 4 agents, profiles/r05u_wino4_threshold_x3.txt)
    WINO4_MIN_WGS_PER_IMAGE_T = Where2ComEngine.WINO4_MIN_WGS_PER_IMAGE
    def _init_config(self, args):
        mf = args["modality_fusion"]
        self.bb, self.sh = mf["base_bev_backbone"], mf["shrink_header"]
        self.fcfg = {"fully": False}
        from ..synth import model_compression
        self.compression = model_compression(args)     # airv2x_v2vnet.py:42-44, 180-181: NaiveCompressor(256, args["compression"]) behind the shrink header
        if self.compression and (256 % self.compression or (256 // self.compression) % 32):
            raise NotImplementedError(f"compression {self.compression}: 256/ratio must be a multiple of 32 channels")
        self.v2v = args["v2vfusion"]
        if self.v2v["conv_gru"]["num_layers"] != 1 or not self.v2v["gru_flag"]:
            raise NotImplementedError("one ConvGRU layer with gru_flag (every shipped v2vfusion block)")
        if self.v2v["agg_operator"] not in ("avg", "max"):
            raise NotImplementedError("agg_operator 'weight' needs a weight input the AirV2X forward never passes (airv2x_v2vnet.py:209)")

    FUSION_WEIGHTS = ("msg_a", "msg_b", "gru_u", "gru_c", "mlp_lin", "compressor")

    def _load_fusion(self, sd, up, prefix="fusion_net."):
        self.compressor = self._load_compressor(sd, up) if getattr(self, "compression", 0) else None
        c = self.v2v["in_channels"]
        wm, bm = sd[prefix + "msg_cnn.weight"].detach().float(), sd[prefix + "msg_cnn.bias"].detach().float()

        def conv(w, bias, act):
            wp, coutp = pack_conv_weight(w.contiguous())
            return ConvLayer(up(wp), None, up(bias), w.shape[1], w.shape[0], coutp, 3, 1, 1, act)

        self.msg_a = conv(wm[:, :c], torch.zeros(c), 0)                      # neighbour half, bias goes with the ego half
        self.msg_b = conv(wm[:, c:], bm, 0)
        g = prefix + "conv_gru.cell_list.0"
        wg, bg = sd[g + ".conv_gates.weight"].detach().float(), sd[g + ".conv_gates.bias"].detach().float()
        wc, bc = sd[g + ".conv_can.weight"].detach().float(), sd[g + ".conv_can.bias"].detach().float()
        self.gru_u = conv(wg[c:2 * c, :2 * c], bg[c:2 * c], 3)               # update gate (beta half), input channels x only
        self.gru_c = conv(wc[:, :2 * c], bc, 4)                              # candidate, tanh, gated by the update gate
        wl = sd[prefix + "mlp.weight"].detach().float()
        wp, coutp = pack_conv_weight(wl.view(c, c, 1, 1))
        self.mlp_lin = ConvLayer(up(wp), None, up(sd[prefix + "mlp.bias"].detach().float()), c, c, coutp, 1, 1, 0, 0)

    def _count(self, t, nz):
        _lib.check(self.lib.av2x_count_nonzero(_ptr(t), t.numel(), _ptr(nz), self.stream()), "av2x_count_nonzero")

    def fuse_sample(self, nodes, theta, n, H, W, trace=None):
        """One sample: nodes (n,H,W,C) -> fused (H,W,C) node 0 after num_iteration rounds, + the reference's comm counter."""
        C = self.v2v["in_channels"]
        iters = self.v2v["num_iteration"]
        op = 1 if self.v2v["agg_operator"] == "max" else 0
        th_all = torch.from_numpy(np.ascontiguousarray(theta[:n, :n], dtype=np.float32)).to(self.device)   # (n,n,2,3)
        nz = self.buf("v2v_nz", (1,), torch.int64)
        comm = self.buf("v2v_comm", (1,), torch.int64)
        _lib.check(self.lib.av2x_fill_zero(_ptr(comm), 8, self.stream()), "av2x_fill_zero")
        cur = nodes
        for it in range(iters):
            last = it == iters - 1
            upd = self.buf(f"v2v_nodes{it % 2}", (n, H, W, C))
            _lib.check(self.lib.av2x_fill_zero(_ptr(nz), 8, self.stream()), "av2x_fill_zero")
            self._count(cur, nz)
            comm.add_(nz * n)                                           # :139 appends the same count once per node
            for i in range(1 if last else n):
                warped = self.buf("v2v_warped", (n, H, W, C))
                _lib.check(self.lib.av2x_warp_affine_simple(_ptr(cur), _ptr(th_all[i]), _ptr(warped), n, H, W, C, self.stream()),
                           "av2x_warp_affine_simple")
                ma = self.buf("v2v_msg_a", (n, H, W, C))
                self.conv(self.msg_a, warped, n, H, W, ma)
                mb = self.buf("v2v_msg_b", (1, H, W, C))
                self.conv(self.msg_b, cur[i:i + 1], 1, H, W, mb)
                xcat = self.buf("v2v_xcat", (1, H, W, 2 * C))
                agg = self.buf("v2v_agg", (H, W, C))
                _lib.check(self.lib.av2x_v2v_aggregate(_ptr(ma), _ptr(mb), _ptr(th_all[i]), n, H, W, C, op, _ptr(agg), self.stream()),
                           "av2x_v2v_aggregate")
                xcat[0, :, :, :C].copy_(cur[i])                         # [x_i | agg]: data movement only
                xcat[0, :, :, C:].copy_(agg)
                gate = self.buf("v2v_gate", (1, H, W, C))
                self.conv(self.gru_u, xcat, 1, H, W, gate)
                self.conv(self.gru_c, xcat, 1, H, W, upd[i:i + 1], residual=gate)
                if trace is not None and i == 0:
                    trace[f"agg_it{it}"] = agg.permute(2, 0, 1).clone()
                    trace[f"node0_it{it}"] = upd[0].permute(2, 0, 1).clone()
            cur = upd
        fused = self.buf("v2v_fused", (1, H, W, C))
        self.conv(self.mlp_lin, cur[0:1], 1, H, W, fused)
        return fused, comm

    def _heads_out(self, fused, B, H, W):
        heads = torch.empty((B, self.heads.cout, H, W), dtype=torch.float32, device=self.device)
        self.conv(self.heads, fused, B, H, W, heads)
        outs = torch.split(heads, self.head_splits, dim=1)
        if B > 1:
            outs = [o.contiguous() for o in outs]
        out = {"psm": outs[0], "rm": outs[1]}
        if self.args["obj_head"]:
            out["obj"] = outs[2]
        return out

    @torch.no_grad()
    def forward(self, data_dict, trace=None, sync_comm_rate=False):
        if not self.weights_ready:
            raise RuntimeError("load_state_dict() must be called before forward()")
        record_len, slots = self.frame_layout(data_dict)
        B, n_total = len(record_len), sum(record_len)
        canvas, ny, nx = self.encode(data_dict, record_len, slots)
        dims = self.level_dims(ny, nx)
        H, W = self.cat_hw(dims)
        C = self.feat_c
        if C != self.v2v["in_channels"] or (H, W) != (self.v2v["conv_gru"]["H"], self.v2v["conv_gru"]["W"]):
            raise ValueError(f"v2vfusion is configured for {self.v2v['in_channels']}x{self.v2v['conv_gru']['H']}x"
                             f"{self.v2v['conv_gru']['W']}, the trunk delivers {C}x{H}x{W}")
        s_all = self.buf("v2v_shrink", (n_total, H, W, C))
        self.trunk(canvas, n_total, ny, nx, shrink_out=s_all)
        if self.compression:                                   # airv2x_v2vnet.py:180-181
            self.run_compressor(s_all, n_total, H, W)
        pair = data_dict["img_pairwise_t_matrix_collab"]
        pair = pair.detach().cpu().numpy() if isinstance(pair, torch.Tensor) else np.asarray(pair)
        if pair.shape[0] != B:
            raise ValueError("img_pairwise_t_matrix_collab batch size does not match record_len")
        theta = normalized_pairwise(pair, H, W, self.v2v["voxel_size"][0], self.v2v["downsample_rate"])
        fused_all = self.buf("v2v_fused_all", (B, H, W, C))
        total = torch.zeros(1, dtype=torch.int64, device=self.device)
        off = 0
        for b, n in enumerate(record_len):
            fused, comm = self.fuse_sample(s_all[off:off + n], theta[b], n, H, W, trace if b == 0 else None)
            fused_all[b].copy_(fused[0])
            total += comm
            off += n
        if trace is not None:
            trace["shrink"] = s_all.permute(0, 3, 1, 2).clone()
            trace["fused"] = fused_all.permute(0, 3, 1, 2).clone()
        out = self._heads_out(fused_all, B, H, W)
        rate = float(total.item()) / B if sync_comm_rate else (total[0].clone() if B == 1 else total[0].double() / B)
        out.update({"mask": 0, "comm_rate": rate})
        return out
