"""Host-side driver of the When2com-LiDAR path (models/airv2x_when2com.py:112-151) on one MI355X.

Per-agent trunk = the Where2Comm engine's.  Fusion = When2comFusion (when2com_modules/when2com.py:60-134, shipped
mode 'softmax'): every agent's shrink output is warped into the ego frame (warp_affine_simple with the normalised
pairwise matrix), the five-layer policy network (Conv3x3 + BN + ReLU, 190 GFLOP per agent at the default grid -- twice
the backbone) runs on conv_igemm, the two km_generator MLPs stream their 577 MB first-layer weights once per frame
through av2x_linear_rows, and the ego's query attends the agents' keys; the fused map is the attention-weighted sum of
the warped maps.

Mode 'activated' is not built: the reference's activated_select (:45-58) raises IndexError for the single-query layout
its own forward builds, so there is no behaviour to reproduce.
"""
from __future__ import annotations

from ctypes import c_void_p

import numpy as np
import torch

from .. import _lib
from .engine import ConvLayer, Where2ComEngine, _ptr
from .packing import fold_bn, pack_conv_weight

POLICY_BN_EPS = 1e-5    # nn.BatchNorm2d default (conv2DBatchNormRelu, when2com.py:160-163); the backbone uses 1e-3
POLICY_STRIDES = (1, 1, 2, 1, 2)


def normalized_pairwise(pairwise, H, W, discrete_ratio, downsample_rate):
    """when2com.py:86-104 in fp32 like the reference: (B,L,L,4,4) -> (B,L,L,2,3) thetas of F.affine_grid."""
    m = np.asarray(pairwise, dtype=np.float32)[:, :, :, [0, 1], :][:, :, :, :, [0, 1, 3]].copy()
    m[..., 0, 1] = m[..., 0, 1] * np.float32(H) / np.float32(W)
    m[..., 1, 0] = m[..., 1, 0] * np.float32(W) / np.float32(H)
    m[..., 0, 2] = m[..., 0, 2] / np.float32(downsample_rate * discrete_ratio * W) * np.float32(2)
    m[..., 1, 2] = m[..., 1, 2] / np.float32(downsample_rate * discrete_ratio * H) * np.float32(2)
    return m


class When2comEngine(Where2ComEngine):
    # throughput mode keeps the latency-mode Winograd classes here: handing the 128- / 256-channel backbone layers to F(4x4,3x3) with frames in
    # flight measured SLOWER for this model (240 vs 249 / 134 vs 136 frames/s at 4 agents, profiles/r05u_wino4_threshold_x3.txt)
    WINO4_MIN_WGS_PER_IMAGE_T = Where2ComEngine.WINO4_MIN_WGS_PER_IMAGE
    def _init_config(self, args):
        mf = args["modality_fusion"]
        self.bb, self.sh = mf["base_bev_backbone"], mf["shrink_header"]
        self.fcfg = {"fully": False}
        from ..synth import model_compression
        self.compression = model_compression(args)     # airv2x_when2com.py:50-52: NaiveCompressor(256, args["compression"]) in front of the fusion
        if self.compression and (256 % self.compression or (256 // self.compression) % 32):
            raise NotImplementedError(f"compression {self.compression}: 256/ratio must be a multiple of 32 channels")
        self.w2 = args["when2com_fusion"]
        if self.w2["mode"] != "softmax":
            raise NotImplementedError("When2com mode %r: only the shipped 'softmax' mode is built (the reference's "
                                      "'activated' branch raises IndexError, when2com.py:58)" % (self.w2["mode"],))

    FUSION_WEIGHTS = ("policy", "key_fc", "query_fc", "att_lin", "compressor")

    def _load_fusion(self, sd, up, prefix="fusion_net."):
        self.compressor = self._load_compressor(sd, up) if getattr(self, "compression", 0) else None
        self.policy = []
        for i, stride in enumerate(POLICY_STRIDES, 1):
            p = f"{prefix}query_key_net.conv{i}.cbr_unit"
            w = sd[p + ".0.weight"].detach().float()
            sc, sh = fold_bn(sd, p + ".1", eps=POLICY_BN_EPS)
            sh = sh + sd[p + ".0.bias"].detach().float().cpu() * sc             # conv bias goes through the BN scale
            wp, coutp = pack_conv_weight(w)
            self.policy.append(ConvLayer(up(wp), up(sc), up(sh), w.shape[1], w.shape[0], coutp, 3, stride, 1, 1))
        h5, w5 = self.w2["H"] // 4, self.w2["W"] // 4
        cq = self.policy[-1].cout

        def fc(net):
            layers = []
            for li, act in ((0, 1), (2, 1), (4, 0)):
                w = sd[f"{prefix}{net}.fc.{li}.weight"].detach().float()
                if li == 0:
                    # the reference flattens NCHW (c, y, x); the policy output here is NHWC (y, x, c): permute the columns once
                    if w.shape[1] != cq * h5 * w5:
                        raise ValueError(f"{net}.fc.0 expects {w.shape[1]} inputs, the policy map has {cq}x{h5}x{w5}")
                    w = w.view(w.shape[0], cq, h5, w5).permute(0, 2, 3, 1).reshape(w.shape[0], -1)
                layers.append((up(w), up(sd[f"{prefix}{net}.fc.{li}.bias"].detach().float()), act))
            return layers

        self.key_fc, self.query_fc = fc("key_net"), fc("query_net")
        self.att_lin = (up(sd[prefix + "attention_net.linear.weight"].detach().float()),
                        up(sd[prefix + "attention_net.linear.bias"].detach().float()), 0)

    # ------------------------------------------------------------------ kernels
    def linear_rows(self, x, m, layer, tag):
        w, b, act = layer
        n, k = w.shape
        need = int(self.lib.av2x_linear_rows_workspace_bytes(m, n, k))
        ws = self.buf("linrows_ws", (max(need // 4, 1),))
        y = self.buf(f"linrows_{tag}", (m, n))
        _lib.check(self.lib.av2x_linear_rows(_ptr(x), _ptr(w), _ptr(b), m, n, k, act, _ptr(y), _ptr(ws), ws.numel() * 4,
                                             self.stream()), "av2x_linear_rows")
        return y

    def mlp(self, x, m, layers, tag):
        for i, layer in enumerate(layers):
            x = self.linear_rows(x, m, layer, f"{tag}{i}")
        return x

    def policy_keys(self, warped, n, H, W, tag=""):
        """policy_net4 + key_net on n warped maps -> (policy map (n,h,w,256), keys (n,key_size))."""
        cur, h, w = warped, H, W
        for i, L in enumerate(self.policy):
            ho, wo = (h + 2 - 3) // L.stride + 1, (w + 2 - 3) // L.stride + 1
            out = self.buf(f"policy{i}{tag}", (n, ho, wo, L.cout))
            self.conv(L, cur, n, h, w, out)
            cur, h, w = out, ho, wo
        if (h, w) != (self.w2["H"] // 4, self.w2["W"] // 4):
            raise ValueError(f"policy map is {h}x{w}, when2com_fusion.H/W promise {self.w2['H'] // 4}x{self.w2['W'] // 4}")
        keys = self.mlp(cur.view(n, -1), n, self.key_fc, "key" + tag)
        return cur, keys

    def query_of(self, qk_ego):
        """query_net on the ego's policy map, then attention_net.linear: (1, key_size)."""
        q = self.mlp(qk_ego.reshape(1, -1), 1, self.query_fc, "query")
        return self.linear_rows(q, 1, self.att_lin, "att")

    def fuse(self, keys, q, maps, out, coef=None):
        n = len(maps)
        arr = (c_void_p * n)(*[t.data_ptr() for t in maps])
        _lib.check(self.lib.av2x_when2com_fuse(_ptr(keys), _ptr(q), n, keys.shape[1], arr, maps[0].numel(), _ptr(out),
                                               _ptr(coef), self.stream()), "av2x_when2com_fuse")

    def warp(self, x, theta, n, H, W, C, out=None):
        warped = out if out is not None else self.buf("w2_warped", (n, H, W, C))
        th = torch.from_numpy(np.ascontiguousarray(theta, dtype=np.float32)).to(self.device)
        _lib.check(self.lib.av2x_warp_affine_simple(_ptr(x), _ptr(th), _ptr(warped), n, H, W, C, self.stream()),
                   "av2x_warp_affine_simple")
        return warped

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

    # ------------------------------------------------------------------ agent sharding
    @torch.no_grad()
    def shard_local_stage(self, data_dict_local, has_ego, n_pad=None):
        """Per-rank half: trunk, warp into the ego frame, policy network and key MLP for THIS rank's agents -- i.e. all
        of the per-agent work, 290 of the frame's ~300 GFLOP per agent.  Send buffer = [n_loc warped maps (36.0 MB
        each at the default grid) | n_loc keys (1 KB each) | the ego's projected query (1 KB; zeros on the other
        ranks)], so that every rank can finish the frame (SPMD).  ``data_dict_local`` carries the frame-level
        ``img_pairwise_t_matrix_collab`` and ``shard_rank`` (set by ShardedFrame): global agent index = rank * n_loc + j."""
        n, record_len, slots = self.shard_frame_agents(data_dict_local)
        n_pad = n if n_pad is None else int(n_pad)
        if n_pad < max(n, 1):
            raise ValueError(f"n_pad = {n_pad} is smaller than this rank's {n} agents")
        if n > 0:
            canvas, ny, nx = self.encode(data_dict_local, record_len, slots)
        else:
            ny, nx = self.canvas_dims()
        dims = self.level_dims(ny, nx)
        H, W = self.cat_hw(dims)
        C, ks = self.feat_c, self.key_fc[-1][0].shape[0]
        hwc = H * W * C
        # message layout: [n_pad warped maps | n_pad keys | the ego's query]; only the first n slots are written
        send = self.buf("shard_send", (n_pad * (hwc + ks) + ks,))
        meta = {"n_loc": n_pad, "H": H, "W": W, "C": C, "ks": ks}
        if n == 0:
            _lib.check(self.lib.av2x_fill_zero(_ptr(send[n_pad * (hwc + ks):]), ks * 4, self.stream()), "av2x_fill_zero")
            return send, torch.zeros(2, dtype=torch.int64, device=self.device), meta
        s = self.buf("w2_shrink", (n, H, W, C))
        self.trunk(canvas, n, ny, nx, shrink_out=s)
        if getattr(self, "compression", 0):
            # NaiveCompressor (airv2x_when2com.py:50-52, 122-123): everything When2com does per agent after it -- the warp into the ego
            # frame, the policy network, the key MLP: 290 of the ~300 GFLOP per agent -- works on the DECODED map, and this split keeps that
            # work on the sender, so encoder and decoder both run here and the message stays the warped 256-channel map.  (Shipping the
            # C / ratio-channel encoder output instead would move the warp and the policy network of every agent to the ego's rank.)
            self.run_compressor(s, n, H, W)
        st = self.stream()
        nz = self.buf("nonzero", (1,), torch.int64)
        _lib.check(self.lib.av2x_fill_zero(_ptr(nz), 8, st), "av2x_fill_zero")
        _lib.check(self.lib.av2x_count_nonzero(_ptr(s), s.numel(), _ptr(nz), st), "av2x_count_nonzero")
        pair = data_dict_local["img_pairwise_t_matrix_collab"]
        pair = pair.detach().cpu().numpy() if isinstance(pair, torch.Tensor) else np.asarray(pair)
        theta = normalized_pairwise(pair, H, W, self.w2["voxel_size"][0], self.w2["downsample_rate"])
        off = data_dict_local.get("shard_agent_offset")     # global index of this rank's first agent
        off = int(data_dict_local.get("shard_rank", 0)) * n if off is None else int(off)
        warped = self.warp(s, theta[0, 0, off:off + n], n, H, W, C, out=send[:n * hwc].view(n, H, W, C))
        qk, keys = self.policy_keys(warped, n, H, W)
        send[n_pad * hwc:n_pad * hwc + n * ks].view(n, ks).copy_(keys)
        if has_ego:
            send[n_pad * (hwc + ks):].copy_(self.query_of(qk[0:1]).view(-1))
        else:
            _lib.check(self.lib.av2x_fill_zero(_ptr(send[n_pad * (hwc + ks):]), ks * 4, st), "av2x_fill_zero")
        stats = torch.stack([torch.zeros((), dtype=torch.int64, device=self.device), nz[0]])
        return send, stats, meta

    @torch.no_grad()
    def shard_ego_stage(self, recv, stats, meta, world, trace=None, sync_comm_rate=False):
        """Ego half: softmax over the gathered keys, weighted sum of the gathered warped maps (read in place from the
        all-gather result), heads."""
        n_loc, H, W, C, ks = meta["n_loc"], meta["H"], meta["W"], meta["C"], meta["ks"]
        counts = meta.get("counts") or [n_loc] * world
        hwc, N = H * W * C, sum(counts)
        chunk = n_loc * (hwc + ks) + ks
        if recv.numel() != world * chunk or len(counts) != world or max(counts) > n_loc:
            raise ValueError("gathered buffer has the wrong size")
        if N > 32:
            raise ValueError(f"{N} agents exceed the 32 the fusion kernel takes")
        per_rank = recv.view(world, chunk)
        keys = self.buf("w2_keys_all", (N, ks))
        a = 0
        for r, c in enumerate(counts):
            if c:
                keys[a:a + c].copy_(per_rank[r, n_loc * hwc:n_loc * hwc + c * ks].view(c, ks))
                a += c
        maps = [per_rank[r, j * hwc:(j + 1) * hwc] for r in range(world) for j in range(counts[r])]
        fused = self.buf("w2_fused", (1, H, W, C))
        coef = self.buf("w2_coef", (1, 32))
        self.fuse(keys, per_rank[0, n_loc * (hwc + ks):], maps, fused[0], coef[0])   # rank 0 holds the ego
        if trace is not None:
            trace["coef0"], trace["fused"] = coef[0, :N].clone(), fused.permute(0, 3, 1, 2).clone()
        out = self._heads_out(fused, 1, H, W)
        out.update({"mask": 0, "comm_rate": int(stats[1].item()) / 1 if sync_comm_rate else stats[1]})
        return out

    # ------------------------------------------------------------------ full forward
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
        s_all = self.buf("w2_shrink", (n_total, H, W, C))
        self.trunk(canvas, n_total, ny, nx, shrink_out=s_all)
        if self.compression:                                   # airv2x_when2com.py:122-123
            self.run_compressor(s_all, n_total, H, W)
        st = self.stream()
        nz = self.buf("nonzero", (1,), torch.int64)            # communication_rates: non-zeros of the shared maps (:118)
        _lib.check(self.lib.av2x_fill_zero(_ptr(nz), 8, st), "av2x_fill_zero")
        _lib.check(self.lib.av2x_count_nonzero(_ptr(s_all), s_all.numel(), _ptr(nz), st), "av2x_count_nonzero")
        pair = data_dict["img_pairwise_t_matrix_collab"]
        pair = pair.detach().cpu().numpy() if isinstance(pair, torch.Tensor) else np.asarray(pair)
        if pair.shape[0] != B:
            raise ValueError("img_pairwise_t_matrix_collab batch size does not match record_len")
        theta = normalized_pairwise(pair, H, W, self.w2["voxel_size"][0], self.w2["downsample_rate"])
        fused = self.buf("w2_fused", (B, H, W, C))
        coef = self.buf("w2_coef", (B, 32))
        off = 0
        for b, n in enumerate(record_len):
            warped = self.warp(s_all[off:off + n], theta[b, 0, :n], n, H, W, C)
            qk, keys = self.policy_keys(warped, n, H, W)
            q = self.query_of(qk[0:1])
            self.fuse(keys, q, [warped[j] for j in range(n)], fused[b], coef[b])
            if trace is not None:
                trace[f"warped{b}"] = warped.permute(0, 3, 1, 2).clone()
                trace[f"policy{b}"] = qk.permute(0, 3, 1, 2).clone()
                trace[f"keys{b}"], trace[f"query{b}"], trace[f"coef{b}"] = keys.clone(), q.clone(), coef[b, :n].clone()
            off += n
        if trace is not None:
            trace["shrink"] = s_all.permute(0, 3, 1, 2).clone()
            trace["fused"] = fused.permute(0, 3, 1, 2).clone()
        out = self._heads_out(fused, B, H, W)
        if sync_comm_rate:
            rate = int(nz[0].item()) / B                      # np.sum(counts) / B  (:132)
        else:
            rate = nz[0].clone() if B == 1 else nz[0].double() / B
        out.update({"mask": 0, "comm_rate": rate})
        return out
