"""Host-side driver of the CoBEVT-LiDAR path (models/airv2x_cobevt.py:112-156) on one MI355X.

Per-agent trunk = the Where2Comm engine's (same encoders / backbone / shrink header).  Fusion =
SwapFusionEncoder (swap_fusion_modules.py:233-280): the token tensor stays in ONE NHWC buffer
``x (L, H, W, C)`` for the whole encoder; 'regroup' (fuse_utils.py:13-64) is the shrink conv
writing the real agents into ``x[:n]`` plus a zero fill of the padded agents, and every einops
rearrange of the reference is index arithmetic inside av2x_fax_attention.  The Linear layers are
1x1 convolutions over tokens on conv_igemm_f32 (bias / GELU / residual fused in its epilogue).

Padded agents cannot be skipped: their query tokens do attend to the valid keys and the final
``mean`` over the agent axis includes them (swap_fusion_modules.py:270), exactly as in the reference.
"""
from __future__ import annotations

import torch

from .. import _lib
from .engine import ConvLayer, Where2ComEngine, _ptr
from .packing import fold_bn, pack_conv_weight

LN_EPS = 1e-5
_FAX_X3 = __import__("os").environ.get("AV2X_FAX_X3", "1") not in ("0", "off", "")


class CoBEVTEngine(Where2ComEngine):
    def _init_config(self, args):
        self.bb = args["base_bev_backbone"]
        self.sh = args["shrink_header"]
        self.fcfg = {"fully": False}
        self.fax = args["fax_fusion"]
        self.compression = int(args.get("compression", 0) or 0)
        if self.compression:
            c = self.fax["input_dim"]
            if c % self.compression or (c // self.compression) % 32:
                raise NotImplementedError(f"compression {self.compression}: {c}/ratio must be a multiple of 32 channels")
        if not self.fax.get("mask", False):
            raise NotImplementedError("SwapFusionBlock without mask (not the shipped AirV2X config)")
        self.L = int(sum(args["max_cav"].values()))
        self.heads_n = self.fax["input_dim"] // self.fax["dim_head"]

    FUSION_WEIGHTS = ("fax_layers", "head_ln", "head_lin", "compressor")

    def fax_x3(self):
        """x3 mode (every product of the frame from three bf16 terms per operand on the bf16 matrix cores): the attention contractions too
        (fax_attention_x3_kernel, > 4 valid agents; AV2X_FAX_X3=0 keeps them on the fp32-input MFMA)."""
        return bool(self.x3p) and not self.amp and _FAX_X3

    def _linear(self, sd, wkey, bkey, act, up, rows=None):
        """Linear -> 1x1 conv layer; ``rows`` = slice of output features (used to split to_qkv into q | k,v)."""
        w = sd[wkey].detach().float()
        if rows is not None:
            w = w[rows]
        wp, coutp = pack_conv_weight(w.view(w.shape[0], w.shape[1], 1, 1))
        b = sd[bkey].detach().float() if bkey else torch.zeros(w.shape[0])
        return ConvLayer(up(wp), None, up(b), w.shape[1], w.shape[0], coutp, 1, 1, 0, act)

    def _load_fusion(self, sd, up, prefix="fusion_net."):
        if self.compression:
            self.compressor = self._load_compressor(sd, up)
        C, ws, L = self.fax["input_dim"], self.fax["window_size"], self.L
        from ..synth import _relative_position_index
        expect = torch.from_numpy(_relative_position_index(L, ws))
        self.fax_layers = []
        for i in range(self.fax["depth"]):
            blk = {}
            for part in ("window", "grid"):
                a, f = f"{prefix}layers.{i}.{part}_attention", f"{prefix}layers.{i}.{part}_ffd"
                idx = sd[a + ".fn.relative_position_index"].cpu()
                if idx.shape != expect.shape or not torch.equal(idx, expect):
                    raise ValueError(f"{a}.fn.relative_position_index is not the index of a ({L},{ws},{ws}) window")
                blk[part] = {
                    "ln1": (up(sd[a + ".norm.weight"].float()), up(sd[a + ".norm.bias"].float())),
                    "qkv": self._linear(sd, a + ".fn.to_qkv.weight", None, 0, up),
                    # padded agents are never keys (swap_fusion_modules.py:103-108): their K / V columns are not computed
                    "q": self._linear(sd, a + ".fn.to_qkv.weight", None, 0, up, rows=slice(0, C)),
                    "kv": self._linear(sd, a + ".fn.to_qkv.weight", None, 0, up, rows=slice(C, 3 * C)),
                    "out": self._linear(sd, a + ".fn.to_out.0.weight", None, 0, up),
                    "table": up(sd[a + ".fn.relative_position_bias_table.weight"].detach().float()),
                    "ln2": (up(sd[f + ".norm.weight"].float()), up(sd[f + ".norm.bias"].float())),
                    "ff1": self._linear(sd, f + ".fn.net.0.weight", f + ".fn.net.0.bias", 2, up),   # + GELU
                    "ff2": self._linear(sd, f + ".fn.net.3.weight", f + ".fn.net.3.bias", 0, up),
                }
            self.fax_layers.append(blk)
        self.head_ln = (up(sd[prefix + "mlp_head.2.weight"].float()), up(sd[prefix + "mlp_head.2.bias"].float()))
        self.head_lin = self._linear(sd, prefix + "mlp_head.3.weight", prefix + "mlp_head.3.bias", 0, up)

    def ln(self, x, gb, y, n_tokens, c):
        _lib.check(self.lib.av2x_layernorm(_ptr(x), _ptr(gb[0]), _ptr(gb[1]), _ptr(y), n_tokens, c, LN_EPS, self.stream()),
                   "av2x_layernorm")

    def fax_encoder(self, x, n_valid, H, W, trace=None):
        """x (L,H,W,C) updated in place through the depth x {window, grid} x {attention, FFN} blocks;
        returns the (1,H,W,C) output of mlp_head."""
        L, C, ws = self.L, self.fax["input_dim"], self.fax["window_size"]
        if H % ws or W % ws:
            raise ValueError(f"BEV map {H}x{W} is not divisible by the window size {ws}")
        nt = L * H * W
        qkv = self.buf("fax_qkv", (L, H, W, 3 * C))
        att = self.buf("fax_att", (L, H, W, C))
        hid = self.buf("fax_hid", (L, H, W, self.fax["mlp_dim"]))
        for i, blk in enumerate(self.fax_layers):
            for gi, part in enumerate(("window", "grid")):
                P = blk[part]
                # PreNorm: the LayerNorm of the residual stream is applied by the consuming Linear while it loads its rows (engine.conv ln=);
                # only the per-token (mean, rstd) pass over x remains of it
                ln1 = self.ln_operand(x, nt, C, P["ln1"][0], P["ln1"][1], LN_EPS)
                if n_valid == L:
                    self.conv(P["qkv"], x, L, H, W, qkv, ln=ln1)
                else:   # q for all L agents (padded query tokens attend the valid keys), k | v only for the valid ones
                    self.conv(P["q"], x, L, H, W, qkv, out_ctot=3 * C, out_coff=0, ln=ln1)
                    self.conv(P["kv"], x, n_valid, H, W, qkv, out_ctot=3 * C, out_coff=C, ln=ln1.rows(0, n_valid * H * W))
                _lib.check(self.lib.av2x_fax_attention(_ptr(qkv), _ptr(P["table"]), _ptr(att), L, n_valid, H, W, ws,
                                                       self.heads_n, self.fax["dim_head"], gi | (32 if self.fax_x3() else 0), self.stream()),
                           "av2x_fax_attention")
                self.conv(P["out"], att, L, H, W, x, residual=x)            # to_out(.) + x   (PreNormResidual)
                ln2 = self.ln_operand(x, nt, C, P["ln2"][0], P["ln2"][1], LN_EPS)
                self.conv(P["ff1"], x, L, H, W, hid, ln=ln2)                 # LayerNorm + Linear + bias + GELU
                self.conv(P["ff2"], hid, L, H, W, x, residual=x)            # Linear + bias + x
            if trace is not None:
                trace[f"fax_block{i}"] = x.permute(0, 3, 1, 2).unsqueeze(0).clone()
        mean = self.buf("fax_mean", (1, H, W, C))
        _lib.check(self.lib.av2x_agent_mean(_ptr(x), _ptr(mean), L, H * W * C, self.stream()), "av2x_agent_mean")
        fused = self.buf("fax_fused", (1, H, W, C))
        self.conv(self.head_lin, mean, 1, H, W, fused, ln=self.ln_operand(mean, H * W, C, self.head_ln[0], self.head_ln[1], LN_EPS, "ln_stats_head"))
        return fused

    def _heads_out(self, fused, H, W, B=1):
        heads = torch.empty((B, self.heads.cout, H, W), dtype=torch.float32, device=self.device)
        self.conv(self.heads, fused, B, H, W, heads)
        outs = torch.split(heads, self.head_splits, dim=1)
        out = {"psm": outs[0], "rm": outs[1]}
        if self.args["obj_head"]:
            out["obj"] = outs[2]
        return out

    # ------------------------------------------------------------------ agent sharding
    @torch.no_grad()
    def shard_local_stage(self, data_dict_local, has_ego, n_pad=None):
        """Per-rank half: encoders + backbone + shrink header for THIS rank's agents, written straight into the
        all-gather send buffer (n_pad,H,W,C) -- 36.0 MB per agent at the default grid; n_pad >= the local count pads
        an uneven frame's message (sharded.py).  With message compression
        the buffer holds the NaiveCompressor ENCODER output instead (C/ratio channels: 9.0 MB at ratio 4); the
        decoder runs on the receiving side.  Replaces regroup()'s in-process concat (fuse_utils.py:13-64)."""
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
        C = self.fax["input_dim"]
        cm = self.compressor[0].cout if self.compression else C
        send = self.buf("shard_send", (n_pad * H * W * cm,), self.msg_dtype())     # autocast: bf16, 18.0 MB per agent (uncompressed)
        stats = torch.zeros(2, dtype=torch.int64, device=self.device)
        if n == 0:
            return send, stats, {"n_loc": n_pad, "H": H, "W": W, "cm": cm}
        if self.compression:
            s = self.buf("shard_shrink", (n, H, W, C))
            self.trunk(canvas, n, ny, nx, shrink_out=s)
            self.conv(self.compressor[0], s, n, H, W, send[:n * H * W * cm].view(n, H, W, cm))
        else:
            self.trunk(canvas, n, ny, nx, shrink_out=send[:n * H * W * C].view(n, H, W, C))
        return send, stats, {"n_loc": n_pad, "H": H, "W": W, "cm": cm}

    @torch.no_grad()
    def shard_ego_stage(self, recv, stats, meta, world, trace=None, **_):
        """Ego half: the gathered buffer is already in frame order (rank-major = agent-major); decode it (if
        compressed) or copy it into the padded token tensor, then fusion + heads."""
        msg, N, H, W, C = self._gathered_tokens(recv, meta, world, decode=False)
        x = self.buf("fax_x", (self.L, H, W, C))
        if N < self.L:
            _lib.check(self.lib.av2x_fill_zero(_ptr(x[N:]), (self.L - N) * H * W * C * 4, self.stream()), "av2x_fill_zero")
        if self.compression:
            mid = self.buf("compress_mid", (N, H, W, C))
            self.conv(self.compressor[1], msg, N, H, W, mid)
            self.conv(self.compressor[2], mid, N, H, W, x[:N])
        elif msg.dtype == torch.bfloat16:
            self.widen(msg, x[:N])
        else:
            x[:N].copy_(msg)
        fused = self.fax_encoder(x, N, H, W, trace)
        return self._heads_out(fused, H, W)

    # second level: the fusion itself is split over the ranks (sharded.fusion_column_shards), 1/world of the 1.3-1.6 TFLOP each
    fusion_sharding = True

    def _gathered_tokens(self, recv, meta, world, decode=True):
        """The real agents' messages as one (N,H,W,cm) tensor in frame order: a view of the gathered buffer when every
        rank holds n_loc agents, else (uneven frame, meta["counts"]) the valid slots compacted into a workspace."""
        n_loc, H, W, cm = meta["n_loc"], meta["H"], meta["W"], meta["cm"]
        counts = meta.get("counts") or [n_loc] * world
        N, C = sum(counts), self.fax["input_dim"]
        if recv.numel() != world * n_loc * H * W * cm or len(counts) != world or max(counts) > n_loc:
            raise ValueError("gathered buffer has the wrong size")
        if N > self.L:
            raise ValueError(f"{N} agents exceed max_cav_num = {self.L}")
        msg = recv.view(world * n_loc, H, W, cm)
        if N != world * n_loc:
            from .sharded import valid_slots
            cmp = self.buf("shard_compact", (N, H, W, cm), recv.dtype)
            for a, slot in enumerate(valid_slots(counts, n_loc)):
                cmp[a].copy_(msg[slot])
            msg = cmp
        if self.compression and decode:   # decode on the receiving side
            mid = self.buf("compress_mid", (N, H, W, C))
            dec = self.buf("compress_dec", (N, H, W, C))
            self.conv(self.compressor[1], msg, N, H, W, mid)
            self.conv(self.compressor[2], mid, N, H, W, dec)
            msg = dec
        return msg, N, H, W, C

    @torch.no_grad()
    def shard_ego_partial(self, recv, stats, meta, world, rank):
        """This rank's share of the fusion: its residue-group columns of every agent's map, compacted to (L, H, Wc, C), go
        through the ordinary encoder and heads; returns the flat (heads, H, Wc) result for the second all-gather."""
        from .sharded import fusion_column_shards
        msg, N, H, W, C = self._gathered_tokens(recv, meta, world)
        shards = fusion_column_shards(W, self.fax["window_size"], world)
        cols, _ = shards[rank]
        Wc = len(cols)
        key = ("fusion_cols", W, world, rank)
        ci = self.ws.get(key)
        if ci is None:
            ci = torch.tensor(cols, dtype=torch.int64, device=self.device)
            self.ws[key] = ci
        xc = self.buf("fax_xc", (self.L, H, Wc, C))
        if msg.dtype == torch.bfloat16:                                  # the uncompressed autocast message: gather, then widen
            xc16 = self.buf("fax_xc16", (N, H, Wc, C), torch.bfloat16)
            torch.index_select(msg, 2, ci, out=xc16)
            self.widen(xc16, xc[:N])
        else:
            torch.index_select(msg, 2, ci, out=xc[:N])                  # column gather (data movement only)
        if N < self.L:
            _lib.check(self.lib.av2x_fill_zero(_ptr(xc[N:]), (self.L - N) * H * Wc * C * 4, self.stream()), "av2x_fill_zero")
        fused = self.fax_encoder(xc, N, H, Wc)
        heads = torch.empty((1, self.heads.cout, H, Wc), dtype=torch.float32, device=self.device)
        self.conv(self.heads, fused, 1, H, Wc, heads)
        return heads.view(-1), {"H": H, "W": W, "Wc": Wc, "shards": shards}

    @torch.no_grad()
    def shard_ego_finish(self, parts, ctx, world, **_):
        H, W, Wc, shards = ctx["H"], ctx["W"], ctx["Wc"], ctx["shards"]
        nh = self.heads.cout
        per_rank = parts.view(world, nh, H, Wc)
        full = torch.empty((1, nh, H, W), dtype=torch.float32, device=self.device)
        ws = self.fax["window_size"]
        for r, (cols, valid) in enumerate(shards):
            if valid == 0:
                continue
            strip = Wc // ws                                             # compact columns per w2 strip
            keep = [w2 * strip + j for w2 in range(ws) for j in range(valid)]
            src = torch.tensor(keep, dtype=torch.int64, device=self.device)
            dst = torch.tensor([cols[k] for k in keep], dtype=torch.int64, device=self.device)
            full[0].index_copy_(2, dst, per_rank[r].index_select(2, src))
        outs = torch.split(full, self.head_splits, dim=1)
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
        if max(record_len) > self.L:
            raise ValueError(f"{max(record_len)} agents exceed max_cav_num = {self.L}")
        canvas, ny, nx = self.encode(data_dict, record_len, slots)
        dims = self.level_dims(ny, nx)
        H, W = self.cat_hw(dims)
        C = self.fax["input_dim"]
        x = self.buf("fax_x", (self.L, H, W, C))
        if B == 1:
            # regroup (fuse_utils.py:13-64): real agents first, zero padding after; the shrink conv writes x[:n]
            n = record_len[0]
            if n < self.L:
                _lib.check(self.lib.av2x_fill_zero(_ptr(x[n:]), (self.L - n) * H * W * C * 4, self.stream()), "av2x_fill_zero")
            m16 = self.msg_dtype() == torch.bfloat16 and not self.compression
            if m16:   # autocast: the shrink header's output is the bf16 message (with compression: the encoder's, in run_compressor)
                x16 = self.buf("fax_x16", (n, H, W, C), torch.bfloat16)
                _, s, H2, W2 = self.trunk(canvas, n, ny, nx, shrink_out=x16)
                self.widen(x16, x[:n])
            else:
                _, s, H2, W2 = self.trunk(canvas, n, ny, nx, shrink_out=x[:n])
            assert (H2, W2) == (H, W)
            if self.compression:
                self.run_compressor(x[:n], n, H, W)
            if trace is not None:
                trace["shrink"] = x[:n].permute(0, 3, 1, 2).clone()
            fused = self.fax_encoder(x, n, H, W, trace)
            if trace is not None:
                trace["fused"] = fused.permute(0, 3, 1, 2).clone()
            return self._heads_out(fused, H, W)
        # B > 1 (the reference's collate layout): the trunk runs on all agents at once, the fusion per sample
        # (regroup pads every sample to L agents, SwapFusionEncoder never mixes samples)
        s_all = self.buf("shrink_batch", (n_total, H, W, C))
        if self.msg_dtype() == torch.bfloat16 and not self.compression:
            s16 = self.buf("shrink_batch16", (n_total, H, W, C), torch.bfloat16)
            self.trunk(canvas, n_total, ny, nx, shrink_out=s16)
            self.widen(s16, s_all)
        else:
            self.trunk(canvas, n_total, ny, nx, shrink_out=s_all)
        if self.compression:
            self.run_compressor(s_all, n_total, H, W)
        fused_all = self.buf("fused_batch", (B, H, W, C))
        off = 0
        for b, n in enumerate(record_len):
            x[:n].copy_(s_all[off:off + n])
            if n < self.L:
                _lib.check(self.lib.av2x_fill_zero(_ptr(x[n:]), (self.L - n) * H * W * C * 4, self.stream()), "av2x_fill_zero")
            fused_all[b:b + 1].copy_(self.fax_encoder(x, n, H, W))
            off += n
        return self._heads_out(fused_all, H, W, B)
