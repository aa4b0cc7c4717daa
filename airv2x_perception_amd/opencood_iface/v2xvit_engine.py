"""Host-side driver of the V2X-ViT-LiDAR path (models/airv2x_v2xvit.py:108-167) on one MI355X.

Per-agent trunk = the Where2Comm engine's.  Fusion = V2XTransformer (v2xvit_basic.py:135-213) on one
NHWC token buffer ``x (n, H, W, 256)`` holding only the REAL agents: padded agents are masked out as
keys of every agent-wise attention (cav mask) and the window attention / FFN act per agent, so they
can never influence agent 0, which is the only output (``output[:, 0]``, :212).

HGT (hmsa.py): the per-relation matrices are folded into the per-type Linear weights once at load
time (fp64 products rounded to fp32): for an agent of node type t the projection GEMM emits
[q'(t->0) | q'(t->1) | k_t | v'(0<-t) | v'(1<-t)] so that the per-pixel kernel is a plain masked
softmax attention over agents and the (B,M,H,W,L,L,C) ``v_msg`` tensor of the reference (8.1 GB at
L = 15) never exists.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_int32, c_void_p

import torch

from .. import _lib
from . import warp as warp_host
from .engine import ConvLayer, Where2ComEngine, _ptr, _w16i
from .packing import pack_conv_weight

LN_EPS = 1e-5


class V2XViTEngine(Where2ComEngine):
    def _init_config(self, args):
        self.bb = args["modality_fusion"]["base_bev_backbone"]
        self.sh = args["modality_fusion"]["shrink_header"]
        self.fcfg = {"fully": False}
        from ..synth import model_compression
        self.compression = model_compression(args)     # airv2x_v2xvit.py:42-44: NaiveCompressor(256, args["compression"]) in front of the fusion
        if self.compression and (256 % self.compression or (256 // self.compression) % 32):
            raise NotImplementedError(f"compression {self.compression}: 256/ratio must be a multiple of 32 channels")
        self.enc = args["transformer"]["encoder"]
        self.cav, self.pw = self.enc["cav_att_config"], self.enc["pwindow_att_config"]
        if not self.cav["use_hetero"] or self.pw["fusion_method"] != "split_attn" or not self.pw["relative_pos_embedding"]:
            raise NotImplementedError("only the shipped V2X-ViT configuration (hetero attention, split_attn, relative pos)")
        self.L = int(args["max_cav_num"])
        # the encoder's output is the ego's feature map only: skip what the last layer computes for the other agents
        self.ego_only_last = True

    FUSION_WEIGHTS = ("layers", "rte_table", "rte_lin", "compressor")

    def _lin(self, w, b, act, up):
        w = w.detach().float()
        wp, coutp = pack_conv_weight(w.reshape(w.shape[0], w.shape[1], 1, 1))
        b = b.detach().float() if b is not None else torch.zeros(w.shape[0])
        return ConvLayer(up(wp), None, up(b), w.shape[1], w.shape[0], coutp, 1, 1, 0, act)

    def _load_fusion(self, sd, up, p="fusion_net.encoder"):
        self.compressor = self._load_compressor(sd, up) if getattr(self, "compression", 0) else None
        heads, dh = self.cav["heads"], self.cav["dim_head"]
        self.layers = []
        for d in range(self.enc["depth"]):
            blocks = []
            for nb in range(self.enc["num_blocks"]):
                q = f"{p}.layers.{d}.0.layers.{nb}"
                h = q + ".0.fn"
                ratt, rmsg = sd[h + ".relation_att"].double().cpu(), sd[h + ".relation_msg"].double().cpu()
                proj, aout, proj_kv = [], [], []
                for t in range(2):
                    Wq, bq = sd[f"{h}.q_linears.{t}.weight"].double().cpu(), sd[f"{h}.q_linears.{t}.bias"].double().cpu()
                    Wk, bk = sd[f"{h}.k_linears.{t}.weight"].double().cpu(), sd[f"{h}.k_linears.{t}.bias"].double().cpu()
                    Wv, bv = sd[f"{h}.v_linears.{t}.weight"].double().cpu(), sd[f"{h}.v_linears.{t}.bias"].double().cpu()
                    rows_w, rows_b = [], []
                    for tj in range(2):      # q' = w_att[e = t*2 + tj]^T q   per head (einsum 'i p, p q, j q', hmsa.py:139-141)
                        e = t * 2 + tj
                        for m in range(heads):
                            sl = slice(m * dh, (m + 1) * dh)
                            rows_w.append(ratt[e, m].t() @ Wq[sl])
                            rows_b.append(ratt[e, m].t() @ bq[sl])
                    rows_w.append(Wk); rows_b.append(bk)
                    for ti in range(2):      # v' = w_msg[e = ti*2 + t]^T v   (einsum 'i j p c, j p', :150)
                        e = ti * 2 + t
                        for m in range(heads):
                            sl = slice(m * dh, (m + 1) * dh)
                            rows_w.append(rmsg[e, m].t() @ Wv[sl])
                            rows_b.append(rmsg[e, m].t() @ bv[sl])
                    proj.append(self._lin(torch.cat(rows_w, 0), torch.cat(rows_b, 0), 0, up))
                    # k | v' columns only (proj[:, 512:1280]): all a non-ego agent contributes to the LAST layer
                    proj_kv.append(self._lin(torch.cat(rows_w, 0)[512:], torch.cat(rows_b, 0)[512:], 0, up))
                    aout.append(self._lin(sd[f"{h}.a_linears.{t}.weight"], sd[f"{h}.a_linears.{t}.bias"], 0, up))
                w = q + ".1.fn"
                qkv_w = torch.cat([sd[f"{w}.pwmsa.{i}.to_qkv.weight"] for i in range(3)], 0)
                blk = {
                    "ln1": (up(sd[q + ".0.norm.weight"].float()), up(sd[q + ".0.norm.bias"].float())),
                    "proj": proj, "aout": aout, "proj_kv": proj_kv,
                    "ln2": (up(sd[q + ".1.norm.weight"].float()), up(sd[q + ".1.norm.bias"].float())),
                    "qkv3": self._lin(qkv_w, None, 0, up),
                    "pos": [up(sd[f"{w}.pwmsa.{i}.pos_embedding"].float()) for i in range(3)],
                    "wout": [self._lin(sd[f"{w}.pwmsa.{i}.to_out.0.weight"], sd[f"{w}.pwmsa.{i}.to_out.0.bias"], 0, up) for i in range(3)],
                    "fc1": self._lin(sd[w + ".split_attn.fc1.weight"], None, 0, up),
                    "bn1": (up(sd[w + ".split_attn.bn1.weight"].float()), up(sd[w + ".split_attn.bn1.bias"].float())),
                    "fc2": self._lin(sd[w + ".split_attn.fc2.weight"], None, 0, up),
                }
                blocks.append(blk)
            f = f"{p}.layers.{d}.1"
            ffn = {"ln": (up(sd[f + ".norm.weight"].float()), up(sd[f + ".norm.bias"].float())),
                   "ff1": self._lin(sd[f + ".fn.net.0.weight"], sd[f + ".fn.net.0.bias"], 2, up),
                   "ff2": self._lin(sd[f + ".fn.net.3.weight"], sd[f + ".fn.net.3.bias"], 0, up)}
            self.layers.append((blocks, ffn))
        self.rte_table = up(sd[p + ".rte.emb.emb.weight"].float())
        self.rte_lin = self._lin(sd[p + ".rte.emb.lin.weight"], sd[p + ".rte.emb.lin.bias"], 0, up)

    def ln(self, x, gb, y, n_tokens, c, relu=0):
        _lib.check(self.lib.av2x_layernorm_act(_ptr(x), _ptr(gb[0]), _ptr(gb[1]), _ptr(y), n_tokens, c, LN_EPS, relu,
                                               self.stream()), "av2x_layernorm")

    @staticmethod
    def _groups(types):
        """Consecutive agents of equal node type -> [(start, stop, type)] (the per-type Linear runs once per group)."""
        out, s = [], 0
        for i in range(1, len(types) + 1):
            if i == len(types) or types[i] != types[s]:
                out.append((s, i, types[s]))
                s = i
        return out

    # second level of the agent-sharded frame: the encoder blocks run on a column strip of the map (windows of 2 / 4 are
    # aligned to 4 columns, everything else is per pixel) and the one global quantity -- the split-attention mean over
    # the map -- is averaged over the ranks (``gap_exchange(gap, world, index)``, default: RCCL all-reduce)
    fusion_sharding = True
    gap_exchange = None
    gap_record = None      # list -> the (m,1,1,C) split-attention means of a run are appended (tests)

    def encoder(self, x, n, H, W, prior, scm, trace=None, strip=None):
        """``strip`` = (first column, number of columns, world): after RTE / STTF / ROI mask on the full maps only these
        columns go through the blocks (returns the (1,H,Wc,C) strip of the ego's fused map)."""
        C, hw, st = 256, H * W, self.stream
        types = [int(prior[i, 2]) for i in range(n)]                      # infra flag -> node type (hmsa.py:123-127)
        dts = [int(prior[i, 1]) for i in range(n)]
        # ---- RTE: x[i] += lin(emb[dt_i * ratio])      (v2xvit_basic.py:58-80)
        vec = None
        if self.cav["use_RTE"]:
            rows = self.rte_table[[dt * self.cav["RTE_ratio"] for dt in dts]].contiguous()     # table rows (gather only)
            vec = self.buf("rte_vec", (n, 1, 1, C))
            self.conv(self.rte_lin, rows.view(n, 1, 1, C), n, 1, 1, vec)
            if n == 1:
                _lib.check(self.lib.av2x_add_agent_vector(_ptr(x), _ptr(vec), n, hw * C, C, st()), "av2x_add_agent_vector")
            # n > 1: the add happens where the maps are read anyway -- in the STTF warp's taps (agents 1..) and in the ego's move
        # ---- STTF: warp agents 1.. into the ego frame; ROI x cav mask
        d = warp_host.discretized_matrix(scm[:n], self.enc["sttf"]["voxel_size"][0], self.enc["sttf"]["downsample_rate"])
        T = warp_host.transformation_matrix(d, (H, W))
        theta = torch.from_numpy(warp_host.affine_theta(T, (H, W), (H, W))).to(self.device)
        if n > 1:
            # the warped maps land in a second buffer that becomes the stream from here on (the ego's map is moved there: 1 / n of the
            # copy-back of all the warped maps)
            xw = self.buf("sttf_out", (n, H, W, C))
            if vec is not None:
                _lib.check(self.lib.av2x_warp_affine_add(_ptr(x[1:]), _ptr(theta[1:]), _ptr(vec[1:]), _ptr(xw[1:]), n - 1, H, W, C, st()),
                           "av2x_warp_affine_add")
                _lib.check(self.lib.av2x_add_agent_vector_to(_ptr(x[0:1]), _ptr(vec[0:1]), _ptr(xw[0:1]), 1, hw * C, C, st()), "av2x_add_agent_vector_to")
            else:
                _lib.check(self.lib.av2x_warp_affine(_ptr(x[1:]), _ptr(theta[1:]), _ptr(xw[1:]), n - 1, H, W, C, st()), "av2x_warp_affine")
                xw[0].copy_(x[0])
            x = xw
        mask = self.buf("com_mask", (n, H, W))
        ones = self.buf("cav_ones", (n,), torch.int32)
        if self.enc["use_roi_mask"]:
            ones.fill_(1)
            _lib.check(self.lib.av2x_roi_mask(_ptr(theta), _ptr(ones), _ptr(mask), n, H, W, st()), "av2x_roi_mask")
        else:
            mask.fill_(1.0)
        if trace is not None:
            trace["after_sttf"] = x.clone()
            trace["com_mask"] = mask.clone()
        world = 1
        if strip is not None:
            c0, Wc, world = strip
            xs = self.buf("vit_x_strip", (n, H, Wc, C))
            ms = self.buf("com_mask_strip", (n, H, Wc))
            xs.copy_(x[:, :, c0:c0 + Wc])                                   # column strip (data movement only)
            ms.copy_(mask[:, :, c0:c0 + Wc])
            x, mask, W, hw = xs, ms, Wc, H * Wc
        if self.amp and self.bf16_activations and self.enc["feed_forward"]["mlp_dim"] == 256:
            return self._blocks_bf16(x, mask, n, H, W, types, world, trace)
        tarr = (c_int32 * n)(*types)
        proj = self.buf("vit_proj", (n, H, W, 1280))
        att = self.buf("vit_att", (n, H, W, C))
        qkv3 = self.buf("vit_qkv3", (n, H, W, 2304))
        wat = self.buf("vit_wat", (n, H, W, C))
        br = [self.buf(f"vit_br{i}", (n, H, W, C)) for i in range(3)]
        gap = self.buf("vit_gap", (n, 1, 1, C))
        gap_scratch = self.buf("vit_gap_scratch", (n, 128, C))
        g1 = self.buf("vit_g1", (n, 1, 1, C))
        g2 = self.buf("vit_g2", (n, 1, 1, C))
        logits = self.buf("vit_logits", (n, 1, 1, 3 * C))
        hid = self.buf("vit_hid", (n, H, W, self.enc["feed_forward"]["mlp_dim"]))
        groups = self._groups(types)
        last = len(self.layers) - 1
        for di, (blocks, ffn) in enumerate(self.layers):
            for bi, blk in enumerate(blocks):
                # V2XTransformer returns output[:, 0] (the ego) only: in the last block of the last layer the other agents
                # are needed as KEYS / VALUES of the HGT attention and nowhere else -> m = 1 agent from there on
                ego_only = self.ego_only_last and di == last and bi == len(blocks) - 1 and trace is None and n > 1
                # ---- x = HGT(LN(x)) + x
                # PreNorm: the consuming Linears normalise their rows while they load them (engine.conv ln=); what is left of the LayerNorm
                # launch is the per-token (mean, rstd) pass
                s1 = self.ln_operand(x, n * hw, C, blk["ln1"][0], blk["ln1"][1], LN_EPS)
                ln1 = lambda a, b: s1.rows(a * hw, b * hw)
                if ego_only:
                    self.conv(blk["proj"][types[0]], x[0:1], 1, H, W, proj[0:1], ln=ln1(0, 1))
                    for (a, b, t) in self._groups(types[1:]):
                        self.conv(blk["proj_kv"][t], x[a + 1:b + 1], b - a, H, W, proj[a + 1:b + 1], out_ctot=1280, out_coff=512, ln=ln1(a + 1, b + 1))
                else:
                    for (a, b, t) in groups:
                        self.conv(blk["proj"][t], x[a:b], b - a, H, W, proj[a:b], ln=ln1(a, b))
                m = 1 if ego_only else n
                _lib.check(self.lib.av2x_hgt_attention_q(_ptr(proj), _ptr(mask), ctypes.cast(tarr, c_void_p), _ptr(att), n, m, hw,
                                                         self.cav["heads"], self.cav["dim_head"], st()), "av2x_hgt_attention")
                for (a, b, t) in (self._groups(types[:1]) if ego_only else groups):
                    self.conv(blk["aout"][t], att[a:b], b - a, H, W, x[a:b], residual=x[a:b])
                if trace is not None:
                    trace[f"hgt{di}"] = x.clone()
                # ---- x = SplitAttn(window attentions(LN(x))) + x
                self.conv(blk["qkv3"], x, m, H, W, qkv3, ln=self.ln_operand(x, m * hw, C, blk["ln2"][0], blk["ln2"][1], LN_EPS))
                for i, (h, dh, ws) in enumerate(zip(self.pw["heads"], self.pw["dim_head"], self.pw["window_size"])):
                    _lib.check(self.lib.av2x_window_attention(_ptr(qkv3), 2304, 768 * i, _ptr(blk["pos"][i]), _ptr(wat), m, H, W,
                                                              h, dh, ws, st()), "av2x_window_attention")
                    self.conv(blk["wout"][i], wat, m, H, W, br[i])
                _lib.check(self.lib.av2x_split_attn_gap(_ptr(br[0]), _ptr(br[1]), _ptr(br[2]), _ptr(gap), _ptr(gap_scratch), m, hw, C,
                                                        st()), "gap")
                if world > 1:       # equal strips: the global mean is the mean of the ranks' means
                    (self.gap_exchange or self._gap_allreduce)(gap[:m], world, di * len(blocks) + bi)
                if self.gap_record is not None:
                    self.gap_record.append(gap[:m].clone())
                self.conv(blk["fc1"], gap, m, 1, 1, g1)
                self.ln(g1, blk["bn1"], g2, m, C, relu=1)
                self.conv(blk["fc2"], g2, m, 1, 1, logits)
                _lib.check(self.lib.av2x_split_attn_combine(_ptr(br[0]), _ptr(br[1]), _ptr(br[2]), _ptr(logits), _ptr(x), _ptr(x),
                                                            m, hw, C, st()), "combine")
            # ---- x = FFN(LN(x)) + x
            m = 1 if (self.ego_only_last and di == last and trace is None and n > 1) else n
            self.conv(ffn["ff1"], x, m, H, W, hid, ln=self.ln_operand(x, m * hw, C, ffn["ln"][0], ffn["ln"][1], LN_EPS))
            self.conv(ffn["ff2"], hid, m, H, W, x, residual=x)
            if trace is not None:
                trace[f"layer{di}"] = x.clone()
        return x[0:1]

    # AMP mode: the outputs of every Linear and of the attention products are stored as bf16 (what torch.autocast stores for
    # nn.Linear / matmul; LayerNorm statistics, softmax, accumulators and the residual stream x stay fp32), on the token-panel
    # GEMM of csrc/linear_bf16.hip.  These layers are HBM-bound (K = 256), so bytes are what the mode saves.
    bf16_activations = True

    @property
    def act16_trunk(self):          # the trunk follows the fusion's storage mode
        return bool(self.bf16_activations)

    def lin16(self, L, a16, m_rows, out, residual=None, out_ctot=None, out_coff=0):
        """out = act(a16 (m_rows, 256) bf16 . W + b) (+ residual): ``out`` bf16 or fp32 (dtype decides)."""
        w, coutp = _w16i(L)
        octot = out_ctot if out_ctot is not None else L.cout
        esz = 2 if out.dtype == torch.bfloat16 else 4
        self.timed_hbm(f"linear_bf16 256->{L.cout}", m_rows * (L.cin * 2 + L.cout * esz + (L.cout * 4 if residual is not None else 0)) + w.numel() * 2,
                       2.0 * m_rows * L.cin * L.cout,
                       lambda: _lib.check(self.lib.av2x_linear_bf16(_ptr(a16), _ptr(w), _ptr(L.shift), _ptr(residual), _ptr(out), m_rows, L.cin,
                                                                    L.cout, coutp, 1 if esz == 2 else 0, octot, out_coff,
                                                                    L.cout if residual is not None else 0, 0, L.relu, self.stream()),
                                          "av2x_linear_bf16"))

    # LayerNorm (+ the pending residual add) folded into the panel load of the Linear that consumes it, FeedForward's two Linears in
    # one launch (csrc/linear_bf16.hip linear_bf16_occ_kernel<LN, FFN>): bit-identical to the separate launches
    fuse_ln = os.environ.get("AV2X_FUSE_LN", "1") != "0"
    fuse_window_out = os.environ.get("AV2X_FUSE_WINDOW_OUT", "1") != "0"
    # ... and LayerNorm -> QKV -> window attention -> to_out of a 4 x 16-pixel block in one workgroup (ln_qkv_window_out_slab_kernel)
    fuse_qkv_window = os.environ.get("AV2X_FUSE_QKV_WINDOW", "1") != "0"
    # ... and SplitAttn's combine computed in the FeedForward launch's panel load (linear_bf16_occ_kernel<SRC_LNC, FFN>)
    fuse_combine_ffn = os.environ.get("AV2X_FUSE_COMBINE_FFN", "1") != "0"

    def _wout3(self, blk):
        """the three to_out Linears as one packed 256 -> 768 weight (columns [256 b, 256 b + 256) = branch b) + the (768,) bias"""
        if "wout3" not in blk:
            ws = [_w16i(L)[0] for L in blk["wout"]]                      # each [32][256][8]: interleaving acts inside groups of 64 columns
            blk["wout3"] = (torch.cat(ws, -2).contiguous(), torch.cat([L.shift for L in blk["wout"]]).contiguous())
        return blk["wout3"]

    def ln_lin16(self, gb, L, x_rows, delta_rows, add_rows, m_rows, out, out_ctot=None, out_coff=0, L2=None, write_back=True):
        """out = Linear(LayerNorm(x_rows (+ delta_rows on the first add_rows rows, written back to x_rows if ``write_back``))) [-> second
        Linear L2]"""
        w, coutp = _w16i(L)
        w2 = _w16i(L2)[0] if L2 is not None else None
        octot = out_ctot if out_ctot is not None else (L2.cout if L2 is not None else L.cout)
        nbytes = (m_rows * (L.cin * 4 + (L2.cout if L2 is not None else L.cout) * 2) + add_rows * (L.cin * 2 + (L.cin * 4 if write_back else 0))
                  + w.numel() * 2)
        flops = 2.0 * m_rows * L.cin * L.cout
        if L2 is not None:
            nbytes += w2.numel() * 2
            flops += 2.0 * m_rows * L2.cin * L2.cout
        self.timed_hbm(f"linear_bf16 ln+256->{L.cout}" + (f"->{L2.cout}" if L2 is not None else ""), nbytes, flops,
                       lambda: _lib.check(self.lib.av2x_ln_linear_bf16(
                           _ptr(x_rows), _ptr(delta_rows) if add_rows else c_void_p(0), add_rows, 1 if write_back else 0, _ptr(gb[0]), _ptr(gb[1]), LN_EPS, _ptr(w),
                           _ptr(L.shift), L.relu, L.cout, coutp, _ptr(w2) if L2 is not None else c_void_p(0),
                           _ptr(L2.shift) if L2 is not None else c_void_p(0), L2.relu if L2 is not None else 0, _ptr(out), octot, out_coff,
                           m_rows, self.stream()), "av2x_ln_linear_bf16"))

    def _blocks_bf16(self, x, mask, n, H, W, types, world, trace):
        """Every Linear writes bf16 (as under autocast); the residual adds `x + fn(x)` of the fp32 stream are folded into the NEXT
        LayerNorm pass (av2x_add_layernorm_bf16: one read-modify-write of x instead of one in the Linear's epilogue and a read
        in the LayerNorm), the last one into a plain add."""
        C, hw, st, bf = 256, H * W, self.stream, torch.bfloat16
        tarr = (c_int32 * n)(*types)
        xn = self.buf("vit16_xn", (n, H, W, C), bf)
        proj = self.buf("vit16_proj", (n, H, W, 1280), bf)
        att = self.buf("vit16_att", (n, H, W, C), bf)
        qkv3 = self.buf("vit16_qkv3", (n, H, W, 2304), bf)
        wat = self.buf("vit16_wat", (n, H, W, C), bf)
        br = [self.buf(f"vit16_br{i}", (n, H, W, C), bf) for i in range(3)]
        hid = self.buf("vit16_hid", (n, H, W, C), bf)
        delta = self.buf("vit16_delta", (n, H, W, C), bf)
        gap = self.buf("vit_gap", (n, 1, 1, C))
        gap_scratch = self.buf("vit_gap_scratch", (n, 128, C))
        g1 = self.buf("vit_g1", (n, 1, 1, C))
        g2 = self.buf("vit_g2", (n, 1, 1, C))
        logits = self.buf("vit_logits", (n, 1, 1, 3 * C))
        groups = self._groups(types)
        last = len(self.layers) - 1
        pending = [0]            # agents 0 .. pending-1 of `delta` still have to be added to x

        def add_ln(gb, m_tok_agents):
            """x[:pending] += delta[:pending]; xn[:m] = LayerNorm(x[:m])"""
            k = pending[0]
            gp, bp = (_ptr(gb[0]), _ptr(gb[1])) if gb is not None else (c_void_p(0), c_void_p(0))
            if k:
                _lib.check(self.lib.av2x_add_layernorm_bf16(_ptr(x), _ptr(delta), gp, bp, _ptr(xn) if gb is not None else c_void_p(0),
                                                            min(k, m_tok_agents) * hw, C, LN_EPS, st()), "av2x_add_layernorm_bf16")
                if k > m_tok_agents:     # agents whose LayerNorm is not needed any more still receive their residual
                    _lib.check(self.lib.av2x_add_layernorm_bf16(_ptr(x[m_tok_agents:k]), _ptr(delta[m_tok_agents:k]), c_void_p(0), c_void_p(0),
                                                                c_void_p(0), (k - m_tok_agents) * hw, C, LN_EPS, st()), "av2x_add_layernorm_bf16")
            if gb is not None and m_tok_agents > k:
                _lib.check(self.lib.av2x_add_layernorm_bf16(_ptr(x[k:m_tok_agents]), c_void_p(0), gp, bp, _ptr(xn[k:m_tok_agents]),
                                                            (m_tok_agents - k) * hw, C, LN_EPS, st()), "av2x_add_layernorm_bf16")
            pending[0] = 0

        fuse = self.fuse_ln

        def ln_lin(gb, L, a, b, out, out_ctot=None, out_coff=0, L2=None, write_back=True):
            """agents [a, b): pending residual + LayerNorm + Linear(s) in one launch (no xn in HBM)"""
            self.ln_lin16(gb, L, x[a:b], delta[a:b], max(0, min(pending[0], b) - a) * hw, (b - a) * hw, out, out_ctot, out_coff, L2, write_back)

        def finish_pending(m_cov):
            """agents [m_cov, pending) were not covered by the fused launches: they still receive their residual"""
            if pending[0] > m_cov:
                _lib.check(self.lib.av2x_add_layernorm_bf16(_ptr(x[m_cov:pending[0]]), _ptr(delta[m_cov:pending[0]]), c_void_p(0), c_void_p(0),
                                                            c_void_p(0), (pending[0] - m_cov) * hw, C, LN_EPS, st()), "av2x_add_layernorm_bf16")
            pending[0] = 0

        for di, (blocks, ffn) in enumerate(self.layers):
            for bi, blk in enumerate(blocks):
                ego_only = self.ego_only_last and di == last and bi == len(blocks) - 1 and trace is None and n > 1
                # ---- x = HGT(LN(x)) + x
                if not fuse:
                    add_ln(blk["ln1"], n)
                if ego_only:
                    if fuse:
                        ln_lin(blk["ln1"], blk["proj"][types[0]], 0, 1, proj[0:1])
                    else:
                        self.lin16(blk["proj"][types[0]], xn[0:1], hw, proj[0:1])
                    for (a, b, t) in self._groups(types[1:]):
                        if fuse:
                            ln_lin(blk["ln1"], blk["proj_kv"][t], a + 1, b + 1, proj[a + 1:b + 1], out_ctot=1280, out_coff=512)
                        else:
                            self.lin16(blk["proj_kv"][t], xn[a + 1:b + 1], (b - a) * hw, proj[a + 1:b + 1], out_ctot=1280, out_coff=512)
                else:
                    for (a, b, t) in groups:
                        if fuse:
                            ln_lin(blk["ln1"], blk["proj"][t], a, b, proj[a:b])
                        else:
                            self.lin16(blk["proj"][t], xn[a:b], (b - a) * hw, proj[a:b])
                if fuse:
                    finish_pending(n)
                m = 1 if ego_only else n
                self.timed_hbm("hgt_attention_bf16", hw * 2 * (m * 512 + n * (256 + 256 * len(set(types[:m]))) + m * 256), 4.0 * m * n * hw * 256,
                               lambda: _lib.check(self.lib.av2x_hgt_attention_bf16(_ptr(proj), _ptr(mask), ctypes.cast(tarr, c_void_p), _ptr(att), n, m,
                                                                                   hw, self.cav["heads"], self.cav["dim_head"], st()),
                                                  "av2x_hgt_attention_bf16"))
                for (a, b, t) in (self._groups(types[:1]) if ego_only else groups):
                    self.lin16(blk["aout"][t], att[a:b], (b - a) * hw, delta[a:b])
                pending[0] = m
                if trace is not None:
                    add_ln(None, 0)
                    trace[f"hgt{di}"] = x.clone()
                # ---- x = SplitAttn(window attentions(LN(x))) + x
                pwc = list(zip(self.pw["heads"], self.pw["dim_head"], self.pw["window_size"]))
                mega = (fuse and self.fuse_qkv_window and H % 4 == 0 and W % 16 == 0 and len(pwc) == 3
                        and all(h_ * dh_ == 256 and (dh_, ws_) in ((16, 2), (32, 4), (64, 4)) for h_, dh_, ws_ in pwc))
                if mega:
                    assert pending[0] in (0, m)
                    w3, b3 = self._wout3(blk)
                    Lq = blk["qkv3"]
                    posv = (c_void_p * 3)(*[t_.data_ptr() for t_ in blk["pos"]])
                    outv = (c_void_p * 3)(*[t_.data_ptr() for t_ in br])
                    hv, dv, wv = ((c_int32 * 3)(*[c_[k_] for c_ in pwc]) for k_ in range(3))
                    self.timed_hbm("ln_qkv_window_out_bf16", m * hw * (1024 + (512 if pending[0] else 0) + 3 * 512) + (2304 + 768) * 512,
                                   2.0 * m * hw * 256 * (2304 + 768) + sum(4.0 * m * hw * ws_ * ws_ * 256 for _, _, ws_ in pwc),
                                   lambda: _lib.check(self.lib.av2x_ln_qkv_window_attention_bf16(
                                       _ptr(x), _ptr(delta) if pending[0] else c_void_p(0), _ptr(blk["ln2"][0]), _ptr(blk["ln2"][1]), LN_EPS,
                                       _ptr(_w16i(Lq)[0]), _ptr(Lq.shift), _ptr(w3), _ptr(b3), ctypes.cast(posv, c_void_p), ctypes.cast(outv, c_void_p),
                                       ctypes.cast(hv, c_void_p), ctypes.cast(dv, c_void_p), ctypes.cast(wv, c_void_p), m, H, W, st()),
                                       "av2x_ln_qkv_window_attention_bf16"))
                elif fuse:
                    # the HGT residual (pending == m agents here) feeds the LayerNorm but is NOT written back to x by this write-bound
                    # launch: the combine kernel below reads x anyway and adds it there (a 16-bit read instead of an fp32 write)
                    assert pending[0] in (0, m)
                    ln_lin(blk["ln2"], blk["qkv3"], 0, m, qkv3, write_back=False)
                else:
                    add_ln(blk["ln2"], m)
                    self.lin16(blk["qkv3"], xn, m * hw, qkv3)
                for i, (h, dh, ws) in enumerate([] if mega else pwc):
                    if self.fuse_window_out and H % 4 == 0 and W % 16 == 0 and h * dh == 256 and (dh, ws) in ((16, 2), (32, 4), (64, 4)):
                        # attention + its output projection in one launch: a rule of the map shape, same bits as the two launches
                        L = blk["wout"][i]
                        self.timed_hbm(f"window_attention_linear_bf16 ws{ws} dh{dh}", m * hw * 2 * (768 + 256) + 256 * 256 * 2,
                                       4.0 * m * hw * ws * ws * 256 + 2.0 * m * hw * 256 * 256,
                                       lambda: _lib.check(self.lib.av2x_window_attention_linear_bf16(
                                           _ptr(qkv3), 2304, 768 * i, _ptr(blk["pos"][i]), _ptr(_w16i(L)[0]), _ptr(L.shift), _ptr(br[i]), C, 0,
                                           m, H, W, h, dh, ws, st()), "av2x_window_attention_linear_bf16"))
                        continue
                    self.timed_hbm(f"window_attention_bf16 ws{ws} dh{dh}", m * hw * 2 * (768 + 256), 4.0 * m * hw * ws * ws * 256,
                                   lambda: _lib.check(self.lib.av2x_window_attention_bf16(_ptr(qkv3), 2304, 768 * i, _ptr(blk["pos"][i]), _ptr(wat),
                                                                                          m, H, W, h, dh, ws, st()), "av2x_window_attention_bf16"))
                    self.lin16(blk["wout"][i], wat, m * hw, br[i])
                _lib.check(self.lib.av2x_split_attn_gap_bf16(_ptr(br[0]), _ptr(br[1]), _ptr(br[2]), _ptr(gap), _ptr(gap_scratch), m, hw, C,
                                                             st()), "gap")
                if world > 1:
                    (self.gap_exchange or self._gap_allreduce)(gap[:m], world, di * len(blocks) + bi)
                if self.gap_record is not None:
                    self.gap_record.append(gap[:m].clone())
                self.conv(blk["fc1"], gap, m, 1, 1, g1)
                self.ln(g1, blk["bn1"], g2, m, C, relu=1)
                self.conv(blk["fc2"], g2, m, 1, 1, logits)
                defer_combine = (fuse and self.fuse_combine_ffn and trace is None and bi == len(blocks) - 1 and hw % 64 == 0
                                 and ffn["ff1"].cout == 256 and ffn["ff2"].cout == 256)
                if defer_combine:
                    pass        # the FeedForward launch below produces x = combine(...) in its panel load (one pass over x less)
                elif pending[0]:
                    _lib.check(self.lib.av2x_split_attn_combine_delta_bf16(_ptr(br[0]), _ptr(br[1]), _ptr(br[2]), _ptr(logits), _ptr(x), _ptr(delta),
                                                                           _ptr(x), m, hw, C, st()), "combine")
                    pending[0] = 0
                else:
                    _lib.check(self.lib.av2x_split_attn_combine_bf16(_ptr(br[0]), _ptr(br[1]), _ptr(br[2]), _ptr(logits), _ptr(x), _ptr(x),
                                                                     m, hw, C, st()), "combine")
            # ---- x = FFN(LN(x)) + x
            m = 1 if (self.ego_only_last and di == last and trace is None and n > 1) else n
            if blocks and defer_combine:
                assert pending[0] in (0, m)
                L1, L2 = ffn["ff1"], ffn["ff2"]
                w1, cp1 = _w16i(L1)
                add = pending[0] * hw
                self.timed_hbm("linear_bf16 combine+ln+256->256->256", m * hw * (1024 * 2 + 3 * 512 + 512) + add * 512 + 2 * 256 * 512,
                               4.0 * m * hw * 256 * 256,
                               lambda: _lib.check(self.lib.av2x_combine_ln_linear_bf16(
                                   _ptr(x), _ptr(delta) if add else c_void_p(0), add, _ptr(br[0]), _ptr(br[1]), _ptr(br[2]), _ptr(logits), hw,
                                   _ptr(ffn["ln"][0]), _ptr(ffn["ln"][1]), LN_EPS, _ptr(w1), _ptr(L1.shift), L1.relu, L1.cout, cp1,
                                   _ptr(_w16i(L2)[0]), _ptr(L2.shift), L2.relu, _ptr(delta), C, 0, m * hw, st()), "av2x_combine_ln_linear_bf16"))
                pending[0] = 0
            elif fuse and ffn["ff1"].cout == 256 and ffn["ff2"].cout == 256:
                ln_lin(ffn["ln"], ffn["ff1"], 0, m, delta, L2=ffn["ff2"])     # delta rows are read (pending) before they are written: per panel
                finish_pending(m)
            else:
                add_ln(ffn["ln"], m)
                self.lin16(ffn["ff1"], xn, m * hw, hid)
                self.lin16(ffn["ff2"], hid, m * hw, delta)
            pending[0] = m
            if trace is not None:
                add_ln(None, 0)
                trace[f"layer{di}"] = x.clone()
        add_ln(None, 0)
        return x[0:1]

    shard_group = None

    def _gap_allreduce(self, gap, world, index):
        import torch.distributed as dist
        dist.all_reduce(gap, op=dist.ReduceOp.SUM, group=self.shard_group)
        gap.mul_(1.0 / world)

    @torch.no_grad()
    def shard_local_stage(self, data_dict_local, has_ego, n_pad=None):
        """Per-rank half of an agent-sharded frame: encoders + backbone + shrink header of THIS rank's
        agents straight into the all-gather send buffer (n_pad,H,W,256): 36.0 MB per agent at the default grid
        (n_pad >= the local count pads an uneven frame's message, sharded.py).
        ``data_dict_local`` carries the frame-level ``prior_encoding`` / ``spatial_correction_matrix`` (host
        metadata of ALL agents, (1,L,.)) even on a rank without agents; stats = [0, canvas non-zeros] (summed over
        ranks = comm_rate)."""
        n, record_len, slots = self.shard_frame_agents(data_dict_local)
        n_pad = n if n_pad is None else int(n_pad)
        if n_pad < max(n, 1):
            raise ValueError(f"n_pad = {n_pad} is smaller than this rank's {n} agents")
        if n > 0:
            canvas, ny, nx = self.encode(data_dict_local, record_len, slots)
        else:
            ny, nx = self.canvas_dims()
        dims = self.level_dims(ny, nx)
        H, Wd = self.cat_hw(dims)
        # with a NaiveCompressor (airv2x_v2xvit.py:42-44, 122-123) the message is its ENCODER's output: 256 / ratio channels (36.0 / ratio MB per
        # agent at the default grid, half of that under autocast); the two decoder layers run on the receiving side (_gathered_maps)
        cm = self.compressor[0].cout if getattr(self, "compression", 0) else 256
        send = self.buf("shard_send", (n_pad * H * Wd * cm,), self.msg_dtype())      # autocast: the bf16 message, 18.0 MB per agent
        meta = {"n_loc": n_pad, "H": H, "W": Wd, "cm": cm,
                "prior": data_dict_local["prior_encoding"][0].detach().cpu().numpy(),
                "scm": data_dict_local["spatial_correction_matrix"][0].detach().cpu().numpy()}
        if n == 0:
            return send, torch.zeros(2, dtype=torch.int64, device=self.device), meta
        st = self.stream()
        nz = self.count_canvas(canvas, st)
        if cm != 256:
            s = self.buf("shard_shrink", (n, H, Wd, 256))
            if self.msg_dtype() == torch.bfloat16:    # as forward(): the shrink header's output is stored as bf16 and widened into the fp32 stream
                s16 = self.buf("shard_shrink16", (n, H, Wd, 256), torch.bfloat16)
                self.trunk(canvas, n, ny, nx, shrink_out=s16)
                self.widen(s16, s)
            else:
                self.trunk(canvas, n, ny, nx, shrink_out=s)
            self.conv(self.compressor[0], s, n, H, Wd, send[:n * H * Wd * cm].view(n, H, Wd, cm))
        else:
            self.trunk(canvas, n, ny, nx, shrink_out=send[:n * H * Wd * 256].view(n, H, Wd, 256))
        stats = torch.stack([torch.zeros((), dtype=torch.int64, device=self.device), nz[0]])
        return send, stats, meta

    def _gathered_maps(self, recv, meta, world):
        """(N,H,W,256) maps of the real agents in frame order: the gathered buffer itself, or (uneven frame) its valid
        slots compacted into a workspace."""
        n_loc, H, Wd = meta["n_loc"], meta["H"], meta["W"]
        cm = meta.get("cm", 256)
        counts = meta.get("counts") or [n_loc] * world
        N = sum(counts)
        if recv.numel() != world * n_loc * H * Wd * cm or len(counts) != world or max(counts) > n_loc:
            raise ValueError("gathered buffer has the wrong size")
        if N > self.L:
            raise ValueError(f"{N} agents exceed max_cav_num = {self.L}")
        if cm != 256:       # compressed message: the real agents' encoder outputs (compacted when the frame is uneven) through the decoder
            msg = recv.view(world * n_loc, H, Wd, cm)
            if N != world * n_loc:
                from .sharded import valid_slots
                cmp = self.buf("shard_compact", (N, H, Wd, cm), recv.dtype)
                for a, slot in enumerate(valid_slots(counts, n_loc)):
                    cmp[a].copy_(msg[slot])
                msg = cmp
            mid = self.buf("compress_mid", (N, H, Wd, 256))
            dec = self.buf("compress_dec", (N, H, Wd, 256))
            self.conv(self.compressor[1], msg, N, H, Wd, mid)
            self.conv(self.compressor[2], mid, N, H, Wd, dec)
            return dec, N, H, Wd
        maps = recv.view(world * n_loc, H, Wd, 256)
        if maps.dtype == torch.bfloat16:        # autocast message: widened into the fp32 stream (and compacted on the way)
            from .sharded import valid_slots
            wide = self.buf("shard_maps32", (N, H, Wd, 256))
            if N == world * n_loc:
                self.widen(maps, wide)
            else:
                for a, slot in enumerate(valid_slots(counts, n_loc)):
                    self.widen(maps[slot], wide[a])
            return wide, N, H, Wd
        if N != world * n_loc:
            from .sharded import valid_slots
            cmp = self.buf("shard_compact", (N, H, Wd, 256))
            for a, slot in enumerate(valid_slots(counts, n_loc)):
                cmp[a].copy_(maps[slot])
            maps = cmp
        return maps, N, H, Wd

    @torch.no_grad()
    def shard_ego_stage(self, recv, stats, meta, world, trace=None, sync_comm_rate=False):
        """Ego half: the gathered (N,H,W,256) buffer is in frame order and is consumed in place by the encoder."""
        maps, N, H, Wd = self._gathered_maps(recv, meta, world)
        fused = self.encoder(maps, N, H, Wd, meta["prior"], meta["scm"], trace)
        heads = torch.empty((1, self.heads.cout, H, Wd), dtype=torch.float32, device=self.device)
        self.conv(self.heads, fused, 1, H, Wd, heads)
        outs = torch.split(heads, self.head_splits, dim=1)
        out = {"psm": outs[0], "rm": outs[1]}
        if self.args["obj_head"]:
            out["obj"] = outs[2]
        out["comm_rate"] = int(stats[1].item()) if sync_comm_rate else stats[1]
        return out

    def fusion_strip(self, W, world, rank):
        """Equal column strips aligned to the 4-column windows, or None when the map does not split evenly."""
        if world <= 1 or W % (4 * world):
            return None
        Wc = W // world
        return (rank * Wc, Wc, world)

    @torch.no_grad()
    def shard_ego_partial(self, recv, stats, meta, world, rank, fusion_world=None, fusion_rank=None):
        """This rank's column strip of the fusion (one tiny all-reduce per block for the split-attention mean) + heads.
        ``fusion_world`` / ``fusion_rank`` default to the agent-sharding world / rank (tests split differently)."""
        maps, N, H, Wd = self._gathered_maps(recv, meta, world)
        fw, fr = (world, rank) if fusion_world is None else (fusion_world, fusion_rank)
        strip = self.fusion_strip(Wd, fw, fr)
        if strip is None:
            raise ValueError(f"map width {Wd} does not split into {fw} strips of whole windows")
        fused = self.encoder(maps, N, H, Wd, meta["prior"], meta["scm"], strip=strip)
        Wc = strip[1]
        heads = torch.empty((1, self.heads.cout, H, Wc), dtype=torch.float32, device=self.device)
        self.conv(self.heads, fused, 1, H, Wc, heads)
        return heads.view(-1), {"H": H, "W": Wd, "Wc": Wc, "stats": stats}

    @torch.no_grad()
    def shard_ego_finish(self, parts, ctx, world, sync_comm_rate=False):
        H, W, Wc = ctx["H"], ctx["W"], ctx["Wc"]
        nh = self.heads.cout
        full = torch.empty((1, nh, H, W), dtype=torch.float32, device=self.device)
        full.view(nh, H, world, Wc).copy_(parts.view(world, nh, H, Wc).permute(1, 2, 0, 3))     # strips side by side
        outs = torch.split(full, self.head_splits, dim=1)
        out = {"psm": outs[0], "rm": outs[1]}
        if self.args["obj_head"]:
            out["obj"] = outs[2]
        st = ctx["stats"]
        out["comm_rate"] = int(st[1].item()) if sync_comm_rate else st[1]
        return out

    @torch.no_grad()
    def forward(self, data_dict, trace=None, sync_comm_rate=False):
        if not self.weights_ready:
            raise RuntimeError("load_state_dict() must be called before forward()")
        record_len, slots = self.frame_layout(data_dict)
        B, n_total = len(record_len), sum(record_len)
        if max(record_len) > self.L:
            raise ValueError(f"{max(record_len)} agents exceed max_cav_num = {self.L}")
        # host metadata first: a device -> host copy waits for everything queued on the stream, so it is read BEFORE this frame's
        # encoders and trunk are launched, not between the trunk and the fusion
        prior_all = data_dict["prior_encoding"].detach().cpu().numpy()           # (B,L,3) per-agent scalars (appendix A #12)
        scm_all = data_dict["spatial_correction_matrix"].detach().cpu().numpy()  # (B,L,4,4) f64
        canvas, ny, nx = self.encode(data_dict, record_len, slots)
        st = self.stream()
        nz = self.count_canvas(canvas, st)
        dims = self.level_dims(ny, nx)
        H, Wd = self.cat_hw(dims)
        x = self.buf("vit_x", (n_total, H, Wd, 256))
        if self.msg_dtype() == torch.bfloat16:   # the shrink header's output is the (bf16) message: same rounding as the sharded frame
            x16 = self.buf("vit_x16", (n_total, H, Wd, 256), torch.bfloat16)
            self.trunk(canvas, n_total, ny, nx, shrink_out=x16)
            self.widen(x16, x)
        else:
            self.trunk(canvas, n_total, ny, nx, shrink_out=x)                    # all agents of the batch at once
        if self.compression:                                                     # airv2x_v2xvit.py:122-123
            self.run_compressor(x, n_total, H, Wd)
        fused_all = self.buf("vit_fused", (B, H, Wd, 256))
        off = 0
        for b, n in enumerate(record_len):                                       # the fusion never mixes samples
            fused = self.encoder(x[off:off + n], n, H, Wd, prior_all[b], scm_all[b], trace if B == 1 else None)
            if B == 1:
                fused_all = fused
            else:
                fused_all[b:b + 1].copy_(fused)
            off += n
        heads = torch.empty((B, self.heads.cout, H, Wd), dtype=torch.float32, device=self.device)
        self.conv(self.heads, fused_all, B, H, Wd, heads)
        outs = torch.split(heads, self.head_splits, dim=1)
        out = {"psm": outs[0], "rm": outs[1]}
        if self.args["obj_head"]:
            out["obj"] = outs[2]
        out["comm_rate"] = int(nz[0].item()) if sync_comm_rate else nz[0].clone()
        return out
