"""Host-side driver of the Where2Comm-LiDAR hot path on one MI355X.

This is plumbing only: it owns device buffers (torch tensors used as raw HBM allocations),
packs the weights once, and enqueues the C-ABI kernels of include/airv2x_hip.h on torch's
current HIP stream.  All arithmetic happens inside libairv2x_hip.so; there is no CPU or
ATen fallback for any stage.

Schedule (eval mode; results identical to the reference's three backbone passes,
airv2x_where2com.py:117-179 + where2comm_fuse.py:198-263):

  canvas  <- pillar VFE + scatter                      (per agent type)
  b0,b1,b2 <- backbone blocks on all agents            (unmasked pass, ONCE)
  cat     <- deblocks(b0,b1,b2) written into one 384-channel NHWC buffer (no torch.cat)
  psm_single <- cls_head(shrink(cat))
  mask    <- communication(psm_single)                 (ego forced to 1)
  b0     *= mask                                       (in place; ego unchanged)
  b1m,b2m <- blocks 1,2 on the masked non-ego agents   (ego rows re-use b1,b2: mask == 1)
  f_i     <- per-pixel attention over agents at scale i (ego row)
  out     <- heads(shrink(deblocks(f_0,f_1,f_2)))
"""
from __future__ import annotations

import ctypes
import os
from ctypes import byref, c_float, c_void_p

import numpy as np
import torch

from .. import _lib
from .packing import fold_bn, interleave2_columns, pack_conv_weight, pack_deconv_weight, to_bf16_koct, to_bf16x3_koct

AGENT_TYPES = ("vehicle", "rsu", "drone")
TYPE_PREFIX = {"vehicle": "veh_models", "rsu": "rsu_models", "drone": "drone_models"}


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None) if hasattr(torch._C, "_cuda_getDevice") else None


class ConvLayer:
    __slots__ = ("w", "scale", "shift", "cin", "cout", "coutp", "ks", "stride", "pad", "relu", "mode", "up", "_w16", "_w3", "_wu", "_w16i", "_wu4", "_w16h", "_wu3", "_wu43")

    def __init__(self, w, scale, shift, cin, cout, coutp, ks, stride, pad, relu, mode=_lib.AV2X_CONV, up=1):
        self.w, self.scale, self.shift = w, scale, shift
        self._w16 = None
        self._w3 = None
        self._wu = None
        self._w16i = None
        self._wu4 = None
        self._wu3 = None
        self._wu43 = None
        self._w16h = None
        self.cin, self.cout, self.coutp = cin, cout, coutp
        self.ks, self.stride, self.pad, self.relu, self.mode, self.up = ks, stride, pad, relu, mode, up


def _w16(L):
    """bf16 k-oct packing of the layer's weights (AMP mode), built from the fp32 packing on first use."""
    if L._w16 is None:
        L._w16 = to_bf16_koct(L.w)
    return L._w16


def _w16i(L):
    """(bf16 k-oct packing with the columns interleaved for av2x_linear_bf16, padded column count), built on first use."""
    if L._w16i is None:
        L._w16i = interleave2_columns(_w16(L))
    return L._w16i


def _w16h(L):
    """(bf16 k-oct packing with the columns zero-padded to a multiple of 128, that column count): what the halo-tile direct convolution
    on bf16 activations walks (csrc/conv_halo_bf16.inc)."""
    if L._w16h is None:
        w = _w16(L)
        T, q, n, eight = w.shape
        npad = (n + 127) // 128 * 128
        if npad != n:
            w = torch.cat([w, torch.zeros(T, q, npad - n, 8, dtype=w.dtype, device=w.device)], 2).contiguous()
        L._w16h = (w, npad)
    return L._w16h


def _w3(L):
    """split-3 bf16 planes (hi, mid, lo) of the layer's weights, built from the fp32 packing on first use."""
    if L._w3 is None:
        L._w3 = to_bf16x3_koct(L.w)
    return L._w3


def _wu(L, lib, stream):
    """Winograd-transformed weights G g G^T in the k-quad packing per position (av2x_wino_pack_weights), built on the device
    from the fp32 packing on first use."""
    if L._wu is None:
        u = torch.empty(lib.av2x_wino_weight_bytes(L.cin, L.coutp) // 4, dtype=torch.float32, device=L.w.device)
        _lib.check(lib.av2x_wino_pack_weights(c_void_p(L.w.data_ptr()), L.cin, L.coutp, c_void_p(u.data_ptr()), stream),
                   "av2x_wino_pack_weights")
        # once per layer: engines that share these weights launch on OTHER streams (FramePipeline, ShardedPipeline), so the
        # transformed copy must be complete before anyone else can see it
        torch.cuda.current_stream().synchronize()
        L._wu = u
    return L._wu


def _wu4(L, lib, stream):
    """Winograd F(4x4,3x3)-transformed weights (av2x_wino4_pack_weights: 36 positions), built on the device on first use."""
    if L._wu4 is None:
        u = torch.empty(lib.av2x_wino4_weight_bytes(L.cin, L.coutp) // 4, dtype=torch.float32, device=L.w.device)
        _lib.check(lib.av2x_wino4_pack_weights(c_void_p(L.w.data_ptr()), L.cin, L.coutp, c_void_p(u.data_ptr()), stream),
                   "av2x_wino4_pack_weights")
        torch.cuda.current_stream().synchronize()      # as _wu: other streams may launch with it next
        L._wu4 = u
    return L._wu4


def _wu3(L, lib, stream):
    """Winograd F(2x2,3x3)-transformed weights as split-3 bf16 planes (av2x_wino_x3_pack_weights: G g G^T in fp64, hi / mid / lo there),
    built on the device on first use."""
    if L._wu3 is None:
        u = torch.empty(lib.av2x_wino_x3_weight_bytes(L.cin, L.coutp) // 2, dtype=torch.bfloat16, device=L.w.device)
        _lib.check(lib.av2x_wino_x3_pack_weights(c_void_p(L.w.data_ptr()), L.cin, L.coutp, c_void_p(u.data_ptr()), stream),
                   "av2x_wino_x3_pack_weights")
        torch.cuda.current_stream().synchronize()      # as _wu: other streams may launch with it next
        L._wu3 = u
    return L._wu3


def _wu43(L, lib, stream):
    """Winograd F(4x4,3x3)-transformed weights as split-3 bf16 planes (av2x_wino4_x3_pack_weights: 36 positions, G g G^T in fp64, hi / mid /
    lo there), built on the device on first use."""
    if L._wu43 is None:
        u = torch.empty(lib.av2x_wino4_x3_weight_bytes(L.cin, L.coutp) // 2, dtype=torch.bfloat16, device=L.w.device)
        _lib.check(lib.av2x_wino4_x3_pack_weights(c_void_p(L.w.data_ptr()), L.cin, L.coutp, c_void_p(u.data_ptr()), stream),
                   "av2x_wino4_x3_pack_weights")
        torch.cuda.current_stream().synchronize()      # as _wu: other streams may launch with it next
        L._wu43 = u
    return L._wu43


class LnOperand:
    """nn.LayerNorm over the channel axis of the first ``n_tokens`` rows of ``x`` (tokens x c, fp32), handed to Where2ComEngine.conv(ln=...).
    Nothing is launched until a consumer asks: a kernel class that normalises while it loads its rows (av2x_conv2d_ln) takes ``stats()`` --
    the per-token (mean, rstd) pass, run once --; any other class takes ``materialised()`` -- one av2x_layernorm launch whose result every
    consumer of this operand shares (CoBEVT's q and k|v Linears read the same normalised rows).  ``rows(a, b)`` is the operand of a token
    range (V2X-ViT's per-agent-type projections)."""

    def __init__(self, eng, x, n_tokens, c, gamma, beta, eps, tag="ln_stats", lo=0, hi=None, root=None):
        self.eng, self.x, self.n_tokens, self.c, self.gamma, self.beta, self.eps, self.tag = eng, x, int(n_tokens), int(c), gamma, beta, eps, tag
        self.lo, self.hi, self.root = lo, (int(n_tokens) if hi is None else hi), root
        self._stats = self._xn = None

    def rows(self, a, b):
        r = self.root or self
        if not (0 <= a <= b <= self.hi - self.lo):
            raise ValueError(f"LayerNorm operand rows [{a}, {b}) outside its {self.hi - self.lo} tokens")
        return LnOperand(self.eng, self.x, b - a, self.c, self.gamma, self.beta, self.eps, self.tag, self.lo + a, self.lo + b, r)

    def stats(self):
        r = self.root or self
        if r._stats is None:
            r._stats = r.eng.ln_stats(r.x, r.n_tokens, r.c, r.eps, r.tag)
        return r._stats[self.lo:self.hi] if self.root is not None else r._stats

    def materialised(self):
        r = self.root or self
        if r._xn is None:
            e = r.eng
            r._xn = e.buf("ln_materialised_" + r.tag, (r.n_tokens, r.c))
            _lib.check(e.lib.av2x_layernorm(_ptr(r.x), _ptr(r.gamma), _ptr(r.beta), _ptr(r._xn), r.n_tokens, r.c, r.eps, e.stream()), "av2x_layernorm")
        return r._xn[self.lo:self.hi] if self.root is not None else r._xn


def _ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def frame_layout(collaborators, data_dict):
    """Agent order of the reference's repack_batch (airv2x_base_model.py:209-236):
    sample-major, then vehicle/rsu/drone, then index inside the type.
    Returns (record_len list, {type: canvas slot of each of that type's agents, in the order
    of that type's own agent index = voxel_coords[:, 0]})."""
    per_type = {}
    B = 0
    for t in AGENT_TYPES:
        d = data_dict.get(t)
        if t not in collaborators or d is None or len(d["batch_idxs"]) == 0:
            continue
        rl = d["record_len"]
        rl = [int(v) for v in (rl.tolist() if hasattr(rl, "tolist") else rl)]
        per_type[t] = rl
        B = max(B, len(rl))
    record_len, slots = [0] * B, {t: [] for t in per_type}
    nxt = 0
    for b in range(B):
        for t in AGENT_TYPES:
            if t in per_type and b < len(per_type[t]) and b in data_dict[t]["batch_idxs"]:
                k = per_type[t][b]
                slots[t] += list(range(nxt, nxt + k))
                nxt += k
                record_len[b] += k
    return record_len, slots


WINO_X3_MIN_CIN = int(os.environ.get("AV2X_WINO_X3_MIN_CIN", "64"))    # 128 = the round-4 rule (block-0 layers on the fp32 Winograd kernels)


class Where2ComEngine:
    def __init__(self, args, device="cuda"):
        self.args = args
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("Where2ComEngine runs on a HIP device only (no CPU path exists)")
        self.lib = _lib.load()
        self._init_config(args)
        self.A, self.C = args["anchor_number"], args["num_class"]
        self.ws = {}
        # workspace pool: one buffer per (name, shape, dtype), re-used across frames.  A scenario stream changes its frame
        # layout (agent count) every few frames, so the pool is bounded: beyond ws_limit bytes the least recently used
        # buffers that the CURRENT frame has not touched are dropped (and re-allocated if that layout comes back)
        self.ws_limit = int(float(os.environ.get("AV2X_WS_LIMIT_GB", "32")) * (1 << 30))
        self._ws_bytes = 0
        self._ws_used = {}
        self._frame = 0
        self.weights_ready = False
        self.conv_tile = 0          # 0 = autotune / pick_tile(); else forced BM<<16|BN (tests / tuning)
        self.conv_sk_wgs = 0        # persistent workgroups when conv_tile carries the stream-K flag 0x2000
        self.autotune = True        # time the candidate tiles once per distinct conv shape, keep the fastest
        self.tile_cache = {}
        # Stream-K (a K-split changes the fp32 summation order, <= ~1e-5 relative, so WHICH schedule runs must not depend on
        # a wall-clock measurement):
        #   "rule" (default) a layer runs stream-K iff its shape says so (sk_rule(): fewer 64x64 tiles than the 768
        #                    persistent workgroups), always with the same tile and workgroup count -> the split points,
        #                    and therefore every output bit, are a function of the layer shape alone; the autotuner only
        #                    picks among implementations of that one schedule (register-staged / LDS-DMA: bit-identical);
        #   False / "off"    data-parallel schedules only (results do not depend on how many agents share a launch:
        #                    what the agent-sharded frame is compared against bit for bit);
        #   "tune"           legacy: stream-K candidates compete by wall-clock (not reproducible across processes).
        self.stream_k = {"0": False, "off": False, "1": "rule", "rule": "rule", "tune": "tune"}[os.environ.get("AV2X_STREAM_K", "rule")]
        # Winograd F(2x2,3x3) for the 3x3 / stride-1 layers (conv_wino.inc): 2.25x fewer matrix-core multiplies, still fp32
        # operands and accumulation, error against fp64 at or below the direct kernel's.  WHICH layers take it is a function
        # of the layer's channels alone (wino_rule), never of a timing, so results stay reproducible and do not depend on
        # how many agents share a launch.  AV2X_WINOGRAD=0: direct implicit GEMM everywhere.
        self.winograd = os.environ.get("AV2X_WINOGRAD", "1") not in ("0", "off")
        # several frames in flight (FramePipeline / ShardedPipeline): other frames' kernels already fill a layer's idle CUs, so
        # the quarter-position Winograd tiling (finer tasks, but twice the input-transform work) is left out of the tuner's
        # candidates.  The tilings are bit-identical, so this changes speed only.
        self.throughput_mode = False
        # AMP mode (what torch.autocast does to Conv2d / Linear): bf16 matrix-core operands, fp32 accumulation and
        # fp32 activations in HBM (conv_igemm_bf16); LayerNorm / softmax / attention stay fp32.  Off = exact fp32.
        self.amp = False
        # split-3 mode: fp32-ACCURATE Conv2d / Linear products on the bf16 matrix cores (every fp32 operand = three bf16
        # terms, six partial products, fp32 accumulation; conv_igemm_bf16x3).  Error vs fp64 is at or below the fp32-MFMA
        # kernel's (tools/split3_bench.py), results are not bit-identical to it.  Opt-in.
        self.split3 = False
        # Winograd F(2x2,3x3) with split-3 operands (csrc/conv_wino_x3.hip): the same 16-position algorithm, every fp32 operand as three
        # bf16 terms on v_mfma_f32_32x32x16_bf16 (2.67x fewer matrix cycles than the fp32-input MFMA, fp32-accurate products, error against
        # fp64 at or below the fp32 Winograd kernel's).  A rule of the layer and the map (wino_x3_rule), never of a timing or the agent
        # count.  ON by default since round 4 (AV2X_X3=0 / AV2X_WINO_X3=0 / bench.py --gemm f32 restore the fp32-input matrix cores): the
        # per-kernel error against fp64 is not above the fp32 Winograd kernel's on any tested shape and every model's goldens hold at their
        # unchanged tolerances (tests/test_gpu_wino_x3.py); results differ from the fp32-MFMA kernels in the last bits.
        x3_default = os.environ.get("AV2X_X3", "1")
        self.wino_x3 = os.environ.get("AV2X_WINO_X3", x3_default) not in ("0", "off", "")
        # ... and its companion for every OTHER convolution / Linear (1x1, strided, transposed): the pipelined split-3 implicit GEMM
        # (csrc/conv_x3p.hip; bit-identical to conv_igemm_bf16x3, 1.1-1.15x faster).  x3p alone leaves the 3x3 layers on the fp32 Winograd
        # kernels; wino_x3 + x3p = the "x3" mode of bench.py: every product of the frame formed from three bf16 terms per operand.
        self.x3p = os.environ.get("AV2X_X3P", x3_default) not in ("0", "off", "")
        # ... and the split-3 form of the F(4x4,3x3) class (csrc/conv_wino4_x3.hip): the layers wino4_rule selects run the 36-position
        # algorithm on the bf16 matrix cores too (needs wino_x3; error against fp64 at or below the fp32 F(4x4) kernel's)
        self.wino4_x3 = os.environ.get("AV2X_WINO4_X3", x3_default) not in ("0", "off", "")
        self.wino2_x3 = True        # the F(2x2,3x3) class follows wino_x3; the training runner keeps it on the fp32-input kernel (single-stream launches)
        self.use_graph = False      # replay everything after the scatter from a captured hipGraph
        self.graphs = {}
        self.profile = None         # list -> (tile, flops, ev0, ev1, workgroups, shape) per conv launch (bench roofline pass)
        self.profile_hbm = None     # list -> (kernel name, algorithmic HBM bytes, flops, ev0, ev1) per launch of an HBM-bound kernel
        self.agent_streams = 1      # >1 (B == 1): agents are split into this many groups that run the per-agent part
                                    # of the frame on separate HIP streams.  Measured: no gain, off by default
        self._streams = None
        self._desc = _lib.ConvDesc()

    def _init_config(self, args):
        self.bb = args["modality_fusion"]["base_bev_backbone"]
        # (BaseBEVBackbone's variants -- deblocks that down-sample, upsample_strides < 1, and the extra deblock on the concatenated map,
        # base_bev_backbone.py:87-121 -- change the resolution of the shared map: cat_hw() / trunk() follow it, and the communication mask
        # is brought to the first block's resolution as where2comm_fuse.py:229-235 does)
        self.sh = args["modality_fusion"]["shrink_header"]
        self.fcfg = args["where2com_fusion"]
        # multi_scale: false = the single-scale branch (where2comm_fuse.py:264-286, airv2x_where2com.py:163-166): the shrunk (and, with
        # a compressor, compressed + decompressed) 256-channel map is masked and fused once, the heads read the fused map directly
        self.multi_scale = bool(self.fcfg["multi_scale"])
        # NaiveCompressor (airv2x_where2com.py:50-52): switched on by modality_fusion.compression > 0, ratio from the TOP-LEVEL
        # ``compression`` key -- a KeyError when only the first is given, in the reference (:52) and here alike
        from ..synth import model_compression
        self.compression = model_compression(args)

    FUSION_WEIGHTS = ()   # attribute names of the packed fusion weights a subclass loads in _load_fusion
    compressor = None     # NaiveCompressor layers (CoBEVT: args["compression"]; V2X-ViT / When2com: synth.model_compression), else None

    def _load_compressor(self, sd, up, prefix="naive_compressor"):
        """NaiveCompressor (naive_compress.py:10-36): three Conv3x3(+bias)+BN(eps 1e-3)+ReLU layers; the conv bias
        is folded into the BN shift.  Layer 0 is the encoder (its C/ratio-channel output is the message a
        multi-GPU deployment would all-gather), layers 1-2 the decoder."""
        layers = []
        pre = prefix + "." if prefix else ""
        for conv, bn in (("encoder.0", "encoder.1"), ("decoder.0", "decoder.1"), ("decoder.3", "decoder.4")):
            w = sd[f"{pre}{conv}.weight"].detach().float()
            sc, sh = fold_bn(sd, f"{pre}{bn}")
            sh = sh + sd[f"{pre}{conv}.bias"].detach().float().cpu() * sc
            wp, coutp = pack_conv_weight(w)
            layers.append(ConvLayer(up(wp), up(sc), up(sh), w.shape[1], w.shape[0], coutp, 3, 1, 1, 1))
        return layers

    def run_compressor(self, x, n, H, W):
        """x (n,H,W,C) -> encoder -> decoder, result written back into x."""
        enc, dec0, dec1 = self.compressor
        msg = self.buf("compress_msg", (n, H, W, enc.cout), self.msg_dtype())     # autocast: the bf16 message the sharded frame sends
        self.conv(enc, x, n, H, W, msg)
        mid = self.buf("compress_mid", (n, H, W, dec0.cout))
        self.conv(dec0, msg, n, H, W, mid)
        self.conv(dec1, mid, n, H, W, x)
        return msg


    def share_weights(self):
        """A second engine on the same device that shares the packed weights but owns its workspaces:
        one engine per in-flight frame (FramePipeline)."""
        other = type(self)(self.args, self.device)
        for k in ("pfn", "blocks", "deblocks", "cat_c", "shrink", "feat_c", "cls_single", "head_splits", "heads",
                  "gauss_w", "gauss_b", "gauss_k", "threshold", "weights_ready", "compressor") + tuple(self.FUSION_WEIGHTS):
            if hasattr(self, k):
                setattr(other, k, getattr(self, k))
        if getattr(self, "cam", None):
            import copy as _copy
            other.cam = {}
            for t, c in self.cam.items():       # same packed weights, the other engine's workspace pool
                cc = _copy.copy(c)
                cc.eng = other
                other.cam[t] = cc
        else:
            other.cam = {}
        other.tile_cache = self.tile_cache
        other.autotune, other.conv_tile, other.stream_k, other.amp = self.autotune, self.conv_tile, self.stream_k, self.amp
        other.winograd, other.throughput_mode = self.winograd, self.throughput_mode
        other.sharded_frame = self.sharded_frame
        other.split3 = self.split3
        other.wino_x3 = self.wino_x3
        other.x3p = self.x3p
        other.wino4_x3 = self.wino4_x3
        other.wino2_x3 = self.wino2_x3
        return other

    def frame_mode(self, throughput, sharded):
        """Context: the engine in the given (throughput_mode, sharded_frame) for the calls inside, the caller's own flags restored afterwards.
        The flags select Winograd classes (wino4_rule), i.e. bits: a pipeline scopes ITS mode to ITS frames with this instead of leaving
        it on the engine the caller handed over (a later direct forward() of that engine keeps the bits it had before)."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            saved = (self.throughput_mode, self.sharded_frame)
            self.throughput_mode, self.sharded_frame = bool(throughput), bool(sharded)
            try:
                yield self
            finally:
                self.throughput_mode, self.sharded_frame = saved
        return scope()

    def graph_active(self):
        return self.use_graph and len(self.graphs) > 0

    @staticmethod
    def pick_tile(m, coutp):
        """Largest workgroup tile that still gives >= 2 workgroups per CU (256 CUs): the deep,
        low-resolution layers would otherwise leave most of the chip idle."""
        bn = 128 if coutp % 128 == 0 else (64 if coutp % 64 == 0 else 32)
        bm = 128
        wgs = lambda a, b: -(-m // a) * (coutp // b)
        if bn == 128 and wgs(128, 128) < 512:
            bn = 64
        if bn == 64 and wgs(128, 64) < 512:
            bm = 64
        return bm, bn

    # ------------------------------------------------------------------ weights
    def _up(self, t):
        return t.to(self.device).contiguous() if t is not None else None

    def load_pfn(self, sd, prefix, voxel_size, lidar_range):
        """PillarVFE weights under ``prefix`` -> (linear weight, folded BN scale / shift, HOST geometry array)."""
        up = self._up
        p = prefix + "pfn_layers.0"
        sc, sh = fold_bn(sd, p + ".norm")
        vs, rng = voxel_size, lidar_range
        geom = (c_float * 6)(vs[0], vs[1], vs[2], vs[0] / 2 + rng[0], vs[1] / 2 + rng[1], vs[2] / 2 + rng[2])
        return (up(sd[p + ".linear.weight"].detach().float()), up(sc), up(sh), geom)

    def load_encoders(self, sd):
        """Airv2xBase.init_encoders (airv2x_base_model.py:36-99): ``<type>_models.<i>`` is the encoder of the type's i-th modality --
        Sequential(PillarVFE, PointPillarScatter) for "lidar", LiftSplatShootEncoder for "cam" (opencood_iface/camera.py)."""
        self.pfn, self.cam = {}, {}
        for t in AGENT_TYPES:
            if t not in self.args["collaborators"]:
                continue
            for mi, m in enumerate(self.args[t]["modalities"]):
                if m == "lidar":
                    cfg = self.args[t]["lidar"]
                    self.pfn[t] = self.load_pfn(sd, f"{TYPE_PREFIX[t]}.{mi}.0.", cfg["voxel_size"], cfg["lidar_range"])
                elif m == "cam":
                    from .camera import CameraEncoder
                    self.cam[t] = CameraEncoder(self, self.args[t]["cam"], sd, f"{TYPE_PREFIX[t]}.{mi}.", t[:3])
                else:
                    raise NotImplementedError(f"Modality {m} not supported for {t}.")

    def load_backbone(self, sd, prefix="backbone.", input_channels=64):
        """BaseBEVBackbone weights (blocks: Conv3x3 + BN + ReLU chains, deblocks: ConvTranspose k = s + BN + ReLU)."""
        up = self._up
        self.blocks, self.deblocks = [], []
        cin = input_channels
        for i, (n, c, s) in enumerate(zip(self.bb["layer_nums"], self.bb["num_filters"], self.bb["layer_strides"])):
            layers = []
            idx = 1
            for li in range(n + 1):
                w, coutp = pack_conv_weight(sd[f"{prefix}blocks.{i}.{idx}.weight"])
                sc, sh = fold_bn(sd, f"{prefix}blocks.{i}.{idx + 1}")
                layers.append(ConvLayer(up(w), up(sc), up(sh), cin if li == 0 else c, c, coutp, 3,
                                        s if li == 0 else 1, 1, 1))
                idx += 3
            self.blocks.append(layers)
            cin = c
        nlev = len(self.bb["layer_nums"])
        for i, (s, cu) in enumerate(zip(self.bb["upsample_strides"][:nlev], self.bb["num_upsample_filter"])):
            sc, sh = fold_bn(sd, f"{prefix}deblocks.{i}.1")
            if s >= 1:
                w, ncol = pack_deconv_weight(sd[f"{prefix}deblocks.{i}.0.weight"])
                self.deblocks.append(ConvLayer(up(w), up(sc), up(sh), self.bb["num_filters"][i], cu, ncol, 1, 1, 0, 1,
                                               _lib.AV2X_DECONV, int(s)))
            else:   # a "deblock" that down-samples: Conv2d(k, stride k), k = round(1 / s)  (base_bev_backbone.py:87-105)
                k = int(round(1.0 / s))
                w, coutp = pack_conv_weight(sd[f"{prefix}deblocks.{i}.0.weight"])
                self.deblocks.append(ConvLayer(up(w), up(sc), up(sh), self.bb["num_filters"][i], cu, coutp, k, k, 0, 1))
        # the concatenated map holds the per-level deblocks' channels only.  The reference asserts len(upsample_strides) ==
        # len(num_upsample_filter) (base_bev_backbone.py:24-27) and sizes the extra ConvTranspose2d with sum(num_upsample_filters): a
        # filter entry beyond the levels would leave channels of the concat buffer unwritten here, so it is rejected instead
        nuf = list(self.bb["num_upsample_filter"])
        if len(nuf) > nlev and sum(nuf[nlev:]) != 0:
            raise ValueError(f"num_upsample_filter has {len(nuf)} entries for {nlev} levels: the entries beyond the levels must be absent or 0 "
                             "(the final deblock works on the concatenation of the per-level maps)")
        self.cat_c = sum(nuf[:nlev])
        self.final_deblock = None
        if len(self.bb["upsample_strides"]) > nlev:   # ConvTranspose2d on the concatenated map (:107-121, :151-152)
            s = int(self.bb["upsample_strides"][-1])
            w, ncol = pack_deconv_weight(sd[f"{prefix}deblocks.{nlev}.0.weight"])
            sc, sh = fold_bn(sd, f"{prefix}deblocks.{nlev}.1")
            self.final_deblock = ConvLayer(up(w), up(sc), up(sh), self.cat_c, self.cat_c, ncol, 1, 1, 0, 1, _lib.AV2X_DECONV, s)

    def load_resnet(self, sd, prefix, layer_nums, layer_strides, num_filters, inplanes=64):
        """ResNetModified(BasicBlock, ...) (coalign_modules/resblock.py:149-268, levels layer0, layer1, ...): per level a list of
        BasicBlocks = (conv3x3/s + BN + ReLU, conv3x3 + BN with the ReLU applied AFTER the residual add [activation code 5], optional
        1x1/s + BN on the identity)."""
        up = self._up
        self.res_layers, cin = [], int(inplanes)
        for li, (n, st, c) in enumerate(zip(layer_nums, layer_strides, num_filters)):
            blocks = []
            for j in range(n):
                q = f"{prefix}layer{li}.{j}."
                s_ = st if j == 0 else 1
                w1, cp1 = pack_conv_weight(sd[q + "conv1.weight"])
                w2, cp2 = pack_conv_weight(sd[q + "conv2.weight"])
                s1, h1 = fold_bn(sd, q + "bn1", 1e-5)
                s2, h2 = fold_bn(sd, q + "bn2", 1e-5)
                blk = {"c1": ConvLayer(up(w1), up(s1), up(h1), cin if j == 0 else c, c, cp1, 3, s_, 1, 1),
                       "c2": ConvLayer(up(w2), up(s2), up(h2), c, c, cp2, 3, 1, 1, 5), "down": None}
                if (q + "downsample.0.weight") in sd:
                    wd, cpd = pack_conv_weight(sd[q + "downsample.0.weight"])
                    sd_, hd = fold_bn(sd, q + "downsample.1", 1e-5)
                    blk["down"] = ConvLayer(up(wd), up(sd_), up(hd), cin, c, cpd, 1, s_, 0, 0)
                blocks.append(blk)
            self.res_layers.append(blocks)
            cin = c

    def run_resnet_layer(self, li, x, n, h, w, tag, out=None):
        """One level of the ResNet backbone on n images; returns (buffer, ho, wo)."""
        cur, ch, cw = x, h, w
        blocks = self.res_layers[li]
        for j, blk in enumerate(blocks):
            s_ = blk["c1"].stride
            ho, wo, c = (ch + 2 - 3) // s_ + 1, (cw + 2 - 3) // s_ + 1, blk["c1"].cout
            a = self.buf(f"res{li}_{j}a_{tag}", (n, ho, wo, c))
            self.conv(blk["c1"], cur, n, ch, cw, a)
            idt = cur
            if blk["down"] is not None:
                idt = self.buf(f"res{li}_{j}d_{tag}", (n, ho, wo, c))
                self.conv(blk["down"], cur, n, ch, cw, idt)
            o = out if (out is not None and j == len(blocks) - 1) else self.buf(f"res{li}_{j}o_{tag}", (n, ho, wo, c))
            self.conv(blk["c2"], a, n, ho, wo, o, residual=idt)
            cur, ch, cw = o, ho, wo
        return cur, ch, cw

    def load_shrink(self, sd, prefix="shrink_conv."):
        """DownsampleConv weights: per layer Conv(k) + ReLU, Conv3x3 + ReLU (biases, no BN).  Returns the output width."""
        up = self._up
        self.shrink = []
        cin = self.sh["input_dim"]
        if self.sh.get("use", True):
            for li, (k, d, s, pd) in enumerate(zip(self.sh["kernal_size"], self.sh["dim"], self.sh["stride"], self.sh["padding"])):
                if s != 1:
                    raise NotImplementedError("shrink_header stride != 1")
                p = f"{prefix}layers.{li}.double_conv"
                w0, cp0 = pack_conv_weight(sd[p + ".0.weight"])
                w1, cp1 = pack_conv_weight(sd[p + ".2.weight"])
                self.shrink.append(ConvLayer(up(w0), None, up(sd[p + ".0.bias"].detach().float()), cin, d, cp0, k, 1, pd, 1))
                self.shrink.append(ConvLayer(up(w1), None, up(sd[p + ".2.bias"].detach().float()), d, d, cp1, 3, 1, 1, 1))
                cin = d
        return cin

    def load_state_dict(self, sd):
        up = self._up
        self.load_encoders(sd)
        self.load_backbone(sd)
        cin = self.load_shrink(sd)
        self.feat_c = cin
        # cls head alone (per-agent confidence) and the three heads fused into one 30-column GEMM
        wc, cpc = pack_conv_weight(sd["cls_head.weight"])
        self.cls_single = ConvLayer(up(wc), None, up(sd["cls_head.bias"].detach().float()), cin, self.A * self.C, cpc, 1, 1, 0, 0)
        names = ["cls_head", "reg_head"] + (["obj_head"] if self.args["obj_head"] else [])
        wcat = torch.cat([sd[n + ".weight"].detach().float().cpu() for n in names], 0)
        bcat = torch.cat([sd[n + ".bias"].detach().float().cpu() for n in names], 0)
        wh, cph = pack_conv_weight(wcat)
        self.head_splits = [sd[n + ".weight"].shape[0] for n in names]
        self.heads = ConvLayer(up(wh), None, up(bcat), cin, wcat.shape[0], cph, 1, 1, 0, 0, _lib.AV2X_CONV_NCHW)
        self._load_fusion(sd, up)
        self.weights_ready = True

    def _load_fusion(self, sd, up):
        dev = self.device
        self.compressor = self._load_compressor(sd, up, "naive_compressor") if getattr(self, "compression", 0) else None
        g = "fusion_net.naive_communication.gaussian_filter"
        comm = self.fcfg["communication"]
        if "gaussian_smooth" in comm:
            self.gauss_w = up(sd[g + ".weight"].detach().float().reshape(-1))
            self.gauss_b = up(sd[g + ".bias"].detach().float().reshape(-1))
            self.gauss_k = int(sd[g + ".weight"].shape[-1])
        else:  # identity smoothing
            self.gauss_w = torch.ones(1, device=dev)
            self.gauss_b = torch.zeros(1, device=dev)
            self.gauss_k = 1
        self.threshold = float(comm["threshold"] or 0.0)

    # ------------------------------------------------------------------ buffers
    def timed_hbm(self, name, nbytes, flops, launch):
        """Run ``launch()``; in the bench's roofline pass (profile_hbm is a list) bracket it with an event pair on the launch stream."""
        if self.profile_hbm is None:
            return launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        self.profile_hbm.append((name, float(nbytes), float(flops), e0, e1))

    def buf(self, name, shape, dtype=torch.float32):
        key = (name, tuple(shape), dtype)
        t = self.ws.get(key)
        if t is None:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self.ws[key] = t
            self._ws_bytes += t.numel() * t.element_size()
            if self._ws_bytes > self.ws_limit:
                self._evict()
        self._ws_used[key] = self._frame
        return t

    def _evict(self):
        """Drop least-recently-used workspace buffers (never one the current frame has touched: its kernels may be queued
        on it -- the caching allocator keeps a freed block alive for the stream that used it, but a later buf() call of the
        same frame must see the same storage).  Captured graphs hold raw pointers into the pool and are dropped with it."""
        old = sorted((f, k) for k, f in self._ws_used.items() if f < self._frame and k in self.ws)
        for _, k in old:
            if self._ws_bytes <= self.ws_limit:
                break
            t = self.ws.pop(k)
            self._ws_used.pop(k, None)
            self._ws_bytes -= t.numel() * t.element_size()
        self.graphs.clear()

    @staticmethod
    def stream():
        """The current HIP stream of the current device as a raw handle (called once per kernel launch: the raw accessor is
        ~10x cheaper than building a torch.cuda.Stream object)."""
        if _RAW_STREAM is not None:
            return c_void_p(_RAW_STREAM(torch._C._cuda_getDevice()))
        return c_void_p(torch.cuda.current_stream().cuda_stream)

    # ------------------------------------------------------------------ kernels
    LN_FOLD = os.environ.get("AV2X_LN_FOLD", "1") != "0"

    def ln_operand(self, x, n_tokens, c, gamma, beta, eps, tag="ln_stats"):
        """nn.LayerNorm(x) of the first n_tokens rows of x as an operand of conv(..., ln=): see LnOperand."""
        return LnOperand(self, x, n_tokens, c, gamma, beta, eps, tag)

    def ln_stats(self, x, n_tokens, c, eps, tag="ln_stats"):
        """(mean, rstd) of every token of x (n_tokens, c): the statistics half of av2x_layernorm, for conv(..., ln=(stats, gamma, beta))."""
        st = self.buf(tag, (n_tokens, 2))
        _lib.check(self.lib.av2x_layernorm_stats(_ptr(x), _ptr(st), n_tokens, c, eps, self.stream()), "av2x_layernorm_stats")
        return st

    def conv(self, L, x, n, h, w, out, in_ctot=None, in_coff=0, out_ctot=None, out_coff=0, residual=None, ln=None):
        """x: NHWC buffer holding >= n images of (h, w, in_ctot); returns (ho, wo).  ``ln`` = (stats, gamma, beta, eps): the layer runs on
        nn.LayerNorm(x) -- normalised while the pipelined split-3 tiles load their operand rows (av2x_conv2d_ln, same bits as the separate
        LayerNorm launch); any other kernel class materialises the LayerNorm first."""
        d = self._desc
        d.n, d.h, d.w, d.cin = n, h, w, L.cin
        d.in_ctot = in_ctot if in_ctot is not None else L.cin
        d.in_coff = in_coff
        if L.mode == _lib.AV2X_DECONV:
            ho, wo = h * L.up, w * L.up
            d.ho, d.wo = h, w
        else:
            ho = (h + 2 * L.pad - L.ks) // L.stride + 1
            wo = (w + 2 * L.pad - L.ks) // L.stride + 1
            d.ho, d.wo = ho, wo
        d.cout, d.coutp = L.cout, L.coutp
        d.out_ctot = out_ctot if out_ctot is not None else L.cout
        d.out_coff = out_coff
        d.ks, d.stride, d.pad, d.relu, d.mode, d.up = L.ks, L.stride, L.pad, L.relu, L.mode, L.up
        d.sk_wgs = 0
        # bf16 ACTIVATIONS (AMP mode only): the tensors' dtypes say how input and output are stored (csrc/conv_igemm_bf16.inc IN16 / OUT16)
        a16 = (1 if x.dtype == torch.bfloat16 else 0) | (2 if out.dtype == torch.bfloat16 else 0)
        if a16 and not self.amp:
            raise RuntimeError("bf16 activation tensors outside AMP mode")
        if a16 & 2 and residual is not None:
            raise RuntimeError("no residual operand with a bf16 output")
        d.act16 = a16
        wgt = _w16(L) if self.amp else (_w3(L) if self.split3 else L.w)
        vflag = 0x0800 if self.amp else (0x0400 if self.split3 else 0)
        # (1 x 1 "maps" -- the per-agent embeddings of V2X-ViT, a few rows per launch -- keep the lighter fp32 tiles: 8 us against 22.  Keyed on
        # the map, not on the launch: the agent-sharded frame and a batch must pick the same class as the single frame)
        x3p_ok = not self.amp and not a16 and not self.conv_tile and L.cin % 16 == 0 and L.coutp % 64 == 0 and d.ho * d.wo >= 16
        if (a16 & 1) and self.halo16 and not self.conv_tile and self.halo16_rule(L) and residual is None:
            wgt, d.coutp = _w16h(L)                     # halo-tile direct convolution on bf16 activations: a rule, not a timing
            d.tile = self.HALO16_TILE
        elif self.winograd and self.wino4 and not self.conv_tile and vflag == 0 and self.wino4_rule(L, n, d.ho, d.wo):
            if self.wino_x3 and self.wino4_x3 and L.cin % 32 == 0:   # the same class on the bf16 matrix cores (split-3 operands)
                wgt = _wu43(L, self.lib, self.stream())
                d.tile = self.WINO4_X3_TILE
            else:
                wgt = _wu4(L, self.lib, self.stream())  # the F(4x4,3x3) class: a pure function of the layer's shape
                d.tile = self.WINO4_TILE
        elif self.winograd and self.wino_x3 and self.wino2_x3 and not self.conv_tile and vflag == 0 and self.wino_x3_rule(L):
            wgt = _wu3(L, self.lib, self.stream())
            d.tile = self.wino_x3_tile(L, d.ho, d.wo)
        elif self.winograd and not self.conv_tile and vflag == 0 and self.wino_rule(L):
            wgt = _wu(L, self.lib, self.stream())
            d.tile = self.WINO_TILE
            if self.autotune:   # the Winograd tilings are bit-identical to each other: which one runs is a speed question only
                key = ("wino" + ("T" if self.throughput_mode else ""), n * d.ho * d.wo, d.ho, d.wo, L.cin, L.coutp)
                t = self.tile_cache.get(key)
                if t is None:
                    t = self._tune(d, x, L, out, "wino", key)
                    self.tile_cache[key] = t
                d.tile, d.sk_wgs = t
        elif x3p_ok and (self.x3p or self.split3):
            wgt = _w3(L)                                 # a rule of the shape (x3p_tile), no timing: reproducible
            d.tile = self.x3p_tile(n * d.ho * d.wo, L.coutp)
        elif self.conv_tile:
            d.tile = self.conv_tile
            d.sk_wgs = self.conv_sk_wgs if (d.tile & 0x2000) else 0
        elif self.autotune:
            skc = self.sk_class(n * d.ho * d.wo, L, vflag)
            key = (L.mode, n * d.ho * d.wo, L.cin, L.coutp, L.ks, L.stride, skc, vflag) + ((f"a16={a16}",) if a16 else ())
            t = self.tile_cache.get(key)
            if t is None:
                t = self._tune(d, x, L, out, skc, key)
                self.tile_cache[key] = t
            d.tile, d.sk_wgs = t
        elif a16:
            d.tile = (128 << 16) | (64 if L.coutp % 128 else 128) | 0x8800
        else:
            bm, bn = self.pick_tile(n * d.ho * d.wo, L.coutp)
            d.tile = (bm << 16) | bn | vflag
        bm, bn = (d.tile >> 16) & 0x7fff, d.tile & 0xffff  # bn keeps the variant flags (profile key); bm & 0x4000 = Winograd
        ln_fold = False
        if ln is not None:
            lazy = isinstance(ln, LnOperand)
            ln_fold = (self.LN_FOLD and (d.tile & 0x1400) == 0x1400 and not (d.tile & 0x2000) and not (d.tile & 0x40000000) and L.ks == 1 and L.stride == 1 and L.pad == 0
                       and L.mode == _lib.AV2X_CONV and d.in_ctot == L.cin and d.in_coff == 0 and L.cin <= 1024 and not a16
                       and (L.cin == 256 or not lazy))      # av2x_layernorm_stats is built for 256 channels; other widths materialise (256 / 512)
            if lazy:
                if ln.c != L.cin or ln.n_tokens < n * h * w:
                    raise ValueError(f"LayerNorm operand of {ln.n_tokens} x {ln.c} tokens for a Linear over {n * h * w} x {L.cin}")
                if ln_fold:     # the statistics pass runs once per LayerNorm, on first use by a folding kernel
                    ln = (ln.stats(), ln.gamma, ln.beta, ln.eps)
                else:           # a kernel class that cannot normalise while it loads: ONE LayerNorm launch, shared by every consumer of this operand
                    x = ln.materialised()
            elif not ln_fold:
                xn = self.buf("ln_materialised", (n, h, w, L.cin))
                _lib.check(self.lib.av2x_layernorm(_ptr(x), _ptr(ln[1]), _ptr(ln[2]), _ptr(xn), n * h * w, L.cin, ln[3], self.stream()), "av2x_layernorm")
                x = xn
        if self.profile is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if d.tile & 0x2000:   # stream-K needs the partial-accumulator workspace (0x1000 = persistent does not)
            ws = self.sk_workspace()
            _lib.check(self.lib.av2x_conv2d_sk(byref(d), _ptr(x), _ptr(wgt), _ptr(L.scale), _ptr(L.shift), _ptr(residual),
                                               _ptr(out), _ptr(ws), ws.numel() * 4, self.stream()), "av2x_conv2d_sk")
        elif ln_fold:
            _lib.check(self.lib.av2x_conv2d_ln(byref(d), _ptr(x), _ptr(ln[0]), _ptr(ln[1]), _ptr(ln[2]), _ptr(wgt), _ptr(L.scale), _ptr(L.shift),
                                               _ptr(residual), _ptr(out), self.stream()), "av2x_conv2d_ln")
        else:
            _lib.check(self.lib.av2x_conv2d_res(byref(d), _ptr(x), _ptr(wgt), _ptr(L.scale), _ptr(L.shift), _ptr(residual),
                                                _ptr(out), self.stream()), "av2x_conv2d")
        if self.profile is not None:
            e1.record()
            # algorithmic FLOPs: 2 * output pixels * real output channels * taps * cin
            ncols = L.coutp if L.mode == _lib.AV2X_DECONV else L.cout
            if bm & 0x2000:   # Winograd F(4x4,3x3): 32-tile blocks of 4x4 outputs x (cout / 64)
                wgs = -(-(n * ((d.ho + 3) // 4) * ((d.wo + 3) // 4)) // 32) * (L.cout // 64)
            elif bm & 0x1000:   # halo-tile direct convolution on bf16 activations: 8 x 16 output pixels x 128 couts
                wgs = n * ((d.ho + 7) // 8) * ((d.wo + 15) // 16) * (d.coutp // 128)
            elif bm & 0x4000:   # Winograd: (32 x TB-tile blocks of 2x2 outputs) x (cout / CB)
                wgs = -(-(n * ((d.ho + 1) // 2) * ((d.wo + 1) // 2)) // (bm & 0x3fff)) * (L.cout // (bn & 0x01ff))
            else:
                wgs = -(-(n * d.ho * d.wo) // bm) * (L.coutp // (bn & 0x01ff))
            if bn & 0x2000:  # sk_schedule() of conv_igemm.hip: whole tiles first, the remainder tiles split evenly
                steps = L.ks * L.ks * (L.cin // 32)
                g = min(d.sk_wgs, wgs * steps)
                dp = (wgs // g) * g
                total = (wgs - dp) * steps
                if dp == 0 and total > 0:
                    per = max(-(-total // g), min(4, steps))
                    g = -(-total // per)
                wgs = g
            self.profile.append(((bm, bn), 2.0 * n * d.ho * d.wo * ncols * L.ks * L.ks * L.cin, e0, e1, wgs,
                                 (n * d.ho * d.wo, L.cin, ncols, L.ks, L.stride, x.element_size(), 2 if self.amp else 4, out.element_size())))
        return ho, wo

    # 32 tiles x 64 couts per workgroup, 8 of the 16 positions per wave (128 accumulation registers: two workgroups per CU,
    # one computes while the other is in its prologue / epilogue); bit-identical to the 16-positions-per-wave tilings
    WINO_TILE = 0x40000000 | (32 << 16) | 64 | 0x8000
    # (0x4000 | tiles, couts | flags) per workgroup: half-position (two workgroups per CU), quarter-position (three or four;
    # finer tasks for the small maps), and the 16-positions-per-wave 32 x 128 form
    WINO_CANDIDATES = ((0x4000 | 32, 64 | 0x8000), (0x4000 | 32, 32 | 0x8000), (0x4000 | 32, 128))

    @staticmethod
    def wino_rule(L):
        """Layers that run as Winograd F(2x2,3x3): 3x3 / stride 1 / pad 1, ReLU / sigmoid / tanh or no activation, >= 64 input channels
        (8+ chunks of 8; measured faster than the direct kernel from there on, tools/wino_bench.py) and a multiple of 64
        output channels."""
        return (L.mode == _lib.AV2X_CONV and L.ks == 3 and L.stride == 1 and L.pad == 1 and L.relu in (0, 1, 3, 4, 5)
                and L.cin >= 64 and L.cin % 8 == 0 and L.cout % 64 == 0 and L.cout == L.coutp)

    @staticmethod
    def x3p_tile(m, coutp):
        """Pipelined split-3 GEMM tile (csrc/conv_x3p.hip): 128 x 128 where that still gives two workgroups per CU, else 128 x 64
        (tools/split3_bench.py); both give the bits of every other split-3 tile."""
        bn = 128 if (coutp % 128 == 0 and -(-m // 128) * (coutp // 128) >= 512) else 64
        return (128 << 16) | bn | 0x1400

    @staticmethod
    def wino_x3_rule(L):
        """Layers the split-3 Winograd kernel takes: the F(2x2,3x3) class with 16-channel chunks and 64-cout blocks.  Round 5: from 64 input
        channels on (the 64 -> 64 block-0 layers at 100 x 352: 51.6 vs 59.6 us at 4 agents, 17.2 vs 21.5 at one, 92 vs 103-108 at eight once
        the kernel's epilogue and memory schedule were fixed, profiles/r05i_b0_layers.txt; round 4 kept them on the fp32 Winograd)."""
        return Where2ComEngine.wino_rule(L) and L.cin % 16 == 0 and L.cin >= WINO_X3_MIN_CIN and L.cout % 64 == 0 and L.coutp == L.cout

    WINO_X3_T32 = os.environ.get("AV2X_WINO_X3_T32", "0") == "1"
    WINO_X3_T32_MAX_PIXELS = 50 * 176

    @classmethod
    def wino_x3_tile(cls, L, h, w):
        """64 x 64 tile (one wave per SIMD with the whole register file).  The 32 x 64 tile (two workgroups per CU, same bits) is faster
        per isolated launch on the small maps but slower at frame level in every mode measured; AV2X_WINO_X3_T32=1
        switches it on for maps of <= 50 x 176 pixels per image (a function of the map only: sharded / batched frames keep their bits)."""
        tb = 32 if (cls.WINO_X3_T32 and h * w <= cls.WINO_X3_T32_MAX_PIXELS) else 64
        return 0x40000400 | (tb << 16) | 64

    # Winograd F(4x4,3x3) (csrc/conv_wino4.inc): 2.25 multiplies per output instead of 4; one workgroup (32 tiles of 4x4 outputs x 64
    # couts, 18 accumulator tiles per wave) occupies a CU, so a launch takes ceil(workgroups / 256) x (14 us + 2.9 us per 8 input
    # channels) (tools/wino4_bench.py: 1.48x faster than F(2x2,3x3) on 4 x 100 x 352 x 256 -> 256, slower on the small maps).
    # Like the F(2x2) rule, the choice is a function of the LAYER and of the map size only, never of the number of agents in the
    # launch: the agent-sharded frame and a batch must give the bits of the single frame (tests/test_gpu_sharded.py,
    # test_gpu_batch_and_single.py).  Taken where ONE image already fills the chip: >= 256 workgroups per image and a K loop of
    # >= 16 chunks -- the two 256 -> 256 shrink convolutions at 100 x 352 of the default grid.
    WINO4_TILE = 0x60000000 | (32 << 16) | 64
    WINO4_X3_TILE = 0x60000400 | (32 << 16) | 64
    WINO4_MIN_WGS_PER_IMAGE = int(os.environ.get("AV2X_WINO4_MIN_WGS", "256"))   # per IMAGE (never per launch): see above
    # THROUGHPUT mode (engine.throughput_mode: FramePipeline / ShardedPipeline with more than one frame in flight): what counts is the
    # CU-time of a layer (workgroups x workgroup time), not how long one launch takes -- other frames' kernels fill the CUs a launch leaves
    # idle -- and there the F(4x4) class is cheaper for the 128 -> 128 layers at 50 x 176 (36 workgroups per image) and the 256 -> 256 layers
    # at 25 x 88 (20) too: round 5, split-3 kernels, 4 agents pipelined 505-519 -> 532-533 frames/s, while one frame at a time falls from 357
    # to 296 (profiles/r05u_wino4_threshold_x3.txt).  Still a function of the layer, the map and the MODE only: every frame of an engine in
    # throughput mode has the same bits, pipelined or not; they differ from the latency-mode frame within the fp32 rounding of the two Winograd
    # classes (both pinned by the same goldens at the same tolerances, tests/test_gpu_forward.py).
    WINO4_MIN_WGS_PER_IMAGE_T = int(os.environ.get("AV2X_WINO4_MIN_WGS_T", "20"))
    throughput_mode = False
    # ... where whole frames of several agents are in flight on ONE GPU (FramePipeline).  The agent-sharded frame (ShardedPipeline: one or two
    # agents per rank, every launch a fraction of the chip, the frame rate set by the length of the rank's kernel chain) keeps the latency-mode
    # classes: `sharded_frame` is set by ShardedPipeline on its engines.
    sharded_frame = False

    def wino4_throughput(self):
        return bool(self.throughput_mode) and not self.sharded_frame
    WINO4_MIN_CIN = 128
    wino4 = os.environ.get("AV2X_WINOGRAD4", "1") != "0"

    WINO4_MIN_CIN_T = int(os.environ.get("AV2X_WINO4_MIN_CIN_T", "128"))      # throughput mode's channel floor

    def wino4_rule(self, L, n, h, w):
        if not (self.wino_rule(L) and L.cin >= (min(self.WINO4_MIN_CIN, self.WINO4_MIN_CIN_T) if self.wino4_throughput() else self.WINO4_MIN_CIN)):
            return False
        need = min(self.WINO4_MIN_WGS_PER_IMAGE, self.WINO4_MIN_WGS_PER_IMAGE_T) if self.wino4_throughput() else self.WINO4_MIN_WGS_PER_IMAGE
        return -(-(((h + 3) // 4) * ((w + 3) // 4)) // 32) * (L.cout // 64) >= need

    # AMP mode with bf16 activation storage: 1x1 / 3x3 stride-1 layers run as the halo-tile direct convolution (csrc/conv_halo_bf16.inc:
    # every input byte crosses L2 -> CU once per 64-channel chunk instead of once per tap).  Its K order differs from conv_igemm_bf16's,
    # so it is chosen by this rule (a function of the layer), never by timing.
    HALO16_TILE = 0x10000000 | (128 << 16) | 128 | 0x0800
    halo16 = os.environ.get("AV2X_HALO16", "1") != "0"

    @staticmethod
    def halo16_rule(L):
        return (L.mode == _lib.AV2X_CONV and L.stride == 1 and L.ks in (1, 3) and L.pad == L.ks // 2 and L.relu in (0, 1)
                and L.cin % 64 == 0 and L.cout >= 128)

    # BM, BN | 0x8000 (8-wave workgroup) | 0x4000 (prefetch distance 2 / third LDS stage) | 0x0200 (LDS-DMA operand path)
    TILE_CANDIDATES = ((128, 128), (128, 64), (64, 64), (64, 128), (128, 128 | 0x8000), (128, 64 | 0x8000),
                       (128, 128 | 0x4000), (128, 64 | 0x4000), (64, 64 | 0x4000), (64, 128 | 0x4000),
                       (128, 128 | 0xc000), (128, 64 | 0xc000), (128, 32),
                       (128, 128 | 0x8200), (128, 128 | 0xc200), (128, 64 | 0x8200), (128, 64 | 0xc200), (128, 128 | 0x0200),
                       (128, 64 | 0x0200), (64, 64 | 0x0200), (64, 64 | 0x4200))

    # "tune" mode only: stream-K candidates (BM, BN | flags | 0x2000, persistent workgroups); tools/sk_bench.py sweep
    SK_CANDIDATES = ((128, 64 | 0xe000, 768), (128, 64 | 0xe000, 512), (128, 128 | 0xe000, 256), (128, 128 | 0xe000, 512),
                     (64, 64 | 0x6000, 1024), (64, 64 | 0x2200, 768), (128, 64 | 0xe200, 512))
    SK_MAX_TILES = 1200   # only layers with at most this many 128x64 tiles are tried with stream-K
    # "rule" mode: THE stream-K schedule (64x64 tiles, 768 persistent workgroups, tools/sk_bench.py) and its three
    # bit-identical implementations the tuner may choose from
    SK_RULE_WGS = 768
    SK_RULE_IMPLS = ((64, 64 | 0x2200, 768), (64, 64 | 0x6200, 768), (64, 64 | 0x6000, 768))
    # persistent whole-tile candidates (flag 0x1000; bit-identical to the data-parallel schedule): short-K GEMMs
    # (1x1 convs / Linears, <= PERSIST_MAX_STEPS K-steps per tile) where the per-tile prologue is a large share
    PERSIST_CANDIDATES = ((128, 64 | 0xd000, 512), (128, 128 | 0xd000, 512), (64, 64 | 0x5000, 1024), (128, 64 | 0x5000, 768))
    PERSIST_MAX_STEPS = 16
    # AMP mode: conv_igemm_bf16 tiles (flag 0x0800; 0x8000 = 8 waves)
    AMP_CANDIDATES = ((128, 128 | 0x8800, 0), (128, 64 | 0x8800, 0), (64, 64 | 0x0800, 0), (128, 64 | 0x0800, 0),
                      (128, 128 | 0x0800, 0), (128, 32 | 0x0800, 0))
    # ... the ones instantiated for bf16 activation storage (ConvDesc.act16)
    AMP_ACT16_CANDIDATES = ((128, 128 | 0x8800, 0), (128, 64 | 0x8800, 0), (64, 64 | 0x0800, 0))

    def sk_class(self, m, L, vflag=0):
        """Numerics class of a conv launch: "rule" = the fixed stream-K schedule applies (a pure function of the shape),
        "tune" = legacy timing-driven stream-K, False = data-parallel (bit-identical whatever tile is picked)."""
        mode = "tune" if self.stream_k is True else self.stream_k
        if vflag or not mode:
            return False
        if mode == "tune":
            return "tune" if -(-m // 128) * (L.coutp // 64 if L.coutp % 64 == 0 else 1 << 30) <= self.SK_MAX_TILES else False
        steps = L.ks * L.ks * (L.cin // 32)
        tiles64 = -(-m // 64) * (L.coutp // 64) if L.coutp % 64 == 0 else 0
        return "rule" if (steps >= 8 and 64 <= tiles64 <= self.SK_RULE_WGS) else False

    def sk_workspace(self):
        """Partial-accumulator scratch of av2x_conv2d_sk, sized for the largest stream-K candidate."""
        cands = self.SK_CANDIDATES + self.SK_RULE_IMPLS
        need = max(int(self.lib.av2x_conv2d_sk_workspace_bytes((bm << 16) | bn, g)) for bm, bn, g in cands)
        need = max(need, int(self.lib.av2x_conv2d_sk_workspace_bytes(self.conv_tile, self.conv_sk_wgs)) if self.conv_tile else 0)
        return self.buf("sk_ws", (need // 4,))

    # ---- persisted tuning results: every candidate inside a numerics class is bit-identical, so a cached (or stale) pick
    # only affects speed.  In-tree table (made on an MI355X by tools/make_tune_table.py) first, then the user's cache.
    TUNE_TABLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tuned_gfx950.json")
    _tune_disk = None

    @classmethod
    def tune_cache_path(cls):
        base = os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache")
        return os.path.join(base, "airv2x_perception_amd", "tune_gfx950.json")

    @classmethod
    def tune_disk(cls):
        if cls._tune_disk is None:
            import json
            tab = {}
            if os.environ.get("AV2X_TUNE_CACHE", "1") != "0":
                for path in (cls.TUNE_TABLE, cls.tune_cache_path()):
                    try:
                        tab.update(json.load(open(path)))
                    except (OSError, ValueError):
                        pass
            cls._tune_disk = tab
        return cls._tune_disk

    @classmethod
    def tune_store(cls, skey, value):
        cls.tune_disk()[skey] = list(value)
        if os.environ.get("AV2X_TUNE_CACHE", "1") == "0":
            return
        import json
        try:
            path = cls.tune_cache_path()
            os.makedirs(os.path.dirname(path), exist_ok=True)
            mine = {}
            try:
                mine = json.load(open(path))
            except (OSError, ValueError):
                pass
            mine[skey] = list(value)
            tmp = f"{path}.{os.getpid()}.tmp"
            json.dump(mine, open(tmp, "w"), indent=0, sort_keys=True)
            os.replace(tmp, path)
        except OSError:
            pass   # a read-only home only costs the next process a re-tune

    def _candidates(self, d, L, skc):
        if skc == "wino":
            return [(bm, bn, 0) for bm, bn in self.WINO_CANDIDATES if L.cout % (bn & 0x01ff) == 0
                    and not (self.throughput_mode and (bn & 0x81ff) == (32 | 0x8000))]
        if self.amp and d.act16:
            cands = list(self.AMP_ACT16_CANDIDATES)
        elif self.amp:
            cands = list(self.AMP_CANDIDATES)
        elif self.split3:   # + the double-buffered forms (0x4000: second LDS buffer set, one barrier per K-step)
            cands = [(bm, (bn & ~0x0800) | 0x0400, g) for bm, bn, g in self.AMP_CANDIDATES]
            cands += [(bm, bn | 0x4000, g) for bm, bn, g in cands]
        elif skc == "rule":
            cands = list(self.SK_RULE_IMPLS)
        else:
            cands = [(bm, bn, 0) for bm, bn in self.TILE_CANDIDATES]
            if L.ks * L.ks * (L.cin // 32) <= self.PERSIST_MAX_STEPS:
                cands += list(self.PERSIST_CANDIDATES)
            if skc == "tune":
                cands += list(self.SK_CANDIDATES)
        # the 32-column tile only where nothing wider divides the GEMM width (32-column heads; the camera trunk's 96 / 160 / 480 / 672)
        return [(bm, bn, g) for bm, bn, g in cands
                if L.coutp % (bn & 0x01ff) == 0 and not ((bn & 0x01ff) == 32 and L.coutp != 32 and L.coutp % 64 == 0)]

    def _tune(self, d, x, L, out, skc=False, key=None):
        """Pick the fastest implementation for this conv shape inside its numerics class (all of a class's candidates give
        bit-identical results: the K order of every output element does not depend on the tile; the "rule" class is ONE
        stream-K schedule).  Runs outside graph capture; the pick is persisted (tune_store)."""
        cands = self._candidates(d, L, skc)
        if torch.cuda.is_current_stream_capturing():
            if skc == "wino":
                return self.WINO_TILE, 0
            if skc == "rule":
                bm, bn, g = cands[-1]
                return (bm << 16) | bn, g
            if d.act16:
                return (128 << 16) | (64 if L.coutp % 128 else 128) | 0x8800, 0
            bm, bn = self.pick_tile(d.n * d.ho * d.wo, L.coutp)
            return (bm << 16) | bn | (0x0800 if self.amp else (0x0400 if self.split3 else 0)), 0
        skey = "|".join(str(v) for v in key) if key is not None else None
        if skey is not None and skc != "tune":
            hit = self.tune_disk().get(skey)
            if hit is not None and any(((bm << 16) | bn, g) == tuple(hit) for bm, bn, g in cands):
                return int(hit[0]), int(hit[1])
        if not getattr(self, "tune_on_miss", True):      # table miss and no timing allowed (training runner): the static rule
            if skc == "wino":
                return self.WINO_TILE, 0
            if skc == "rule":
                bm, bn, g = cands[-1]
                return (bm << 16) | bn, g
            if d.act16:
                return (128 << 16) | (64 if L.coutp % 128 else 128) | 0x8800, 0
            bm, bn = self.pick_tile(d.n * d.ho * d.wo, L.coutp)
            return (bm << 16) | bn | (0x0800 if self.amp else (0x0400 if self.split3 else 0)), 0
        if not cands:
            raise NotImplementedError(f"no conv tile for cin={L.cin} coutp={L.coutp} ks={L.ks} (class {skc})")
        best, best_t = None, float("inf")
        wgt = _wu(L, self.lib, self.stream()) if skc == "wino" else (_w16(L) if self.amp else (_w3(L) if self.split3 else L.w))
        # tune into a scratch output: `out` may alias the input / residual (in-place transformer updates)
        ho = d.ho * (L.up if L.mode == _lib.AV2X_DECONV else 1)
        wo = d.wo * (L.up if L.mode == _lib.AV2X_DECONV else 1)
        scratch = torch.empty(d.n * ho * wo * max(d.out_ctot, L.cout), dtype=torch.bfloat16 if d.act16 & 2 else torch.float32, device=self.device)
        ws = self.sk_workspace()
        st = self.stream()
        for bm, bn, g in cands:
            d.tile, d.sk_wgs = (bm << 16) | bn, g
            call = lambda: _lib.check(self.lib.av2x_conv2d_sk(byref(d), _ptr(x), _ptr(wgt), _ptr(L.scale), _ptr(L.shift), None,
                                                              _ptr(scratch), _ptr(ws), ws.numel() * 4, st), "av2x_conv2d")
            call()  # warm-up (module load, L2)
            t = float("inf")
            for _rep in range(3):   # best of three short bursts: one burst alone is noisy enough to flip close candidates
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    call()
                e1.record()
                e1.synchronize()
                t = min(t, e0.elapsed_time(e1))
            if skc == "tune" and bn & 0x2000:
                t *= 1.03  # prefer the bit-reproducible schedules unless stream-K is clearly faster
            if os.environ.get("AV2X_TUNE_LOG"):
                print(f"[tune] M={d.n * d.ho * d.wo} cin={L.cin} coutp={L.coutp} ks={L.ks} tile={bm}x{bn & 0x1ff} flags={bn & 0xfe00:#x} "
                      f"wgs={g}: {t / 3 * 1e3:.1f} us", flush=True)
            if t < best_t:
                best, best_t = (d.tile, g), t
        if skey is not None and skc != "tune":
            self.tune_store(skey, best)
        return best

    # AMP mode of an engine whose fusion keeps bf16 activations (V2XViTEngine): the trunk's intermediate maps are stored as bf16 too
    # (what autocast stores between two Conv2d); buffers handed in by the caller keep their own dtype
    act16_trunk = False

    def trunk_dtype(self):
        return torch.bfloat16 if (self.amp and self.act16_trunk) else torch.float32

    # The feature-sharing message of the autocast frame (CoBEVT / V2X-ViT: the shrink header's -- or the compressor encoder's -- output) is
    # bf16: what torch.autocast stores for that Conv2d in the reference, and half the bytes per xGMI link (18.0 MB per agent at the default
    # grid).  The single-process autocast frame rounds the same tensor the same way, so the sharded frame keeps its bits.
    bf16_message = True

    def msg_dtype(self):
        return torch.bfloat16 if (self.amp and self.bf16_message) else torch.float32

    def widen(self, src16, dst32):
        """dst32 = float(src16): bf16 message -> the fusion's fp32 stream (exact)."""
        assert src16.dtype == torch.bfloat16 and dst32.dtype == torch.float32 and src16.numel() == dst32.numel()
        _lib.check(self.lib.av2x_bf16_to_f32(_ptr(src16), _ptr(dst32), src16.numel(), self.stream()), "av2x_bf16_to_f32")
        return dst32

    def run_block(self, i, x, n, h, w, tag, out=None):
        """backbone.blocks[i] on n images; returns (buffer, ho, wo).  ``out``: write the block's
        result into this (n,ho,wo,c) buffer (e.g. a slice of the all-gather send buffer)."""
        layers = self.blocks[i]
        c = layers[0].cout
        ho = (h + 2 - 3) // layers[0].stride + 1
        wo = (w + 2 - 3) // layers[0].stride + 1
        td = self.trunk_dtype()
        ping = self.buf(f"blk{i}_ping_{tag}", (n, ho, wo, c), td)
        pong = self.buf(f"blk{i}_pong_{tag}", (n, ho, wo, c), td)
        if out is None:
            out = self.buf(f"blk{i}_out_{tag}", (n, ho, wo, c), td)
        cur, ch, cw = x, h, w
        for li, L in enumerate(layers):
            dst = out if li == len(layers) - 1 else (ping if li % 2 == 0 else pong)
            if not (i == 0 and li == 0 and self.sparse_first_conv(L, cur, n, ch, cw, dst)):
                self.conv(L, cur, n, ch, cw, dst)
            cur, ch, cw = dst, ho, wo
        return out, ho, wo

    def run_deblocks(self, feats, n, cat):
        """feats: [(buffer, h, w)] per level; writes the channel-concatenated map into cat (n,H,W,384)."""
        coff = 0
        for i, (x, h, w) in enumerate(feats):
            L = self.deblocks[i]
            self.conv(L, x, n, h, w, cat, out_ctot=self.cat_c, out_coff=coff)
            coff += L.cout

    def run_shrink(self, x, n, h, w, tag, out=None):
        cur, cin_tot = x, self.cat_c
        for li, L in enumerate(self.shrink):
            last = li == len(self.shrink) - 1
            dst = out if (out is not None and last) else self.buf(f"shrink{li}_{tag}", (n, h, w, L.cout),
                                                                   torch.float32 if last else self.trunk_dtype())
            self.conv(L, cur, n, h, w, dst, in_ctot=cin_tot)
            cur, cin_tot = dst, L.cout
        return cur

    # ------------------------------------------------------------------ stages
    def frame_layout(self, data_dict):
        self._frame += 1          # every forward / sharded stage starts here: the epoch the workspace LRU counts in
        pf = data_dict.get("points")
        if pf is not None:   # raw-cloud input (voxelizer.points_frame): one frame, agents already in frame order
            rank = {t: i for i, t in enumerate(AGENT_TYPES)}
            ts = list(pf["types"])
            if len(ts) == 0 or len(ts) != len(pf["clouds"]):
                raise ValueError("points frame: one type per cloud, at least one agent")
            if any(t not in self.pfn for t in ts):
                raise ValueError("points frame: agent type without a lidar encoder in this model")
            if any(rank[a] > rank[b] for a, b in zip(ts, ts[1:])):
                raise ValueError("points frame: agents must come in frame order (vehicles, rsus, drones; ego first)")
            return [len(ts)], {"__points__": pf}
        return frame_layout(self.args["collaborators"], data_dict)

    def encode_points(self, pf):
        """Raw clouds -> canvas with NO host round trip: per agent av2x_prepare_voxelize (ego-box mask, projection by the
        agent's pose, range crop, pillar voxelizer) -> av2x_voxelize_dummy_if_empty (the reference's empty-cloud branch,
        sp_voxel_preprocessor.py:80-90) -> av2x_pillar_vfe_scatter_dev, which reads the pillar count from device memory.
        Same kernels and therefore the same canvas, bit for bit, as voxelize_frame + the (M,32,4) input contract."""
        clouds, types = pf["clouds"], list(pf["types"])
        rng, vs = [float(v) for v in pf["lidar_range"]], [float(v) for v in pf["voxel_size"]]
        mp, mv = int(pf.get("max_points", 32)), int(pf.get("max_voxels", 70000))
        poses, perms, mask_ego = pf.get("poses"), pf.get("perms"), bool(pf.get("mask_ego", True))
        if mp != 32:
            raise ValueError("the pillar feature net is built for 32 points per pillar")
        n = len(clouds)
        g = [int(v) for v in self.args[types[0]]["lidar"]["point_pillar_scatter"]["grid_size"]]
        nx, ny = g[0], g[1]
        grid = [int(round((rng[3 + j] - rng[j]) / vs[j])) for j in range(3)]
        if grid[0] != nx or grid[1] != ny or grid[2] != 1:
            raise ValueError(f"voxel grid {grid} of the preprocess range does not match the model's canvas {nx}x{ny}x1")
        canvas = self.buf("canvas", (n, ny, nx, 64))
        st = self.stream()
        _lib.check(self.lib.av2x_fill_zero(_ptr(canvas), canvas.numel() * 4, st), "av2x_fill_zero")
        nz = self._nz_begin(st)
        occ = self._occ_begin(st, canvas, n, ny, nx)
        r6, v3 = (c_float * 6)(*rng), (c_float * 3)(*vs)
        for i, (pts, t) in enumerate(zip(clouds, types)):
            if pts.device != self.device or pts.dtype != torch.float32 or not pts.is_contiguous():
                pts = pts.to(self.device, torch.float32).contiguous()
            P = int(pts.shape[0])
            pr = max(16384, -(-P // 16384) * 16384)         # capacity classes: real clouds change size every frame
            cap = max(2, min(pr, mv))
            ws = self.buf(f"vox_ws{i}", (int(self.lib.av2x_voxelize_workspace_bytes(pr, nx, ny, 1)),), torch.uint8)
            voxels = self.buf(f"vox_f{i}", (cap, mp, 4))
            coords = self.buf(f"vox_c{i}", (cap, 3), torch.int32)
            num = self.buf(f"vox_n{i}", (cap,), torch.int32)
            cnt = self.buf(f"vox_m{i}", (1,), torch.int32)
            t16 = None
            if poses is not None and poses[i] is not None:
                t16 = (c_float * 16)(*np.asarray(poses[i], dtype=np.float32).reshape(-1).tolist())
            pm = None
            if perms is not None and perms[i] is not None:
                pm = perms[i].to(device=self.device, dtype=torch.int32).contiguous()
            _lib.check(self.lib.av2x_prepare_voxelize(_ptr(pts), _ptr(pm), P, ctypes.cast(t16, c_void_p) if t16 is not None else None,
                                                      ctypes.cast(r6, c_void_p), 1 if mask_ego else 0, ctypes.cast(r6, c_void_p),
                                                      ctypes.cast(v3, c_void_p), mp, mv, _ptr(ws), _ptr(voxels), _ptr(coords),
                                                      _ptr(num), _ptr(cnt), st), "av2x_prepare_voxelize")
            _lib.check(self.lib.av2x_voxelize_dummy_if_empty(ctypes.cast(r6, c_void_p), ctypes.cast(v3, c_void_p), mp, mv, cap,
                                                             _ptr(voxels), _ptr(coords), _ptr(num), _ptr(cnt), st),
                       "av2x_voxelize_dummy_if_empty")
            w, sc, sh, geom = self.pfn[t]
            _lib.check(self.lib.av2x_pillar_vfe_scatter_dev_count(_ptr(voxels), _ptr(coords), _ptr(num), _ptr(cnt), cap, _ptr(w), _ptr(sc),
                                                                  _ptr(sh), ctypes.cast(geom, c_void_p), _ptr(canvas), i, ny, nx, _ptr(nz), _ptr(occ), st),
                       "av2x_pillar_vfe_scatter_dev_count")
        self._nz_canvas = canvas if nz is not None else None
        return canvas, ny, nx

    def encode(self, data_dict, record_len, slots):
        if "__points__" in slots:
            return self.encode_points(slots["__points__"])
        n_total = sum(record_len)
        t0 = next(iter(slots))
        if "lidar" in self.args[t0]:
            g = [int(v) for v in self.args[t0]["lidar"]["point_pillar_scatter"]["grid_size"]]
        else:
            g = [int(v) for v in self.cam[t0].nx]
        nx, ny = g[0], g[1]
        canvas = self.buf("canvas", (n_total, ny, nx, 64))
        cam = getattr(self, "cam", {})
        both = [t for t in slots if t in cam and t in self.pfn]
        lidar_canvas = self.buf("canvas_lidar", (n_total, ny, nx, 64)) if both else canvas
        st = self.stream()
        if any(t in self.pfn for t in slots):
            self.timed_hbm("canvas clear (hipMemsetAsync)", lidar_canvas.numel() * 4, 0.0,
                           lambda: _lib.check(self.lib.av2x_fill_zero(_ptr(lidar_canvas), lidar_canvas.numel() * 4, st), "av2x_fill_zero"))
        # a LiDAR-only frame: the scatter counts the non-zeros it writes (comm_rate, airv2x_where2com.py:122) -- no read-back pass over the canvas
        lidar_only = not any(t in cam for t in slots)
        nz = self._nz_begin(st) if lidar_only else None
        occ = self._occ_begin(st, canvas, n_total, ny, nx) if lidar_only else self._occ_begin(st, None, 0, 0, 0)
        self.encode_lidar(data_dict, slots, lidar_canvas, ny, nx, nz, occ)
        self._nz_canvas = canvas if nz is not None else None
        per = ny * nx * 64
        for t, sl in slots.items():
            if t not in cam:
                if both:    # a LiDAR-only type next to multimodal ones: its rows move from the scatter buffer to the frame canvas
                    for s_ in sl:
                        _lib.check(self.lib.av2x_mean2(_ptr(lidar_canvas[s_]), None, _ptr(canvas[s_]), per, st), "av2x_mean2")
                continue
            ci = data_dict[t].get("batch_merged_cam_inputs")
            if ci is None:
                raise ValueError(f"{t}: the model has a camera encoder but the frame carries no batch_merged_cam_inputs")
            if int(ci["imgs"].shape[0]) != len(sl):
                raise ValueError(f"{t}: {int(ci['imgs'].shape[0])} camera rigs for {len(sl)} agents")
            contiguous = sl == list(range(sl[0], sl[0] + len(sl)))
            if t not in self.pfn and contiguous:     # camera only: BevEncode's last conv writes the canvas rows
                cam[t].forward(ci, out=canvas[sl[0]:sl[0] + len(sl)])
                continue
            bev = cam[t].forward(ci)
            for j, s_ in enumerate(sl):             # Airv2xBase.fuse_bev (airv2x_base_model.py:167-177): mean over the modality maps
                _lib.check(self.lib.av2x_mean2(_ptr(bev[j]), _ptr(lidar_canvas[s_]) if t in self.pfn else None, _ptr(canvas[s_]), per, st),
                           "av2x_mean2")
        return canvas, ny, nx

    FOLD_COUNT = os.environ.get("AV2X_FOLD_COUNT", "1") != "0"
    _nz_canvas = None       # the canvas whose non-zero count the scatter has already put into buf("nonzero") (see count_canvas)

    def _nz_begin(self, st):
        """Zeroed counter for a counting scatter, or None when the fold is switched off (AV2X_FOLD_COUNT=0: count by read-back)."""
        if not self.FOLD_COUNT:
            return None
        nz = self.buf("nonzero_slots", (32 * 16,), torch.int64)       # AV2X_NZ_SLOTS counters, AV2X_NZ_STRIDE apart (include/airv2x_hip.h)
        _lib.check(self.lib.av2x_fill_zero(_ptr(nz), nz.numel() * 8, st), "av2x_fill_zero")
        return nz

    def count_canvas(self, canvas, st):
        """comm_rate = spatial_features.count_nonzero() (airv2x_where2com.py:122) -> device counter (1,) i64.  encode() of a LiDAR-only frame
        has counted while scattering (every cell is written at most once into a zeroed canvas); any other canvas (camera rows, a caller's
        own tensor) is counted by a read-back pass."""
        nz = self.buf("nonzero", (1,), torch.int64)
        if self._nz_canvas is canvas and canvas is not None:
            self._nz_canvas = None
            _lib.check(self.lib.av2x_nonzero_slots_sum(_ptr(self.buf("nonzero_slots", (32 * 16,), torch.int64)), _ptr(nz), st), "av2x_nonzero_slots_sum")
            return nz
        _lib.check(self.lib.av2x_fill_zero(_ptr(nz), 8, st), "av2x_fill_zero")
        self.timed_hbm("count_nonzero (comm_rate)", canvas.numel() * 4, 0.0,
                       lambda: _lib.check(self.lib.av2x_count_nonzero(_ptr(canvas), canvas.numel(), _ptr(nz), st), "av2x_count_nonzero"))
        return nz

    SPARSE_CONV0 = os.environ.get("AV2X_SPARSE_CONV0", "1") != "0"
    _occ_info = None        # (frame, canvas data_ptr, agents, ny, nx, occupancy bytes): the scattered canvas of THIS frame (see sparse_first_conv)

    def _occ_begin(self, st, canvas, n, ny, nx):
        """Zeroed occupancy bytes (one per canvas cell) for the counting scatter of a LiDAR-only frame -- the first backbone convolution then
        gathers the occupied taps only (csrc/sparse_conv.hip) --, or None (camera rows in the canvas, AV2X_SPARSE_CONV0=0, autocast)."""
        self._occ_info = None
        if canvas is None or not self.SPARSE_CONV0 or self.amp or canvas.dtype != torch.float32:
            return None
        occ = self.buf("canvas_occ", (n, ny, nx), torch.uint8)
        _lib.check(self.lib.av2x_fill_zero(_ptr(occ), occ.numel(), st), "av2x_fill_zero")
        self._occ_info = (self._frame, canvas.data_ptr(), n, ny, nx, occ)
        return occ

    def sparse_eligible(self):
        """This frame's canvas came from a counting LiDAR-only scatter (its occupancy bytes are valid): what sparse_first_conv keys on.  Part of
        every hipGraph key: a capture made on a LiDAR-only frame holds the sparse gather, and must not replay over a frame with camera rows."""
        return self._occ_info is not None and self._occ_info[0] == self._frame

    def sparse_first_conv(self, L, x, n, h, w, out):
        """Conv2d(64, 64, 3, stride 2) + folded BatchNorm + ReLU of block 0 on (rows of) this frame's scattered canvas: the gather over occupied
        taps (av2x_conv3x3s2_sparse).  A rule of the layer and of the frame kind (LiDAR-only, fp32-accurate mode), never of the agent count:
        batches, agent groups and the sharded frame take it alike.  Returns False when the input is not that canvas."""
        info = self._occ_info
        if (info is None or info[0] != self._frame or self.amp or self.conv_tile or L.ks != 3 or L.stride != 2 or L.pad != 1 or L.cin != 64
                or L.cout != 64 or L.coutp != 64 or L.mode != _lib.AV2X_CONV or x.dtype != torch.float32 or out.dtype != torch.float32
                or (h, w) != (info[3], info[4]) or L.relu not in (0, 1)):
            return False
        per = h * w * 64 * 4
        off = x.data_ptr() - info[1]
        if off < 0 or off % per or off // per + n > info[2]:
            return False
        occ = info[5][off // per: off // per + n]
        _lib.check(self.lib.av2x_conv3x3s2_sparse(_ptr(x), _ptr(occ), _ptr(L.w), _ptr(L.scale), _ptr(L.shift), L.relu, _ptr(out), n, h, w, 64, 64,
                                                  self.stream()), "av2x_conv3x3s2_sparse")
        return True

    def encode_lidar(self, data_dict, slots, canvas, ny, nx, nz=None, occ=None):
        """Sequential(PillarVFE, PointPillarScatter) of every agent type with a LiDAR encoder, into the (zeroed) canvas rows; ``nz``: the
        device counter the scatter adds its written non-zeros to; ``occ``: the occupancy bytes it sets."""
        st = self.stream()
        for t, sl in slots.items():
            if t not in self.pfn:
                continue
            lid = data_dict[t]["batch_merged_lidar_features_torch"]
            vf, vc, vn = lid["voxel_features"], lid["voxel_coords"], lid["voxel_num_points"]
            if vf.device != self.device:
                vf, vc, vn = vf.to(self.device), vc.to(self.device), vn.to(self.device)
            vf = vf.contiguous().float()
            vc = vc.contiguous().to(torch.int32)
            vn = vn.contiguous().to(torch.int32)
            if vf.shape[1:] != (32, 4):
                raise ValueError(f"voxel_features must be (M,32,4), got {tuple(vf.shape)}")
            w, sc, sh, geom = self.pfn[t]
            contiguous = sl == list(range(sl[0], sl[0] + len(sl)))
            smap = None
            if not contiguous:
                smap = torch.tensor(sl, dtype=torch.int32, device=self.device)
            self.timed_hbm("pillar_vfe_scatter", vf.shape[0] * (512 + 12 + 4 + 256), 2.0 * 32 * 10 * 64 * vf.shape[0],
                           lambda: _lib.check(self.lib.av2x_pillar_vfe_scatter_count(_ptr(vf), _ptr(vc), _ptr(vn), vf.shape[0], _ptr(w), _ptr(sc),
                                                                                     _ptr(sh), ctypes.cast(geom, c_void_p), _ptr(canvas), sl[0],
                                                                                     _ptr(smap), len(sl), ny, nx, _ptr(nz), _ptr(occ), st),
                                              "av2x_pillar_vfe_scatter_count"))

    @staticmethod
    def _deblock_hw(L, h, w):
        """Output size of one deblock: ConvTranspose2d(k = s, stride s) or, for upsample_strides < 1, Conv2d(k, stride k)."""
        if L.mode == _lib.AV2X_DECONV:
            return h * L.up, w * L.up
        return (h + 2 * L.pad - L.ks) // L.stride + 1, (w + 2 * L.pad - L.ks) // L.stride + 1

    def cat_hw(self, dims):
        """(H, W) of spatial_features_2d for the per-level (h, w, c) of level_dims(): the per-level deblocks' common output size, times the
        final deblock's stride when the backbone has one (base_bev_backbone.py:141-152)."""
        H, W = self._deblock_hw(self.deblocks[0], dims[0][0], dims[0][1])
        for L, (h, w, _) in zip(self.deblocks, dims):
            if self._deblock_hw(L, h, w) != (H, W):
                raise ValueError("BaseBEVBackbone: the deblock outputs do not share one resolution (torch.cat would fail in the reference)")
        fd = getattr(self, "final_deblock", None)
        return (H * fd.up, W * fd.up) if fd is not None else (H, W)

    def run_cat(self, feats, n, tag, trunk_dtype=None):
        """deblocks -> channel concatenation (-> the extra deblock on the concatenated map): spatial_features_2d of n images."""
        Hc, Wc = self._deblock_hw(self.deblocks[0], feats[0][1], feats[0][2])
        cat = self.buf(f"cat_{tag}", (n, Hc, Wc, self.cat_c), trunk_dtype or torch.float32)
        self.run_deblocks(feats, n, cat)
        fd = getattr(self, "final_deblock", None)
        if fd is None:
            return cat, Hc, Wc
        out = self.buf(f"cat_final_{tag}", (n, Hc * fd.up, Wc * fd.up, self.cat_c), trunk_dtype or torch.float32)
        self.conv(fd, cat, n, Hc, Wc, out)
        return out, Hc * fd.up, Wc * fd.up

    def trunk(self, canvas, n, ny, nx, tag="all", block_out=None, shrink_out=None):
        """blocks -> deblocks -> shrink for n agents.  Returns (feats per level, shrink out, H, W)."""
        feats = []
        x, h, w = canvas, ny, nx
        for i in range(len(self.blocks)):
            x, h, w = self.run_block(i, x, n, h, w, tag, out=(block_out or {}).get(i))
            feats.append((x, h, w))
        cat, H, W = self.run_cat(feats, n, tag, self.trunk_dtype())
        s = self.run_shrink(cat, n, H, W, tag, out=shrink_out) if self.shrink else cat
        return feats, s, H, W

    def comm_layout(self, record_len, has_ego=True):
        """Cached device arrays of a frame layout: (sample of every agent i32, is-ego flag i32, agents per sample f32)."""
        key = ("layout", tuple(record_len), has_ego)
        lay = self.ws.get(key)
        if lay is None:
            samp, ego = [], []
            for b, k in enumerate(record_len):
                samp += [b] * k
                ego += [1 if has_ego else 0] + [0] * (k - 1) if k > 0 else []
            lay = (torch.tensor(samp, dtype=torch.int32, device=self.device),
                   torch.tensor(ego, dtype=torch.int32, device=self.device),
                   torch.tensor(record_len, dtype=torch.float32, device=self.device))
            self.ws[key] = lay
        return lay

    def comm_mask(self, psm_single, n, H, W, record_len, has_ego=True, tag="", count=None, topk=None):
        """Communication.forward (where2comm_fuse.py:83-149).  ``topk``: the TRAINING branch (:104-121) -- one K per sample
        (int(H * W * random.uniform(0, 1)) in the reference): every agent of the sample transmits its K most confident cells
        instead of the thresholded ones."""
        B = len(record_len)
        lay = self.comm_layout(record_len, has_ego)
        conf = self.buf("comm_conf" + tag, (n, H, W))
        smooth = self.buf("comm_smooth" + tag, (n, H, W))
        mask = self.buf("comm_mask" + tag, (n, H, W))
        st = self.stream()
        if count is None:  # else: a shared counter the caller zeroed before forking the agent groups
            count = self.buf("comm_count", (B,), torch.int32)
            _lib.check(self.lib.av2x_fill_zero(_ptr(count), B * 4, st), "av2x_fill_zero")
        # reads the (n,H,W,A*C) single-agent scores once, writes confidence, smoothed map and mask (1.97 MB + 0.14 MB x 3 per agent)
        self.timed_hbm("comm_mask (confidence + Gaussian + threshold)", n * H * W * (psm_single.shape[-1] + 3) * 4, 0.0,
                       lambda: _lib.check(self.lib.av2x_comm_mask(_ptr(psm_single), n, H, W, psm_single.shape[-1], self.A * self.C,
                                                                  _ptr(self.gauss_w), _ptr(self.gauss_b), self.gauss_k, self.threshold,
                                                                  _ptr(lay[0]), _ptr(lay[1]), _ptr(conf), _ptr(smooth), _ptr(mask),
                                                                  _ptr(count), st), "av2x_comm_mask"))
        if topk is not None:
            if len(topk) != B:
                raise ValueError("topk: one K per sample")
            # one K per agent, uploaded from a pinned staging buffer: a pageable torch.tensor(..., device=) copy waits for the stream to
            # drain (1 ms of host stall per training step).  Four buffers in rotation: a buffer is rewritten four steps after its copy
            # was queued.
            vals = [int(topk[b]) for b, k in enumerate(record_len) for _ in range(k)]
            ring = getattr(self, "_ks_ring", None)
            if ring is None or ring[0][0].numel() < len(vals):
                ring = self._ks_ring = [[torch.empty(max(len(vals), 16), dtype=torch.int32).pin_memory() for _ in range(4)], 0]
            pin = ring[0][ring[1] % 4]
            ring[1] += 1
            pin[:len(vals)] = torch.tensor(vals, dtype=torch.int32)
            ks = self.buf("comm_topk_k" + tag, (len(vals),), torch.int32)
            ks.copy_(pin[:len(vals)], non_blocking=True)
            _lib.check(self.lib.av2x_fill_zero(_ptr(count), B * 4, st), "av2x_fill_zero")
            _lib.check(self.lib.av2x_comm_mask_topk(_ptr(smooth), n, H * W, _ptr(ks), _ptr(lay[0]), _ptr(lay[1]), _ptr(mask), _ptr(count),
                                                    st), "av2x_comm_mask_topk")
        return mask, count, smooth, lay[2]

    def comm_rate(self, count, agents_per_sample, B, hw):
        """0-dim fp32 tensor: mean over samples of count / (agents * H * W) (one tiny launch, no ATen arithmetic)."""
        com = self.buf("comm_rate_out", (1,))
        _lib.check(self.lib.av2x_comm_rate(_ptr(count), _ptr(agents_per_sample), B, hw, _ptr(com), self.stream()), "av2x_comm_rate")
        return com[0].clone()   # fresh 0-dim tensor: the caller may keep it across frames

    def attn(self, ptrs, hw, c, out):
        arr = (c_void_p * len(ptrs))(*ptrs)
        # every agent's map read once, the ego's fused row written (15.77 MB x N + 15.77 MB over the three scales)
        self.timed_hbm("pixel_attn_fuse", (len(ptrs) + 1) * hw * c * 4, 4.0 * len(ptrs) * hw * c,
                       lambda: _lib.check(self.lib.av2x_pixel_attn_fuse(arr, len(ptrs), hw, c, _ptr(out), self.stream()), "av2x_pixel_attn_fuse"))

    # ------------------------------------------------------------------ full forward
    @torch.no_grad()
    def forward(self, data_dict, trace=None, sync_comm_rate=False):
        if not self.weights_ready:
            raise RuntimeError("load_state_dict() must be called before forward()")
        record_len, slots = self.frame_layout(data_dict)
        B, n = len(record_len), sum(record_len)
        if n == 0:
            raise ValueError("empty frame: no agent has lidar input")
        canvas, ny, nx = self.encode(data_dict, record_len, slots)
        if self.use_graph and trace is None and self.profile is None:
            # the first convolution's class (sparse gather over this frame's occupancy bytes, or the dense kernel: camera rows in the
            # canvas) is decided per frame in Python and baked into the capture -> part of the key
            key = (tuple(record_len), ny, nx, self.sparse_eligible())
            ent = self.graphs.get(key)
            if ent is None:
                # one eager pass allocates every workspace buffer and layout tensor outside the capture
                self._post_encode(canvas, ny, nx, record_len, None)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    static = self._post_encode(canvas, ny, nx, record_len, None)
                ent = (g, static)
                self.graphs[key] = ent
            g, static = ent
            g.replay()
            heads, com, nz = static
            heads = heads.clone()  # outputs are fresh tensors, as in the reference
            com, nz = com.clone(), nz.clone()
        else:
            heads, com, nz = self._post_encode(canvas, ny, nx, record_len, trace)
        outs = torch.split(heads, self.head_splits, dim=1)
        if B > 1:
            outs = [o.contiguous() for o in outs]
        out = {"psm": outs[0], "rm": outs[1]}
        if self.args["obj_head"]:
            out["obj"] = outs[2]
        if sync_comm_rate:
            comm_rate = int(nz[0].item())
        else:
            comm_rate = nz[0].clone()   # fresh tensor, not a view of the workspace the next frame overwrites
        out.update({"mask": 0, "com": com, "comm_rate": comm_rate})
        return out

    def _post_encode(self, canvas, ny, nx, record_len, trace):
        """Everything after the scatter; static shapes for a given record_len -> capturable."""
        B, n = len(record_len), sum(record_len)
        if (B == 1 and n >= 2 and self.agent_streams > 1 and trace is None and self.profile is None and self.multi_scale
                and getattr(self, "final_deblock", None) is None and all(L.mode == _lib.AV2X_DECONV for L in self.deblocks)
                and not self.fcfg["fully"] and not torch.cuda.is_current_stream_capturing()):
            return self._post_encode_groups(canvas, ny, nx, n)
        st = self.stream()
        nz = self.count_canvas(canvas, st)

        feats, s, H, W = self.trunk(canvas, n, ny, nx)
        psm_single = self.buf("psm_single", (n, H, W, self.A * self.C))
        self.conv(self.cls_single, s, n, H, W, psm_single)
        if trace is not None:
            trace["spatial_features"] = canvas.permute(0, 3, 1, 2).clone()
            for i, (x, _, _) in enumerate(feats):
                trace[f"block{i}"] = x.permute(0, 3, 1, 2).clone()
            trace["spatial_features_2d"] = self.buf("cat_final_all" if getattr(self, "final_deblock", None) is not None else "cat_all",
                                                    (n, H, W, self.cat_c)).permute(0, 3, 1, 2).clone()
            trace["shrink"] = s.permute(0, 3, 1, 2).clone()
            trace["psm_single"] = psm_single.permute(0, 3, 1, 2).clone()

        if not self.multi_scale:
            return self._post_encode_single_scale(s, psm_single, n, H, W, record_len, nz, trace)
        # (multi-scale: the reference runs its compressor on the shrunk map here too, airv2x_where2com.py:147-150, and then fuses
        # batch_dict["spatial_features"] -- the result is dead; nothing is launched for it)
        (b0, h0, w0), (b1, h1, w1), (b2, h2, w2) = feats
        if self.fcfg["fully"]:
            com = torch.tensor(1, device=self.device)
            mask = None
        else:
            mask, count, smooth, rl = self.comm_mask(psm_single, n, H, W, record_len)
            com = self.comm_rate(count, rl, B, H * W)                    # where2comm_fuse.py:137,147
            if trace is not None:
                trace["comm_mask"] = mask.unsqueeze(1).clone()
                trace["comm_map"] = smooth.unsqueeze(1).clone()
            if (h0, w0) != (H, W):      # where2comm_fuse.py:229-235: the mask at the first block's resolution (bilinear, align_corners=False)
                mask_r = self.buf("comm_mask_resized", (n, h0, w0))
                _lib.check(self.lib.av2x_mask_resize_bilinear(_ptr(mask), n, H, W, h0, w0, _ptr(mask_r), st), "av2x_mask_resize_bilinear")
                mask = mask_r
            self.timed_hbm("apply_mask (in place)", n * h0 * w0 * (2 * b0.shape[-1] + 1) * 4, 0.0,
                           lambda: _lib.check(self.lib.av2x_apply_mask(_ptr(b0), _ptr(mask), n, h0 * w0, b0.shape[-1], st), "av2x_apply_mask"))

        # masked pass through blocks 1, 2.  B == 1: agent 0 is the ego (mask == 1) and is skipped.
        skip_ego = (B == 1) and mask is not None
        first = 1 if skip_ego else 0
        nm = n - first
        if mask is None:
            m1, m2 = b1, b2
            first, nm = 0, n
        else:
            if nm > 0:
                x0m = b0[first:]
                m1, _, _ = self.run_block(1, x0m, nm, h0, w0, "masked")
                m2, _, _ = self.run_block(2, m1, nm, h1, w1, "masked")
            else:
                m1 = m2 = None

        levels = [(b0, None, h0, w0), (b1, m1, h1, w1), (b2, m2, h2, w2)]
        fused = []
        for i, (unm, msk, h, w) in enumerate(levels):
            c = unm.shape[-1]
            f = self.buf(f"fused{i}", (B, h, w, c))
            a0 = 0
            for b, k in enumerate(record_len):
                ptrs = []
                for j in range(a0, a0 + k):
                    if i == 0 or mask is None:
                        ptrs.append(unm[j].data_ptr())
                    elif skip_ego:
                        ptrs.append(unm[j].data_ptr() if j == 0 else msk[j - 1].data_ptr())
                    else:
                        ptrs.append(msk[j].data_ptr())
                self.attn(ptrs, h * w, c, f[b])
                a0 += k
            fused.append((f, h, w))
            if trace is not None:
                trace[f"fused{i}"] = f.permute(0, 3, 1, 2).clone()
        catf, _, _ = self.run_cat(fused, B, "fused")
        fs = self.run_shrink(catf, B, H, W, "fused") if self.shrink else catf
        nh = self.heads.cout
        heads = torch.empty((B, nh, H, W), dtype=torch.float32, device=self.device)
        self.conv(self.heads, fs, B, H, W, heads)
        if trace is not None:
            trace["fused_2d"] = catf.permute(0, 3, 1, 2).clone()
            trace["fused_shrink"] = fs.permute(0, 3, 1, 2).clone()
        return heads, com, nz

    def _post_encode_single_scale(self, s, psm_single, n, H, W, record_len, nz, trace):
        """``multi_scale: false`` (airv2x_where2com.py:147-150, 163-166; where2comm_fuse.py:264-286): compressor (if any) on the shrunk map,
        communication mask x map, ONE per-pixel attention over each sample's agents at 256 channels, heads on the fused map (no second
        shrink).  Same kernels as the multi-scale levels (av2x_apply_mask, pixel_attn_kernel<4>)."""
        B = len(record_len)
        st = self.stream()
        c = s.shape[-1]
        if self.compressor:
            self.run_compressor(s, n, H, W)                      # in place: encoder -> decoder writes s back (naive_compress.py:38-42)
            if trace is not None:
                trace["compressed"] = s.permute(0, 3, 1, 2).clone()
        if self.fcfg["fully"]:
            com = torch.tensor(1, device=self.device)
        else:
            mask, count, smooth, rl = self.comm_mask(psm_single, n, H, W, record_len)
            com = self.comm_rate(count, rl, B, H * W)
            self.timed_hbm("apply_mask (in place)", n * H * W * (2 * c + 1) * 4, 0.0,
                           lambda: _lib.check(self.lib.av2x_apply_mask(_ptr(s), _ptr(mask), n, H * W, c, st), "av2x_apply_mask"))
            if trace is not None:
                trace["comm_mask"] = mask.unsqueeze(1).clone()
                trace["comm_map"] = smooth.unsqueeze(1).clone()
        fused = self.buf("fused_single", (B, H, W, c))
        a0 = 0
        for b, k in enumerate(record_len):
            self.attn([s[j].data_ptr() for j in range(a0, a0 + k)], H * W, c, fused[b])
            a0 += k
        heads = torch.empty((B, self.heads.cout, H, W), dtype=torch.float32, device=self.device)
        self.conv(self.heads, fused, B, H, W, heads)
        if trace is not None:
            trace["fused_2d"] = fused.permute(0, 3, 1, 2).clone()
        return heads, com, nz

    def _post_encode_groups(self, canvas, ny, nx, n):
        """B == 1 frame with the per-agent work split into agent groups on separate HIP streams.
        Agents are independent until the attention, and every layer launch ends in a partially filled
        last wave of workgroups; running two half-batches concurrently lets one group's layer tail overlap
        the other group's next layer.  Results are bit-identical to the single-stream schedule."""
        G = min(self.agent_streams, n)
        if self._streams is None or len(self._streams) < G:
            self._streams = pooled_streams(self.device, G)
        main = torch.cuda.current_stream()
        st = self.stream()
        nz = self.count_canvas(canvas, st)
        count = self.buf("comm_count", (1,), torch.int32)
        _lib.check(self.lib.av2x_fill_zero(_ptr(count), 4, st), "av2x_fill_zero")
        bounds = [(g * n) // G for g in range(G + 1)]
        fork = torch.cuda.Event()
        fork.record(main)
        groups, H, W = [], None, None
        for g in range(G):
            g0, g1 = bounds[g], bounds[g + 1]
            ng, tag = g1 - g0, f"g{g}of{G}"
            s_g = self._streams[g]
            s_g.wait_event(fork)
            with torch.cuda.stream(s_g):
                cg = canvas[g0:g1]
                feats, s, H, W = self.trunk(cg, ng, ny, nx, tag=tag)
                psm_single = self.buf("psm_single" + tag, (ng, H, W, self.A * self.C))
                self.conv(self.cls_single, s, ng, H, W, psm_single)
                (b0, h0, w0), (b1, h1, w1), (b2, h2, w2) = feats
                if (h0, w0) != (H, W):
                    raise NotImplementedError("mask/feature size mismatch (where2comm_fuse.py:230) never happens for AirV2X")
                mask, _, _, _ = self.comm_mask(psm_single, ng, H, W, [ng], has_ego=(g == 0), tag=tag, count=count)
                _lib.check(self.lib.av2x_apply_mask(_ptr(b0), _ptr(mask), ng, h0 * w0, b0.shape[-1], self.stream()),
                           "av2x_apply_mask")
                first = 1 if g == 0 else 0
                m1 = m2 = None
                if ng - first > 0:
                    m1, _, _ = self.run_block(1, b0[first:], ng - first, h0, w0, "masked" + tag)
                    m2, _, _ = self.run_block(2, m1, ng - first, h1, w1, "masked" + tag)
                done = torch.cuda.Event()
                done.record(s_g)
            groups.append((g0, g1, first, b0, b1, b2, m1, m2, done))
        for gr in groups:
            main.wait_event(gr[-1])
        dims = self.level_dims(ny, nx)
        fused = []
        for i, (h, w, c) in enumerate(dims):
            f = self.buf(f"fused{i}", (1, h, w, c))
            ptrs = []
            for (g0, g1, first, b0, b1, b2, m1, m2, _) in groups:
                unm, msk = (b0, None) if i == 0 else ((b1, m1) if i == 1 else (b2, m2))
                for l in range(g1 - g0):
                    if i == 0 or l < first:
                        ptrs.append(unm[l].data_ptr())
                    else:
                        ptrs.append(msk[l - first].data_ptr())
            self.attn(ptrs, h * w, c, f[0])
            fused.append((f, h, w))
        catf = self.buf("cat_fused", (1, H, W, self.cat_c))
        self.run_deblocks(fused, 1, catf)
        fs = self.run_shrink(catf, 1, H, W, "fused") if self.shrink else catf
        heads = torch.empty((1, self.heads.cout, H, W), dtype=torch.float32, device=self.device)
        self.conv(self.heads, fs, 1, H, W, heads)
        com = self.comm_rate(count, self.comm_layout((n,), True)[2], 1, H * W)
        return heads, com, nz

    # ------------------------------------------------------------------ agent-sharded frame (one frame over N GPUs)
    def level_dims(self, ny, nx):
        dims, h, w = [], ny, nx
        for layers in self.blocks:
            h, w = (h + 2 - 3) // layers[0].stride + 1, (w + 2 - 3) // layers[0].stride + 1
            dims.append((h, w, layers[0].cout))
        return dims

    def canvas_dims(self):
        """(ny, nx) of the pillar canvas from the model config (needed by a rank that holds no agent of a sharded frame)."""
        t = next(iter(self.pfn))
        g = [int(v) for v in self.args[t]["lidar"]["point_pillar_scatter"]["grid_size"]]
        return g[1], g[0]

    def shard_frame_agents(self, data_dict_local):
        """Number of agents this rank holds of a sharded frame (0 for ``None`` / a dict without lidar input) + layout."""
        if data_dict_local is None:
            return 0, None, None
        record_len, slots = self.frame_layout(data_dict_local)
        if len(record_len) > 1:
            raise ValueError("agent sharding handles one collaborative frame (B = 1) per step")
        n = record_len[0] if record_len else 0
        return n, record_len, slots

    @torch.no_grad()
    def shard_local_stage(self, data_dict_local, has_ego, n_pad=None):
        """Per-rank half of an agent-sharded frame: encode + trunk + confidence mask +
        masked blocks for THIS rank's agents, written straight into the all-gather send buffer
        [level0: n_pad maps | level1 | level2] (15.77 MB per agent at the default grid; ``n_pad`` >= the local
        agent count is the largest count of any rank: an uneven frame pads the message, a rank without agents sends
        only padding).
        Returns (send flat f32, stats int64[2] = [mask ones before ego override, canvas non-zeros], meta)."""
        n, record_len, slots = self.shard_frame_agents(data_dict_local)
        n_pad = n if n_pad is None else int(n_pad)
        if n_pad < max(n, 1):
            raise ValueError(f"n_pad = {n_pad} is smaller than this rank's {n} agents")
        if has_ego and n == 0:
            raise ValueError("the ego's rank holds no agent")
        if n > 0:
            canvas, ny, nx = self.encode(data_dict_local, record_len, slots)
        else:
            ny, nx = self.canvas_dims()
        dims = self.level_dims(ny, nx)
        H, W = self.cat_hw(dims)
        if not self.multi_scale:
            # single-scale Where2comm (multi_scale: false): ONE level -- the shrunk (compressed + decompressed) 256-channel map times the
            # agent's communication mask, 36.0 MB per agent at the default grid.  The mask multiplies the DECODED map
            # (airv2x_where2com.py:147-150, where2comm_fuse.py:264-275), so the compressor runs whole on the sender.
            C = self.feat_c
            send = self.buf("shard_send", (n_pad * H * W * C,))
            meta = {"dims": [(H, W, C)], "n_loc": n_pad, "H": H, "W": W}
            if n == 0:
                return send, torch.zeros(2, dtype=torch.int64, device=self.device), meta
            st = self.stream()
            nz = self.count_canvas(canvas, st)
            s = send[:n * H * W * C].view(n, H, W, C)
            self.trunk(canvas, n, ny, nx, shrink_out=s)
            psm_single = self.buf("psm_single", (n, H, W, self.A * self.C))
            self.conv(self.cls_single, s, n, H, W, psm_single)
            if self.compressor:
                self.run_compressor(s, n, H, W)
            if self.fcfg["fully"]:
                ones = torch.zeros((), dtype=torch.int64, device=self.device)
            else:
                mask, count, _, _ = self.comm_mask(psm_single, n, H, W, record_len, has_ego=has_ego)
                _lib.check(self.lib.av2x_apply_mask(_ptr(s), _ptr(mask), n, H * W, C, st), "av2x_apply_mask")
                ones = count.sum().to(torch.int64)
            return send, torch.stack([ones, nz[0]]), meta
        sizes = [h * w * c for h, w, c in dims]
        send = self.buf("shard_send", (n_pad * sum(sizes),))
        meta = {"dims": dims, "n_loc": n_pad, "H": H, "W": W}
        if n == 0:   # nothing to compute; the padding is never read by the fusion
            return send, torch.zeros(2, dtype=torch.int64, device=self.device), meta
        body = lambda: self._shard_local_body(canvas, send, n, n_pad, has_ego, ny, nx, record_len, dims, sizes)
        stats = self._graphed(("shard_local", n, n_pad, bool(has_ego), ny, nx, canvas.data_ptr(), send.data_ptr(), self.sparse_eligible()), body)
        return send, stats, meta

    def _graphed(self, key, body):
        """``body()`` -> tensor(s), eagerly or -- with ``use_graph`` -- replayed from a hipGraph captured on first use (the per-rank stages of
        an agent-sharded frame are 29 / 33 launches of one agent's share of the work: launch-bound from ~4 ranks on, bench.py --dry-run).
        Everything ``body`` touches must live at addresses that are part of ``key`` (pool buffers); results are cloned out of the graph's pool."""
        if not self.use_graph or self.profile is not None or torch.cuda.is_current_stream_capturing():
            return body()
        ent = self.graphs.get(key)
        if ent is None:
            body()                                  # one eager pass: workspace buffers, tile choices, packed weights exist before the capture
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                static = body()
            ent = self.graphs[key] = (g, static)
        g, static = ent
        g.replay()
        return static.clone() if isinstance(static, torch.Tensor) else tuple(t.clone() for t in static)

    def _shard_local_body(self, canvas, send, n, n_pad, has_ego, ny, nx, record_len, dims, sizes):
        st = self.stream()
        nz = self.count_canvas(canvas, st)
        lv, off = [], 0
        for (h, w, c), f in zip(dims, sizes):
            lv.append(send[off:off + n * f].view(n, h, w, c))
            off += n_pad * f
        if self.fcfg["fully"]:
            # fully connected communication graph (where2comm_fuse.py:222-223): no mask, the UNMASKED block outputs are the message (and the
            # per-agent deblocks / shrink header / confidence head, whose only consumer is the mask, are not launched); com = 1
            x, h, w = canvas, ny, nx
            for i in range(len(self.blocks)):
                x, h, w = self.run_block(i, x, n, h, w, "all", out=lv[i])
            return torch.stack([torch.zeros((), dtype=torch.int64, device=self.device), nz[0]])
        feats, s, H, W = self.trunk(canvas, n, ny, nx, block_out={0: lv[0]})
        psm_single = self.buf("psm_single", (n, H, W, self.A * self.C))
        self.conv(self.cls_single, s, n, H, W, psm_single)
        mask, count, _, _ = self.comm_mask(psm_single, n, H, W, record_len, has_ego=has_ego)
        (b0, h0, w0), (b1, h1, w1), (b2, h2, w2) = feats
        if (h0, w0) != (H, W):      # where2comm_fuse.py:229-235 (backbone variants whose shared map is not at the first block's resolution)
            mask_r = self.buf("comm_mask_resized", (n, h0, w0))
            _lib.check(self.lib.av2x_mask_resize_bilinear(_ptr(mask), n, H, W, h0, w0, _ptr(mask_r), st), "av2x_mask_resize_bilinear")
            mask = mask_r
        _lib.check(self.lib.av2x_apply_mask(_ptr(b0), _ptr(mask), n, h0 * w0, b0.shape[-1], st), "av2x_apply_mask")
        first = 1 if has_ego else 0
        if n - first > 0:
            m1, _, _ = self.run_block(1, b0[first:], n - first, h0, w0, "masked", out=lv[1][first:])
            self.run_block(2, m1, n - first, h1, w1, "masked", out=lv[2][first:])
        if has_ego:  # the ego's mask is all ones: its "masked" features are the unmasked ones
            lv[1][0].copy_(b1[0])
            lv[2][0].copy_(b2[0])
        return torch.stack([count.sum().to(torch.int64), nz[0]])

    @torch.no_grad()
    def shard_ego_stage(self, recv, stats, meta, world, sync_comm_rate=False):
        """Ego half: per-pixel attention over all gathered agents (pointer arithmetic into the all-gather buffer, no
        regroup copy; ``meta["counts"]`` = agents per rank when the frame is uneven, else every rank holds n_loc),
        deblocks, shrink, heads."""
        dims, n_loc, H, W = meta["dims"], meta["n_loc"], meta["H"], meta["W"]
        counts = meta.get("counts") or [n_loc] * world
        sizes = [h * w * c for h, w, c in dims]
        per_rank = n_loc * sum(sizes)
        if recv.numel() != world * per_rank or len(counts) != world or max(counts) > n_loc:
            raise ValueError("gathered buffer has the wrong size")
        def body():
            base = recv.data_ptr()
            fused, off = [], 0
            for i, ((h, w, c), f) in enumerate(zip(dims, sizes)):
                out = self.buf(f"fused{i}", (1, h, w, c))
                ptrs = [base + 4 * (r * per_rank + off + j * f) for r in range(world) for j in range(counts[r])]
                self.attn(ptrs, h * w, c, out[0])
                fused.append((out, h, w))
                off += n_loc * f
            if not self.multi_scale:        # single scale: the heads read the fused map directly (airv2x_where2com.py:163-169)
                fs = fused[0][0]
            else:
                catf, _, _ = self.run_cat(fused, 1, "fused")
                fs = self.run_shrink(catf, 1, H, W, "fused") if self.shrink else catf
            heads = torch.empty((1, self.heads.cout, H, W), dtype=torch.float32, device=self.device)
            self.conv(self.heads, fs, 1, H, W, heads)
            return heads
        # the gathered buffer's address is part of the key: the sharded frame receives into a pool buffer (EngineBackend.recv_buffer)
        heads = self._graphed(("shard_ego", recv.data_ptr(), world, tuple(counts), n_loc, H, W), body)
        outs = torch.split(heads, self.head_splits, dim=1)
        out = {"psm": outs[0], "rm": outs[1]}
        if self.args["obj_head"]:
            out["obj"] = outs[2]
        n_total = sum(counts)
        com = torch.tensor(1, device=self.device) if self.fcfg["fully"] else stats[0].to(torch.float32) / float(n_total * H * W)
        comm_rate = int(stats[1].item()) if sync_comm_rate else stats[1]
        out.update({"mask": 0, "com": com, "comm_rate": comm_rate})
        return out


_STREAM_POOL = {}


def pooled_streams(device, k):
    """The process's HIP side streams of ``device``, the first ``k`` of them.  HIP multiplexes streams onto FOUR hardware queues
    (GPU_MAX_HW_QUEUES): the null stream plus three frame streams use exactly four, and every further stream a process has ever created makes
    two frames share a queue -- measured: a fifth stream costs a 3-deep FramePipeline 25 % (profiles/r05z_deblock_overlap.txt), and
    --inflight 4 runs slower than 3.  Hence ONE pool per device: FramePipeline, ShardedPipeline, the agent-group streams and the training
    step's weight-gradient stream all take their streams from here instead of creating their own."""
    dev = torch.device(device)
    dev = torch.device("cuda", torch.cuda.current_device()) if dev.index is None else dev
    pool = _STREAM_POOL.setdefault(dev, [])
    while len(pool) < k:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[:k]


class FramePipeline:
    """Throughput mode: ``depth`` independent frames in flight, each on its own HIP stream with its own
    workspaces (weights shared).  Every layer launch ends in a partially filled last round of
    workgroups and the ego stage of a frame is a single-image tail; a second frame's kernels fill those
    gaps.  Per-frame results are bit-identical to the sequential forward of the same engine (the "rule" stream-K
    schedule is a function of the layer shapes and stays on); per-frame latency grows.  Only the legacy timing-driven
    stream-K mode ("tune") is switched off here: its persistent workgroups fill the chip on their own, so overlapping
    frames gained nothing on top (measured: -0.8 % vs +5 % single-stream)."""

    def __init__(self, engine, depth=2):
        # the pipeline's mode, applied around each of ITS frames (engine.frame_mode): whole frames per GPU -> never the agent-sharded classes;
        # more than one frame in flight -> throughput mode (engine.wino4_rule).  The caller's engine keeps its own flags for its own calls.
        self.throughput_mode = depth > 1
        self.engines = [engine] + [engine.share_weights() for _ in range(depth - 1)]
        if depth > 3:
            import warnings
            warnings.warn(f"FramePipeline(depth={depth}): HIP multiplexes streams onto four hardware queues (the null stream + three frame "
                          "streams); a fourth frame stream shares a queue and measured SLOWER than depth 3 (433-441 against 521 frames/s)")
        self.streams = pooled_streams(engine.device, depth)
        self.events = [None] * depth
        self.i = 0

    def submit(self, data_dict, after=None, **kw):
        """Enqueue one frame; returns (output dict, event recorded on the frame's stream).  ``after(out, slot)`` is
        called inside the frame's stream context (e.g. VoxelPostprocessor.launch): its kernels join the frame."""
        k = self.i % len(self.engines)
        self.i += 1
        s = self.streams[k]
        s.wait_stream(torch.cuda.current_stream())  # inputs produced on the caller's stream
        eng = self.engines[k]
        # "rule" stays on (a frame's bits must not depend on whether it was pipelined); the legacy timing-driven stream-K
        # gains nothing with several frames in flight and is switched off
        legacy = eng.stream_k is True or eng.stream_k == "tune"
        saved, eng.stream_k = eng.stream_k, (eng.stream_k if (not legacy or len(self.engines) == 1 or os.environ.get("AV2X_PIPE_SK") == "1") else False)
        try:
            with torch.cuda.stream(s), eng.frame_mode(self.throughput_mode, False):
                out = eng.forward(data_dict, **kw)
        finally:
            eng.stream_k = saved
        with torch.cuda.stream(s):
            if after is not None:
                out = after(out, k)
            ev = torch.cuda.Event()
            ev.record(s)
        self.events[k] = ev
        return out, ev

    def drain(self):
        for s in self.streams:
            torch.cuda.current_stream().wait_stream(s)
