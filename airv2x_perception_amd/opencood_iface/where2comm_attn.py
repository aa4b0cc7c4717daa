"""OPV2V-style Where2comm fusion: the mirror of the reference's
``opencood/models/where2comm_modules/where2comm_attn.py`` (Where2comm :222-404, AttenFusion :55-67, MaxFusion :70-75) with
``where2comm_modules/where2comm.py``'s Communication (:10-116), running in libairv2x_hip.so.

Same constructor arguments, ``state_dict`` keys (``naive_communication.gaussian_filter.{weight,bias}`` when the
configuration smooths the confidence map, otherwise none) and call convention:

    fused, communication_rates, {} = fusion_net(x, rm, record_len, pairwise_t_matrix, backbone, heads)

with the reference's as-written semantics (oracle/where2comm_attn_oracle.py lists them; tests/golden/w2c_attn.npz pins
them): the second return value is the communication VOLUME of where2comm.py:93-116 as a numpy float64, the even agents of
every sample transmit everything, and the single-scale branch indexes the concatenated mask tensor with the sample index.

Device schedule per level: backbone block (implicit-GEMM convs) -> [level 0: confidence -> smoothing -> threshold in one
mask kernel, gated count of the non-zero cells, in-place masking] -> ONE ``av2x_warp_fuse`` launch per sample that samples
every agent through its ego-row matrix and applies the per-pixel attention (or max) on the fly -> deblock written straight
into its channel slice of the concatenated output.  The warped maps never exist in HBM.

The ResNet backbone variant (``backbone.resnet``, base_bev_backbone_resnet.py -> submodules.ResNetBEVBackbone) is built.
Raises, as the reference does: the 'Transformer' aggregation -- ``Where2comm.forward`` calls ``self.fuse_modules[i](neighbor_feature)``
with ONE argument (:360) where ``TransformerFusion.forward`` takes four (:130-136: TypeError), its EncodeLayer passes a ``quality_map``
keyword that ``nn.MultiheadAttention`` does not accept (:105-107), and the single-scale form stores the module as ``fuse_network`` but
calls ``fuse_modules`` (:262, :399): no configuration of that mode can run in the reference.  GPU only; ``.train()`` forwards are
differentiable in x and in the backbone's parameters (``_forward_train``), over either backbone.
"""
from __future__ import annotations

from ctypes import c_void_p

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from .engine import _ptr
from .submodules import BaseBEVBackbone, _HipModule, _Runner, _declare, _lens, _nchw, _nhwc, _nhwc_grad

_MODES = {"ATTEN": 0, "MAX": 1}


def _agent_ptrs(x, a0, k):
    return (c_void_p * k)(*[x[j].data_ptr() for j in range(a0, a0 + k)])


class AttenFusion(nn.Module):
    """where2comm_attn.py:55-67: (cav_num, C, H, W) aligned maps -> (C, H, W), the ego's row of the per-pixel attention."""

    def __init__(self, feature_dim):
        super().__init__()
        self.feature_dim = feature_dim

    @torch.no_grad()
    def forward(self, x):
        lib = _lib.load()
        y = _nhwc(x)
        n, h, w, c = y.shape
        out = torch.empty((h, w, c), dtype=torch.float32, device=y.device)
        st = c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.av2x_pixel_attn_fuse(_agent_ptrs(y, 0, n), n, h * w, c, _ptr(out), st), "av2x_pixel_attn_fuse")
        return out.permute(2, 0, 1)


class MaxFusion(nn.Module):
    """where2comm_attn.py:70-75."""

    @torch.no_grad()
    def forward(self, x):
        lib = _lib.load()
        y = _nhwc(x)
        n, h, w, c = y.shape
        out = torch.empty((h, w, c), dtype=torch.float32, device=y.device)
        st = c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.av2x_agent_max(_agent_ptrs(y, 0, n), n, h * w * c, _ptr(out), st), "av2x_agent_max")
        return out.permute(2, 0, 1)


def normalized_pairwise(pairwise_t_matrix, H, W, discrete_ratio, downsample_rate):
    """where2comm_attn.py:293-307 on the host (a few floats per agent pair): (B,L,L,4,4) -> (B,L,L,2,3) fp32 numpy,
    computed in fp32 like the reference does for the fp32 matrices the datasets ship.  The caller's tensor is not touched
    (the reference's fancy indexing copies too)."""
    m = pairwise_t_matrix.detach().to("cpu", torch.float32).numpy()
    m = m[:, :, :, [0, 1], :][:, :, :, :, [0, 1, 3]].copy()
    H32, W32 = np.float32(H), np.float32(W)
    m[..., 0, 1] = m[..., 0, 1] * H32 / W32
    m[..., 1, 0] = m[..., 1, 0] * W32 / H32
    m[..., 0, 2] = m[..., 0, 2] / np.float32(downsample_rate * discrete_ratio * W) * np.float32(2)
    m[..., 1, 2] = m[..., 1, 2] / np.float32(downsample_rate * discrete_ratio * H) * np.float32(2)
    return np.ascontiguousarray(m, dtype=np.float32)


class Where2comm(_HipModule):
    """where2comm_attn.py:222-404."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.communication = "communication" in args
        self.round = args["communication"].get("round", 1) if self.communication else 1
        self.discrete_ratio = args["voxel_size"][0]
        self.downsample_rate = args["downsample_rate"]
        self.agg_mode = args["agg_operator"]["mode"]
        if self.agg_mode == "Transformer":
            raise NotImplementedError("agg_operator mode 'Transformer' cannot run in the reference either: Where2comm.forward calls the "
                                      "fusion module with one argument where TransformerFusion.forward takes four "
                                      "(where2comm_attn.py:360 vs :130-136), and EncodeLayer passes nn.MultiheadAttention a "
                                      "quality_map keyword it does not have (:105-107)")
        if self.agg_mode not in _MODES:
            raise ValueError(f"agg_operator mode {self.agg_mode!r}: ATTEN or MAX")
        self.multi_scale = args["multi_scale"]
        make = (lambda c: AttenFusion(c)) if self.agg_mode == "ATTEN" else (lambda c: MaxFusion())
        if self.multi_scale:
            self.num_levels = len(args["layer_nums"])
            self.fuse_modules = nn.ModuleList([make(c) for c in args["num_filters"]])
        else:
            self.fuse_modules = make(args["agg_operator"].get("feature_dim"))
        self.thre = float(args["communication"]["thre"]) if self.communication else 0.0
        if self.communication and "gaussian_smooth" in args["communication"]:
            g = args["communication"]["gaussian_smooth"]
            k, s = g["k_size"], g["c_sigma"]
            _declare(self, [("naive_communication.gaussian_filter.weight", (1, 1, k, k), "zeros"),
                            ("naive_communication.gaussian_filter.bias", (1,), "zeros")])
            c = k // 2
            gx, gy = np.mgrid[0 - c:k - c, 0 - c:k - c]
            # init_gaussian_filter (where2comm.py:28-45); checkpoints overwrite it
            gk = 1 / (2 * np.pi * s) * np.exp(-(np.square(gx) + np.square(gy)) / (2 * np.square(s)))
            with torch.no_grad():
                self.naive_communication.gaussian_filter.weight.copy_(torch.Tensor(gk).view(1, 1, k, k))

    # ------------------------------------------------------------------ device state
    def _make_runner(self, device):
        return _Runner(device)

    def _pack(self, r, sd):
        w = sd.get("naive_communication.gaussian_filter.weight")
        if w is None:   # no smoothing: a 1 x 1 identity filter, so the same mask kernel serves both forms
            r.gauss_w, r.gauss_b, r.gauss_k = torch.ones(1, device=r.device), torch.zeros(1, device=r.device), 1
        else:
            r.gauss_w = w.detach().to(r.device, torch.float32).reshape(-1).contiguous()
            r.gauss_b = sd["naive_communication.gaussian_filter.bias"].detach().to(r.device, torch.float32).reshape(1).contiguous()
            r.gauss_k = int(w.shape[-1])

    def runner(self, train_ok=False):
        if self._tensors():
            return super().runner(train_ok=train_ok)
        if self.training and not train_ok:
            raise NotImplementedError("Where2comm: this forward has no training form; call .eval()")
        if self._runner_obj is None:
            self.__dict__["_runner_obj"] = self._make_runner(torch.device("cuda", torch.cuda.current_device()))
            self._pack(self._runner_obj, {})
        return self._runner_obj

    def regroup(self, x, record_len):
        cum = torch.cumsum(torch.as_tensor(record_len), dim=0)
        return torch.tensor_split(x, cum[:-1].cpu())

    # ------------------------------------------------------------------ Communication.forward (where2comm.py:47-116)
    def _communicate(self, r, x, rm, lens):
        """x (n,h,w,c) features, rm (n,A,h,w)-shaped confidence logits -> (mask (n,h,w) with the even agents of every sample
        forced to one, volume counter (1,) u64 on the device)."""
        psm = _nhwc(rm)
        n, h, w, a = psm.shape
        if (n, h, w) != tuple(x.shape[:3]):
            raise ValueError(f"confidence maps {tuple(psm.shape[:3])} do not match the features {tuple(x.shape[:3])}")
        key = ("w2c_attn_layout", tuple(lens))
        lay = r.ws.get(key)
        if lay is None:
            samp = [b for b, k in enumerate(lens) for _ in range(k)]
            even = [1 - (j & 1) for k in lens for j in range(k)]          # communication_mask_nodiag[::2] = 1 (:104-108)
            lay = (torch.tensor(samp, dtype=torch.int32, device=r.device), torch.tensor(even, dtype=torch.int32, device=r.device))
            r.ws[key] = lay
        conf, smooth, mask = (r.buf("wa_" + t, (n, h, w)) for t in ("conf", "smooth", "mask"))
        count = r.buf("wa_count", (len(lens),), torch.int32)
        vol = r.buf("wa_vol", (1,), torch.int64)
        st = r.stream()
        _lib.check(r.lib.av2x_fill_zero(_ptr(count), count.numel() * 4, st), "av2x_fill_zero")
        _lib.check(r.lib.av2x_fill_zero(_ptr(vol), 8, st), "av2x_fill_zero")
        # the kernel's "threshold <= 0: all ones" shortcut equals `maps > thre` for thre <= 0 only while the smoothed
        # confidence is positive (sigmoid outputs through a positive filter); a non-positive thre is not a useful setting
        if self.thre <= 0:
            raise NotImplementedError("communication.thre must be positive")
        _lib.check(r.lib.av2x_comm_mask(_ptr(psm), n, h, w, a, a, _ptr(r.gauss_w), _ptr(r.gauss_b), r.gauss_k, self.thre,
                                        _ptr(lay[0]), _ptr(lay[1]), _ptr(conf), _ptr(smooth), _ptr(mask), _ptr(count), st),
                   "av2x_comm_mask")
        _lib.check(r.lib.av2x_count_nonzero_where(_ptr(x), _ptr(smooth), self.thre, n * h * w, x.shape[3], _ptr(vol), st),
                   "av2x_count_nonzero_where")
        return mask, vol

    def _fuse(self, r, x, lens, theta, out):
        """One warp + fuse launch per sample.  x (n,h,w,c); theta (B,L,L,2,3) numpy; out (B,h,w,c)."""
        n, h, w, c = x.shape
        mode = _MODES[self.agg_mode]
        a0 = 0
        for b, k in enumerate(lens):
            th = np.ascontiguousarray(theta[b, 0, :k], dtype=np.float32)
            _lib.check(r.lib.av2x_warp_fuse(_agent_ptrs(x, a0, k), th.ctypes.data_as(c_void_p), k, h, w, c, mode, _ptr(out[b]),
                                            r.stream()), "av2x_warp_fuse")
            a0 += k

    # ------------------------------------------------------------------ train mode
    def _forward_train(self, x, rm, lens, pairwise_t_matrix, backbone):
        """Train mode of where2comm_attn.py:275-404: the same schedule on differentiable nodes (HIP forward / backward pairs) --
        backbone blocks / deblocks with BatchNorm batch statistics (submodules.BaseBEVBackbone), the communication mask as a constant
        (torch.where over constants carries no gradient, where2comm.py:84-86), ``warp_affine_simple`` with its adjoint, the per-pixel
        attention / the maximum over the warped agents.  Differentiable in x and in the backbone's parameters (the module itself has
        none that receive a gradient: the smoothing filter only shapes the mask)."""
        from . import train_ops as T
        from .train_v2vnet import AgentMaxFn
        from .train_when2com import warp_affine_simple
        r = self.runner(train_ok=True)
        _, C, H, W = x.shape
        B = len(lens)
        theta = normalized_pairwise(pairwise_t_matrix, H, W, self.discrete_ratio, self.downsample_rate)
        th = [torch.from_numpy(np.ascontiguousarray(theta[b, 0, :k], dtype=np.float32)).to(r.device) for b, k in enumerate(lens)]

        def fuse(cur):
            outs, a0 = [], 0
            for b, k in enumerate(lens):
                warped = warp_affine_simple(cur[a0:a0 + k], th[b])
                outs.append(T.PixelAttn.apply(warped) if self.agg_mode == "ATTEN" else AgentMaxFn.apply(warped)[0])
                a0 += k
            return torch.stack(outs)

        cur = _nhwc_grad(x)
        vol = None
        if not self.multi_scale:
            if self.communication:
                with torch.no_grad():
                    mask, vol = self._communicate(r, cur.detach(), rm, lens)
                    # as written (:394): the mask of GLOBAL agent b for every agent of sample b
                    m = torch.cat([mask[b:b + 1].expand(k, -1, -1) for b, k in enumerate(lens)]).contiguous()
                cur = T.MaskMul.apply(cur, m)
            return _nchw(fuse(cur)), self._volume(vol, B, r), {}
        if not isinstance(backbone, BaseBEVBackbone):
            raise TypeError("Where2comm (MI355X build): `backbone` must be the BaseBEVBackbone / ResNetBEVBackbone of this build")
        with_resnet = hasattr(backbone, "resnet")                  # :312-314: every level from the UNMASKED input, as written
        if with_resnet:
            feats = backbone._train_resnet(cur)
        ups = []
        for i in range(self.num_levels):
            cur = feats[i] if with_resnet else backbone._train_block(i, cur)
            if i == 0 and self.communication:
                with torch.no_grad():
                    mask, vol = self._communicate(r, cur.detach(), rm, lens)
                    mask = mask.clone()
                cur = T.MaskMul.apply(cur, mask)
            fused = fuse(cur)
            ups.append(backbone._train_deblock(i, fused) if backbone.model_cfg.get("upsample_strides") else fused)
        if len(ups) > 1 and not backbone.model_cfg.get("upsample_strides"):
            raise NotImplementedError("multi-scale Where2comm without deblocks needs a single level")
        out = torch.cat(ups, -1) if len(ups) > 1 else ups[0]
        if len(backbone.model_cfg.get("upsample_strides", [])) > self.num_levels:            # where2comm_attn.py:369-370
            out = backbone._train_deblock(self.num_levels, out)
        return _nchw(out), self._volume(vol, B, r), {}

    # ------------------------------------------------------------------ forward
    def forward(self, x, rm, record_len, pairwise_t_matrix, backbone=None, heads=None):
        if x.device.type != "cuda":
            raise RuntimeError("Where2comm (MI355X build) has no CPU path: move the module and its inputs to the GPU")
        if self.training:
            lens = _lens(record_len)
            if any(k < 1 for k in lens):
                raise ValueError("every sample needs at least the ego agent")
            if pairwise_t_matrix.shape[0] != len(lens) or max(lens) > pairwise_t_matrix.shape[1] or sum(lens) != x.shape[0]:
                raise ValueError("pairwise_t_matrix / record_len do not match the agents")
            return self._forward_train(x, rm, lens, pairwise_t_matrix, backbone)
        with torch.no_grad():
            return self._forward_eval(x, rm, record_len, pairwise_t_matrix, backbone, heads)

    def _forward_eval(self, x, rm, record_len, pairwise_t_matrix, backbone=None, heads=None):
        r = self.runner()
        lens = _lens(record_len)
        if any(k < 1 for k in lens):
            raise ValueError("every sample needs at least the ego agent")
        _, C, H, W = x.shape
        B, L = pairwise_t_matrix.shape[:2]
        if B != len(lens) or max(lens) > L:
            raise ValueError("pairwise_t_matrix does not match record_len")
        theta = normalized_pairwise(pairwise_t_matrix, H, W, self.discrete_ratio, self.downsample_rate)
        cur = _nhwc(x)
        if sum(lens) != cur.shape[0]:
            raise ValueError("record_len does not sum to the number of agents")
        st = r.stream()
        vol = None
        if not self.multi_scale:
            if self.communication:
                mask, vol = self._communicate(r, cur, rm, lens)
                # as written (:394): `node_features * communication_masks[b]` -- the mask of GLOBAL agent b for the whole sample
                masked = r.buf("wa_masked", tuple(cur.shape))
                masked.copy_(cur)
                a0 = 0
                for b, k in enumerate(lens):
                    for j in range(a0, a0 + k):
                        _lib.check(r.lib.av2x_apply_mask(_ptr(masked[j]), _ptr(mask[b]), 1, cur.shape[1] * cur.shape[2],
                                                         cur.shape[3], st), "av2x_apply_mask")
                    a0 += k
                cur = masked
            out = torch.empty((B,) + tuple(cur.shape[1:]), dtype=torch.float32, device=r.device)
            self._fuse(r, cur, lens, theta, out)
            return _nchw(out), self._volume(vol, B, r), {}
        with_resnet = hasattr(backbone, "resnet")                  # where2comm_attn.py:312-314
        if not isinstance(backbone, BaseBEVBackbone):
            raise TypeError("Where2comm (MI355X build): `backbone` must be the BaseBEVBackbone of this build")
        br = backbone.runner()
        if len(br.blocks) < self.num_levels:
            raise ValueError("the backbone has fewer levels than the fusion configuration")
        cat, coff, ups = None, 0, []
        feats = backbone.resnet_nhwc(cur) if with_resnet else None    # every level from the UNMASKED input, as written
        for i in range(self.num_levels):
            cur = feats[i] if with_resnet else backbone.block_nhwc(i, cur)
            n, h, w, c = cur.shape
            if i == 0 and self.communication:
                mask, vol = self._communicate(r, cur, rm, lens)
                _lib.check(r.lib.av2x_apply_mask(_ptr(cur), _ptr(mask), n, h * w, c, st), "av2x_apply_mask")
            fused = r.buf(f"wa_fused{i}", (B, h, w, c))
            self._fuse(r, cur, lens, theta, fused)
            if len(br.deblocks) > 0:
                Ld = br.deblocks[i]
                if cat is None:
                    cat = torch.empty((B, h * Ld.up, w * Ld.up, br.cat_c), dtype=torch.float32, device=r.device)
                backbone.deblock_nhwc(i, fused, out=cat, out_ctot=br.cat_c, out_coff=coff)
                coff += Ld.cout
            else:
                ups.append(fused.clone())
        if cat is not None:
            out = cat
        elif len(ups) == 1:
            out = ups[0]
        else:
            raise NotImplementedError("multi-scale Where2comm without deblocks needs a single level")
        return _nchw(out), self._volume(vol, B, r), {}

    def _volume(self, vol, B, r):
        """`communication_vol = np.sum(per-sample counts) / B` (where2comm.py:109): one 8-byte read-back, where the
        reference synchronises once per sample (.item()).  Without a communication block: tensor(0) on the device (:344)."""
        if vol is None:
            return torch.tensor(0).to(r.device)
        return np.float64(int(vol.cpu()[0])) / B
