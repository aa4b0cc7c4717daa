"""Stand-alone mirrors of the reference SUB-modules on the hot path (SURVEY §8b "sub-module signatures the build's
counterparts must keep"): other reference code reaches into these (``backbone.blocks[i](x)`` from
where2comm_fuse.py:218, ``fusion_net(x, psm_single, record_len, t, backbone)`` ...), so each one keeps the
reference's constructor signature, ``state_dict`` keys and call convention while its forward runs in
libairv2x_hip.so.  The three top-level models (airv2x_{where2com,cobevt,v2xvit}.py) do NOT go through
these classes: they drive the kernels directly with fused buffers.

Layout convention: feature maps are NCHW-*shaped* tensors, exactly as in the reference, but every map these
modules return is stored ``channels_last`` (NHWC in memory = the kernels' native layout).  An input that is
already channels_last is consumed in place; a plain-contiguous NCHW input costs one layout copy.  Chains
of these modules therefore never transpose.

CUDA/HIP device only; there is no CPU path.  ``.eval()`` forwards everywhere; ``.train()`` forwards (autograd graph of HIP
forward / backward ops, ``train_ops.py``) for PillarVFE, PointPillarScatter, BaseBEVBackbone, DownsampleConv and Where2comm; the other modules refuse them.
"""
from __future__ import annotations

import ctypes
import weakref
from ctypes import c_void_p

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from ..synth import (backbone_param_spec, compressor_param_spec, fax_param_spec, pfn_param_spec, shrink_param_spec,
                     synthetic_tensor, v2xvit_encoder_spec)
from .airv2x_where2com import _amp_requested, _install, _Node
from .cobevt_engine import CoBEVTEngine
from .engine import Where2ComEngine, _ptr
from .v2xvit_engine import V2XViTEngine

_TILE_CACHE = {}   # autotune results shared by every sub-module runner of the process


# ----------------------------------------------------------------------------------------------- helpers
def _nhwc(x):
    """NCHW-shaped CUDA tensor -> (N,H,W,C) contiguous fp32 tensor; zero-copy when x is channels_last fp32."""
    if x.dim() != 4:
        raise ValueError(f"expected a (N,C,H,W) tensor, got shape {tuple(x.shape)}")
    if x.device.type != "cuda":
        raise RuntimeError("MI355X build: feature maps must live on the GPU (no CPU path)")
    y = x.detach().permute(0, 2, 3, 1)
    if y.dtype != torch.float32:
        y = y.float()
    return y if y.is_contiguous() else y.contiguous()


def _nhwc_grad(x):
    """_nhwc that stays on the autograd graph (train mode)."""
    if x.dim() != 4:
        raise ValueError(f"expected a (N,C,H,W) tensor, got shape {tuple(x.shape)}")
    if x.device.type != "cuda":
        raise RuntimeError("MI355X build: feature maps must live on the GPU (no CPU path)")
    return x.permute(0, 2, 3, 1).float().contiguous()


def _nchw(y):
    """(N,H,W,C) buffer -> NCHW-shaped channels_last view (no copy)."""
    return y.permute(0, 3, 1, 2)


def _lens(record_len):
    if isinstance(record_len, torch.Tensor):
        return [int(v) for v in record_len.detach().cpu().tolist()]
    return [int(v) for v in record_len]


def _declare(root, spec):
    """Register parameters / buffers under the reference's key names (identity-like defaults; real values come
    from load_state_dict)."""
    for key, shape, kind in spec:
        if kind == "count":
            t, buf = torch.zeros(shape, dtype=torch.long), True
        elif kind.startswith("relidx:"):
            t, buf = torch.from_numpy(synthetic_tensor(key, shape, kind)), True
        elif kind in ("bn_m", "bn_v"):
            t, buf = (torch.ones(shape) if kind == "bn_v" else torch.zeros(shape)), True
        elif kind in ("bn_w", "ln_w"):
            t, buf = torch.ones(shape), False
        elif kind in ("gauss_w", "rte_table"):
            t, buf = torch.from_numpy(synthetic_tensor(key, shape, kind)), False
        else:
            t, buf = torch.zeros(shape), False
        _install(root, key, t, buf)


class _Runner(Where2ComEngine):
    """Conv launcher + workspace pool without a model configuration."""

    def __init__(self, device, **cfg):
        super().__init__({"anchor_number": 0, "num_class": 0}, device)
        self.tile_cache = _TILE_CACHE
        for k, v in cfg.items():
            setattr(self, k, v)

    def _init_config(self, args):
        pass


class _HipModule(nn.Module):
    """Parameters under the reference's names + a lazily (re)packed device copy for the kernels."""

    _runner_obj = None
    _packed = None

    def _make_runner(self, device):
        return _Runner(device)

    def _pack(self, runner, sd):   # pragma: no cover - abstract
        raise NotImplementedError

    def _tensors(self):
        return list(self.state_dict(keep_vars=True).values())

    def runner(self, train_ok=False):
        ts = self._tensors()
        dev = ts[0].device if ts else torch.device("cuda", torch.cuda.current_device())
        if dev.type != "cuda":
            raise RuntimeError(f"{type(self).__name__} (MI355X build) has no CPU path: move the module to the GPU")
        if self.training and not train_ok:
            raise NotImplementedError(f"{type(self).__name__}: training is not built; call .eval()")
        ver = tuple(t._version for t in ts) + (dev,)
        if self._runner_obj is None or self._runner_obj.device != dev:
            self.__dict__["_runner_obj"] = self._make_runner(dev)
            self.__dict__["_packed"] = None
        if self._packed != ver:
            self._pack(self._runner_obj, self.state_dict())
            self.__dict__["_packed"] = ver
        self._runner_obj.amp = _amp_requested(self)
        return self._runner_obj


class _Stage(_Node):
    """One entry of a ModuleList of the reference that callers invoke directly (``backbone.blocks[i](x)``)."""

    def bind(self, owner, method, index):
        self.__dict__["_owner"] = weakref.ref(owner)
        self.__dict__["_method"] = method
        self.__dict__["_index"] = index

    def forward(self, x):
        return getattr(self._owner(), self._method)(self._index, x)


def _retype(node, cls):
    node.__class__ = cls
    return node


# ----------------------------------------------------------------------------------------------- PillarVFE / scatter
class PillarVFE(_HipModule):
    """models/common_modules/airv2x_pillar_vfe.py:52-160.  ``forward(batch_dict)`` reads
    ``batch_dict[agent_type]["batch_merged_lidar_features_torch"]`` and stores ``pillar_features`` (M,64) there
    (always 2-D: the reference's ``squeeze()`` quirk for M == 1 is not reproduced)."""

    def __init__(self, model_cfg, num_point_features, voxel_size, point_cloud_range, agent_type):
        super().__init__()
        if (not model_cfg["use_norm"] or model_cfg["with_distance"] or not model_cfg["use_absolute_xyz"]
                or list(model_cfg["num_filters"]) != [64] or num_point_features != 4):
            raise NotImplementedError("PillarVFE: only the shipped configuration (4 point features, absolute xyz, "
                                      "one 64-wide PFN layer with BatchNorm) is built")
        self.model_cfg = model_cfg
        self.agent_type = agent_type
        self.num_filters = list(model_cfg["num_filters"])
        self.voxel_size, self.point_cloud_range = list(voxel_size), list(point_cloud_range)
        _declare(self, pfn_param_spec(""))
        for p_ in self.parameters():
            p_.requires_grad_(True)          # trainable, as the reference's nn.Module is (train mode below)

    def get_output_feature_dim(self):
        return self.num_filters[-1]

    def _pack(self, r, sd):
        r.pfn_w = r.load_pfn(sd, "", self.voxel_size, self.point_cloud_range)

    def forward(self, batch_dict):
        if self.training:
            return self._forward_train(batch_dict)
        with torch.no_grad():
            return self._forward_eval(batch_dict)

    def _forward_train(self, batch_dict):
        """Train mode: BatchNorm1d batch statistics (and the running-statistics update), ``pillar_features`` attached to the
        autograd graph of the Linear / BatchNorm1d parameters (train_ops.PillarFeatures)."""
        from ctypes import c_float

        from . import train_ops as T
        P = dict(self.named_parameters())
        dev = next(iter(P.values())).device
        if dev.type != "cuda":
            raise RuntimeError("PillarVFE (MI355X build) has no CPU path: move the module to the GPU")
        bd = batch_dict[self.agent_type]["batch_merged_lidar_features_torch"]
        vf = bd["voxel_features"].to(dev).contiguous().float()
        vc = bd["voxel_coords"].to(dev).contiguous().to(torch.int32)
        vn = bd["voxel_num_points"].to(dev).contiguous().to(torch.int32)
        if vf.shape[1:] != (32, 4):
            raise ValueError(f"voxel_features must be (M,32,4), got {tuple(vf.shape)}")
        vs, rng = self.voxel_size, self.point_cloud_range
        geom = (c_float * 6)(vs[0], vs[1], vs[2], vs[0] / 2 + rng[0], vs[1] / 2 + rng[1], vs[2] / 2 + rng[2])
        st = []
        p = "pfn_layers.0"
        out = T.PillarFeatures.apply(vf, vc, vn, geom, T.BN_EPS, st, P[p + ".linear.weight"], P[p + ".norm.weight"], P[p + ".norm.bias"])
        sd = self.state_dict(keep_vars=True)
        T.update_running_stats(sd[p + ".norm.running_mean"], sd[p + ".norm.running_var"], sd.get(p + ".norm.num_batches_tracked"), st[0], 1)
        bd["pillar_features"] = out
        return bd

    def _forward_eval(self, batch_dict):
        r = self.runner()
        bd = batch_dict[self.agent_type]["batch_merged_lidar_features_torch"]
        vf = bd["voxel_features"].to(r.device).contiguous().float()
        vc = bd["voxel_coords"].to(r.device).contiguous().to(torch.int32)
        vn = bd["voxel_num_points"].to(r.device).contiguous().to(torch.int32)
        if vf.shape[1:] != (32, 4):
            raise ValueError(f"voxel_features must be (M,32,4), got {tuple(vf.shape)}")
        out = torch.empty((vf.shape[0], 64), dtype=torch.float32, device=r.device)
        w, sc, sh, geom = r.pfn_w
        _lib.check(r.lib.av2x_pillar_vfe(_ptr(vf), _ptr(vc), _ptr(vn), vf.shape[0], _ptr(w), _ptr(sc), _ptr(sh),
                                         ctypes.cast(geom, c_void_p), _ptr(out), r.stream()), "av2x_pillar_vfe")
        bd["pillar_features"] = out
        return bd


class PointPillarScatter(nn.Module):
    """models/common_modules/point_pillar_scatter.py:5-82 (no parameters)."""

    def __init__(self, model_cfg):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = model_cfg["num_features"]
        self.nx, self.ny, self.nz = [int(v) for v in model_cfg["grid_size"]]
        assert self.nz == 1

    def forward(self, batch_dict):
        feats = batch_dict["pillar_features"]
        if feats.device.type != "cuda":
            raise RuntimeError("PointPillarScatter (MI355X build) has no CPU path")
        if torch.is_grad_enabled() and feats.requires_grad:    # train mode upstream: keep the graph (train_ops.PillarScatter)
            from . import train_ops as T
            coords = batch_dict["voxel_coords"].to(feats.device).contiguous().to(torch.int32)
            if feats.shape[1] != self.num_bev_features:
                raise ValueError(f"pillar_features has {feats.shape[1]} channels, expected {self.num_bev_features}")
            batch_size = int(coords[:, 0].max().item()) + 1
            sf = _nchw(T.PillarScatter.apply(feats.float(), coords, batch_size, self.ny, self.nx))
            batch_dict["spatial_features_3d"] = sf.unsqueeze(2)
            batch_dict["spatial_features"] = sf
            return batch_dict
        with torch.no_grad():
            return self._forward_eval(batch_dict)

    def _forward_eval(self, batch_dict):
        feats, coords = batch_dict["pillar_features"], batch_dict["voxel_coords"]
        lib = _lib.load()
        feats = feats.contiguous().float()
        coords = coords.to(feats.device).contiguous().to(torch.int32)
        C = feats.shape[1]
        if C != self.num_bev_features:
            raise ValueError(f"pillar_features has {C} channels, expected {self.num_bev_features}")
        batch_size = int(coords[:, 0].max().item()) + 1                      # point_pillar_scatter.py:43 (host sync)
        canvas = torch.empty((batch_size, self.ny, self.nx, C), dtype=torch.float32, device=feats.device)
        st = c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.av2x_fill_zero(_ptr(canvas), canvas.numel() * 4, st), "av2x_fill_zero")
        _lib.check(lib.av2x_pillar_scatter(_ptr(feats), _ptr(coords), feats.shape[0], C, _ptr(canvas), batch_size,
                                           self.ny, self.nx, st), "av2x_pillar_scatter")
        sf = _nchw(canvas)
        batch_dict["spatial_features_3d"] = sf.unsqueeze(2)
        batch_dict["spatial_features"] = sf
        return batch_dict


# ----------------------------------------------------------------------------------------------- backbone
class BaseBEVBackbone(_HipModule):
    """models/common_modules/base_bev_backbone.py:6-154.  ``blocks[i]`` / ``deblocks[i]`` are callable on
    NCHW-shaped maps like the reference's Sequentials; ``forward(data_dict)`` adds ``spatial_features_2d``."""

    def __init__(self, model_cfg, input_channels):
        super().__init__()
        self.model_cfg = model_cfg
        self.input_channels = input_channels
        ups = model_cfg.get("upsample_strides", [])
        nlev = len(model_cfg["layer_nums"])
        if len(ups) not in (0, nlev, nlev + 1):
            raise ValueError("upsample_strides: one per level, optionally one more for the final deblock")
        # variants of base_bev_backbone.py:87-121 (OPV2V-style configs): deblocks that DOWN-sample (stride < 1 -> Conv2d(k, stride k)) and
        # one more ConvTranspose2d on the concatenated map; eval mode (their training is not built)
        self.variant = len(ups) == nlev + 1 or any(s < 1 for s in ups)
        _declare(self, backbone_param_spec(model_cfg, input_channels, ""))
        for p_ in self.parameters():
            p_.requires_grad_(True)          # trainable, as the reference's nn.Module is (train mode below)
        if "deblocks" not in self._modules:
            self.add_module("deblocks", _Node())
        for i in range(len(model_cfg["layer_nums"])):
            _retype(self.blocks[i], _Stage).bind(self, "_run_block", i)
        for i in range(min(len(ups), nlev)):
            _retype(self.deblocks[i], _Stage).bind(self, "_run_deblock", i)
        if len(ups) == nlev + 1:
            _retype(self.deblocks[nlev], _Stage).bind(self, "_run_deblock", nlev)
        self.num_bev_features = sum(model_cfg.get("num_upsample_filter", [])) if ups else model_cfg["num_filters"][-1]

    def _make_runner(self, device):
        cfg = dict(self.model_cfg)
        cfg.setdefault("upsample_strides", [])
        cfg.setdefault("num_upsample_filter", [])
        return _Runner(device, bb=cfg)

    def _layer(self, i):
        r = self.runner()
        return r.final_deblock if i == len(r.blocks) else r.deblocks[i]

    @staticmethod
    def _out_hw(L, h, w):
        if L.mode == _lib.AV2X_DECONV:
            return h * L.up, w * L.up
        return (h - L.ks) // L.stride + 1, (w - L.ks) // L.stride + 1

    def _pack(self, r, sd):
        r.load_backbone(sd, "", self.input_channels)

    # NHWC-level entry points (also used by Where2comm below)
    def block_nhwc(self, i, x, out=None):
        r = self.runner()
        n, h, w, _ = x.shape
        L0 = r.blocks[i][0]
        ho, wo = (h + 2 - 3) // L0.stride + 1, (w + 2 - 3) // L0.stride + 1
        if out is None:
            out = torch.empty((n, ho, wo, L0.cout), dtype=torch.float32, device=r.device)
        r.run_block(i, x, n, h, w, "sub", out=out)
        return out

    def deblock_nhwc(self, i, x, out=None, out_ctot=None, out_coff=0):
        r = self.runner()
        n, h, w, _ = x.shape
        L = self._layer(i)
        if out is None:
            ho, wo = self._out_hw(L, h, w)
            out = torch.empty((n, ho, wo, L.cout), dtype=torch.float32, device=r.device)
        r.conv(L, x, n, h, w, out, out_ctot=out_ctot, out_coff=out_coff)
        return out

    # ---- train mode: BatchNorm batch statistics + running-statistic updates, differentiable in the input and in
    #      every parameter; the nodes are train_ops' HIP forward / backward Functions
    def _train_state(self):
        P = dict(self.named_parameters())
        if next(iter(P.values())).device.type != "cuda":
            raise RuntimeError("BaseBEVBackbone (MI355X build) has no CPU path: move the module to the GPU")
        return P, self.state_dict(keep_vars=True)

    def _train_block(self, i, x):
        from .train_where2com import _block
        P, sd = self._train_state()
        return _block(P, sd, i, x, self.model_cfg["layer_nums"][i], self.model_cfg["layer_strides"][i], 1, prefix="")

    def _train_deblock(self, i, x):
        """deblocks[i] in train mode: ConvTranspose(k = s) + BN + ReLU; a DOWN-sampling one (stride 1 / k, base_bev_backbone.py:87-105) is
        Conv2d(k, stride k) + BN + ReLU = a 1x1 convolution of the space-to-depth map (k x k x cin channels per coarse pixel), so its data
        and weight gradients run on the same nodes; the final deblock (index = number of levels) is an ordinary one on the concatenation."""
        from . import train_ops as T
        from .train_where2com import _deblock, _running
        P, sd = self._train_state()
        ups = self.model_cfg.get("upsample_strides", [])
        if i < len(self.model_cfg["layer_nums"]) and ups[i] < 1:
            k = int(round(1.0 / ups[i]))
            n, h, w, c = x.shape
            if h % k or w % k:
                raise ValueError(f"down-sampling deblock {i}: the {h}x{w} map is not divisible by {k}")
            wt = P[f"deblocks.{i}.0.weight"]                                        # (cout, cin, k, k)
            w1 = wt.permute(0, 2, 3, 1).reshape(wt.shape[0], k * k * c, 1, 1)       # (dh, dw, c) = _space_to_depth's channel order
            bn = f"deblocks.{i}.1"
            return T.conv_bn_act(T._space_to_depth(x, k).contiguous(), w1, P[bn + ".weight"], P[bn + ".bias"], 1, 0, running=_running(sd, bn, 1))
        return _deblock(P, sd, i, x, 1, prefix="")

    def _run_block(self, i, x):
        if self.training:
            return _nchw(self._train_block(i, _nhwc_grad(x)))
        with torch.no_grad():
            return _nchw(self.block_nhwc(i, _nhwc(x)))

    def _run_deblock(self, i, x):
        if self.training:
            return _nchw(self._train_deblock(i, _nhwc_grad(x)))
        with torch.no_grad():
            return _nchw(self.deblock_nhwc(i, _nhwc(x)))

    def forward(self, data_dict):
        if self.training:
            x = _nhwc_grad(data_dict["spatial_features"])
            ups, nlev = [], len(self.model_cfg["layer_nums"])
            for i in range(nlev):
                x = self._train_block(i, x)
                if self.model_cfg.get("upsample_strides"):
                    ups.append(self._train_deblock(i, x))
            out = torch.cat(ups, -1) if len(ups) > 1 else (ups[0] if ups else x)
            if len(self.model_cfg.get("upsample_strides", [])) > nlev:          # base_bev_backbone.py:151-152
                out = self._train_deblock(nlev, out)
            data_dict["spatial_features_2d"] = _nchw(out)
            return data_dict
        with torch.no_grad():
            return self._forward_eval(data_dict)

    def _forward_eval(self, data_dict):
        r = self.runner()
        x = _nhwc(data_dict["spatial_features"])
        n = x.shape[0]
        feats = []
        for i in range(len(r.blocks)):
            x = self.block_nhwc(i, x)
            feats.append(x)
        if len(r.deblocks) > 0:
            H, W = self._out_hw(r.deblocks[0], feats[0].shape[1], feats[0].shape[2])
            cat = torch.empty((n, H, W, r.cat_c), dtype=torch.float32, device=r.device)
            coff = 0
            for i, f in enumerate(feats):                                   # torch.cat(ups, dim=1) written in place
                if self._out_hw(r.deblocks[i], f.shape[1], f.shape[2]) != (H, W):
                    raise ValueError("deblock outputs do not share one resolution")
                self.deblock_nhwc(i, f, out=cat, out_ctot=r.cat_c, out_coff=coff)
                coff += r.deblocks[i].cout
            out = cat
            if r.final_deblock is not None:                                 # base_bev_backbone.py:151-152
                out = self.deblock_nhwc(len(r.blocks), cat)
        else:
            if len(feats) != 1:
                raise NotImplementedError("BaseBEVBackbone without deblocks: a single level only")
            out = feats[0]
        data_dict["spatial_features_2d"] = _nchw(out)
        return data_dict


class ResNetBEVBackbone(BaseBEVBackbone):
    """models/common_modules/base_bev_backbone_resnet.py:16-128 (the backbone of airv2x_heal / airv2x_stamp / point_pillar_coalign):
    ``resnet`` = coalign_modules.resblock.ResNetModified(BasicBlock, layer_nums, layer_strides, num_filters, inplanes) returning the
    level maps (levels ``layer0``, ``layer1``, ...), then BaseBEVBackbone's deblocks.  ``resnet(x)`` is callable on NCHW-shaped maps like the reference's module
    (where2comm_attn.py:312-314 calls it).  Eval and train mode."""

    def __init__(self, model_cfg, input_channels=64):
        _HipModule.__init__(self)
        from ..synth import resnet_backbone_param_spec
        self.model_cfg = model_cfg
        self.input_channels = int(model_cfg.get("inplanes", input_channels))      # base_bev_backbone_resnet.py:50
        ups = model_cfg.get("upsample_strides", [])
        nlev = len(model_cfg["layer_nums"])
        if len(ups) not in (0, nlev, nlev + 1):
            raise ValueError("upsample_strides: one per level, optionally one more for the final deblock")
        self.variant = True               # BaseBEVBackbone's blocks[i] / packed deblock variants do not apply
        _declare(self, resnet_backbone_param_spec(model_cfg, "", self.input_channels))
        if "deblocks" not in self._modules:
            self.add_module("deblocks", _Node())
        for i in range(min(len(ups), nlev)):
            _retype(self.deblocks[i], _Stage).bind(self, "_run_deblock", i)
        if len(ups) == nlev + 1:
            _retype(self.deblocks[nlev], _Stage).bind(self, "_run_deblock", nlev)
        _retype(self.resnet, _Stage).bind(self, "_run_resnet", 0)
        self.num_levels = nlev
        # as the reference (base_bev_backbone_resnet.py:80-81): sum(num_upsample_filters), 0 without deblocks
        self.num_bev_features = sum(model_cfg.get("num_upsample_filter", [])) if ups else 0

    def _pack(self, r, sd):
        cfg = self.model_cfg
        r.load_resnet(sd, "resnet.", cfg["layer_nums"], cfg["layer_strides"], cfg["num_filters"], self.input_channels)
        r.blocks = [None] * len(cfg["layer_nums"])                        # level count for the shared deblock code
        only_deblocks = {k: v for k, v in sd.items() if k.startswith("deblocks.")}
        self._load_deblocks(r, only_deblocks)

    @staticmethod
    def _load_deblocks(r, sd):
        """BaseBEVBackbone's deblock loader without its blocks: a runner whose ``bb`` lists zero layers per level."""
        saved = r.bb
        r.bb = dict(saved, layer_nums=[-1] * len(saved["layer_nums"]))   # range(n + 1) is empty: no block weights are read
        try:
            blocks = r.blocks
            r.load_backbone(sd, "", 64)
            r.blocks = blocks
        finally:
            r.bb = saved

    def resnet_nhwc(self, x):
        """x (n,h,w,64) -> [level maps]"""
        r = self.runner()
        n, h, w, _ = x.shape
        feats, cur = [], x
        for li in range(len(r.res_layers)):
            c = r.res_layers[li][0]["c1"].cout
            s_ = r.res_layers[li][0]["c1"].stride
            out = torch.empty((n, (h + 2 - 3) // s_ + 1, (w + 2 - 3) // s_ + 1, c), dtype=torch.float32, device=r.device)
            cur, h, w = r.run_resnet_layer(li, cur, n, h, w, "sub", out=out)
            feats.append(cur)
        return feats

    # ---- train mode: resblock.py's BasicBlocks (conv3x3 - BN - ReLU - conv3x3 - BN, 1x1 downsample, add, ReLU; nn.BatchNorm2d defaults)
    #      on train_camera's nodes (the same block BevEncode trains with), BaseBEVBackbone's deblocks
    def _train_resnet(self, x):
        from . import train_camera as TC
        P, sd = self._train_state()
        feats, cur = [], x
        for li, (nb, stride) in enumerate(zip(self.model_cfg["layer_nums"], self.model_cfg["layer_strides"])):
            for bi in range(nb):
                cur = TC.basic_block(P, sd, f"resnet.layer{li}.{bi}.", cur, stride if bi == 0 else 1)
            feats.append(cur)
        return feats

    def _run_resnet(self, _i, x):
        if self.training:
            return tuple(_nchw(f) for f in self._train_resnet(_nhwc_grad(x)))
        with torch.no_grad():
            return tuple(_nchw(f) for f in self.resnet_nhwc(_nhwc(x)))

    def _run_deblock(self, i, x):
        if self.training:
            return _nchw(self._train_deblock(i, _nhwc_grad(x)))
        with torch.no_grad():
            return _nchw(self.deblock_nhwc(i, _nhwc(x)))

    def block_nhwc(self, i, x, out=None):
        raise NotImplementedError("ResNetBEVBackbone has no blocks[i]: call resnet(x)")

    def forward(self, data_dict):
        if self.training:
            feats = self._train_resnet(_nhwc_grad(data_dict["spatial_features"]))
            if self.model_cfg.get("upsample_strides"):
                feats = [self._train_deblock(i, f) for i, f in enumerate(feats)]
            elif len(feats) > 1 and not all(f.shape[1:3] == feats[0].shape[1:3] for f in feats):
                raise ValueError("ResNetBEVBackbone without deblocks: the level maps have different resolutions and cannot be concatenated")
            out = torch.cat(feats, -1) if len(feats) > 1 else feats[0]
            if len(self.model_cfg.get("upsample_strides", [])) > self.num_levels:      # base_bev_backbone_resnet.py:127-128
                out = self._train_deblock(self.num_levels, out)
            data_dict["spatial_features_2d"] = _nchw(out)
            return data_dict
        with torch.no_grad():
            r = self.runner()
            feats = self.resnet_nhwc(_nhwc(data_dict["spatial_features"]))
            n = feats[0].shape[0]
            if len(r.deblocks) > 0:
                H, W = self._out_hw(r.deblocks[0], feats[0].shape[1], feats[0].shape[2])
                cat = torch.empty((n, H, W, r.cat_c), dtype=torch.float32, device=r.device)
                coff = 0
                for i, f in enumerate(feats):
                    self.deblock_nhwc(i, f, out=cat, out_ctot=r.cat_c, out_coff=coff)
                    coff += r.deblocks[i].cout
                out = cat
                if r.final_deblock is not None:
                    out = self.deblock_nhwc(len(r.blocks), cat)
            else:
                # no deblocks: the reference concatenates the level maps (base_bev_backbone_resnet.py:112-120), which raises when their
                # resolutions differ -- as here, instead of silently dropping levels
                if len(feats) > 1 and not all(f.shape[1:3] == feats[0].shape[1:3] for f in feats):
                    raise ValueError("ResNetBEVBackbone without deblocks: the level maps have different resolutions "
                                     f"({[tuple(f.shape[1:3]) for f in feats]}) and cannot be concatenated")
                out = torch.cat(feats, -1) if len(feats) > 1 else feats[0]
            data_dict["spatial_features_2d"] = _nchw(out)
            return data_dict


# ----------------------------------------------------------------------------------------------- shrink / compressor
class DownsampleConv(_HipModule):
    """models/common_modules/downsample_conv.py:34-54 (DoubleConv layers: Conv k + ReLU, Conv3x3 + ReLU)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        _declare(self, shrink_param_spec(config, ""))
        for p_ in self.parameters():
            p_.requires_grad_(True)

    def _make_runner(self, device):
        return _Runner(device, sh=dict(self.config))

    def _pack(self, r, sd):
        r.load_shrink(sd, "")

    def nhwc(self, x, out=None):
        r = self.runner()
        n, h, w, c = x.shape
        r.cat_c = c
        y = r.run_shrink(x, n, h, w, "sub", out=out if out is not None else
                         torch.empty((n, h, w, r.shrink[-1].cout), dtype=torch.float32, device=r.device))
        return y

    def forward(self, x):
        if self.training:   # differentiable Conv + bias + ReLU pairs (train_ops.ConvBiasAct)
            from .train_where2com import _shrink
            P = dict(self.named_parameters())
            if next(iter(P.values())).device.type != "cuda":
                raise RuntimeError("DownsampleConv (MI355X build) has no CPU path: move the module to the GPU")
            return _nchw(_shrink(P, self.config, _nhwc_grad(x), prefix=""))
        with torch.no_grad():
            return _nchw(self.nhwc(_nhwc(x)))


class NaiveCompressor(_HipModule):
    """models/common_modules/naive_compress.py:5-42: encoder Conv3x3+BN+ReLU (C -> C/ratio), decoder 2 x Conv3x3+BN+ReLU."""

    def __init__(self, input_dim, compress_raito):
        super().__init__()
        if input_dim % compress_raito or (input_dim // compress_raito) % 32:
            raise NotImplementedError("NaiveCompressor: input_dim / ratio must be a multiple of 32 channels")
        self.input_dim, self.ratio = input_dim, compress_raito
        _declare(self, compressor_param_spec(input_dim, compress_raito, prefix=""))

    def _make_runner(self, device):
        return _CompRunner(device)

    def _pack(self, r, sd):
        r.compressor = r._load_compressor(sd, r._up, prefix="")

    @torch.no_grad()
    def forward(self, x):
        r = self.runner()
        xin = _nhwc(x)
        n, h, w, c = xin.shape
        enc, dec0, dec1 = r.compressor
        msg = r.buf("compress_msg", (n, h, w, enc.cout))
        mid = r.buf("compress_mid", (n, h, w, dec0.cout))
        out = torch.empty((n, h, w, dec1.cout), dtype=torch.float32, device=r.device)
        r.conv(enc, xin, n, h, w, msg)
        r.conv(dec0, msg, n, h, w, mid)
        r.conv(dec1, mid, n, h, w, out)
        return _nchw(out)


class _CompRunner(CoBEVTEngine):
    def __init__(self, device):
        Where2ComEngine.__init__(self, {"anchor_number": 0, "num_class": 0}, device)
        self.tile_cache = _TILE_CACHE

    def _init_config(self, args):
        pass


# ----------------------------------------------------------------------------------------------- regroup
def regroup(dense_feature, record_len, max_len):
    """models/common_modules/fuse_utils.py:13-64 (= cobevt_modules/fuse_utils.py): (sum N, C, H, W) ->
    ((B, L, C, H, W) zero padded on the agent axis, mask (B, L) int64).  The result is stored (B, L, H, W, C) in
    memory (the layout SwapFusionEncoder / V2XTransformer consume), returned as a (B, L, C, H, W)-shaped view."""
    lens = _lens(record_len)
    if sum(lens) != dense_feature.shape[0]:
        raise ValueError("record_len does not sum to the number of agents")
    if max(lens) > max_len:
        raise ValueError(f"{max(lens)} agents exceed max_len = {max_len}")
    N, C, H, W = dense_feature.shape
    B = len(lens)
    buf = torch.empty((B, max_len, H, W, C), dtype=dense_feature.dtype, device=dense_feature.device)
    src = dense_feature.permute(0, 2, 3, 1)
    off = 0
    for b, n in enumerate(lens):
        buf[b, :n].copy_(src[off:off + n])
        if n < max_len:
            buf[b, n:].zero_()
        off += n
    mask = torch.tensor([[1] * n + [0] * (max_len - n) for n in lens], dtype=torch.int64, device=dense_feature.device)
    return buf.permute(0, 1, 4, 2, 3), mask


def _valid_prefix(mask_row):
    """(L,) 0/1 validity -> number of valid agents; they must come first (what regroup produces)."""
    m = [int(v) for v in mask_row]
    n = sum(m)
    if m != [1] * n + [0] * (len(m) - n):
        raise NotImplementedError("agent mask must be 1...10...0 (valid agents first), as produced by regroup()")
    return n


# ----------------------------------------------------------------------------------------------- Where2comm
class Where2comm(_HipModule):
    """models/where2comm_modules/where2comm_fuse.py:166-288.  ``forward(x, psm_single, record_len, pairwise_t_matrix,
    backbone)`` -> (fused map, communication rate).  ``backbone`` must be this build's BaseBEVBackbone mirror."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.discrete_ratio = args["voxel_size"][0]
        self.downsample_rate = args["downsample_rate"]
        self.fully = args["fully"]
        self.multi_scale = args["multi_scale"]
        if self.multi_scale:
            self.num_levels = len(args["layer_nums"])
        comm = args["communication"]
        self.threshold = comm["threshold"]
        if "gaussian_smooth" in comm:
            k, s = comm["gaussian_smooth"]["k_size"], float(comm["gaussian_smooth"]["c_sigma"])
            _declare(self, [("naive_communication.gaussian_filter.weight", (1, 1, k, k), "zeros"),
                            ("naive_communication.gaussian_filter.bias", (1,), "zeros")])
            c = k // 2
            gx, gy = np.mgrid[0 - c:k - c, 0 - c:k - c]
            # constructor default of init_gaussian_filter (:66-81); checkpoints overwrite it
            g = 1.0 / (2.0 * np.pi * s) * np.exp(-(np.square(gx) + np.square(gy)) / (2.0 * np.square(s)))
            with torch.no_grad():
                self.naive_communication.gaussian_filter.weight.copy_(torch.from_numpy(g.astype(np.float32)).view(1, 1, k, k))

    def _make_runner(self, device):
        return _Runner(device, fcfg={"communication": self.args["communication"], "fully": self.fully, "multi_scale": True})

    def _pack(self, r, sd):
        r._load_fusion({"fusion_net." + k: v for k, v in sd.items()}, r._up)

    def runner(self, train_ok=False):
        if not self._tensors():   # no gaussian filter: nothing to version
            if self._runner_obj is None:
                dev = torch.device("cuda", torch.cuda.current_device())
                self.__dict__["_runner_obj"] = self._make_runner(dev)
                self._pack(self._runner_obj, {})
            return self._runner_obj
        return super().runner(train_ok)

    def _mask(self, r, psm_single, lens, H, W, topk=None):
        psm = _nhwc(psm_single)
        n, h, w, c = psm.shape
        if (h, w) != (H, W):
            raise NotImplementedError("mask / feature size mismatch (the bilinear resize branch, where2comm_fuse.py:230-236)")
        r.A, r.C = c, 1
        mask, count, _, rl = r.comm_mask(psm, n, H, W, lens, topk=topk)
        rate = r.comm_rate(count, rl, len(lens), H * W).clone()
        return mask, rate

    def _forward_train(self, x, psm_single, lens, backbone, topk=None, mask=None):
        """Train mode of the multi-scale fusion (where2comm_fuse.py:214-263 with ``self.training``): random top-K masks (K drawn
        from python's ``random`` as the reference draws it, one per sample), the backbone's blocks / deblocks with batch
        statistics, everything differentiable in x and in the backbone's parameters (train_ops' HIP forward / backward nodes).
        ``topk`` / ``mask``: pin K / replay a recorded mask (tests)."""
        import random

        from . import train_ops as T
        if not self.multi_scale or not isinstance(backbone, BaseBEVBackbone):
            raise NotImplementedError("Where2comm training: the multi-scale fusion over this build's BaseBEVBackbone")
        r = self.runner(train_ok=True)
        cur = _nhwc_grad(x)
        rate = torch.tensor(1, device=r.device)
        ups = []
        for i in range(self.num_levels):
            cur = backbone._train_block(i, cur)
            n, h, w, c = cur.shape
            if i == 0 and not self.fully:
                with torch.no_grad():
                    if topk is None:
                        topk = [int(h * w * random.uniform(0, 1)) for _ in lens]          # where2comm_fuse.py:106
                    m, rate = self._mask(r, psm_single, lens, h, w, topk=topk)
                    m = m.clone() if mask is None else mask.to(r.device, torch.float32).reshape(n, h, w).contiguous()
                cur = T.MaskMul.apply(cur, m)
            outs, a0 = [], 0
            for k in lens:
                outs.append(T.PixelAttn.apply(cur[a0:a0 + k]))
                a0 += k
            fused = torch.stack(outs)
            ups.append(backbone._train_deblock(i, fused) if backbone.model_cfg.get("upsample_strides") else fused)
        return _nchw(torch.cat(ups, -1) if len(ups) > 1 else ups[0]), rate

    def _fuse(self, r, x, lens, out):
        n, h, w, c = x.shape
        a0 = 0
        for b, k in enumerate(lens):
            r.attn([x[j].data_ptr() for j in range(a0, a0 + k)], h * w, c, out[b])
            a0 += k

    def forward(self, x, psm_single, record_len, pairwise_t_matrix, backbone=None):
        lens = _lens(record_len)
        B = pairwise_t_matrix.shape[0]
        if B != len(lens):
            raise ValueError("pairwise_t_matrix batch size does not match record_len")
        if any(k < 1 for k in lens):
            raise ValueError("every sample needs at least the ego agent")
        if self.training:
            if sum(lens) != x.shape[0]:
                raise ValueError("record_len does not sum to the number of agents")
            return self._forward_train(x, psm_single, lens, backbone)
        with torch.no_grad():
            return self._forward_eval(x, psm_single, lens, B, backbone)

    def _forward_eval(self, x, psm_single, lens, B, backbone):
        r = self.runner()
        cur = _nhwc(x)
        if sum(lens) != cur.shape[0]:
            raise ValueError("record_len does not sum to the number of agents")
        rate = torch.tensor(1, device=r.device)
        st = r.stream()
        if not self.multi_scale:
            if not self.fully:
                cur = cur.clone() if cur.data_ptr() == x.data_ptr() else cur    # never modify the caller's tensor
                mask, rate = self._mask(r, psm_single, lens, cur.shape[1], cur.shape[2])
                _lib.check(r.lib.av2x_apply_mask(_ptr(cur), _ptr(mask), cur.shape[0], cur.shape[1] * cur.shape[2],
                                                 cur.shape[3], st), "av2x_apply_mask")
            out = torch.empty((B,) + tuple(cur.shape[1:]), dtype=torch.float32, device=r.device)
            self._fuse(r, cur, lens, out)
            return _nchw(out), rate
        if not isinstance(backbone, BaseBEVBackbone):
            raise TypeError("Where2comm (MI355X build): `backbone` must be the BaseBEVBackbone of this build")
        br = backbone.runner()
        ups, cat, coff = [], None, 0
        for i in range(self.num_levels):
            cur = backbone.block_nhwc(i, cur)
            n, h, w, c = cur.shape
            if i == 0 and not self.fully:
                mask, rate = self._mask(r, psm_single, lens, h, w)
                _lib.check(r.lib.av2x_apply_mask(_ptr(cur), _ptr(mask), n, h * w, c, st), "av2x_apply_mask")
            fused = r.buf(f"w2c_fused{i}", (B, h, w, c))
            self._fuse(r, cur, lens, fused)
            if len(br.deblocks) > 0:
                L = br.deblocks[i]
                if cat is None:
                    cat = torch.empty((B, h * L.up, w * L.up, br.cat_c), dtype=torch.float32, device=r.device)
                backbone.deblock_nhwc(i, fused, out=cat, out_ctot=br.cat_c, out_coff=coff)
                coff += L.cout
            else:
                ups.append(fused.clone())
        if cat is not None:
            return _nchw(cat), rate
        if len(ups) == 1:
            return _nchw(ups[0]), rate
        raise NotImplementedError("multi-scale Where2comm without deblocks needs equal-resolution levels")


# ----------------------------------------------------------------------------------------------- SwapFusionEncoder
class _FaxRunner(CoBEVTEngine):
    def __init__(self, device, fax):
        self._fax_cfg = fax
        Where2ComEngine.__init__(self, {"anchor_number": 0, "num_class": 0}, device)
        self.tile_cache = _TILE_CACHE

    def _init_config(self, args):
        fax = self._fax_cfg
        self.fax, self.compression = fax, 0
        self.fcfg = {"fully": False}
        self.L = int(fax["agent_size"])
        self.heads_n = fax["input_dim"] // fax["dim_head"]


class SwapFusionEncoder(_HipModule):
    """models/cobevt_modules/swap_fusion_modules.py:233-280 (mask variant).  ``forward(x (B,L,C,H,W), mask)``;
    ``mask`` is the reference's (B,H,W,1,L) key mask (airv2x_cobevt.py:136-141) or a (B,L) validity array; it must
    be constant over the map and of the form 1..10..0."""

    def __init__(self, args):
        super().__init__()
        if not args.get("mask", False):
            raise NotImplementedError("SwapFusionEncoder: only the masked variant (the shipped AirV2X config) is built")
        self.args = args
        self.depth = args["depth"]
        self.mask = True
        _declare(self, fax_param_spec(args, ""))

    def _make_runner(self, device):
        return _FaxRunner(device, self.args)

    def _pack(self, r, sd):
        r._load_fusion(sd, r._up, prefix="")

    @torch.no_grad()
    def forward(self, x, mask=None):
        r = self.runner()
        B, L, C, H, W = x.shape
        if L != r.L or C != self.args["input_dim"]:
            raise ValueError(f"expected (B,{r.L},{self.args['input_dim']},H,W), got {tuple(x.shape)}")
        if mask is None:
            valid = [[1] * L] * B
        else:
            m = mask.detach()
            if m.dim() == 5:                                   # (B,H,W,1,L)
                if not bool((m == m[:, :1, :1]).all()):
                    raise NotImplementedError("spatially varying key masks are not built")
                m = m[:, 0, 0, 0, :]
            valid = m.cpu().tolist()
        out = torch.empty((B, H, W, C), dtype=torch.float32, device=r.device)
        tok = r.buf("fax_x", (L, H, W, C))
        for b in range(B):
            n = _valid_prefix(valid[b])
            tok.copy_(x[b].permute(0, 2, 3, 1))                # the encoder works in place on its own token buffer
            out[b:b + 1].copy_(r.fax_encoder(tok, n, H, W))
        return _nchw(out)


# ----------------------------------------------------------------------------------------------- V2XTransformer
class _VitRunner(V2XViTEngine):
    def __init__(self, device, enc):
        self._enc_cfg = enc
        Where2ComEngine.__init__(self, {"anchor_number": 0, "num_class": 0}, device)
        self.tile_cache = _TILE_CACHE

    def _init_config(self, args):
        self.enc = self._enc_cfg
        self.cav, self.pw = self.enc["cav_att_config"], self.enc["pwindow_att_config"]
        if self.cav["dim"] != 256 or self.cav["heads"] * self.cav["dim_head"] != 256:
            raise NotImplementedError("V2XTransformer: the kernels are built for dim = heads * dim_head = 256")
        if not self.cav["use_hetero"] or self.pw["fusion_method"] != "split_attn" or not self.pw["relative_pos_embedding"]:
            raise NotImplementedError("only the shipped V2X-ViT configuration (hetero attention, split_attn, relative pos)")
        self.fcfg = {"fully": False}
        self.L = None
        self.ego_only_last = True


class V2XTransformer(_HipModule):
    """models/v2xvit_modules/v2xvit_basic.py:202-213.  ``forward(x (B,L,H,W,C+3), mask (B,L), spatial_correction_matrix
    (B,L,4,4))`` -> (B,H,W,C): the ego's fused map.  The three trailing channels are the broadcast prior encoding
    (velocity, time delay, infra flag), read at pixel (0,0) like the reference (hmsa.py:123-128)."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        _declare(self, v2xvit_encoder_spec(args["encoder"], "encoder"))

    def _make_runner(self, device):
        return _VitRunner(device, self.args["encoder"])

    def _pack(self, r, sd):
        r._load_fusion(sd, r._up, p="encoder")

    @torch.no_grad()
    def forward(self, x, mask, spatial_correction_matrix):
        r = self.runner()
        B, L, H, W, C3 = x.shape
        C = C3 - 3
        if C != r.cav["dim"]:
            raise ValueError(f"expected {r.cav['dim']}+3 channels, got {C3}")
        prior = x[:, :, 0, 0, C:].detach().float().cpu().numpy()                  # (B,L,3)
        scm = spatial_correction_matrix.detach().cpu().numpy()
        valid = mask.detach().cpu().tolist()
        out = torch.empty((B, H, W, C), dtype=torch.float32, device=r.device)
        for b in range(B):
            n = _valid_prefix(valid[b])
            if n < 1:
                raise ValueError("every sample needs at least the ego agent")
            tok = r.buf("vit_x", (n, H, W, C))
            tok.copy_(x[b, :n, :, :, :C])
            out[b:b + 1].copy_(r.encoder(tok, n, H, W, prior[b], scm[b]))
        return out
