"""``Airv2xCoBEVT`` — drop-in for opencood/models/airv2x_cobevt.py:15-156 (det task, LiDAR) running
in libairv2x_hip.so (``.eval()``: the packed inference engine; ``.train()``: the autograd graph of train_cobevt.py).  Same constructor argument, input contract, output keys (``psm``, ``rm``,
``obj``) and state_dict keys/shapes (236 tensors at max_cav 3/2/2) as the reference."""
from __future__ import annotations

import torch
import torch.nn as nn

from ..synth import cobevt_param_spec
from .airv2x_where2com import _amp_requested, _install
from .cobevt_engine import CoBEVTEngine


class Airv2xCoBEVT(nn.Module):
    def __init__(self, args):
        super().__init__()
        if args.get("task", "det") != "det":
            raise NotImplementedError("only the det task is on the MI355X hot path")
        for t in args["collaborators"]:
            if not args[t]["modalities"] or any(m not in ("lidar", "cam") for m in args[t]["modalities"]):
                raise NotImplementedError(f"Modality {args[t]['modalities']} not supported for {t}.")   # airv2x_base_model.py:57,78,99
        self.args = args
        self.collaborators = args["collaborators"]
        self.active_sensors = args["active_sensors"]
        self.max_cav_num = sum(args["max_cav"].values())
        args["fax_fusion"]["agent_size"] = self.max_cav_num  # airv2x_cobevt.py:50
        self.outC = args["outC"]
        for key, shape, kind in cobevt_param_spec(args):
            if kind == "count":
                t, buf = torch.zeros(shape, dtype=torch.long), True
            elif kind.startswith("relidx:"):
                from ..synth import synthetic_tensor
                t, buf = torch.from_numpy(synthetic_tensor(key, shape, kind)), True
            elif kind in ("bn_m", "bn_v"):
                t, buf = (torch.ones(shape) if kind == "bn_v" else torch.zeros(shape)), True
            elif kind in ("bn_w", "ln_w"):
                t, buf = torch.ones(shape), False
            else:
                t, buf = torch.zeros(shape), False
            _install(self, key, t, buf, requires_grad=True)    # trainable, as the reference's nn.Modules are
        if args.get("backbone_fix"):
            self.backbone_fix()
        self._engine = None
        self._packed_version = None

    def _version(self):
        return tuple(t._version for t in self.state_dict(keep_vars=True).values()) + (next(iter(self.parameters())).device,)

    def engine(self):
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("Airv2xCoBEVT (MI355X build) has no CPU path: move the module to the GPU (model.to('cuda'))")
        ver = self._version()
        if self._engine is None or self._engine.device != dev:
            self._engine = CoBEVTEngine(self.args, dev)
            self._packed_version = None
        if self._packed_version != ver:
            self._engine.load_state_dict(self.state_dict())
            self._packed_version = ver
        return self._engine

    def backbone_fix(self):
        """airv2x_cobevt.py:77-110 (fine-tuning on time delay): freeze the encoders, backbone, shrink header, compressor and heads; the
        fusion net stays trainable.  (As written the reference's method names ``self.veh_model`` and fails; this is what it intends.)"""
        for name, p in self.named_parameters():
            if not name.startswith("fusion_net."):
                p.requires_grad = False

    def forward(self, data_dict):
        if self.training:   # the graph torch autograd differentiates, on HIP forward / backward ops (train_cobevt.py)
            from .train_cobevt import forward_train
            return forward_train(self, data_dict)
        eng = self.engine()
        eng.amp = _amp_requested(self)
        return eng.forward(data_dict)
