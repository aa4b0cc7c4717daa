"""``Airv2xWhere2com`` — drop-in for opencood/models/airv2x_where2com.py:19-227 (det task; LiDAR, camera
or both modalities per agent type) whose forward runs entirely in libairv2x_hip.so.

Same constructor argument (``hypes["model"]["args"]``), same ``forward(data_dict)`` input
contract, same output keys (``psm``, ``rm``, ``obj``, ``mask``, ``com``,
``comm_rate``), and the SAME ``state_dict`` keys/shapes as the reference (162 tensors) so
released raw-state_dict checkpoints load with ``load_state_dict``.

Parameters are registered under the reference's names.  ``.eval()`` forwards run the packed inference engine (the packed
device copies are rebuilt lazily whenever a parameter changes: ``load_state_dict``, ``.to()``, optimiser steps and other
in-place edits bump the version counters); ``.train()`` forwards build the autograd graph of ``train_where2com.py``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..synth import where2com_param_spec
from .engine import Where2ComEngine


class _Node(nn.Module):
    """Pure container used to reproduce the reference's module tree (and thus its keys)."""

    def forward(self, *a, **k):  # pragma: no cover - containers are never called
        raise RuntimeError("parameter container: the compute lives in libairv2x_hip.so")

    def __getitem__(self, i):  # ModuleList / Sequential style access (backbone.blocks[1][4])
        return self._modules[str(i)]

    def __len__(self):
        return len(self._modules)


def _amp_requested(module):
    """AMP mode = the caller wrapped the forward in ``torch.autocast("cuda", ...)`` (as the reference's train.py
    does for its validation pass, tools/train.py:212-220) or set ``module.amp = True``: Conv2d / Linear products
    then run on the bf16 matrix cores with fp32 accumulation (conv_igemm_bf16), everything else stays fp32."""
    if getattr(module, "amp", None) is not None:
        return bool(module.amp)
    try:
        return bool(torch.is_autocast_enabled("cuda"))
    except TypeError:  # older signature
        return bool(torch.is_autocast_enabled())


def _install(root, key, tensor, is_buffer, requires_grad=False):
    parts = key.split(".")
    node = root
    for p in parts[:-1]:
        nxt = node._modules.get(p)
        if nxt is None:
            nxt = _Node()
            node.add_module(p, nxt)
        node = nxt
    if is_buffer:
        node.register_buffer(parts[-1], tensor)
    else:
        node.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=requires_grad))


class Airv2xWhere2com(nn.Module):
    def __init__(self, args):
        super().__init__()
        if args.get("task", "det") != "det":
            raise NotImplementedError("only the det task is on the MI355X hot path (seg branch out of scope)")
        for t in args["collaborators"]:
            if not args[t]["modalities"] or any(m not in ("lidar", "cam") for m in args[t]["modalities"]):
                raise NotImplementedError(f"Modality {args[t]['modalities']} not supported for {t}.")   # airv2x_base_model.py:57,78,99
        self.args = args
        self.collaborators = args["collaborators"]
        self.active_sensors = args["active_sensors"]
        self.multi_scale = args["where2com_fusion"]["multi_scale"]
        self.outC = args["outC"]
        for key, shape, kind in where2com_param_spec(args):
            buf = kind in ("bn_m", "bn_v", "count")
            if kind == "count":
                t = torch.zeros(shape, dtype=torch.long)
            elif kind == "bn_v" or kind == "bn_w":
                t = torch.ones(shape)
            else:
                t = torch.zeros(shape)
            _install(self, key, t, buf, requires_grad=True)   # trainable, as the reference's nn.Modules are
        if args.get("backbone_fix"):
            self.backbone_fix()                                    # airv2x_where2com.py:77-78
        self._engine = None
        self._packed_version = None
        self.sync_comm_rate = True  # the reference returns a python int (airv2x_where2com.py:122)

    def backbone_fix(self):
        """airv2x_where2com.py:80-115: freeze the encoders, the backbone, the shrink header and the heads (fine-tuning on time
        delay); what is left trainable is the fusion net."""
        for name, p in self.named_parameters():
            if not name.startswith("fusion_net."):
                p.requires_grad = False

    def _version(self):
        return tuple(t._version for t in self.state_dict(keep_vars=True).values()) + (
            next(iter(self.parameters())).device,)

    def engine(self):
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("Airv2xWhere2com (MI355X build) has no CPU path: move the module to the GPU (model.to('cuda'))")
        ver = self._version()
        if self._engine is None or self._engine.device != dev:
            self._engine = Where2ComEngine(self.args, dev)
            self._packed_version = None
        if self._packed_version != ver:
            self._engine.load_state_dict(self.state_dict())
            self._packed_version = ver
        return self._engine

    def forward(self, data_dict):
        if self.training:   # the graph torch autograd differentiates, on the HIP forward / backward ops (train_where2com.py)
            # camera branches train too (train_camera.py): EfficientNet-B0 CamEncode, the ground-truth-depth lift, BevEncode
            from .train_where2com import forward_train
            return forward_train(self, data_dict)
        eng = self.engine()
        eng.amp = _amp_requested(self)
        return eng.forward(data_dict, sync_comm_rate=self.sync_comm_rate)
