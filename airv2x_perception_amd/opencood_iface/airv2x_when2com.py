"""``Airv2xWhen2com`` — drop-in for opencood/models/airv2x_when2com.py:24-151 (det task, LiDAR) running in
libairv2x_hip.so.  Same constructor argument, input contract (incl. ``img_pairwise_t_matrix_collab``), output keys
(``psm``, ``rm``, ``obj``, ``mask``, ``comm_rate``) and state_dict keys/shapes as the reference."""
from __future__ import annotations

import torch.nn as nn

from ..synth import when2com_param_spec
from .airv2x_where2com import _amp_requested
from .submodules import _declare
from .when2com_engine import When2comEngine


class Airv2xWhen2com(nn.Module):
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
        self.outC = args["outC"]
        _declare(self, when2com_param_spec(args))
        for p in self.parameters():        # trainable, as the reference's nn.Modules are (train_when2com.py is the train-mode forward)
            p.requires_grad_(True)
        if args.get("backbone_fix"):
            self.backbone_fix()
        self._engine = None
        self._packed_version = None
        self.sync_comm_rate = True   # the reference returns a python float (when2com.py:132)

    def _version(self):
        return tuple(t._version for t in self.state_dict(keep_vars=True).values()) + (next(iter(self.parameters())).device,)

    def engine(self):
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("Airv2xWhen2com (MI355X build) has no CPU path: move the module to the GPU (model.to('cuda'))")
        ver = self._version()
        if self._engine is None or self._engine.device != dev:
            self._engine = When2comEngine(self.args, dev)
            self._packed_version = None
        if self._packed_version != ver:
            self._engine.load_state_dict(self.state_dict())
            self._packed_version = ver
        return self._engine

    def backbone_fix(self):
        """airv2x_when2com.py:77-110 (fine-tuning on time delay): freeze the encoders, backbone, shrink header and heads; the fusion net
        stays trainable."""
        for name, p in self.named_parameters():
            if not name.startswith("fusion_net."):
                p.requires_grad = False

    def forward(self, data_dict):
        if self.training:      # HIP forward / backward ops under torch autograd (train_when2com.py)
            from .train_when2com import forward_train
            return forward_train(self, data_dict)
        eng = self.engine()
        eng.amp = _amp_requested(self)
        return eng.forward(data_dict, sync_comm_rate=self.sync_comm_rate)
