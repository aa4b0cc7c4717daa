"""``VoxelPostprocessor`` — inference half of data_utils/post_processor/voxel_postprocessor.py
(anchors :33-86, ``post_process_airv2x`` :666-839) with the box decoding, filters and the rotated
NMS running on the device (av2x_postprocess) instead of torch ops + a shapely loop on the host.

Same constructor (``VoxelPostprocessor(hypes["postprocess"], dataset, train)``), same
``generate_anchor_box()`` result (numpy float64, host-side constants as in the reference) and the
same ``post_process_airv2x(data_dict, output_dict)`` return tuple
``(pred_box3d (K,8,3), scores (K,), labels (K,) int64, boxes3d (K,7))`` in NMS pick order.

Two ways to use it:
* ``VoxelPostprocessor`` -- stand-alone (this repo's harness, the GPU box): anchors + post_process_airv2x only.
* ``bind_device_postprocess(RefVoxelPostprocessor)`` -- the drop-in binding inside the reference tree: a SUBCLASS of the
  reference's own class that overrides only ``post_process_airv2x`` (+ ``launch`` / ``finish``), so everything else the
  dataset calls on the same object -- ``generate_label_airv2x`` (intermediate_fusion_dataset.py:360),
  ``generate_object_center_airv2x`` (:482), ``collate_batch_airv2x`` (:806), ``generate_gt_bbx_airv2x`` (:935-936),
  ``post_process_segmentation_airv2x`` (:960) -- stays the reference's.
"""
from __future__ import annotations

import ctypes
import math
from ctypes import c_float, c_void_p

import numpy as np
import torch

from .. import _lib


class DevicePostprocess:
    """``post_process_airv2x`` on the device (mixin).  Needs the attributes the reference's constructor sets
    (voxel_postprocessor.py:26-31): ``params``, ``num_class``, ``lidar_range``."""

    nms_top = 1000  # box_utils.py:849

    @property
    def _ws(self):
        ws = self.__dict__.get("_av2x_ws")
        if ws is None:
            ws = self.__dict__["_av2x_ws"] = {}
        return ws

    def _device_anchors(self, anchors, dev):
        """fp32 device copy of the anchor tensor (``anchors.float()`` as delta_to_boxes3d :612), cached:
        the dataset hands over the same constants every frame (a fresh 3.9 MB float64 host tensor per
        batch in the reference), so the conversion + upload is done once.  A new tensor object is
        recognised as the cached anchors by shape and 64 sampled rows."""
        anchors = anchors if isinstance(anchors, torch.Tensor) else torch.from_numpy(np.asarray(anchors))
        flat = anchors.reshape(-1, 7)
        c = self._ws.get("anchors")
        if c is not None and c["dev"].device == dev and c["shape"] == tuple(flat.shape):
            if c["ptr"] == flat.data_ptr() or torch.equal(flat[c["rows"]].double().cpu(), c["sample"]):
                return c["dev"]
        rows = torch.linspace(0, flat.shape[0] - 1, 64).long()
        d = flat.float().to(dev).contiguous()
        self._ws["anchors"] = {"dev": d, "shape": tuple(flat.shape), "ptr": flat.data_ptr(), "rows": rows,
                               "sample": flat[rows].double().cpu()}
        return d

    def _buffers(self, dev, H, W, A, slot=0):
        key = (str(dev), H, W, A, slot)
        b = self._ws.get(key)
        if b is None:
            lib = _lib.load()
            top = self.nms_top
            b = {
                "ws": torch.empty(int(lib.av2x_postprocess_workspace_bytes(H, W, A, top)), dtype=torch.uint8, device=dev),
                "corners": torch.empty((top, 8, 3), dtype=torch.float32, device=dev),
                "scores": torch.empty((top,), dtype=torch.float32, device=dev),
                "labels": torch.empty((top,), dtype=torch.int32, device=dev),
                "boxes": torch.empty((top, 7), dtype=torch.float32, device=dev),
                "index": torch.empty((top,), dtype=torch.int32, device=dev),
                "counts": torch.zeros((8,), dtype=torch.int32, device=dev),
            }
            self._ws[key] = b
        return b

    @torch.no_grad()
    def post_process_airv2x(self, data_dict, output_dict, return_counts=False):
        return self.finish(self.launch(data_dict, output_dict), return_counts)

    @staticmethod
    def finish(handle, return_counts=False):
        """Second half of post_process_airv2x: the one host read-back (5 counters) and the exact-shape slices."""
        b = handle
        counts = b["counts"][:5].tolist()
        if counts[0] == 0:
            res = (None, None, None, None)
        else:
            k = counts[4]
            res = (b["corners"][:k].clone(), b["scores"][:k].clone(), b["labels"][:k].to(torch.int64), b["boxes"][:k].clone())
        return res + (counts, b["index"][:counts[4]].clone()) if return_counts else res

    @torch.no_grad()
    def launch(self, data_dict, output_dict, slot=0):
        """First half: enqueue av2x_postprocess on the current stream into buffer set ``slot`` and return the handle
        for ``finish`` -- no host synchronisation (frames kept in flight by FramePipeline finish later)."""
        if len(data_dict) != 1:
            raise NotImplementedError("intermediate fusion hands over exactly one entry ('ego'); late fusion is out of scope")
        (cav_id, cav), = data_dict.items()
        out = output_dict[cav_id]
        psm, rm, obj = out["psm"], out["rm"], out["obj"]
        if psm.device.type != "cuda":
            raise RuntimeError("VoxelPostprocessor (MI355X build) has no CPU path")
        if psm.shape[0] != 1:
            raise ValueError(f"inference only has 1 batch, but got {tuple(psm.shape)}")
        dev = psm.device
        _, AC, H, W = psm.shape
        C = self.num_class
        A = AC // C
        anchors = self._device_anchors(cav["anchor_box"], dev)
        if anchors.shape[0] != H * W * A:
            raise ValueError("anchor_box does not match the head resolution")
        T = cav["transformation_matrix"]
        t_dev = t16 = None
        if isinstance(T, torch.Tensor) and T.device.type == "cuda":
            # the batch is on the GPU (inference_utils.py / train_utils.to_device): the kernel reads the matrix from
            # device memory, no host read-back that would drain the frame's stream
            t_dev = T.detach().to(dev, torch.float32).contiguous().view(-1)
            if t_dev.numel() != 16:
                raise ValueError("transformation_matrix must be 4x4")
        else:
            T = (T.detach().numpy() if isinstance(T, torch.Tensor) else np.asarray(T)).astype(np.float32).reshape(16)
            t16 = (c_float * 16)(*[float(v) for v in T])
        r6 = (c_float * 6)(*[float(v) for v in self.lidar_range])
        b = self._buffers(dev, H, W, A, slot)
        lib = _lib.load()
        psm, rm, obj = psm.contiguous().float(), rm.contiguous().float(), obj.contiguous().float()
        st = c_void_p(torch.cuda.current_stream().cuda_stream)
        P = lambda t: c_void_p(t.data_ptr())
        fn = lib.av2x_postprocess_devt if t_dev is not None else lib.av2x_postprocess
        _lib.check(fn(P(psm), P(rm), P(obj), P(anchors), H, W, A, C, P(t_dev) if t_dev is not None else ctypes.cast(t16, c_void_p),
                      ctypes.cast(r6, c_void_p), float(self.params["target_args"]["obj_threshold"]),
                      float(self.params["nms_thresh"]), 1 if self.params["order"] == "hwl" else 0,
                      self.nms_top, P(b["ws"]), P(b["corners"]), P(b["scores"]), P(b["labels"]),
                      P(b["boxes"]), P(b["index"]), P(b["counts"]), st), "av2x_postprocess")
        if t_dev is not None:
            b["t_dev"] = t_dev   # keep the (possibly converted) matrix alive until the kernel has run
        return b


class VoxelPostprocessor(DevicePostprocess):
    """Stand-alone form: the reference's constructor fields + anchors + the device post-process."""

    def __init__(self, anchor_params, dataset="airv2x", train=False):
        self.params = anchor_params
        self.dataset = dataset
        self.train = train
        self.anchor_num = self.params["anchor_args"].get("num", 2)
        self.num_class = self.params["anchor_args"].get("num_class", 7)  # voxel_postprocessor.py:29, appendix A #19
        self.lidar_range = self.params["anchor_args"]["cav_lidar_range"]

    def generate_anchor_box(self):
        a = self.params["anchor_args"]
        W, H = a["W"], a["H"]
        r = [math.radians(e) for e in a["r"]]
        assert self.anchor_num == len(r)
        fs = a.get("feature_stride", 2)
        rng = self.lidar_range
        x = np.linspace(rng[0] + a["vw"], rng[3] - a["vw"], W // fs)
        y = np.linspace(rng[1] + a["vh"], rng[4] - a["vh"], H // fs)
        cx, cy = np.meshgrid(x, y)
        cx = np.tile(cx[..., np.newaxis], self.anchor_num)
        cy = np.tile(cy[..., np.newaxis], self.anchor_num)
        cz = np.ones_like(cx) * -1.0
        w, l, h = np.ones_like(cx) * a["w"], np.ones_like(cx) * a["l"], np.ones_like(cx) * a["h"]
        r_ = np.ones_like(cx)
        for i in range(self.anchor_num):
            r_[..., i] = r[i]
        if self.params["order"] == "hwl":
            return np.stack([cx, cy, cz, h, w, l, r_], axis=-1)
        if self.params["order"] == "lhw":
            return np.stack([cx, cy, cz, l, h, w, r_], axis=-1)
        raise ValueError("Unknown bbx order.")


def bind_device_postprocess(reference_cls):
    """``VoxelPostprocessor = bind_device_postprocess(VoxelPostprocessor)`` at the end of the reference's
    data_utils/post_processor/voxel_postprocessor.py: a subclass of the reference's class whose
    ``post_process_airv2x`` runs on the device; every other method (label generation, collate, GT boxes,
    ``generate_anchor_box``, the seg branch) is inherited from the reference unchanged."""
    return type(reference_cls.__name__, (DevicePostprocess, reference_cls), {"__doc__": reference_cls.__doc__,
                                                                            "__module__": reference_cls.__module__})
