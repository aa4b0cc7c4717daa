"""``roiaware_pool3d_cuda`` -- drop-in for the reference's compiled extension module of that name
(opencood/pcdet_utils/roiaware_pool3d/src/roiaware_pool3d.cpp:179-184: ``forward``, ``backward``, ``points_in_boxes_gpu``,
``points_in_boxes_cpu``; imported at pcdet_utils/roiaware_pool3d/roiaware_pool3d_utils.py:5 and through it by
data_utils/datasets/opv2v/intermediate_fusion_dataset.py:24-26 before any dataset can be built).  Same positional arguments (torch
tensors, results written into the caller's tensors, return 1); the work is done by libairv2x_hip.so (csrc/roiaware.hip) through the C-ABI.

``install_import_shims()`` (this package's __init__) registers this module as ``opencood.pcdet_utils.roiaware_pool3d.roiaware_pool3d_cuda``
and box_overlaps.py as ``opencood.utils.box_overlaps`` so that the reference's unmodified imports resolve on a ROCm box."""
from __future__ import annotations

from ctypes import c_void_p

import torch

from .. import _lib


def _p(t):
    return c_void_p(t.data_ptr())


def _f32(t, name, device_type):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise TypeError(f"{name}: a contiguous float32 tensor is expected (got {t.dtype}, contiguous={t.is_contiguous()})")
    if t.device.type != device_type:
        raise TypeError(f"{name}: expected a {device_type} tensor, got {t.device}")
    return t


def _i32(t, name, device_type):
    if t.dtype != torch.int32 or not t.is_contiguous():
        raise TypeError(f"{name}: a contiguous int32 tensor is expected (got {t.dtype})")
    if t.device.type != device_type:
        raise TypeError(f"{name}: expected a {device_type} tensor, got {t.device}")
    return t


def _stream(t):
    return c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def points_in_boxes_cpu(boxes_tensor, pts_tensor, pts_indices_tensor):
    """roiaware_pool3d.cpp:143-177: boxes (N,7), pts (P,3) on the HOST; pts_indices (N,P) int32 <- 1 where the point lies in the box."""
    b, p, o = _f32(boxes_tensor, "boxes", "cpu"), _f32(pts_tensor, "pts", "cpu"), _i32(pts_indices_tensor, "pts_indices", "cpu")
    if b.dim() != 2 or b.shape[1] != 7 or p.dim() != 2 or p.shape[1] != 3 or tuple(o.shape) != (b.shape[0], p.shape[0]):
        raise ValueError("points_in_boxes_cpu: boxes (N,7), pts (P,3), pts_indices (N,P)")
    _lib.check(_lib.load().av2x_points_in_boxes_cpu(_p(b), _p(p), b.shape[0], p.shape[0], _p(o)), "av2x_points_in_boxes_cpu")
    return 1


def points_in_boxes_gpu(boxes_tensor, pts_tensor, box_idx_of_points_tensor):
    """roiaware_pool3d.cpp:95-114: boxes (B,N,7), pts (B,P,3) on the device; box_idx_of_points (B,P) int32 (pre-filled with -1 by the
    caller) <- index of the first box holding each point."""
    b, p = _f32(boxes_tensor, "boxes", "cuda"), _f32(pts_tensor, "pts", "cuda")
    o = _i32(box_idx_of_points_tensor, "box_idx_of_points", "cuda")
    if b.dim() != 3 or b.shape[2] != 7 or p.dim() != 3 or p.shape[2] != 3 or b.shape[0] != p.shape[0] or tuple(o.shape) != tuple(p.shape[:2]):
        raise ValueError("points_in_boxes_gpu: boxes (B,N,7), pts (B,P,3), box_idx_of_points (B,P)")
    with torch.cuda.device(p.device):
        _lib.check(_lib.load().av2x_points_in_boxes_gpu(_p(b), _p(p), b.shape[0], b.shape[1], p.shape[1], _p(o), _stream(p)),
                   "av2x_points_in_boxes_gpu")
    return 1


def forward(rois, pts, pts_feature, argmax, pts_idx_of_voxels, pooled_features, pool_method):
    """roiaware_pool3d.cpp:27-63: rois (N,7), pts (P,3), pts_feature (P,C); argmax (N,ox,oy,oz,C) int32, pts_idx_of_voxels
    (N,ox,oy,oz,max_pts) int32 and pooled_features (N,ox,oy,oz,C) are the caller's zero tensors; pool_method 0 = max, 1 = avg."""
    r, p, f = _f32(rois, "rois", "cuda"), _f32(pts, "pts", "cuda"), _f32(pts_feature, "pts_feature", "cuda")
    am, vx = _i32(argmax, "argmax", "cuda"), _i32(pts_idx_of_voxels, "pts_idx_of_voxels", "cuda")
    pf = _f32(pooled_features, "pooled_features", "cuda")
    if vx.dim() != 5 or pf.dim() != 5:
        raise ValueError("forward: pts_idx_of_voxels (N,ox,oy,oz,max_pts), pooled_features (N,ox,oy,oz,C)")
    n, ox, oy, oz, maxp = (int(v) for v in vx.shape)
    # every extent the kernels index with comes from the shapes below: a mismatched caller gets an exception, not an out-of-bounds access
    # (the reference takes boxes_num from rois.size(0), roiaware_pool3d.cpp:41)
    if r.dim() != 2 or tuple(r.shape) != (n, 7) or p.dim() != 2 or p.shape[1] != 3 or f.dim() != 2 or f.shape[0] != p.shape[0]:
        raise ValueError(f"forward: rois (N,7), pts (P,3), pts_feature (P,C) with N = {n} (got {tuple(r.shape)}, {tuple(p.shape)}, {tuple(f.shape)})")
    c = int(f.shape[1])
    if tuple(pf.shape) != (n, ox, oy, oz, c) or tuple(am.shape) != (n, ox, oy, oz, c):
        raise ValueError(f"forward: argmax and pooled_features must be ({n},{ox},{oy},{oz},{c}) (got {tuple(am.shape)}, {tuple(pf.shape)})")
    if maxp < 1 or len({t.device for t in (r, p, f, am, vx, pf)}) != 1:
        raise ValueError("forward: max_pts >= 1 and every tensor on the same device")
    lib = _lib.load()
    with torch.cuda.device(p.device):
        ws = torch.empty(max(1, int(lib.av2x_roiaware_pool3d_workspace_bytes(n, p.shape[0])) // 4), dtype=torch.int32, device=p.device)
        _lib.check(lib.av2x_roiaware_pool3d_forward(_p(r), _p(p), _p(f), n, p.shape[0], f.shape[1], maxp, ox, oy, oz, _p(am), _p(vx), _p(pf),
                                                    int(pool_method), _p(ws), _stream(p)), "av2x_roiaware_pool3d_forward")
    return 1


def backward(pts_idx_of_voxels, argmax, grad_out, grad_in, pool_method):
    """roiaware_pool3d.cpp:65-93: grad_in (P,C) += the gradient of the pooled features (N,ox,oy,oz,C)."""
    vx, am = _i32(pts_idx_of_voxels, "pts_idx_of_voxels", "cuda"), _i32(argmax, "argmax", "cuda")
    go, gi = _f32(grad_out, "grad_out", "cuda"), _f32(grad_in, "grad_in", "cuda")
    if vx.dim() != 5 or go.dim() != 5 or gi.dim() != 2:
        raise ValueError("backward: pts_idx_of_voxels (N,ox,oy,oz,max_pts), grad_out (N,ox,oy,oz,C), grad_in (P,C)")
    n, ox, oy, oz, maxp = (int(v) for v in vx.shape)
    c = int(go.shape[4])
    if tuple(go.shape) != (n, ox, oy, oz, c) or tuple(am.shape) != (n, ox, oy, oz, c) or gi.shape[1] != c:
        raise ValueError(f"backward: argmax / grad_out ({n},{ox},{oy},{oz},C) and grad_in (P,C) with one C (got {tuple(am.shape)}, "
                         f"{tuple(go.shape)}, {tuple(gi.shape)})")
    if len({t.device for t in (vx, am, go, gi)}) != 1:
        raise ValueError("backward: every tensor on the same device")
    with torch.cuda.device(go.device):
        _lib.check(_lib.load().av2x_roiaware_pool3d_backward(_p(vx), _p(am), _p(go), n, ox, oy, oz, go.shape[4], maxp, _p(gi), int(pool_method),
                                                             _stream(go)), "av2x_roiaware_pool3d_backward")
    return 1
