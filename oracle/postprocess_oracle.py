"""ORACLE (test infrastructure, not product code): anchors, box decoding, filters, rotated NMS.

Restates data_utils/post_processor/voxel_postprocessor.py (generate_anchor_box :33-86,
delta_to_boxes3d :585-634, post_process_airv2x :666-839) and the helpers it calls in
utils/box_utils.py (boxes_to_corners_3d :195-258, project_box3d :332-366,
corner_to_standup_box_torch :305-329, remove_large_pred_bbx :981-1014, remove_bbx_abnormal_z
:1017-1035, nms_rotated :823-868, get_mask_for_boxes_within_range_torch :399-430) with the same
torch CPU ops.  Parity: decode + filters are PINNED by golden vectors captured from the reference
(tools/gen_golden.py monkey-patches only ``nms_rotated``, whose shapely dependency is absent);
the polygon IoU inside the NMS is UNPINNED (see oracle/nms_oracle.c).
"""
from __future__ import annotations

import ctypes
import math

import numpy as np
import torch

from . import build_oracle

_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build_oracle.build())
        _LIB.av2x_oracle_quad_iou.restype = ctypes.c_double
        _LIB.av2x_oracle_quad_iou.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _LIB.av2x_oracle_nms_rotated.restype = ctypes.c_int
        _LIB.av2x_oracle_nms_rotated.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                                 ctypes.c_int, ctypes.c_void_p]
    return _LIB


def generate_anchor_box(pp):
    """voxel_postprocessor.py:33-86 -> (H/stride, W/stride, A, 7) float64, order hwl."""
    a = pp["anchor_args"]
    W, H = a["W"], a["H"]
    l, w, h = a["l"], a["w"], a["h"]
    r = [math.radians(e) for e in a["r"]]
    num = a.get("num", 2)
    assert num == len(r)
    vh, vw = a["vh"], a["vw"]
    rng = a["cav_lidar_range"]
    fs = a.get("feature_stride", 2)
    x = np.linspace(rng[0] + vw, rng[3] - vw, W // fs)
    y = np.linspace(rng[1] + vh, rng[4] - vh, H // fs)
    cx, cy = np.meshgrid(x, y)
    cx = np.tile(cx[..., np.newaxis], num)
    cy = np.tile(cy[..., np.newaxis], num)
    cz = np.ones_like(cx) * -1.0
    ww, ll, hh = np.ones_like(cx) * w, np.ones_like(cx) * l, np.ones_like(cx) * h
    r_ = np.ones_like(cx)
    for i in range(num):
        r_[..., i] = r[i]
    if pp["order"] == "hwl":
        return np.stack([cx, cy, cz, hh, ww, ll, r_], axis=-1)
    if pp["order"] == "lhw":
        return np.stack([cx, cy, cz, ll, hh, ww, r_], axis=-1)
    raise ValueError("Unknown bbx order.")


def delta_to_boxes3d(deltas, anchors):
    """voxel_postprocessor.py:585-634.  deltas (N,14,H,W) f32, anchors (H,W,2,7) -> (N, H*W*2, 7)."""
    N = deltas.shape[0]
    deltas = deltas.permute(0, 2, 3, 1).contiguous().view(N, -1, 7)
    boxes3d = torch.zeros_like(deltas)
    ar = anchors.view(-1, 7).float()
    ad = torch.sqrt(ar[:, 4] ** 2 + ar[:, 5] ** 2)
    ad = ad.repeat(N, 2, 1).transpose(1, 2)
    ar = ar.repeat(N, 1, 1)
    boxes3d[..., [0, 1]] = torch.mul(deltas[..., [0, 1]], ad) + ar[..., [0, 1]]
    boxes3d[..., [2]] = torch.mul(deltas[..., [2]], ar[..., [3]]) + ar[..., [2]]
    boxes3d[..., [3, 4, 5]] = torch.exp(deltas[..., [3, 4, 5]]) * ar[..., [3, 4, 5]]
    boxes3d[..., 6] = deltas[..., 6] + ar[..., 6]
    return boxes3d


def boxes_to_corners_3d(boxes3d, order):
    """box_utils.py:195-258 (+ common_utils.rotate_points_along_z :60-82)."""
    b = boxes3d[:, [0, 1, 2, 5, 4, 3, 6]] if order == "hwl" else boxes3d
    template = b.new_tensor(([1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, -1],
                             [1, -1, 1], [1, 1, 1], [-1, 1, 1], [-1, -1, 1])) / 2
    c = b[:, None, 3:6].repeat(1, 8, 1) * template[None, :, :]
    ang = b[:, 6]
    cosa, sina = torch.cos(ang), torch.sin(ang)
    z, o = ang.new_zeros(c.shape[0]), ang.new_ones(c.shape[0])
    rot = torch.stack((cosa, sina, z, -sina, cosa, z, z, z, o), dim=1).view(-1, 3, 3).float()
    c = torch.matmul(c.view(-1, 8, 3)[:, :, 0:3].float(), rot).view(-1, 8, 3)
    return c + b[:, None, 0:3]


def project_box3d(box3d, T):
    """box_utils.py:332-366."""
    c = box3d.transpose(1, 2)
    c = torch.cat((c, torch.ones((c.shape[0], 1, 8))), dim=1)
    return torch.matmul(T, c)[:, :3, :].transpose(1, 2)


def nms_rotated(corners8, scores, threshold, top=1000):
    """box_utils.py:823-868 with the C restatement of the shapely IoU (oracle/nms_oracle.c)."""
    n = corners8.shape[0]
    if n == 0:
        return np.array([], dtype=np.int32)
    quads = np.ascontiguousarray(corners8[:, :4, :2].detach().cpu().numpy(), dtype=np.float32)
    sc = np.ascontiguousarray(scores.detach().cpu().numpy(), dtype=np.float32)
    keep = np.empty(n, dtype=np.int32)
    k = _lib().av2x_oracle_nms_rotated(quads.ctypes.data, sc.ctypes.data, n, float(threshold), top, keep.ctypes.data)
    return keep[:k].copy()


def quad_iou(qa, qb):
    a = np.ascontiguousarray(qa, dtype=np.float64)
    b = np.ascontiguousarray(qb, dtype=np.float64)
    return float(_lib().av2x_oracle_quad_iou(a.ctypes.data, b.ctypes.data))


def post_process(psm, rm, obj, anchors, T, pp, lidar_range, num_class=7, stages=None):
    """voxel_postprocessor.py:666-839 for the single 'ego' entry of intermediate fusion.
    psm (1,A*C,H,W), rm (1,A*7,H,W), obj (1,A,H,W) f32; anchors (H,W,A,7) f64; T (4,4) f32.
    Returns (corners (K,8,3), scores (K,), labels (K,), boxes3d (K,7)) or (None,)*4.
    ``stages`` (dict) receives the intermediate tensors for the golden tests."""
    C = num_class
    objectness = torch.sigmoid(obj.permute(0, 2, 3, 1).contiguous()).view(1, -1)
    B, AC, H, W = psm.shape
    A = AC // C
    p = psm.view(B, C, A, H, W).permute(0, 3, 4, 2, 1).contiguous()
    prob = torch.sigmoid(p).view(1, -1, C)[:, :, 1:]
    _, labels = torch.max(prob, dim=-1)
    labels = labels + 1
    mask = objectness > pp["target_args"]["obj_threshold"]
    if mask.sum() == 0:
        return None, None, None, None
    batch_box3d = delta_to_boxes3d(rm, anchors)
    boxes3d = torch.masked_select(batch_box3d[0], mask[0].unsqueeze(-1).repeat(1, 7)).view(-1, 7)
    scores = torch.masked_select(objectness[0], mask[0])
    labels3d = torch.masked_select(labels[0], mask[0])
    corners = project_box3d(boxes_to_corners_3d(boxes3d, pp["order"]), T)
    x_len = corners[:, :, 0].max(1)[0] - corners[:, :, 0].min(1)[0]
    y_len = corners[:, :, 1].max(1)[0] - corners[:, :, 1].min(1)[0]
    z_len = corners[:, :, 2].max(1)[0] - corners[:, :, 2].min(1)[0]
    keep1 = torch.logical_and(torch.logical_and(x_len <= 6, y_len <= 6), z_len)  # box_utils.py:1011-1012
    keep2 = torch.logical_and(corners[:, :, 2].min(1)[0] >= lidar_range[2], corners[:, :, 2].max(1)[0] <= lidar_range[5])
    keep = torch.logical_and(keep1, keep2)
    if stages is not None:
        stages.update({"cand_index": torch.nonzero(mask[0]).view(-1), "cand_boxes3d": boxes3d, "cand_scores": scores,
                       "cand_labels": labels3d, "cand_corners": corners, "cand_keep": keep})
    corners, scores, labels3d, boxes3d = corners[keep], scores[keep], labels3d[keep], boxes3d[keep]
    k = torch.from_numpy(nms_rotated(corners, scores, pp["nms_thresh"]).astype(np.int64))
    if stages is not None:
        stages.update({"nms_in_corners": corners, "nms_in_scores": scores, "nms_keep": k})
    corners, scores, labels3d, boxes3d = corners[k], scores[k], labels3d[k], boxes3d[k]
    lo = torch.tensor(lidar_range[:2], dtype=torch.float32).view(1, 1, -1)
    hi = torch.tensor(lidar_range[3:5], dtype=torch.float32).view(1, 1, -1)
    inr = torch.all(torch.all(corners[:, :, :2] >= lo, dim=-1) & torch.all(corners[:, :, :2] <= hi, dim=-1), dim=-1)
    return corners[inr], scores[inr], labels3d[inr], boxes3d[inr]
