"""ORACLE (test infrastructure, not product code): training-label assignment.

numpy restatement of VoxelPostprocessor.generate_label_airv2x (data_utils/post_processor/voxel_postprocessor.py:217-354),
bbox_overlaps (utils/box_overlaps.pyx:17-57, fp32, with its "+ 1" pixel convention), boxes_to_corners_3d
(utils/box_utils.py:195-258, torch fp32) and corner2d_to_standup_box (:279-302).
Parity: PINNED by tests/golden/labels_*.npz -- tools/gen_golden.py runs the reference's own method with its own
box_overlaps.pyx compiled by oracle/build_ref.py.
"""
from __future__ import annotations

import numpy as np
import torch

from .postprocess_oracle import boxes_to_corners_3d


def bbox_overlaps(boxes, query_boxes):
    """box_overlaps.pyx:17-57 as Cython 3 compiles it: the operands are C floats, the literal 1 becomes the double 1.0,
    so every "+ 1" (and the whole union expression inside float(...)) is evaluated in double and rounded to float on
    assignment to the float-typed iw / ih / box_area / ua; the differences and iw * ih are float operations."""
    b, q = boxes.astype(np.float32), query_boxes.astype(np.float32)
    f32, f64 = np.float32, np.float64
    box_area = ((q[:, 2] - q[:, 0]).astype(f64) + 1.0) * ((q[:, 3] - q[:, 1]).astype(f64) + 1.0)
    box_area = box_area.astype(f32)
    iw = ((np.minimum(b[:, None, 2], q[None, :, 2]) - np.maximum(b[:, None, 0], q[None, :, 0])).astype(f64) + 1.0).astype(f32)
    ih = ((np.minimum(b[:, None, 3], q[None, :, 3]) - np.maximum(b[:, None, 1], q[None, :, 1])).astype(f64) + 1.0).astype(f32)
    inter = iw * ih                                                                     # float * float
    barea = ((b[:, 2] - b[:, 0]).astype(f64) + 1.0) * ((b[:, 3] - b[:, 1]).astype(f64) + 1.0)
    ua = (barea[:, None] + box_area[None, :].astype(f64) - inter.astype(f64)).astype(f32)
    out = np.zeros((b.shape[0], q.shape[0]), f32)
    ok = (iw > 0) & (ih > 0)
    out[ok] = inter[ok] / ua[ok]
    return out


def standup(boxes7, order="hwl"):
    c = boxes_to_corners_3d(torch.from_numpy(np.asarray(boxes7)).float(), order).numpy()
    s = np.zeros((c.shape[0], 4))
    s[:, 0], s[:, 1] = c[:, :, 0].min(1), c[:, :, 1].min(1)
    s[:, 2], s[:, 3] = c[:, :, 0].max(1), c[:, :, 1].max(1)
    return np.ascontiguousarray(s).astype(np.float32)


def generate_label(gt_box_center, anchors, mask, class_ids_padded, pos_threshold, neg_threshold):
    A = anchors.shape[2]
    shape = anchors.shape[:2]
    class_ids_valid = class_ids_padded[mask == 1]
    anchors = anchors.reshape(-1, 7)
    anchors_d = np.sqrt(anchors[:, 4] ** 2 + anchors[:, 5] ** 2)
    pos, neg = np.zeros((*shape, A)), np.zeros((*shape, A))
    targets = np.zeros((*shape, A * 7))
    gt = gt_box_center[mask == 1]
    iou = bbox_overlaps(standup(anchors), standup(gt))
    id_highest = np.argmax(iou.T, axis=1)
    id_highest_gt = np.arange(iou.T.shape[0])
    m = iou.T[id_highest_gt, id_highest] > 0
    id_highest, id_highest_gt = id_highest[m], id_highest_gt[m]
    id_pos, id_pos_gt = np.where(iou > pos_threshold)
    id_neg = np.where(np.sum(iou < neg_threshold, axis=1) == iou.shape[1])[0]
    id_pos = np.concatenate([id_pos, id_highest])
    id_pos_gt = np.concatenate([id_pos_gt, id_highest_gt])
    id_pos, index = np.unique(id_pos, return_index=True)
    id_pos_gt = id_pos_gt[index]
    ix, iy, iz = np.unravel_index(id_pos, (*shape, A))
    pos[ix, iy, iz] = 1
    cls = np.zeros((*shape, A), dtype=int)
    cls[ix, iy, iz] = class_ids_valid[id_pos_gt]
    g = gt_box_center[mask == 1] if False else gt_box_center   # the reference indexes the PADDED array with valid-box indices (:311-330)
    targets[ix, iy, iz * 7] = (g[id_pos_gt, 0] - anchors[id_pos, 0]) / anchors_d[id_pos]
    targets[ix, iy, iz * 7 + 1] = (g[id_pos_gt, 1] - anchors[id_pos, 1]) / anchors_d[id_pos]
    targets[ix, iy, iz * 7 + 2] = (g[id_pos_gt, 2] - anchors[id_pos, 2]) / anchors[id_pos, 3]
    targets[ix, iy, iz * 7 + 3] = np.log(g[id_pos_gt, 3] / anchors[id_pos, 3])
    targets[ix, iy, iz * 7 + 4] = np.log(g[id_pos_gt, 4] / anchors[id_pos, 4])
    targets[ix, iy, iz * 7 + 5] = np.log(g[id_pos_gt, 5] / anchors[id_pos, 5])
    targets[ix, iy, iz * 7 + 6] = g[id_pos_gt, 6] - anchors[id_pos, 6]
    ix, iy, iz = np.unravel_index(id_neg, (*shape, A))
    neg[ix, iy, iz] = 1
    ix, iy, iz = np.unravel_index(id_highest, (*shape, A))
    neg[ix, iy, iz] = 0
    return {"pos_equal_one": pos, "neg_equal_one": neg, "targets": targets, "cls_labels": cls}
