"""``box_overlaps`` -- drop-in for the reference's Cython module ``opencood.utils.box_overlaps`` (utils/box_overlaps.pyx:17-143, imported
at utils/box_utils.py / data_utils/post_processor/voxel_postprocessor.py:20 before a post-processor can be built): ``bbox_overlaps``,
``bbox_intersections``, ``box_vote`` on float32 numpy arrays, boxes [x1, y1, x2, y2] with the "+1" pixel convention.  Host code in
libairv2x_hip.so (csrc/roiaware.hip) behind the C-ABI; pinned to the reference's own compiled module by tests/golden/box_overlaps_pin.npz."""
from __future__ import annotations

from ctypes import c_void_p

import numpy as np

from .. import _lib


def _arr(a, name, cols):
    if not isinstance(a, np.ndarray) or a.dtype != np.float32 or a.ndim != 2:
        raise ValueError(f"Buffer dtype mismatch / wrong number of dimensions: {name} must be a 2-d float32 ndarray")   # what the typed Cython signature raises
    if a.shape[1] < cols:
        raise IndexError(f"{name}: rows of at least {cols} values")
    return np.ascontiguousarray(a)


def _ptr(a):
    return c_void_p(a.ctypes.data)


def bbox_overlaps(boxes, query_boxes):
    """(N,4), (K,4) -> (N,K) IoU (box_overlaps.pyx:17-57)."""
    b, q = _arr(boxes, "boxes", 4), _arr(query_boxes, "query_boxes", 4)
    if b.shape[1] != 4 or q.shape[1] != 4:
        b, q = np.ascontiguousarray(b[:, :4]), np.ascontiguousarray(q[:, :4])
    out = np.zeros((b.shape[0], q.shape[0]), np.float32)
    _lib.check(_lib.load().av2x_bbox_overlaps(_ptr(b), _ptr(q), b.shape[0], q.shape[0], _ptr(out), 0), "av2x_bbox_overlaps")
    return out


def bbox_intersections(boxes, query_boxes):
    """(N,4), (K,4) -> (N,K) intersection / area(query box) (box_overlaps.pyx:59-97)."""
    b, q = _arr(boxes, "boxes", 4), _arr(query_boxes, "query_boxes", 4)
    if b.shape[1] != 4 or q.shape[1] != 4:
        b, q = np.ascontiguousarray(b[:, :4]), np.ascontiguousarray(q[:, :4])
    out = np.zeros((b.shape[0], q.shape[0]), np.float32)
    _lib.check(_lib.load().av2x_bbox_overlaps(_ptr(b), _ptr(q), b.shape[0], q.shape[0], _ptr(out), 1), "av2x_bbox_overlaps")
    return out


def box_vote(dets_NMS, dets_all):
    """(N,C), (M,C) rows [x1, y1, x2, y2, score, ...] -> (N,C): score-weighted mean of the dets_all boxes with IoU >= 0.5 to each kept
    detection, its original score in column 4 (box_overlaps.pyx:99-143)."""
    a, b = _arr(dets_NMS, "dets_NMS", 5), _arr(dets_all, "dets_all", 5)
    if a.shape[1] != b.shape[1]:
        raise ValueError("box_vote: dets_NMS and dets_all must have the same number of columns")
    out = np.zeros_like(a)
    _lib.check(_lib.load().av2x_box_vote(_ptr(a), _ptr(b), a.shape[0], b.shape[0], a.shape[1], _ptr(out)), "av2x_box_vote")
    return out
