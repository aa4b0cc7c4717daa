"""Drop-in for opencood/utils/eval_utils_opv2v.py (voc_ap :15-38, caluclate_tp_fp :41-97, calculate_ap
:100-151, eval_final_results :154-189) with the per-frame matching on the MI355X.

``caluclate_tp_fp`` (the reference's spelling) keeps the call contract of inference*.py: detections and
ground truth as (N,8,3) corner tensors, ``result_stat[iou] = {"tp": [], "fp": [], "gt": 0, "score": []}``
updated in place.  The polygon IoU matrix and the greedy matching run in libairv2x_hip.so
(av2x_eval_tp_fp); one (n_det,) int read-back per call replaces the O(n_det x n_gt) shapely loop.
There is no CPU path: tensors must be (or are moved) on the GPU."""
from __future__ import annotations

import os
from ctypes import c_void_p

import numpy as np
import torch

from .. import _lib


def voc_ap(rec, prec):
    """VOC 2010 average precision (eval_utils_opv2v.py:15-38).  Like the reference it returns
    (ap, mrec, mpre) with the 0/1 sentinels added; unlike it the caller's lists are not modified."""
    mrec = np.concatenate(([0.0], np.asarray(rec, dtype=np.float64), [1.0]))
    mpre = np.concatenate(([0.0], np.asarray(prec, dtype=np.float64), [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]            # monotone precision envelope (right to left running max)
    ap = 0.0
    for i in (np.flatnonzero(mrec[1:] != mrec[:-1]) + 1).tolist():   # recall steps, summed left to right in Python floats
        ap += float(mrec[i] - mrec[i - 1]) * float(mpre[i])
    mrec, mpre = mrec.tolist(), mpre.tolist()
    return ap, mrec, mpre


def _corners(t, dev):
    t = torch.as_tensor(t)
    if t.dim() == 3 and t.shape[1:] == (4, 2):   # (N,4,2) BEV quads: pad to the (N,8,3) layout the kernel reads
        full = torch.zeros((t.shape[0], 8, 3), dtype=torch.float32)
        full[:, :4, :2] = t
        t = full
    if t.dim() != 3 or t.shape[1:] != (8, 3):
        raise ValueError(f"boxes must be (N,8,3) or (N,4,2), got {tuple(t.shape)}")
    return t.to(device=dev, dtype=torch.float32).contiguous()


def match_tp_fp(det_boxes, det_score, gt_boxes, iou_thresh, device=None):
    """(tp (N,) int32 in score order, sorted scores (N,), matched gt index (N,)) as device tensors."""
    dev = torch.device(device) if device is not None else (det_boxes.device if torch.is_tensor(det_boxes) and det_boxes.is_cuda
                                                           else torch.device("cuda"))
    lib = _lib.load()
    det = _corners(det_boxes, dev)
    gt = _corners(gt_boxes, dev)
    score = torch.as_tensor(det_score).to(device=dev, dtype=torch.float32)
    n, g = det.shape[0], gt.shape[0]
    sorted_score, order = torch.sort(score, descending=True, stable=True)
    order32 = order.to(torch.int32)
    tp = torch.zeros(n, dtype=torch.int32, device=dev)
    mg = torch.full((n,), -1, dtype=torch.int32, device=dev)
    ws = torch.empty(max(n * g, 1), dtype=torch.float32, device=dev)
    P = lambda t: c_void_p(t.data_ptr())
    with torch.cuda.device(dev):
        _lib.check(lib.av2x_eval_tp_fp(P(det), P(order32), n, P(gt), g, float(iou_thresh), P(ws), P(tp), P(mg),
                                       c_void_p(torch.cuda.current_stream().cuda_stream)), "av2x_eval_tp_fp")
    return tp, sorted_score, mg


def caluclate_tp_fp(det_boxes, det_score, gt_boxes, result_stat, iou_thresh):
    """eval_utils_opv2v.py:41-97."""
    fp, tp = [], []
    gt = int(gt_boxes.shape[0])
    if det_boxes is not None:
        t, s, _ = match_tp_fp(det_boxes, det_score, gt_boxes, iou_thresh)
        t = t.cpu().numpy()
        tp = t.tolist()
        fp = (1 - t).tolist()
        result_stat[iou_thresh]["score"] += s.cpu().numpy().tolist()
    result_stat[iou_thresh]["fp"] += fp
    result_stat[iou_thresh]["tp"] += tp
    result_stat[iou_thresh]["gt"] += gt


def calculate_ap(result_stat, iou, global_sort_detections):
    """eval_utils_opv2v.py:100-151."""
    st = result_stat[iou]
    if global_sort_detections:
        fp, tp, score = np.array(st["fp"]), np.array(st["tp"]), np.array(st["score"])
        assert len(fp) == len(tp) and len(tp) == len(score)
        idx = np.argsort(-score)
        fp, tp = fp[idx], tp[idx]
    else:
        fp, tp = np.asarray(st["fp"], dtype=np.int64), np.asarray(st["tp"], dtype=np.int64)
        assert len(fp) == len(tp)
    gt_total = st["gt"]
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = [float(t) / gt_total for t in tp]
    prec = [float(t) / (f + t) for t, f in zip(tp, fp)]
    return voc_ap(rec, prec)


def eval_final_results(result_stat, save_path, global_sort_detections=False, eval_epoch=None):
    """eval_utils_opv2v.py:154-189: AP@0.3/0.5/0.7, dumped to ``save_path/eval_epoch{N}.yaml``."""
    res = {thr: calculate_ap(result_stat, thr, global_sort_detections) for thr in (0.30, 0.50, 0.70)}
    ap_30, ap_50, ap_70 = res[0.30][0], res[0.50][0], res[0.70][0]
    dump = {"ap_30": ap_30, "ap_50": ap_50, "ap_70": ap_70, "mpre_50": res[0.50][2], "mrec_50": res[0.50][1],
            "mpre_70": res[0.70][2], "mrec_70": res[0.70][1]}
    name = f"eval_epoch{eval_epoch}.yaml" if not global_sort_detections else "eval_global_sort.yaml"
    if save_path is not None:
        import yaml
        with open(os.path.join(save_path, name), "w") as f:
            yaml.dump(dump, f, default_flow_style=False)
    print("The Average Precision at IOU 0.3 is %.2f, The Average Precision at IOU 0.5 is %.2f, "
          "The Average Precision at IOU 0.7 is %.2f" % (ap_30, ap_50, ap_70))
    return ap_30, ap_50, ap_70
