"""AP evaluation: oracle vs the reference golden (CPU) and the GPU matcher vs both."""
import copy

import numpy as np
import pytest
import torch

from oracle import eval_oracle as eo
from tests.helpers import load_fixture

THS = (0.3, 0.5, 0.7)


def _frames(fx):
    out = []
    for i in range(int(fx["n_frames"])):
        det = fx[f"det_{i}"] if f"det_{i}" in fx else None
        out.append((det, fx[f"score_{i}"] if det is not None else None, fx[f"gt_{i}"]))
    return out


def _check_stats(fx, stat):
    for t in THS:
        k = int(t * 100)
        assert stat[t]["gt"] == int(fx[f"gt_total_{k}"])
        for gs in (False, True):
            ap, mrec, mpre = eo.calculate_ap(copy.deepcopy(stat), t, gs)
            assert ap == float(fx[f"ap_{k}_{int(gs)}"])
            if k == 50:
                assert np.array_equal(np.asarray(mrec), fx[f"mrec_50_{int(gs)}"])
                assert np.array_equal(np.asarray(mpre), fx[f"mpre_50_{int(gs)}"])


def test_oracle_matches_reference_golden():
    fx = load_fixture("eval_small")
    stat = {t: {"tp": [], "fp": [], "gt": 0, "score": []} for t in THS}
    for i, (det, score, gt) in enumerate(_frames(fx)):
        for t in THS:
            n0 = len(stat[t]["tp"])
            eo.caluclate_tp_fp(det, score, gt, stat, t)
            assert stat[t]["tp"][n0:] == fx[f"tp_{i}_{int(t * 100)}"].tolist(), (i, t)
            assert [1 - v for v in stat[t]["tp"][n0:]] == stat[t]["fp"][n0:]
    _check_stats(fx, stat)
    assert float(fx["ap_70_0"]) < float(fx["ap_30_0"])   # the fixture discriminates between thresholds


def test_voc_ap_known_answers():
    # perfect detector: precision 1 up to recall 1
    assert eo.voc_ap([0.5, 1.0], [1.0, 1.0])[0] == 1.0
    # one TP then one FP with 2 GT: recall 0.5 at precision 1 -> area 0.5
    assert eo.voc_ap([0.5, 0.5], [1.0, 0.5])[0] == 0.5
    # no detections: rec/prec empty -> 0
    assert eo.voc_ap([], [])[0] == 0.0
    from airv2x_perception_amd.opencood_iface import eval_utils as ev
    r, p = [0.1, 0.3, 0.3, 0.6], [1.0, 1.0, 0.66, 0.75]
    assert ev.voc_ap(r, p) == eo.voc_ap(r, p) and r == [0.1, 0.3, 0.3, 0.6]


@pytest.mark.gpu
def test_gpu_tp_fp_matches_reference_golden(tmp_path):
    from airv2x_perception_amd.opencood_iface import eval_utils as ev
    fx = load_fixture("eval_small")
    stat = {t: {"tp": [], "fp": [], "gt": 0, "score": []} for t in THS}
    ostat = {t: {"tp": [], "fp": [], "gt": 0, "score": []} for t in THS}
    for i, (det, score, gt) in enumerate(_frames(fx)):
        for t in THS:
            n0 = len(stat[t]["tp"])
            ev.caluclate_tp_fp(None if det is None else torch.from_numpy(det).cuda(),
                               None if det is None else torch.from_numpy(score).cuda(), torch.from_numpy(gt).cuda(), stat, t)
            eo.caluclate_tp_fp(det, score, gt, ostat, t)
            assert stat[t]["tp"][n0:] == fx[f"tp_{i}_{int(t * 100)}"].tolist(), (i, t)
    for t in THS:
        assert stat[t] == ostat[t]
    _check_stats(fx, stat)
    ap = ev.eval_final_results(copy.deepcopy(stat), str(tmp_path), eval_epoch=3)
    assert ap == tuple(float(fx[f"ap_{k}_0"]) for k in (30, 50, 70))
    assert (tmp_path / "eval_epoch3.yaml").exists()
    # matched ground-truth indices and the IoU the kernel saw agree with the oracle's fp64 clipping IoU
    det, score, gt = _frames(fx)[1]
    tp, s, mg = ev.match_tp_fp(torch.from_numpy(det), torch.from_numpy(score), torch.from_numpy(gt), 0.5)
    order = np.argsort(-score, kind="stable")
    for d in np.nonzero(tp.cpu().numpy())[0]:
        g = int(mg[d])
        assert eo.po.quad_iou(det[order[d], :4, :2], gt[g, :4, :2]) >= 0.5
    # (N,4,2) quads are accepted like the reference's convert_format does
    tp2, _, _ = ev.match_tp_fp(torch.from_numpy(det[:, :4, :2].copy()), torch.from_numpy(score), torch.from_numpy(gt[:, :4, :2].copy()), 0.5)
    assert torch.equal(tp, tp2)
    # empty ground truth: everything is a false positive; empty detections: nothing to do
    tp3, _, _ = ev.match_tp_fp(torch.from_numpy(det), torch.from_numpy(score), torch.zeros(0, 8, 3), 0.5)
    assert int(tp3.sum()) == 0
    tp4, _, _ = ev.match_tp_fp(torch.zeros(0, 8, 3), torch.zeros(0), torch.from_numpy(gt), 0.5)
    assert tp4.numel() == 0


@pytest.mark.gpu
def test_gpu_iou_matrix_matches_oracle_and_analytic_properties():
    """The fp64 convex-clipping IoU of av2x_eval_tp_fp (shared with the NMS) against the oracle's C restatement on random
    rotated rectangles, plus the properties any polygon IoU must have (shapely itself is absent: parity unpinned)."""
    from ctypes import c_void_p
    from airv2x_perception_amd import _lib
    lib = _lib.load()
    g = np.random.default_rng(12)

    def boxes(n, spread):
        c = g.uniform(-spread, spread, (n, 2))
        l, w, yaw = g.uniform(1.0, 6.0, n), g.uniform(0.8, 3.0, n), g.uniform(-np.pi, np.pi, n)
        base = np.array([[0.5, -0.5], [0.5, 0.5], [-0.5, 0.5], [-0.5, -0.5]])
        out = np.zeros((n, 8, 3), np.float32)
        for i in range(n):
            R = np.array([[np.cos(yaw[i]), -np.sin(yaw[i])], [np.sin(yaw[i]), np.cos(yaw[i])]])
            out[i, :4, :2] = (base * [l[i], w[i]]) @ R.T + c[i]
        return out

    det, gt = boxes(120, 8.0), boxes(90, 8.0)      # dense: most pairs overlap partially
    det[0], gt[0] = gt[1].copy(), gt[1].copy()     # identical boxes
    d, t = torch.from_numpy(det).cuda(), torch.from_numpy(gt).cuda()
    order = torch.arange(120, dtype=torch.int32, device="cuda")
    iou = torch.empty(120 * 90, device="cuda")
    tp = torch.empty(120, dtype=torch.int32, device="cuda")
    mg = torch.empty(120, dtype=torch.int32, device="cuda")
    P = lambda x: c_void_p(x.data_ptr())
    _lib.check(lib.av2x_eval_tp_fp(P(d), P(order), 120, P(t), 90, 0.5, P(iou), P(tp), P(mg),
                                   c_void_p(torch.cuda.current_stream().cuda_stream)), "eval")
    m = iou.view(120, 90).cpu().numpy()
    ref = np.array([[eo.po.quad_iou(det[i, :4, :2], gt[j, :4, :2]) for j in range(90)] for i in range(120)], np.float32)
    assert np.abs(m - ref).max() <= 1e-6
    assert m.min() >= 0.0 and m.max() <= 1.0 + 1e-6 and abs(m[0, 1] - 1.0) < 1e-6
    assert (m > 0).mean() > 0.05                    # the test is not vacuous
    # symmetry: IoU(a, b) == IoU(b, a)
    iou2 = torch.empty(90 * 120, device="cuda")
    tp2 = torch.empty(90, dtype=torch.int32, device="cuda")
    mg2 = torch.empty(90, dtype=torch.int32, device="cuda")
    o2 = torch.arange(90, dtype=torch.int32, device="cuda")
    _lib.check(lib.av2x_eval_tp_fp(P(t), P(o2), 90, P(d), 120, 0.5, P(iou2), P(tp2), P(mg2),
                                   c_void_p(torch.cuda.current_stream().cuda_stream)), "eval")
    assert np.abs(iou2.view(90, 120).cpu().numpy().T - m).max() <= 1e-6
