"""Pins of the two third-party-backed pieces of the path against REFERENCE-HELD arithmetic:

* polygon IoU (a19 / a20): the reference calls shapely (absent here).  Its repository also holds an independent C++
  rotated-IoU (pcdet_utils/iou3d_nms/src/iou3d_cpu.cpp:128-262).  tests/golden/iou_pin.npz stores THAT code's IoUs
  (compiled in the build container by oracle/build_ref.py) for 4096 box pairs, 1506 of them within 1e-3 of the
  thresholds 0.15 / 0.3 / 0.5 / 0.7; oracle/nms_oracle.c and the device kernels must reproduce them -- to fp32 rounding
  wherever that code forms the exact intersection polygon (98 % of the pairs), and within its own 1 cm corner-acceptance
  margin (:77) on the rest, where the restatement (exact clipping, what GEOS computes) must be the SMALLER value.
* pillar voxelizer (a2): the reference calls spconv (absent).  Its repository also holds a numpy voxelizer
  (data_utils/pre_processor/voxel_preprocessor.py:30-80); tests/golden/voxel_pin.npz stores ITS output on a cloud with
  166 over-full voxels.  The oracle and the device voxelizer must produce the same voxel SET, the same counts and keep
  the same points (order of the voxels is np.unique's there, first appearance in spconv: compared order-independently).
What stays unpinned: shapely's own result on the same pairs, spconv's voxel order / its >max_voxels cut.
"""
import numpy as np
import pytest
import torch

from oracle import postprocess_oracle as po
from oracle import voxelize_oracle as vox
from tests.helpers import load_fixture

IOU_TOL = 2e-5      # the reference C++ works in fp32 with epsilon tests, the restatement clips in fp64
MARGIN = 1e-2       # iou3d_cpu.cpp:77: check_in_box2d counts a corner up to 1 cm OUTSIDE the other box as inside


def _corner_in_margin(box, quad):
    """True when a corner of `quad` lies in the 1 cm band just outside `box` (x,y,z,dx,dy,dz,heading) where the reference's
    check_in_box2d (iou3d_cpu.cpp:75-85) already counts it as inside -- the one place its polygon is not the exact one."""
    c, s_ = np.cos(-box[6]), np.sin(-box[6])
    dx, dy = quad[:, 0] - box[0], quad[:, 1] - box[1]
    rx, ry = np.abs(dx * c - dy * s_), np.abs(dx * s_ + dy * c)
    inside_loose = (rx < box[3] / 2 + MARGIN) & (ry < box[4] / 2 + MARGIN)
    inside_exact = (rx <= box[3] / 2 - 1e-6) & (ry <= box[4] / 2 - 1e-6)
    return bool((inside_loose & ~inside_exact).any())


def _decisions_agree(got, fx, what):
    """`got` must reproduce the reference C++'s IoU to fp32 rounding wherever that code computes the exact polygon, and
    may only differ -- downwards, by less than 1e-2 -- on pairs with a corner inside its 1 cm acceptance margin."""
    ref, thr = fx["iou"], fx["near_threshold"]
    err = np.abs(got - ref)
    loose = err > IOU_TOL
    assert loose.mean() < 0.03, (what, float(loose.mean()))
    assert err.max() < 1e-2, (what, float(err.max()))
    for i in np.nonzero(loose)[0]:
        assert got[i] < ref[i], (what, i)           # the margin only ever ADDS area to the reference's polygon
        assert _corner_in_margin(fx["boxes_a"][i], fx["quads_b"][i]) or _corner_in_margin(fx["boxes_b"][i], fx["quads_a"][i]), (what, i)
    exact = ~loose
    for t in (0.15, 0.3, 0.5, 0.7):          # every threshold the path compares against
        clear = exact & (np.abs(ref - t) > IOU_TOL)
        assert np.array_equal((got > t)[clear], (ref > t)[clear]), (what, t)
        assert np.array_equal((got >= t)[clear], (ref >= t)[clear]), (what, t)
    near = exact & (thr > 0) & (np.abs(ref - thr) < 1e-3)
    assert near.sum() > 1000                    # > 1000 pairs within 1e-3 of 0.15 / 0.3 / 0.5 / 0.7 decide identically
    return float(err[exact].max())


def test_oracle_polygon_iou_matches_the_references_cpp_iou():
    fx = load_fixture("iou_pin")
    qa, qb = fx["quads_a"].astype(np.float64), fx["quads_b"].astype(np.float64)
    got = np.array([po.quad_iou(qa[i], qb[i]) for i in range(qa.shape[0])], np.float64)
    e = _decisions_agree(got.astype(np.float32), fx, "oracle")
    print(f"[iou pin] oracle vs reference iou3d_cpu.cpp: max |diff| {e:.2e} over {qa.shape[0]} pairs")


@pytest.mark.gpu
def test_gpu_eval_iou_kernel_matches_the_references_cpp_iou():
    """av2x_eval_tp_fp's IoU matrix (the kernel behind caluclate_tp_fp; the NMS mask kernel shares its clipping code)."""
    from ctypes import c_void_p

    from airv2x_perception_amd import _lib
    lib = _lib.load()
    fx = load_fixture("iou_pin")
    n = fx["quads_a"].shape[0]
    got = np.empty(n, np.float32)
    P = lambda t: c_void_p(t.data_ptr())
    for i0 in range(0, n, 512):
        k = min(512, n - i0)
        det = torch.zeros(k, 8, 3)
        gt = torch.zeros(k, 8, 3)
        det[:, :4, :2] = torch.from_numpy(fx["quads_a"][i0:i0 + k])
        gt[:, :4, :2] = torch.from_numpy(fx["quads_b"][i0:i0 + k])
        det, gt = det.cuda(), gt.cuda()
        order = torch.arange(k, dtype=torch.int32, device="cuda")
        iou = torch.empty(k * k, device="cuda")
        tp = torch.empty(k, dtype=torch.int32, device="cuda")
        mg = torch.empty(k, dtype=torch.int32, device="cuda")
        _lib.check(lib.av2x_eval_tp_fp(P(det), P(order), k, P(gt), k, 0.5, P(iou), P(tp), P(mg),
                                       c_void_p(torch.cuda.current_stream().cuda_stream)), "av2x_eval_tp_fp")
        got[i0:i0 + k] = torch.diagonal(iou.view(k, k)).cpu().numpy()
    e = _decisions_agree(got, fx, "device")
    print(f"[iou pin] device IoU vs reference iou3d_cpu.cpp: max |diff| {e:.2e}")


def _by_voxel(voxels, coords, num):
    order = np.lexsort((coords[:, 2], coords[:, 1], coords[:, 0]))        # (z, y, x) ascending = np.unique(axis=0) order
    return voxels[order], coords[order], num[order]


def _check_voxels(voxels, coords, num, fx):
    v, c, n = _by_voxel(np.asarray(voxels), np.asarray(coords), np.asarray(num))
    assert np.array_equal(c, fx["coords"])                      # the same SET of occupied voxels
    assert np.array_equal(n, fx["counts"])                      # the same per-voxel counts, capped at 32
    assert int((n == 32).sum()) == 166
    assert np.array_equal(v, fx["points_kept"])                 # the same points survive in over-full voxels, same order


def test_oracle_voxelizer_matches_the_references_numpy_voxelizer():
    fx = load_fixture("voxel_pin")
    rng, vs = [float(v) for v in fx["lidar_range"]], [float(v) for v in fx["voxel_size"]]
    _check_voxels(*vox.points_to_voxels(fx["points"], rng, vs, 32, 70000), fx)


@pytest.mark.gpu
def test_gpu_voxelizer_matches_the_references_numpy_voxelizer():
    from airv2x_perception_amd.opencood_iface.voxelizer import voxelize_points
    fx = load_fixture("voxel_pin")
    rng, vs = [float(v) for v in fx["lidar_range"]], [float(v) for v in fx["voxel_size"]]
    v, c, n = voxelize_points(torch.from_numpy(fx["points"]).cuda(), rng, vs, 32, 70000)
    _check_voxels(v.cpu().numpy(), c.cpu().numpy(), n.cpu().numpy(), fx)
