"""The two import-time native modules of the reference's callers (VERDICT r04 "missing 3"):
``roiaware_pool3d_cuda`` (pcdet_utils/roiaware_pool3d/src/roiaware_pool3d.cpp:27-183 + roiaware_pool3d_kernel.cu) and
``opencood.utils.box_overlaps`` (utils/box_overlaps.pyx:17-143), as Python shims over the C-ABI of libairv2x_hip.so.
box_overlaps is pinned bit for bit to outputs of the reference's OWN compiled module (tests/golden/box_overlaps_pin.npz, made by
tools/gen_golden.py from oracle/_ref's Cython build); the RoI-aware pooling is held to oracle/roiaware_oracle.py (restatement, known
answers below: its CUDA kernels cannot run here)."""
import sys
import types

import numpy as np
import pytest
import torch

from oracle import roiaware_oracle as ro
from tests.helpers import load_fixture


def _boxes3d(g, n, spread=20.0):
    c = g.uniform(-spread, spread, (n, 3)).astype(np.float32)
    c[:, 2] = g.uniform(-1, 1, n)
    d = g.uniform(1.0, 6.0, (n, 3)).astype(np.float32)
    return np.concatenate([c, d, g.uniform(-3.2, 3.2, (n, 1)).astype(np.float32)], 1).astype(np.float32)


def _points_near(g, boxes, per_box, extra):
    pts = [b[:3] + g.normal(0, 1, (per_box, 3)).astype(np.float32) * b[3:6] * 0.45 for b in boxes]
    pts.append(g.uniform(-25, 25, (extra, 3)).astype(np.float32))
    return np.concatenate(pts, 0).astype(np.float32)


# ------------------------------------------------------------------------------------------------------------------ box_overlaps
def test_box_overlaps_equals_the_references_compiled_module_bit_for_bit():
    from airv2x_perception_amd.opencood_iface import box_overlaps as bo
    fx = load_fixture("box_overlaps_pin")
    for si in range(len(fx["scales"])):
        a, b, d1, d2 = fx[f"a{si}"], fx[f"b{si}"], fx[f"d1_{si}"], fx[f"d2_{si}"]
        assert np.array_equal(bo.bbox_overlaps(a, b), fx[f"overlaps{si}"])
        assert np.array_equal(bo.bbox_intersections(a, b), fx[f"intersections{si}"])
        assert np.array_equal(bo.box_vote(d1, d2), fx[f"vote{si}"], equal_nan=True)
    assert np.array_equal(bo.box_vote(fx["lone"], fx["far"]), fx["vote_lone"], equal_nan=True)      # 0 / 0: nan box, score kept


def test_box_overlaps_known_answers_and_argument_checks():
    from airv2x_perception_amd.opencood_iface import box_overlaps as bo
    a = np.float32([[0, 0, 9, 9], [0, 0, 4, 9]])
    b = np.float32([[5, 0, 14, 9], [20, 20, 21, 21]])
    ov = bo.bbox_overlaps(a, b)               # "+1" pixel convention: 10 x 10 boxes, intersection 5 x 10
    assert ov.shape == (2, 2) and ov.dtype == np.float32
    assert abs(ov[0, 0] - 50.0 / 150.0) < 1e-7 and ov[1, 0] == 0.0 and ov[0, 1] == 0.0
    assert abs(bo.bbox_intersections(a, b)[0, 0] - 0.5) < 1e-7
    assert bo.bbox_overlaps(np.zeros((0, 4), np.float32), b).shape == (0, 2)
    with pytest.raises(ValueError):
        bo.bbox_overlaps(a.astype(np.float64), b)          # the typed Cython signature refuses float64 too
    v = bo.box_vote(np.float32([[0, 0, 9, 9, 0.9]]), np.float32([[0, 0, 9, 9, 0.9], [1, 1, 10, 10, 0.3], [50, 50, 60, 60, 1.0]]))
    assert np.allclose(v[0, :4], (0.9 * np.float32([0, 0, 9, 9]) + 0.3 * np.float32([1, 1, 10, 10])) / 1.2, atol=1e-6) and v[0, 4] == np.float32(0.9)


# ------------------------------------------------------------------------------------------------------------------ roiaware_pool3d_cuda (host entry)
def test_points_in_boxes_cpu_matches_the_restatement_and_known_answers():
    from airv2x_perception_amd.opencood_iface import roiaware_pool3d_cuda as rp
    g = np.random.default_rng(5)
    boxes = _boxes3d(g, 12)
    pts = _points_near(g, boxes, 60, 800)
    out = torch.zeros((len(boxes), len(pts)), dtype=torch.int32)
    assert rp.points_in_boxes_cpu(torch.from_numpy(boxes), torch.from_numpy(pts), out) == 1
    want = ro.points_in_boxes_cpu(boxes, pts)
    assert np.array_equal(out.numpy(), want) and 200 < int(want.sum()) < 2000
    # axis-aligned unit case: margin 1e-2 in x / y (roiaware_pool3d.cpp:128), none in z
    box = torch.tensor([[0, 0, 0, 2, 2, 2, 0.0]])
    p = torch.tensor([[1.005, 0, 0], [1.02, 0, 0], [0, -1.009, 0.999], [0, 0, 1.001], [0.5, 0.5, -1.0]])
    o = torch.zeros((1, 5), dtype=torch.int32)
    rp.points_in_boxes_cpu(box, p, o)
    assert o.tolist() == [[1, 0, 1, 0, 1]]
    # heading: a 4 x 1 box turned by 90 degrees covers y in [-2, 2]
    o2 = torch.zeros((1, 2), dtype=torch.int32)
    rp.points_in_boxes_cpu(torch.tensor([[0, 0, 0, 4, 1, 1, np.pi / 2]]), torch.tensor([[0, 1.8, 0], [1.8, 0, 0]]), o2)
    assert o2.tolist() == [[1, 0]]
    with pytest.raises(TypeError):
        rp.points_in_boxes_cpu(box.double(), p, o)
    with pytest.raises(ValueError):
        rp.points_in_boxes_cpu(box, p, torch.zeros((2, 5), dtype=torch.int32))


def test_empty_inputs_are_no_ops():
    from airv2x_perception_amd.opencood_iface import box_overlaps as bo, roiaware_pool3d_cuda as rp
    out = torch.zeros((0, 7), dtype=torch.int32)
    assert rp.points_in_boxes_cpu(torch.zeros((0, 7)), torch.zeros((7, 3)), out) == 1 and out.numel() == 0
    out = torch.full((3, 0), 5, dtype=torch.int32)
    assert rp.points_in_boxes_cpu(torch.zeros((3, 7)), torch.zeros((0, 3)), out) == 1 and out.shape == (3, 0)
    assert bo.bbox_overlaps(np.zeros((2, 4), np.float32), np.zeros((0, 4), np.float32)).shape == (2, 0)
    assert bo.bbox_intersections(np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32)).shape == (0, 0)
    assert bo.box_vote(np.zeros((0, 5), np.float32), np.zeros((4, 5), np.float32)).shape == (0, 5)
    # a degenerate (zero-extent) box holds only what its margin allows
    o = torch.zeros((1, 2), dtype=torch.int32)
    rp.points_in_boxes_cpu(torch.tensor([[0, 0, 0, 0, 0, 0, 0.0]]), torch.tensor([[0.005, 0.0, 0.0], [0.02, 0.0, 0.0]]), o)
    assert o.tolist() == [[1, 0]]


def test_install_import_shims_registers_the_references_module_names():
    import airv2x_perception_amd.opencood_iface as iface
    names = ("opencood", "opencood.pcdet_utils", "opencood.pcdet_utils.roiaware_pool3d", "opencood.utils")
    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k == "opencood" or k.startswith("opencood.")}
    try:
        for k in saved:
            del sys.modules[k]
        for n in names:                               # stand-ins for the reference's (pure Python) packages on a box without the tree
            m = types.ModuleType(n)
            m.__path__ = []
            sys.modules[n] = m
        done = iface.install_import_shims(force=True)
        assert done == ["opencood.pcdet_utils.roiaware_pool3d.roiaware_pool3d_cuda", "opencood.utils.box_overlaps"]
        from opencood.pcdet_utils.roiaware_pool3d import roiaware_pool3d_cuda          # the reference's own import lines
        from opencood.utils.box_overlaps import bbox_overlaps
        assert all(hasattr(roiaware_pool3d_cuda, f) for f in ("forward", "backward", "points_in_boxes_gpu", "points_in_boxes_cpu"))
        assert bbox_overlaps(np.float32([[0, 0, 1, 1]]), np.float32([[0, 0, 1, 1]]))[0, 0] == 1.0
        assert iface.install_import_shims() == []     # already importable: left alone
    finally:
        for k in [k for k in sys.modules if k == "opencood" or k.startswith("opencood.")]:
            del sys.modules[k]
        sys.modules.update({k: v for k, v in saved.items() if v is not None})


# ------------------------------------------------------------------------------------------------------------------ device entries
@pytest.mark.gpu
def test_points_in_boxes_gpu_first_box_wins():
    from airv2x_perception_amd.opencood_iface import roiaware_pool3d_cuda as rp
    g = np.random.default_rng(8)
    B = 3
    boxes = np.stack([_boxes3d(g, 9, spread=8.0) for _ in range(B)])
    boxes[:, 4] = boxes[:, 2]                                     # an overlapping pair: the lower index must win
    boxes[:, 4, 3:6] *= 1.5
    pts = np.stack([_points_near(g, boxes[b], 40, 500) for b in range(B)])
    out = torch.full((B, pts.shape[1]), -1, dtype=torch.int32, device="cuda")
    assert rp.points_in_boxes_gpu(torch.from_numpy(boxes).cuda(), torch.from_numpy(pts).cuda(), out) == 1
    want = ro.points_in_boxes_gpu(boxes, pts)
    assert np.array_equal(out.cpu().numpy(), want)
    assert (want == 2).any() and (want == -1).any()
    with pytest.raises(TypeError):
        rp.points_in_boxes_gpu(torch.from_numpy(boxes), torch.from_numpy(pts).cuda(), out)       # host boxes


@pytest.mark.gpu
@pytest.mark.parametrize("method,out_size,max_pts", [(0, (4, 3, 2), 8), (1, (4, 3, 2), 8), (0, (7, 7, 7), 128), (1, (2, 2, 1), 3)])
def test_roiaware_pool3d_forward_backward(method, out_size, max_pts):
    from airv2x_perception_amd.opencood_iface import roiaware_pool3d_cuda as rp
    g = np.random.default_rng(11 + method)
    rois = _boxes3d(g, 10, spread=10.0)
    pts = _points_near(g, rois, 70, 600)
    C = 5
    feat = g.normal(0, 1, (len(pts), C)).astype(np.float32)
    feat[::7] = feat[3]                                           # ties: the first maximum is the argmax
    ox, oy, oz = out_size
    n = len(rois)
    pooled = torch.zeros((n, ox, oy, oz, C), device="cuda")
    argmax = torch.zeros((n, ox, oy, oz, C), dtype=torch.int32, device="cuda")
    vox = torch.zeros((n, ox, oy, oz, max_pts), dtype=torch.int32, device="cuda")
    assert rp.forward(torch.from_numpy(rois).cuda(), torch.from_numpy(pts).cuda(), torch.from_numpy(feat).cuda(), argmax, vox, pooled, method) == 1
    p_ref, a_ref, v_ref = ro.roiaware_pool3d_forward(rois, pts, feat, out_size, max_pts, method)
    assert np.array_equal(vox.cpu().numpy(), v_ref), "per-voxel point lists (count + indices in increasing order, capped)"
    assert int(v_ref[..., 0].max()) == max_pts - 1 or max_pts > 100      # the cap is exercised in the small-cap cases
    if method == 0:
        assert np.array_equal(argmax.cpu().numpy(), a_ref)
        assert np.array_equal(pooled.cpu().numpy(), p_ref)
    else:
        assert np.allclose(pooled.cpu().numpy(), p_ref, rtol=0, atol=1e-6)
    gout = g.normal(0, 1, (n, ox, oy, oz, C)).astype(np.float32)
    gin = torch.zeros((len(pts), C), device="cuda")
    assert rp.backward(vox, argmax, torch.from_numpy(gout).cuda(), gin, method) == 1
    want = ro.roiaware_pool3d_backward(v_ref, a_ref, gout, len(pts), method)
    assert np.allclose(gin.cpu().numpy(), want, rtol=0, atol=1e-5)
    assert np.abs(want).sum() > 0


@pytest.mark.gpu
def test_roiaware_pool3d_empty_and_argument_errors():
    from airv2x_perception_amd import _lib
    from airv2x_perception_amd.opencood_iface import roiaware_pool3d_cuda as rp
    z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device="cuda")
    # no point inside: max pooling leaves zeros and argmax -1
    pooled, am, vx = z(1, 2, 2, 2, 3), z(1, 2, 2, 2, 3, dt=torch.int32), z(1, 2, 2, 2, 4, dt=torch.int32)
    rp.forward(torch.tensor([[0, 0, 0, 1, 1, 1, 0.0]], device="cuda"), torch.tensor([[5, 5, 5.0]], device="cuda"), z(1, 3) + 1, am, vx, pooled, 0)
    assert float(pooled.abs().sum()) == 0 and bool((am == -1).all()) and int(vx.sum()) == 0
    with pytest.raises(RuntimeError, match="1..255"):
        rp.forward(z(1, 7), z(4, 3), z(4, 2), z(1, 256, 1, 1, 2, dt=torch.int32), z(1, 256, 1, 1, 4, dt=torch.int32), z(1, 256, 1, 1, 2), 0)
    assert b"1..255" in _lib.load().av2x_last_error()
    # mismatched shapes are refused by the shim before any kernel runs (the extents the kernels index with come from these shapes)
    bad = [
        lambda: rp.forward(z(2, 7), z(4, 3), z(4, 2), z(1, 2, 2, 2, 2, dt=torch.int32), z(1, 2, 2, 2, 4, dt=torch.int32), z(1, 2, 2, 2, 2), 0),   # rois N
        lambda: rp.forward(z(1, 7), z(4, 3), z(5, 2), z(1, 2, 2, 2, 2, dt=torch.int32), z(1, 2, 2, 2, 4, dt=torch.int32), z(1, 2, 2, 2, 2), 0),   # feature rows
        lambda: rp.forward(z(1, 7), z(4, 3), z(4, 2), z(1, 2, 2, 2, 3, dt=torch.int32), z(1, 2, 2, 2, 4, dt=torch.int32), z(1, 2, 2, 2, 2), 0),   # argmax C
        lambda: rp.forward(z(1, 7), z(4, 3), z(4, 2), z(1, 2, 2, 2, 2, dt=torch.int32), z(1, 2, 2, 2, 4, dt=torch.int32), z(1, 2, 2, 1, 2), 0),   # pooled grid
        lambda: rp.backward(z(1, 2, 2, 2, 4, dt=torch.int32), z(1, 2, 2, 2, 2, dt=torch.int32), z(1, 2, 2, 2), z(4, 2), 0),                      # grad_out rank
        lambda: rp.backward(z(1, 2, 2, 2, 4, dt=torch.int32), z(1, 2, 2, 2, 2, dt=torch.int32), z(1, 2, 2, 2, 2), z(4, 3), 0),                   # grad_in C
        lambda: rp.backward(z(1, 2, 2, 2, 4, dt=torch.int32), z(1, 2, 2, 2, 3, dt=torch.int32), z(1, 2, 2, 2, 2), z(4, 2), 0),                   # argmax C
    ]
    for call in bad:
        with pytest.raises(ValueError):
            call()
