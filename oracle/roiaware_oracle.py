"""TEST INFRASTRUCTURE (checker only; nothing under airv2x_perception_amd/ imports this).

numpy restatement of the reference's RoI-aware pooling extension, function by function:
  pcdet_utils/roiaware_pool3d/src/roiaware_pool3d.cpp:117-177      lidar_to_local_coords_cpu, check_pt_in_box3d_cpu, points_in_boxes_cpu
  pcdet_utils/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:16-37  lidar_to_local_coords, check_pt_in_box3d (device margin 1e-5)
  ...kernel.cu:40-76    generate_pts_mask_for_box3d   ...:78-113  collect_inside_pts_for_box3d
  ...kernel.cu:116-193  roiaware_maxpool3d / roiaware_avgpool3d   ...:228-284  the two backward kernels   ...:304-327  points_in_boxes_kernel
Parity status: the CUDA kernels cannot run here and roiaware_pool3d.cpp does not link without them (its three launchers live in the
.cu), so this restatement is pinned by known-answer cases (tests/test_native_shims.py), not by outputs of the reference: "parity
unpinned" for this module.  (box_overlaps IS pinned: tests/golden/box_overlaps_pin.npz holds outputs of the reference's own compiled .pyx.)"""
from __future__ import annotations

import numpy as np


def _in_box(pts, box, margin):
    """check_pt_in_box3d(_cpu): float32 rotation, the extent comparisons in double."""
    pts = np.asarray(pts, np.float32)
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    cx, cy, cz, dx, dy, dz, rz = (np.float32(v) for v in box)
    zin = ~(np.abs(z - cz).astype(np.float64) > np.float64(dz) / 2.0)
    cosa, sina = np.cos(-rz, dtype=np.float32), np.sin(-rz, dtype=np.float32)
    sx, sy = (x - cx).astype(np.float32), (y - cy).astype(np.float32)
    lx = (sx * cosa + sy * (-sina)).astype(np.float32)
    ly = (sx * sina + sy * cosa).astype(np.float32)
    m = np.float64(np.float32(margin))
    ins = (np.abs(lx).astype(np.float64) < np.float64(dx) / 2.0 + m) & (np.abs(ly).astype(np.float64) < np.float64(dy) / 2.0 + m)
    return zin & ins, lx, ly


def points_in_boxes_cpu(boxes, pts):
    """roiaware_pool3d.cpp:143-177 -> (N, P) int32."""
    boxes, pts = np.asarray(boxes, np.float32), np.asarray(pts, np.float32)
    out = np.zeros((len(boxes), len(pts)), np.int32)
    for i, b in enumerate(boxes):
        out[i] = _in_box(pts, b, 1e-2)[0]
    return out


def points_in_boxes_gpu(boxes, pts):
    """roiaware_pool3d_kernel.cu:304-327: boxes (B, N, 7), pts (B, P, 3) -> (B, P) int32, -1 = background, else the FIRST box."""
    boxes, pts = np.asarray(boxes, np.float32), np.asarray(pts, np.float32)
    out = np.full(pts.shape[:2], -1, np.int32)
    for b in range(len(pts)):
        for k in range(boxes.shape[1] - 1, -1, -1):        # later boxes first so that the first one wins
            out[b][_in_box(pts[b], boxes[b, k], 1e-5)[0]] = k
    return out


def roiaware_pool3d_forward(rois, pts, feat, out_size, max_pts, method):
    """-> pooled (N, ox, oy, oz, C) float32, argmax (same, int32; max pooling) / None, pts_idx_of_voxels (N, ox, oy, oz, max_pts) int32."""
    rois, pts, feat = np.asarray(rois, np.float32), np.asarray(pts, np.float32), np.asarray(feat, np.float32)
    ox, oy, oz = out_size
    n, c = len(rois), feat.shape[1]
    vox = np.zeros((n, ox, oy, oz, max_pts), np.int32)
    pooled = np.zeros((n, ox, oy, oz, c), np.float32)
    argmax = np.zeros((n, ox, oy, oz, c), np.int32)
    for b, r in enumerate(rois):
        ins, lx, ly = _in_box(pts, r, 1e-5)
        dx, dy, dz = np.float32(r[3]), np.float32(r[4]), np.float32(r[5])
        lz = (pts[:, 2] - np.float32(r[2])).astype(np.float32)
        xr, yr, zr = np.float32(dx / np.float32(ox)), np.float32(dy / np.float32(oy)), np.float32(dz / np.float32(oz))

        def idx(l, d, res, o):      # unsigned(int(.)) then min(max(., 0), o - 1) on unsigned values (:62-68)
            v = np.trunc(((l + d / np.float32(2)).astype(np.float32) / res).astype(np.float32)).astype(np.int64)
            v = np.where(v < 0, v + (1 << 32), v)
            return np.minimum(v, o - 1)
        xi, yi, zi = idx(lx, dx, xr, ox), idx(ly, dy, yr, oy), idx(lz, dz, zr, oz)
        for k in np.nonzero(ins)[0]:                    # increasing point index (:92-104)
            cell = vox[b, xi[k], yi[k], zi[k]]
            if cell[0] < max_pts - 1:
                cell[cell[0] + 1] = k
                cell[0] += 1
        for x in range(ox):
            for y in range(oy):
                for z in range(oz):
                    cell = vox[b, x, y, z]
                    ids = cell[1:1 + cell[0]]
                    if method == 0:
                        if len(ids):
                            f = feat[ids]
                            am = f.argmax(0)           # the first maximum (strict >)
                            pooled[b, x, y, z] = f[am, np.arange(c)]
                            argmax[b, x, y, z] = ids[am]
                        else:
                            argmax[b, x, y, z] = -1
                    elif len(ids):
                        s = np.zeros(c, np.float32)
                        for i in ids:                  # sequential float32 sum (:178-182)
                            s = (s + feat[i]).astype(np.float32)
                        pooled[b, x, y, z] = s / np.float32(len(ids))
    return pooled, (argmax if method == 0 else None), vox


def roiaware_pool3d_backward(vox, argmax, grad_out, n_pts, method):
    """-> grad_in (P, C) float64 (the device accumulates float32 atomics in an unspecified order: compare with a tolerance)."""
    n, ox, oy, oz, c = grad_out.shape
    gin = np.zeros((n_pts, c), np.float64)
    for b in range(n):
        for x in range(ox):
            for y in range(oy):
                for z in range(oz):
                    if method == 0:
                        for ch in range(c):
                            a = argmax[b, x, y, z, ch]
                            if a != -1:
                                gin[a, ch] += grad_out[b, x, y, z, ch]
                    else:
                        cell = vox[b, x, y, z]
                        g = np.float32(1) / np.float32(max(float(cell[0]), 1.0))
                        for i in cell[1:1 + cell[0]]:
                            gin[i] += (grad_out[b, x, y, z] * g).astype(np.float32)
    return gin
