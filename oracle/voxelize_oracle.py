"""ORACLE (test infrastructure, not product code): pillar voxelizer.

The reference delegates voxelization to a third-party dependency that is NOT
vendored under /root/reference: ``spconv.utils.Point2VoxelCPU3d`` (spconv 2.x,
``pip install spconv-cu113`` per doc/INSTALL.md:18-27; cumm for tensorview),
called from data_utils/pre_processor/sp_voxel_preprocessor.py:36-72,93-110.
No wheel is installable here, so **parity with the real spconv binary is
UNPINNED**; this file restates spconv's published CPU algorithm
(``Point2VoxelCPU::point_to_voxel``: a single pass over the points in input
order with a dense coor->voxel-index table) and the observable contract at the
reference's call sites:

  * float32 arithmetic ``c = floor((p[j] - range_min[j]) / voxel_size[j])``,
    a point is dropped if any c is outside [0, grid[j]);
  * a new voxel is created at the first point that falls in it, while fewer
    than ``max_voxels`` exist (points of later voxels are dropped);
  * a voxel keeps its first ``max_points`` points in input order;
  * outputs ``voxels (M,max_points,4) f32`` zero padded, ``coordinates (M,3)
    i32`` in **z,y,x** order, ``num_points_per_voxel (M,) i32``; voxel order =
    first-appearance order.

Downstream (PillarVFE + scatter) is invariant to voxel order, so the only
parity-relevant choices are which points/voxels survive the two caps.
"""
from __future__ import annotations

import numpy as np


def grid_size(lidar_range, voxel_size):
    r = np.asarray(lidar_range, dtype=np.float64)
    v = np.asarray(voxel_size, dtype=np.float64)
    return np.round((r[3:6] - r[0:3]) / v).astype(np.int64)


def points_to_voxels(points, lidar_range, voxel_size, max_points=32, max_voxels=70000):
    """Vectorised equivalent of the sequential spconv pass (see module docstring)."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    n, nf = pts.shape
    rmin = np.asarray(lidar_range[:3], dtype=np.float32)
    vs = np.asarray(voxel_size, dtype=np.float32)
    gs = grid_size(lidar_range, voxel_size)
    c = np.floor((pts[:, :3] - rmin) / vs)  # float32 throughout
    valid = np.all((c >= 0) & (c < gs.astype(np.float32)), axis=1)
    ci = c.astype(np.int64)
    lin = (ci[:, 2] * gs[1] + ci[:, 1]) * gs[0] + ci[:, 0]
    idx = np.nonzero(valid)[0]
    lin_v = lin[idx]
    if idx.size == 0:
        return (np.zeros((0, max_points, nf), np.float32), np.zeros((0, 3), np.int32), np.zeros((0,), np.int32))
    uniq, first, inv = np.unique(lin_v, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")          # voxels by first appearance
    rank_of_uniq = np.empty_like(order)
    rank_of_uniq[order] = np.arange(order.size)
    vrank = rank_of_uniq[inv]                          # per valid point: voxel rank
    keep_v = vrank < max_voxels
    idx, vrank = idx[keep_v], vrank[keep_v]
    m = int(min(order.size, max_voxels))
    srt = np.argsort(vrank, kind="stable")             # group by voxel, input order inside
    idx_s, vr_s = idx[srt], vrank[srt]
    starts = np.searchsorted(vr_s, np.arange(m), side="left")
    pos = np.arange(idx_s.size) - starts[vr_s]
    keep_p = pos < max_points
    voxels = np.zeros((m, max_points, nf), dtype=np.float32)
    voxels[vr_s[keep_p], pos[keep_p]] = pts[idx_s[keep_p]]
    counts = np.bincount(vr_s, minlength=m)
    num = np.minimum(counts, max_points).astype(np.int32)
    # coordinates of each voxel (z,y,x) from its first point
    fp = np.empty(m, dtype=np.int64)
    fp[vrank[::-1]] = idx[::-1]                        # smallest index wins (reverse overwrite)
    coords = np.stack([ci[fp, 2], ci[fp, 1], ci[fp, 0]], axis=1).astype(np.int32)
    return voxels, coords, num


def points_to_voxels_sequential(points, lidar_range, voxel_size, max_points=32, max_voxels=70000):
    """Literal single-pass form (slow, pure Python) used to cross-check the
    vectorised version on small inputs."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    rmin = np.asarray(lidar_range[:3], dtype=np.float32)
    vs = np.asarray(voxel_size, dtype=np.float32)
    gs = grid_size(lidar_range, voxel_size)
    table = {}
    voxels, coords, num = [], [], []
    for i in range(pts.shape[0]):
        c = np.floor((pts[i, :3] - rmin) / vs)
        if np.any(c < 0) or np.any(c >= gs):
            continue
        key = (int(c[2]), int(c[1]), int(c[0]))
        v = table.get(key, -1)
        if v == -1:
            if len(voxels) >= max_voxels:
                continue
            v = len(voxels)
            table[key] = v
            voxels.append(np.zeros((max_points, pts.shape[1]), np.float32))
            coords.append(key)
            num.append(0)
        if num[v] < max_points:
            voxels[v][num[v]] = pts[i]
            num[v] += 1
    if not voxels:
        return (np.zeros((0, max_points, pts.shape[1]), np.float32), np.zeros((0, 3), np.int32),
                np.zeros((0,), np.int32))
    return np.stack(voxels), np.asarray(coords, np.int32), np.asarray(num, np.int32)


def mask_points_by_range(points, limit_range):
    """utils/pcd_utils.py:136-165 — strict inequalities on all six faces."""
    p = points
    m = ((p[:, 0] > limit_range[0]) & (p[:, 0] < limit_range[3]) & (p[:, 1] > limit_range[1])
         & (p[:, 1] < limit_range[4]) & (p[:, 2] > limit_range[2]) & (p[:, 2] < limit_range[5]))
    return p[m]


def mask_ego_points(points):
    """utils/pcd_utils.py:168-190 — drop the ego vehicle's own returns."""
    p = points
    m = (p[:, 0] >= -1.95) & (p[:, 0] <= 2.95) & (p[:, 1] >= -1.1) & (p[:, 1] <= 1.1)
    return p[np.logical_not(m)]


def project_points(points_xyz, transformation_matrix):
    """utils/box_utils.py:1038-1067 (project_points_by_matrix_torch) in fp32 exactly as torch evaluates the
    einsum on the reference's CPU path: one rounded product, then fused multiply-adds over k = 1, 2, 3
    (pinned bit-for-bit by tests/golden/points_small.npz).  float64 holds every fp32 product exactly, so
    fl32(a*b + c) computed in float64 and rounded once IS the fp32 FMA."""
    p = np.asarray(points_xyz, np.float32).astype(np.float64)
    T = np.asarray(transformation_matrix, np.float32).astype(np.float64)
    out = np.empty((p.shape[0], 3), np.float32)
    for j in range(3):
        acc = (p[:, 0] * T[j, 0]).astype(np.float32)
        acc = (p[:, 1] * T[j, 1] + acc.astype(np.float64)).astype(np.float32)
        acc = (p[:, 2] * T[j, 2] + acc.astype(np.float64)).astype(np.float32)
        acc = (T[j, 3] + acc.astype(np.float64)).astype(np.float32)
        out[:, j] = acc
    return out


def prepare_points(points, lidar_range, transformation_matrix=None, mask_ego=True, perm=None):
    """datasets/airv2x/intermediate_fusion_dataset.py:591-603: shuffle -> mask_ego_points -> project (proj_first)
    -> mask_points_by_range.  ``perm`` stands for np.random.permutation (pcd_utils.py:193-197)."""
    p = np.asarray(points, np.float32)
    if perm is not None:
        p = p[np.asarray(perm)]
    if mask_ego:
        p = mask_ego_points(p)
    if transformation_matrix is not None:
        p = p.copy()
        p[:, :3] = project_points(p[:, :3], transformation_matrix)
    return mask_points_by_range(p, lidar_range)
