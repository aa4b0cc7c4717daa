"""ORACLE (test infrastructure, not product code): the tensor part of the camera lift-splat encoder.

CPU restatement of LiftSplatShootEncoder.create_frustum / get_geometry / voxel_pooling
(models/common_modules/airv2x_encoder.py:94-131, 133-167, 208-275) with QuickCumsum.forward
(utils/camera_utils.py:341-358), gen_dx_bx (:238-245), depth_discretization (:303-315) and Airv2xBase.fuse_bev
(common_modules/airv2x_base_model.py:167-177), as plain functions.
Parity: PINNED by tests/golden/lss_*.npz -- tools/gen_golden.py calls the reference's own (unbound) methods on a
namespace carrying dx / bx / nx / frustum, because the class constructor itself needs EfficientNet weights, torchvision
and a CUDA device.  The image trunk and BevEncode are NOT restated (unpinnable here).
`voxel_pooling_exact` is the same pooling in float64 without the cumsum trick: the yardstick for the accuracy comparison.
"""
from __future__ import annotations

import numpy as np
import torch


def gen_dx_bx(xbound, ybound, zbound):
    dx = torch.Tensor([row[2] for row in [xbound, ybound, zbound]])
    bx = torch.Tensor([row[0] + row[2] / 2.0 for row in [xbound, ybound, zbound]])
    nx = torch.LongTensor([int((row[1] - row[0]) / row[2] + 0.5) for row in [xbound, ybound, zbound]])
    return dx, bx, nx


def depth_discretization(depth_min, depth_max, num_bins, mode):
    if mode == "UD":
        return depth_min + (depth_max - depth_min) / num_bins * np.arange(num_bins)
    if mode == "LID":
        bin_size = 2 * (depth_max - depth_min) / (num_bins * (1 + num_bins))
        return depth_min + bin_size * (np.arange(num_bins) * np.arange(1, 1 + num_bins)) / 2
    raise NotImplementedError(mode)


def create_frustum(grid_conf, data_aug_conf, downsample):
    ogfH, ogfW = data_aug_conf["final_dim"]
    fH, fW = ogfH // downsample, ogfW // downsample
    ds = torch.tensor(depth_discretization(*grid_conf["ddiscr"], grid_conf["mode"]), dtype=torch.float).view(-1, 1, 1).expand(-1, fH, fW)
    D = ds.shape[0]
    xs = torch.linspace(0, ogfW - 1, fW, dtype=torch.float).view(1, 1, fW).expand(D, fH, fW)
    ys = torch.linspace(0, ogfH - 1, fH, dtype=torch.float).view(1, fH, 1).expand(D, fH, fW)
    return torch.stack((xs, ys, ds), -1)


def get_geometry(frustum, rots, trans, intrins, post_rots, post_trans):
    B, N, _ = trans.shape
    points = frustum - post_trans.view(B, N, 1, 1, 1, 3)
    points = torch.inverse(post_rots).view(B, N, 1, 1, 1, 3, 3).matmul(points.unsqueeze(-1))
    points = torch.cat((points[:, :, :, :, :, :2] * points[:, :, :, :, :, 2:3], points[:, :, :, :, :, 2:3]), 5)
    combine = rots.matmul(torch.inverse(intrins))
    points = combine.view(B, N, 1, 1, 1, 3, 3).matmul(points).squeeze(-1)
    points += trans.view(B, N, 1, 1, 1, 3)
    return points


def voxel_indices(geom, dx, bx, nx, B):
    """:227-246 -> (voxel coords (Nprime, 4) [x, y, z, b], kept mask)."""
    n = geom.numel() // 3
    g = ((geom - (bx - dx / 2.0)) / dx).long().view(n, 3)
    batch_ix = torch.cat([torch.full([n // B, 1], ix, dtype=torch.long) for ix in range(B)])
    g = torch.cat((g, batch_ix), 1)
    kept = ((g[:, 0] >= 0) & (g[:, 0] < nx[0]) & (g[:, 1] >= 0) & (g[:, 1] < nx[1]) & (g[:, 2] >= 0) & (g[:, 2] < nx[2]))
    return g, kept


def voxel_pooling(geom, x, dx, bx, nx):
    """:208-275 as written (sort by rank, running fp32 sum, differences at the voxel boundaries, scatter)."""
    B, N, D, H, W, C = x.shape
    x = x.reshape(B * N * D * H * W, C)
    g, kept = voxel_indices(geom, dx, bx, nx, B)
    x, g = x[kept], g[kept]
    ranks = g[:, 0] * (nx[1] * nx[2] * B) + g[:, 1] * (nx[2] * B) + g[:, 2] * B + g[:, 3]
    sorts = ranks.argsort()
    x, g, ranks = x[sorts], g[sorts], ranks[sorts]
    x = x.cumsum(0)
    k = torch.ones(x.shape[0], dtype=torch.bool)
    k[:-1] = ranks[1:] != ranks[:-1]
    x, g = x[k], g[k]
    x = torch.cat((x[:1], x[1:] - x[:-1]))
    final = torch.zeros((B, C, int(nx[2]), int(nx[1]), int(nx[0])))
    final[g[:, 3], :, g[:, 2], g[:, 1], g[:, 0]] = x
    return torch.cat(final.unbind(dim=2), 1)


def voxel_pooling_exact(geom, x, dx, bx, nx):
    """The same pooling, every voxel summed on its own in float64 (no running sum): what both fp32 forms approximate."""
    B, N, D, H, W, C = x.shape
    x = x.reshape(-1, C).double()
    g, kept = voxel_indices(geom, dx, bx, nx, B)
    x, g = x[kept], g[kept]
    lin = ((g[:, 3] * int(nx[2]) + g[:, 2]) * int(nx[1]) + g[:, 1]) * int(nx[0]) + g[:, 0]
    acc = torch.zeros(B * int(nx[2]) * int(nx[1]) * int(nx[0]), C, dtype=torch.float64)
    acc.index_add_(0, lin, x)
    final = acc.view(B, int(nx[2]), int(nx[1]), int(nx[0]), C).permute(0, 4, 1, 2, 3)
    return torch.cat(final.unbind(dim=2), 1)


def fuse_bev(spatial_features_list):
    return torch.mean(torch.stack(list(spatial_features_list), dim=0), dim=0)
