"""Host-side matrix chain of the reference's spatial-transform warp
(models/common_modules/torch_transformation_utils.py): 4x4 correction -> pixel-level 2x3
(get_discretized_transformation_matrix :116-143) -> rotation about the image centre + translation
(get_rotation_matrix2d / get_transformation_matrix :265-308) -> normalised, inverted 2x3 ``theta``
that ``warp_affine`` hands to ``F.affine_grid`` (:337-381, normalize_homography :226-262).
These are a handful of fp32 3x3 products per agent (host logic); the sampling itself runs in
``av2x_warp_affine`` / ``av2x_roi_mask`` on the device.  Same signatures as the reference's
``warp_affine`` for NCHW tensors are provided by ``warp_affine`` below.
"""
from __future__ import annotations

from ctypes import c_void_p

import numpy as np
import torch

from .. import _lib


def discretized_matrix(scm, discrete_ratio, downsample_rate):
    m = np.asarray(scm, dtype=np.float64)[..., [0, 1], :][..., [0, 1, 3]].copy()
    m[..., -1] = m[..., -1] / (discrete_ratio * downsample_rate)
    return m.astype(np.float32)


def transformation_matrix(M, dsize):
    H, W = dsize
    M = np.asarray(M, dtype=np.float32)
    B = M.shape[0]
    eye = lambda: np.tile(np.eye(3, dtype=np.float32), (B, 1, 1))
    sh, shi, rot = eye(), eye(), eye()
    sh[:, 0, 2], sh[:, 1, 2] = np.float32(W / 2), np.float32(H / 2)
    shi[:, 0, 2], shi[:, 1, 2] = -np.float32(W / 2), -np.float32(H / 2)
    rot[:, :2, :2] = M[:, :2, :2]
    T = np.matmul(np.matmul(sh, rot), shi)[:, :2, :].copy()
    T[..., 2] += M[..., 2]
    return T


def _norm_pix(h, w):
    t = np.array([[1.0, 0.0, -1.0], [0.0, 1.0, -1.0], [0.0, 0.0, 1.0]], dtype=np.float32)
    t[0, 0] = t[0, 0] * np.float32(2.0) / np.float32(1e-14 if w == 1 else w - 1.0)
    t[1, 1] = t[1, 1] * np.float32(2.0) / np.float32(1e-14 if h == 1 else h - 1.0)
    return t[None]


def affine_theta(M, src_hw, dsize):
    """(B,2,3) pixel-space affine -> (B,2,3) fp32 theta of F.affine_grid (align_corners=True)."""
    M = np.asarray(M, dtype=np.float32)
    B = M.shape[0]
    H3 = np.zeros((B, 3, 3), dtype=np.float32)
    H3[:, :2, :] = M
    H3[:, 2, 2] = 1.0
    sn, dn = _norm_pix(*src_hw), _norm_pix(*dsize)
    d = np.matmul(dn, np.matmul(H3, np.linalg.inv(sn).astype(np.float32))).astype(np.float32)
    return np.linalg.inv(d).astype(np.float32)[:, :2, :].copy()


def warp_affine(src, M, dsize, mode="bilinear", padding_mode="zeros", align_corners=True):
    """Reference signature (torch_transformation_utils.py:337): src (B,C,H,W) CUDA, M (B,2,3) -> (B,C,H,W)."""
    if mode != "bilinear" or padding_mode != "zeros" or not align_corners:
        raise NotImplementedError("only bilinear / zeros / align_corners=True (the mode the AirV2X path uses)")
    B, C, H, W = src.shape
    if tuple(dsize) != (H, W):
        raise NotImplementedError("dsize must equal the source size")
    lib = _lib.load()
    theta = torch.from_numpy(affine_theta(M.detach().cpu().numpy() if isinstance(M, torch.Tensor) else M, (H, W), dsize)).to(src.device)
    x = src.permute(0, 2, 3, 1).contiguous().float()
    y = torch.empty_like(x)
    _lib.check(lib.av2x_warp_affine(c_void_p(x.data_ptr()), c_void_p(theta.data_ptr()), c_void_p(y.data_ptr()), B, H, W, C,
                                    c_void_p(torch.cuda.current_stream().cuda_stream)), "av2x_warp_affine")
    return y.permute(0, 3, 1, 2).contiguous()


def warp_affine_simple(src, M, dsize, mode="bilinear", padding_mode="zeros", align_corners=False):
    """Reference signature (torch_transformation_utils.py:327-334): ``M`` (B,2,3) is ALREADY the normalised theta of
    F.affine_grid; the reference ignores ``mode`` / ``padding_mode`` here (grid_sample defaults: bilinear, zeros)."""
    B, C, H, W = src.shape
    if tuple(dsize) != (H, W):
        raise NotImplementedError("dsize must equal the source size")
    lib = _lib.load()
    theta = torch.as_tensor(M, dtype=torch.float32).to(src.device).contiguous()
    x = src.permute(0, 2, 3, 1).contiguous().float()
    y = torch.empty_like(x)
    fn = lib.av2x_warp_affine if align_corners else lib.av2x_warp_affine_simple
    _lib.check(fn(c_void_p(x.data_ptr()), c_void_p(theta.data_ptr()), c_void_p(y.data_ptr()), B, H, W, C,
                  c_void_p(torch.cuda.current_stream().cuda_stream)), "av2x_warp_affine_simple")
    return y.permute(0, 3, 1, 2).contiguous()
