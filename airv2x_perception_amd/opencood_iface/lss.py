"""Camera lift-splat on the device: the tensor part of ``LiftSplatShootEncoder``
(models/common_modules/airv2x_encoder.py:31-336) -- ``create_frustum`` (:94-131), ``get_geometry`` (:133-167),
``voxel_pooling`` (:208-275) -- and the modality mean ``Airv2xBase.fuse_bev`` (airv2x_base_model.py:167-177).

The image trunk (EfficientNet-b0 / ResNet) and ``BevEncode`` (torchvision ResNet-18 layers) that surround these steps
need weights and packages this image does not have; they are NOT built, so the camera branch does not run end to end
yet (the model classes raise for camera modalities).  What is here is the HBM-bound middle of the encoder as one fused
kernel (``av2x_lss_voxel_pool``): no geometry tensor, no 0.7 M-key sort, no running sum.

Host side = per-frame camera matrices only (3x3 inverses and products in fp32 with the same torch calls as the
reference), exactly like the spatial-correction matrices of V2X-ViT.
"""
from __future__ import annotations

from ctypes import c_float, c_int32, c_void_p

import numpy as np
import torch

from .. import _lib


def gen_dx_bx(xbound, ybound, zbound):
    """utils/camera_utils.py:238-245."""
    dx = torch.Tensor([row[2] for row in [xbound, ybound, zbound]])
    bx = torch.Tensor([row[0] + row[2] / 2.0 for row in [xbound, ybound, zbound]])
    nx = torch.LongTensor([int((row[1] - row[0]) / row[2] + 0.5) for row in [xbound, ybound, zbound]])
    return dx, bx, nx


def depth_discretization(depth_min, depth_max, num_bins, mode):
    """utils/camera_utils.py:303-315 (float64 numpy, as there)."""
    if mode == "UD":
        bin_size = (depth_max - depth_min) / num_bins
        return depth_min + bin_size * np.arange(num_bins)
    if mode == "LID":
        bin_size = 2 * (depth_max - depth_min) / (num_bins * (1 + num_bins))
        return depth_min + bin_size * (np.arange(num_bins) * np.arange(1, 1 + num_bins)) / 2
    raise NotImplementedError(mode)


def create_frustum(grid_conf, data_aug_conf, downsample):
    """airv2x_encoder.py:94-131 -> (D, fH, fW, 3) fp32 host tensor (pixel x, pixel y, depth)."""
    ogfH, ogfW = data_aug_conf["final_dim"]
    fH, fW = ogfH // downsample, ogfW // downsample
    ds = torch.tensor(depth_discretization(*grid_conf["ddiscr"], grid_conf["mode"]), dtype=torch.float).view(-1, 1, 1).expand(-1, fH, fW)
    D = ds.shape[0]
    xs = torch.linspace(0, ogfW - 1, fW, dtype=torch.float).view(1, 1, fW).expand(D, fH, fW)
    ys = torch.linspace(0, ogfH - 1, fH, dtype=torch.float).view(1, fH, 1).expand(D, fH, fW)
    return torch.stack((xs, ys, ds), -1)


class LiftSplat:
    """The frustum / grid constants of one agent type's camera encoder (``args[agent_type]["cam"]``) on the device."""

    def __init__(self, cam_args, device="cuda"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("LiftSplat runs on a HIP device only (no CPU path exists)")
        self.lib = _lib.load()
        g = cam_args["grid_conf"]
        self.dx, self.bx, self.nx = gen_dx_bx(g["xbound"], g["ybound"], g["zbound"])
        self.downsample = cam_args["img_downsample"]
        self.camC = cam_args["img_features"]
        fr = create_frustum(g, cam_args["data_aug_conf"], self.downsample)
        self.D, self.fH, self.fW = fr.shape[:3]
        self.frustum_host = fr
        self.frustum = fr.contiguous().view(-1, 3).to(self.device)
        lo = self.bx - self.dx / 2.0                                  # fp32, as `self.bx - self.dx / 2.0` (:227)
        self._lo = (c_float * 3)(*[float(v) for v in lo])
        self._dx = (c_float * 3)(*[float(v) for v in self.dx])
        self._nx = (c_int32 * 3)(*[int(v) for v in self.nx])
        self._ws = {}

    def cam_params(self, rots, trans, intrins, post_rots, post_trans):
        """(B,N,3,3) / (B,N,3) camera tensors -> (B*N, 24) device rows [inverse(post_rots) | post_trans |
        rots @ inverse(intrins) | trans]; the two matrix expressions are the reference's own (:149, :164) in fp32."""
        f = lambda t: t.detach().to("cpu", torch.float32)
        rots, trans, intrins, post_rots, post_trans = f(rots), f(trans), f(intrins), f(post_rots), f(post_trans)
        B, N = trans.shape[:2]
        ipr = torch.inverse(post_rots).reshape(B * N, 9)
        comb = rots.matmul(torch.inverse(intrins)).reshape(B * N, 9)
        rows = torch.cat([ipr, post_trans.reshape(B * N, 3), comb, trans.reshape(B * N, 3)], 1).contiguous()
        return rows.to(self.device), B, N

    def _call(self, x, params, B, N, out, geom):
        st = c_void_p(torch.cuda.current_stream().cuda_stream)
        P = lambda t: c_void_p(t.data_ptr()) if t is not None else None
        ws = None
        if out is not None:
            need = int(self.lib.av2x_lss_pool_workspace_bytes(B, int(self.nx[0]), int(self.nx[1]), int(self.nx[2]), self.camC))
            ws = self._ws.get(need)
            if ws is None:
                ws = self._ws[need] = torch.empty(need, dtype=torch.uint8, device=self.device)
        import ctypes
        _lib.check(self.lib.av2x_lss_voxel_pool(P(x), P(self.frustum), P(params), B, N, self.D * self.fH * self.fW, self.camC,
                                                ctypes.cast(self._lo, c_void_p), ctypes.cast(self._dx, c_void_p),
                                                ctypes.cast(self._nx, c_void_p), P(ws), P(out), P(geom), st), "av2x_lss_voxel_pool")

    @torch.no_grad()
    def get_geometry(self, rots, trans, intrins, post_rots, post_trans):
        """(B, N, D, fH, fW, 3) ego-frame points (airv2x_encoder.py:133-167)."""
        params, B, N = self.cam_params(rots, trans, intrins, post_rots, post_trans)
        geom = torch.empty((B, N, self.D, self.fH, self.fW, 3), dtype=torch.float32, device=self.device)
        self._call(None, params, B, N, None, geom)
        return geom

    @torch.no_grad()
    def voxel_pooling(self, x, rots, trans, intrins, post_rots, post_trans):
        """x (B, N, D, fH, fW, C) lifted features -> BEV (B, ny, nx, nz*C) NHWC (get_geometry + voxel_pooling fused; the
        reference's (B, nz*C, ny, nx) is ``.permute(0, 3, 1, 2)``)."""
        params, B, N = self.cam_params(rots, trans, intrins, post_rots, post_trans)
        if tuple(x.shape) != (B, N, self.D, self.fH, self.fW, self.camC):
            raise ValueError(f"x must be {(B, N, self.D, self.fH, self.fW, self.camC)}, got {tuple(x.shape)}")
        x = x.to(self.device, torch.float32).contiguous()
        out = torch.empty((B, int(self.nx[1]), int(self.nx[0]), int(self.nx[2]) * self.camC), dtype=torch.float32, device=self.device)
        self._call(x, params, B, N, out, None)
        return out


@torch.no_grad()
def fuse_bev(spatial_features_list):
    """Airv2xBase.fuse_bev (airv2x_base_model.py:167-177): the mean over an agent type's modality encoders (camera BEV,
    LiDAR pillar BEV) of equal-shaped maps -- one pass of av2x_agent_mean, no stacked copy."""
    lib = _lib.load()
    maps = [t.contiguous().float() for t in spatial_features_list]
    if len(maps) == 1:
        return maps[0]
    if any(m.shape != maps[0].shape or m.device != maps[0].device for m in maps) or maps[0].device.type != "cuda":
        raise ValueError("fuse_bev: equal-shaped maps on one HIP device")
    stacked = torch.stack(maps, 0)   # plumbing: the kernel takes one (k, elems) buffer
    out = torch.empty_like(maps[0])
    _lib.check(lib.av2x_agent_mean(c_void_p(stacked.data_ptr()), c_void_p(out.data_ptr()), len(maps), out.numel(),
                                   c_void_p(torch.cuda.current_stream().cuda_stream)), "av2x_agent_mean")
    return out
