"""GPU counterpart of ``SpVoxelPreprocessor.preprocess``
(data_utils/pre_processor/sp_voxel_preprocessor.py:74-116): points -> (voxel_features,
voxel_coords z,y,x, voxel_num_points), computed by av2x_voxelize on the device.

``range_filter=True`` first applies the strict range crop the dataset performs before
voxelising (utils/pcd_utils.py:136-165, called at intermediate_fusion_dataset.py:598-603);
it is an order-preserving boolean mask (torch indexing = plumbing, no arithmetic).
"""
from __future__ import annotations

import ctypes
from ctypes import c_float, c_void_p

import numpy as np
import torch

from .. import _lib


def prepare_points(points, lidar_range, transformation_matrix=None, mask_ego=True, perm=None):
    """The per-agent point preparation of intermediate_fusion_dataset.py:591-603 on the device (av2x_prepare_points):
    [shuffle by ``perm``] -> mask_ego_points -> project to the ego frame -> mask_points_by_range.
    points (P,4) fp32 CUDA tensor; returns the surviving points (P',4), order preserved (one host read of P')."""
    lib = _lib.load()
    if points.device.type != "cuda":
        raise RuntimeError("prepare_points runs on a HIP device only")
    pts = points.contiguous().float()
    n = pts.shape[0]
    dev = pts.device
    out = torch.empty((n, 4), dtype=torch.float32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = torch.empty(int(lib.av2x_prepare_points_workspace_bytes(n)), dtype=torch.uint8, device=dev)
    r6 = (c_float * 6)(*[float(v) for v in lidar_range])
    t16 = None
    if transformation_matrix is not None:
        t16 = (c_float * 16)(*np.asarray(transformation_matrix, dtype=np.float32).reshape(-1).tolist())
    pm = perm.to(device=dev, dtype=torch.int32).contiguous() if perm is not None else None
    _lib.check(lib.av2x_prepare_points(c_void_p(pts.data_ptr()), c_void_p(pm.data_ptr()) if pm is not None else None, n,
                                       ctypes.cast(t16, c_void_p) if t16 is not None else None, ctypes.cast(r6, c_void_p),
                                       1 if mask_ego else 0, c_void_p(ws.data_ptr()), c_void_p(out.data_ptr()),
                                       c_void_p(cnt.data_ptr()), c_void_p(torch.cuda.current_stream().cuda_stream)),
               "av2x_prepare_points")
    return out[:int(cnt.item())]


def voxelize_points(points, lidar_range, voxel_size, max_points=32, max_voxels=70000, range_filter=False):
    """points: (P,4) fp32 CUDA tensor -> (voxels (M,32,4), coords (M,3) i32 zyx, num (M,) i32) on the device.
    Reads M back to the host once (the reference contract has exact-shaped tensors)."""
    lib = _lib.load()
    if points.device.type != "cuda":
        raise RuntimeError("voxelize_points runs on a HIP device only")
    pts = points.contiguous().float()
    if range_filter:
        pts = prepare_points(pts, lidar_range, None, mask_ego=False)
    n = pts.shape[0]
    if n == 0:
        # dummy points of the reference's empty-cloud branch (sp_voxel_preprocessor.py:80-90)
        d = np.array([[0, 0, 0, 0], [-0.218277, -11.13425732, -80.05884552, 1.230595649e-38]], dtype=np.float32)
        pts = torch.from_numpy(d).to(points.device)
        n = 2
    grid = np.round((np.asarray(lidar_range[3:6], np.float64) - np.asarray(lidar_range[0:3], np.float64))
                    / np.asarray(voxel_size, np.float64)).astype(np.int64)
    cap = min(n, max_voxels)
    dev = pts.device
    ws = torch.empty(int(lib.av2x_voxelize_workspace_bytes(n, int(grid[0]), int(grid[1]), int(grid[2]))),
                     dtype=torch.uint8, device=dev)
    voxels = torch.empty((cap, max_points, 4), dtype=torch.float32, device=dev)
    coords = torch.empty((cap, 3), dtype=torch.int32, device=dev)
    num = torch.empty((cap,), dtype=torch.int32, device=dev)
    m_dev = torch.zeros(1, dtype=torch.int32, device=dev)
    r6 = (c_float * 6)(*[float(v) for v in lidar_range])
    v3 = (c_float * 3)(*[float(v) for v in voxel_size])
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.av2x_voxelize(c_void_p(pts.data_ptr()), n, ctypes.cast(r6, c_void_p), ctypes.cast(v3, c_void_p),
                                 max_points, max_voxels, c_void_p(ws.data_ptr()), c_void_p(voxels.data_ptr()),
                                 c_void_p(coords.data_ptr()), c_void_p(num.data_ptr()), c_void_p(m_dev.data_ptr()), st),
               "av2x_voxelize")
    m = int(m_dev.item())
    return voxels[:m], coords[:m], num[:m]


def voxelize_frame(clouds, lidar_range, voxel_size, poses=None, mask_ego=True, perms=None, max_points=32, max_voxels=70000):
    """Whole-frame front end: for every agent's raw (P,4) CUDA cloud run av2x_prepare_voxelize (ego mask -> projection by
    ``poses[i]`` -> range crop -> pillar voxelizer, one fused pass) back to back, then read ALL pillar counts with ONE
    host read-back and return the exact-shaped (voxels (M,32,4), coords (M,3), num (M,)) triples of the reference's
    input contract.  Equivalent to prepare_points + voxelize_points per agent (tests/test_gpu_voxelizer.py)."""
    lib = _lib.load()
    grid = np.round((np.asarray(lidar_range[3:6], np.float64) - np.asarray(lidar_range[0:3], np.float64))
                    / np.asarray(voxel_size, np.float64)).astype(np.int64)
    r6 = (c_float * 6)(*[float(v) for v in lidar_range])
    v3 = (c_float * 3)(*[float(v) for v in voxel_size])
    n_agents = len(clouds)
    dev = clouds[0].device
    if dev.type != "cuda":
        raise RuntimeError("voxelize_frame runs on a HIP device only")
    counts = torch.zeros(n_agents, dtype=torch.int32, device=dev)
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    pend = []
    for i, pts in enumerate(clouds):
        pts = pts.contiguous().float()
        n = pts.shape[0]
        cap = max(1, min(n, max_voxels))
        ws = torch.empty(int(lib.av2x_voxelize_workspace_bytes(max(n, 1), int(grid[0]), int(grid[1]), int(grid[2]))),
                         dtype=torch.uint8, device=dev)
        voxels = torch.empty((cap, max_points, 4), dtype=torch.float32, device=dev)
        coords = torch.empty((cap, 3), dtype=torch.int32, device=dev)
        num = torch.empty((cap,), dtype=torch.int32, device=dev)
        t16 = None
        if poses is not None and poses[i] is not None:
            t16 = (c_float * 16)(*np.asarray(poses[i], dtype=np.float32).reshape(-1).tolist())
        pm = perms[i].to(device=dev, dtype=torch.int32).contiguous() if perms is not None and perms[i] is not None else None
        _lib.check(lib.av2x_prepare_voxelize(c_void_p(pts.data_ptr()), c_void_p(pm.data_ptr()) if pm is not None else None, n,
                                             ctypes.cast(t16, c_void_p) if t16 is not None else None, ctypes.cast(r6, c_void_p),
                                             1 if mask_ego else 0, ctypes.cast(r6, c_void_p), ctypes.cast(v3, c_void_p),
                                             max_points, max_voxels, c_void_p(ws.data_ptr()), c_void_p(voxels.data_ptr()),
                                             c_void_p(coords.data_ptr()), c_void_p(num.data_ptr()),
                                             c_void_p(counts[i:i + 1].data_ptr()), st), "av2x_prepare_voxelize")
        pend.append((voxels, coords, num, ws, pts, pm))
    ms = counts.tolist()                                   # the one host read-back of the frame's front end
    out = []
    for i, (m, (voxels, coords, num, _, pts, pm)) in enumerate(zip(ms, pend)):
        if m == 0:   # empty cloud after the crop: the reference's dummy-point branch (sp_voxel_preprocessor.py:80-90)
            out.append(voxelize_points(torch.zeros((0, 4), device=dev), lidar_range, voxel_size, max_points, max_voxels))
        else:
            out.append((voxels[:m], coords[:m], num[:m]))
    return out


def points_frame(clouds, types, preprocess_params, poses=None, perms=None, mask_ego=True, train=False, **frame_metadata):
    """Model input for ONE collaborative frame straight from raw clouds: ``model(points_frame(...))`` runs point
    preparation, voxelizer, pillar feature net and scatter back to back on the device without any host read-back (the
    (M,32,4) tensors of the reference's input contract never take their exact shape; the pillar counts stay in HBM).

    clouds: list of (P,4) fp32 CUDA tensors in frame order (vehicles, rsus, drones; ego first), types: their agent
    types, preprocess_params: ``hypes["preprocess"]`` (cav_lidar_range, voxel_size, max_points_per_voxel, max_voxel_*),
    poses[i]: 4x4 sensor -> ego transform or None, perms[i]: the shuffle permutation or None, mask_ego: drop the
    ego-vehicle box (intermediate_fusion_dataset.py:591-603).  ``frame_metadata``: the frame-level entries the model reads
    besides the lidar features (``prior_encoding``, ``spatial_correction_matrix``, ``img_pairwise_t_matrix_collab``)."""
    a = preprocess_params["args"]
    pf = {"clouds": list(clouds), "types": list(types), "lidar_range": list(preprocess_params["cav_lidar_range"]),
          "voxel_size": list(a["voxel_size"]), "max_points": int(a["max_points_per_voxel"]),
          "max_voxels": int(a["max_voxel_train"] if train else a["max_voxel_test"]),
          "poses": poses, "perms": perms, "mask_ego": bool(mask_ego)}
    dd = {"points": pf}
    dd.update(frame_metadata)
    return dd


class SpVoxelPreprocessor:
    """Mirror of data_utils/pre_processor/sp_voxel_preprocessor.py:30-175 for pipelines that keep the clouds on the GPU
    (the reference runs spconv's CPU voxelizer inside DataLoader workers).

    Same constructor (``preprocess_params`` = ``hypes["preprocess"]``, ``train``) and methods:
    ``preprocess(pcd)`` -> ``{"voxel_features" (M,32,4), "voxel_coords" (M,3) z,y,x, "voxel_num_points" (M,)}``
    (empty clouds get the reference's two dummy points, :80-90) and ``collate_batch(batch)`` -> the three arrays
    concatenated over the agents with the agent index as leading column of the coordinates (:142-175).  ``pcd`` may be
    a numpy array or a tensor on any device; results are CUDA tensors (``numpy=True`` returns host arrays like the
    reference).  ``collate_batch`` accepts the dict-of-lists form the AirV2X dataset uses AND the list-of-dicts form,
    which raises NameError in the reference."""

    def __init__(self, preprocess_params, train, device="cuda", numpy=False):
        self.params = preprocess_params
        self.train = train
        self.lidar_range = preprocess_params["cav_lidar_range"]
        a = preprocess_params["args"]
        self.voxel_size = a["voxel_size"]
        self.max_points_per_voxel = a["max_points_per_voxel"]
        self.max_voxels = a["max_voxel_train"] if train else a["max_voxel_test"]
        g = (np.array(self.lidar_range[3:6]) - np.array(self.lidar_range[0:3])) / np.array(self.voxel_size)
        self.grid_size = np.round(g).astype(np.int64)
        self.device = torch.device(device)
        self.numpy = numpy

    def preprocess(self, pcd_np):
        pts = torch.as_tensor(pcd_np, dtype=torch.float32).to(self.device)
        if pts.dim() != 2 or pts.shape[1] != 4:
            raise ValueError(f"point cloud must be (P,4), got {tuple(pts.shape)}")
        v, c, n = voxelize_points(pts, self.lidar_range, self.voxel_size, self.max_points_per_voxel, self.max_voxels)
        if self.numpy:
            v, c, n = v.cpu().numpy(), c.cpu().numpy(), n.cpu().numpy()
        return {"voxel_features": v, "voxel_coords": c, "voxel_num_points": n}

    @staticmethod
    def collate_batch(batch):
        if isinstance(batch, (list, tuple)):
            keys = batch[0].keys()
            batch = {k: [s[k] for s in batch] for k in keys}
        elif not isinstance(batch, dict):
            raise TypeError("batch must be list or dict, got {}".format(type(batch)))
        t = lambda a: a if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a))
        feats = [t(a) for a in batch["voxel_features"]]
        nums = [t(a) for a in batch["voxel_num_points"]]
        coords = [t(a) for a in batch["voxel_coords"]]
        with_idx = [torch.cat([torch.full((c.shape[0], 1), i, dtype=c.dtype, device=c.device), c], 1) for i, c in enumerate(coords)]
        return {"voxel_features": torch.cat(feats), "voxel_coords": torch.cat(with_idx), "voxel_num_points": torch.cat(nums)}
