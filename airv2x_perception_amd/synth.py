"""Synthetic configuration, weights and inputs for the AirV2X Where2Comm hot path.

Nothing here comes from the reference at run time: the hypes dictionary below
restates the *keys and values* the hot path consumes from
``opencood/hypes_yaml/airv2x/lidar/det/airv2x_intermediate_where2com.yaml``
(:114-119 voxel/range, :164-248 per-agent lidar blocks, :251-283 backbone /
shrink / where2comm fusion, :286-297 heads) so that bench.py, the tests and
smoke() can build the model on a box where ``/root/reference`` does not exist.

The weight generator is deterministic per state_dict key (numpy PCG64 seeded by
crc32(key)), so golden fixtures only need to store the key->shape manifest and
the seed, not 29 MB of parameters.
"""
from __future__ import annotations

import copy
import zlib
from collections import OrderedDict

import numpy as np
import torch

AGENT_TYPES = ("vehicle", "rsu", "drone")
TYPE_PREFIX = {"vehicle": "veh_models", "rsu": "rsu_models", "drone": "drone_models"}

# default AirV2X geometry (where2com yaml :114-119, :168-169, :198-199, :228-229)
DEFAULT_RANGE = [-140.8, -40.0, -3.0, 140.8, 40.0, 1.0]
DEFAULT_VOXEL = [0.4, 0.4, 4.0]


def default_hypes(lidar_range=None, max_cav=(5, 5, 5)):
    """Return a hypes dict with the same nesting the reference loader produces
    (yaml_utils.py:224-299 ``load_airv2x_params``): ``model.args`` carries the
    per-type ``grid_size`` and ``max_cav_num``; ``postprocess.anchor_args``
    carries W/H/D and vw/vh/vd.

    ``lidar_range`` lets the tests shrink the BEV grid (x/y extents must give a
    grid divisible by 8); the z extents of every agent type stay the reference's.
    """
    r = list(DEFAULT_RANGE if lidar_range is None else lidar_range)
    xy = lambda z0, z1: [r[0], r[1], z0, r[3], r[4], z1]
    per_type = {
        "vehicle": (xy(-3.0, 1.0), [0.4, 0.4, 4.0]),
        "rsu": (xy(-30.0, 30.0), [0.4, 0.4, 60.0]),
        "drone": (xy(-150.0, -6.0), [0.4, 0.4, 144.0]),
    }
    args = {
        "ego_type": "vehicle",
        "collaborators": list(AGENT_TYPES),
        "active_sensors": ["lidar"],
        "max_cav": {"vehicle": max_cav[0], "rsu": max_cav[1], "drone": max_cav[2]},
        "max_cav_num": int(sum(max_cav)),
        "device": "cuda",
        "train": True,
        "proj_first": True,
        "supervise_single": False,
        "backbone_fix": False,
        "modality_fusion": {
            "base_bev_backbone": {
                "layer_nums": [3, 5, 8],
                "layer_strides": [2, 2, 2],
                "num_filters": [64, 128, 256],
                "upsample_strides": [1, 2, 4],
                "num_upsample_filter": [128, 128, 128],
            },
            "shrink_header": {
                "use": True,
                "input_dim": 384,
                "dim": [256],
                "kernal_size": [1],
                "stride": [1],
                "padding": [0],
            },
            "compression": 0,
        },
        "where2com_fusion": {
            "fully": False,
            "voxel_size": list(DEFAULT_VOXEL),
            "downsample_rate": 4,
            "in_channels": 256,
            "multi_scale": True,
            "layer_nums": [3, 5, 8],
            "num_filters": [64, 128, 256],
            "communication": {
                "round": 1,
                "threshold": 0.01,
                "gaussian_smooth": {"k_size": 5, "c_sigma": 1.0},
            },
        },
        "task": "det",
        "head_dim": 256,
        "outC": 256,
        "anchor_number": 2,
        "num_class": 7,
        "cav_range": list(r),
        "obj_head": True,
    }
    for t, (rng, vs) in per_type.items():
        grid = np.round((np.array(rng[3:6]) - np.array(rng[0:3])) / np.array(vs)).astype(np.int64)
        args[t] = {
            "modalities": ["lidar"],
            "lidar": {
                "voxel_size": vs,
                "lidar_range": rng,
                "compression": 0,
                "backbone_fix": False,
                "pillar_vfe": {
                    "use_norm": True,
                    "with_distance": False,
                    "use_absolute_xyz": True,
                    "num_filters": [64],
                },
                "point_pillar_scatter": {"num_features": 64, "grid_size": grid},
            },
        }
    vw, vh, vd = DEFAULT_VOXEL
    hypes = {
        "name": "airv2x_intermediate_where2comm",
        "preprocess": {
            "core_method": "SpVoxelPreprocessor",
            "ego_type": "vehicle",
            "args": {
                "voxel_size": list(DEFAULT_VOXEL),
                "max_points_per_voxel": 32,
                "max_voxel_train": 32000,
                "max_voxel_test": 70000,
            },
            "cav_lidar_range": list(r),
        },
        "postprocess": {
            "core_method": "VoxelPostprocessor",
            "ego_type": "vehicle",
            "anchor_args": {
                "cav_lidar_range": list(r),
                "l": 3.9, "w": 1.6, "h": 1.56, "r": [0, 90],
                "feature_stride": 2, "num": 2,
                "vw": vw, "vh": vh, "vd": vd,
                "W": int(np.ceil((r[3] - r[0]) / vw - 1e-9)),
                "H": int(np.ceil((r[4] - r[1]) / vh - 1e-9)),
                "D": int(np.ceil((r[5] - r[2]) / vd - 1e-9)),
            },
            "target_args": {"pos_threshold": 0.6, "neg_threshold": 0.45,
                            "score_threshold": 0.2, "obj_threshold": 0.2},
            "order": "hwl",
            "max_num": 300,
            "nms_thresh": 0.15,
        },
        "model": {"core_method": "airv2x_where2com", "args": args},
    }
    return hypes


# --------------------------------------------------------------------------
# state_dict manifest (SURVEY.md §8b key contract; verified against the
# reference's own state_dict by tools/gen_golden.py)
# --------------------------------------------------------------------------

def _bn(prefix, c):
    return [
        (prefix + ".weight", (c,), "bn_w"),
        (prefix + ".bias", (c,), "bn_b"),
        (prefix + ".running_mean", (c,), "bn_m"),
        (prefix + ".running_var", (c,), "bn_v"),
        (prefix + ".num_batches_tracked", (), "count"),
    ]


def pfn_param_spec(prefix=""):
    """PillarVFE with one PFN layer (airv2x_pillar_vfe.py:69-81)."""
    p = prefix + "pfn_layers.0"
    return [(p + ".linear.weight", (64, 10), "lin")] + _bn(p + ".norm", 64)


def backbone_param_spec(bb, input_channels=64, prefix=""):
    """BaseBEVBackbone(model_cfg, input_channels) (base_bev_backbone.py:38-105)."""
    spec = []
    cin = input_channels
    for i, (n, c) in enumerate(zip(bb["layer_nums"], bb["num_filters"])):
        # Sequential: 0 ZeroPad, 1 conv, 2 bn, 3 relu, then (conv,bn,relu)*n
        idx = 1
        spec.append((f"{prefix}blocks.{i}.{idx}.weight", (c, cin, 3, 3), "conv"))
        spec += _bn(f"{prefix}blocks.{i}.{idx + 1}", c)
        idx += 3
        for _ in range(n):
            spec.append((f"{prefix}blocks.{i}.{idx}.weight", (c, c, 3, 3), "conv"))
            spec += _bn(f"{prefix}blocks.{i}.{idx + 1}", c)
            idx += 3
        cin = c
    ups, nuf = list(bb.get("upsample_strides", [])), list(bb.get("num_upsample_filter", []))
    for i, (s, cu) in enumerate(zip(ups[:len(bb["layer_nums"])], nuf)):
        c = bb["num_filters"][i]
        if s >= 1:      # ConvTranspose2d(c, cu, s, stride=s)                                   (base_bev_backbone.py:73-86)
            spec.append((f"{prefix}deblocks.{i}.0.weight", (c, cu, int(s), int(s)), "deconv"))
        else:           # down-sampling "deblock": Conv2d(c, cu, k, stride=k), k = round(1 / s)  (:87-105)
            k = int(round(1.0 / s))
            spec.append((f"{prefix}deblocks.{i}.0.weight", (cu, c, k, k), "conv"))
        spec += _bn(f"{prefix}deblocks.{i}.1", cu)
    if len(ups) > len(bb["layer_nums"]):   # one more ConvTranspose2d on the concatenated map    (:107-121)
        c_in, s = sum(nuf), int(ups[-1])
        i = len(bb["layer_nums"])
        spec.append((f"{prefix}deblocks.{i}.0.weight", (c_in, c_in, s, s), "deconv"))
        spec += _bn(f"{prefix}deblocks.{i}.1", c_in)
    return spec


def resnet_backbone_param_spec(bb, prefix="", input_channels=64):
    """ResNetBEVBackbone(model_cfg, input_channels) (common_modules/base_bev_backbone_resnet.py:16-110): ``resnet`` =
    coalign_modules.resblock.ResNetModified(BasicBlock, layer_nums, layer_strides, num_filters, inplanes) -- levels ``layer0``,
    ``layer1``, ... (:182-189), input width ``model_cfg.get("inplanes", input_channels)`` -- then the deblocks of BaseBEVBackbone."""
    spec, cin = [], int(bb.get("inplanes", input_channels))
    for li, (n, st, c) in enumerate(zip(bb["layer_nums"], bb["layer_strides"], bb["num_filters"])):
        for j in range(n):
            q = f"{prefix}resnet.layer{li}.{j}."
            spec += _basic_block_spec(q, cin if j == 0 else c, c, st if j == 0 else 1)
        cin = c
    return spec + [e for e in backbone_param_spec(bb, 64, prefix) if e[0].startswith(prefix + "deblocks.")]


def shrink_param_spec(sh, prefix=""):
    """DownsampleConv(config) (downsample_conv.py:17-31, :40-54)."""
    spec = []
    cin = sh["input_dim"]
    for li, (k, d) in enumerate(zip(sh["kernal_size"], sh["dim"])):
        p = f"{prefix}layers.{li}.double_conv"
        spec.append((p + ".0.weight", (d, cin, k, k), "conv"))
        spec.append((p + ".0.bias", (d,), "bias"))
        spec.append((p + ".2.weight", (d, d, 3, 3), "conv"))
        spec.append((p + ".2.bias", (d,), "bias"))
        cin = d
    return spec


def encoder_param_spec(args):
    """Airv2xBase.init_encoders (airv2x_base_model.py:36-99): per agent type one encoder per entry of ``modalities``, in
    that order -- ``<type>_models.<i>`` is a LiftSplatShootEncoder for "cam", Sequential(PillarVFE, PointPillarScatter)
    for "lidar"."""
    spec = []
    for t in AGENT_TYPES:
        if t not in args["collaborators"]:
            continue
        for i, m in enumerate(args.get(t, {}).get("modalities", ["lidar"])):
            if m == "lidar":
                spec += pfn_param_spec(f"{TYPE_PREFIX[t]}.{i}.0.")
            elif m == "cam":
                spec += lss_param_spec(args[t]["cam"], f"{TYPE_PREFIX[t]}.{i}.")
            else:
                raise NotImplementedError(f"Modality {m} not supported")
    return spec


def where2com_param_spec(args):
    """Ordered (key, shape, kind) manifest of Airv2xWhere2com's state_dict.

    Key layout follows the reference constructors: airv2x_base_model.py:36-99
    (encoders), base_bev_backbone.py:38-105 (blocks/deblocks),
    downsample_conv.py:17-31 (shrink), where2comm_fuse.py:58-62 (gaussian),
    airv2x_where2com.py:59-69 (heads).
    """
    spec = encoder_param_spec(args)
    spec += backbone_param_spec(args["modality_fusion"]["base_bev_backbone"], 64, "backbone.")
    spec += shrink_param_spec(args["modality_fusion"]["shrink_header"], "shrink_conv.")
    spec += compressor_param_spec(256, model_compression(args))       # airv2x_where2com.py:50-52: NaiveCompressor(256, args["compression"])
    ks = args["where2com_fusion"]["communication"]["gaussian_smooth"]["k_size"]
    spec.append(("fusion_net.naive_communication.gaussian_filter.weight", (1, 1, ks, ks), "gauss_w"))
    spec.append(("fusion_net.naive_communication.gaussian_filter.bias", (1,), "gauss_b"))
    A, C, outC = args["anchor_number"], args["num_class"], args["outC"]
    spec.append(("cls_head.weight", (A * C, outC, 1, 1), "head"))
    spec.append(("cls_head.bias", (A * C,), "cls_bias"))
    spec.append(("reg_head.weight", (7 * A, outC, 1, 1), "head"))
    spec.append(("reg_head.bias", (7 * A,), "bias"))
    if args["obj_head"]:
        spec.append(("obj_head.weight", (A, outC, 1, 1), "head"))
        spec.append(("obj_head.bias", (A,), "obj_bias"))
    return spec


def _rng_for(key, seed):
    return np.random.default_rng((zlib.crc32(key.encode()) + 7919 * int(seed)) & 0xFFFFFFFF)


def synthetic_tensor(key, shape, kind, seed=0):
    """Deterministic fp32 values for one parameter.

    Convolutions use a He-uniform scale (var = 2/fan_in) so the signal keeps
    O(1) magnitude through the 19-layer trunk in eval mode (PyTorch's default
    init shrinks it by ~6x per layer, which would make every fp comparison
    vacuous); BatchNorm statistics are randomised so that BN folding bugs show.
    The cls/obj head biases are shifted negative so the where2comm threshold
    (0.01) and the obj>0.2 gate both split the map instead of being all-ones.
    """
    g = _rng_for(key, seed)
    shape = tuple(shape)
    if kind == "count":
        return np.zeros(shape, dtype=np.int64)
    if kind.startswith("relidx:"):
        _, L, ws = kind.split(":")
        return _relative_position_index(int(L), int(ws))
    if kind == "ln_w":
        return g.uniform(0.7, 1.3, size=shape).astype(np.float32)
    if kind == "ln_b":
        return g.uniform(-0.1, 0.1, size=shape).astype(np.float32)
    if kind == "relbias":
        return g.uniform(-1.0, 1.0, size=shape).astype(np.float32)
    if kind == "rel":  # HGT relation matrices (xavier-uniform-like scale)
        b = np.sqrt(6.0 / (shape[-1] + shape[-2]))
        return g.uniform(-b, b, size=shape).astype(np.float32)
    if kind == "rte_table":  # sinusoid table of RelTemporalEncoding (v2xvit_basic.py:46-53)
        n_hid = shape[1]
        pos = np.arange(0.0, shape[0], dtype=np.float32)[:, None]
        div = np.exp(np.arange(0, n_hid, 2, dtype=np.float32) * np.float32(-(np.log(10000.0) / n_hid)))
        t = np.zeros(shape, dtype=np.float32)
        t[:, 0::2] = np.sin(pos * div) / np.float32(np.sqrt(n_hid))
        t[:, 1::2] = np.cos(pos * div) / np.float32(np.sqrt(n_hid))
        return t
    if kind == "att_lin":   # query -> key projection of When2com: small, so that the softmax over agents stays mixed
        b = 0.25 * np.sqrt(6.0 / shape[1])
        return g.uniform(-b, b, size=shape).astype(np.float32)
    if kind == "se_w":    # squeeze-excite 1x1 convs: modest scale so that the sigmoid gate stays in its mixed range
        b = np.sqrt(3.0 / int(np.prod(shape[1:])))
        return g.uniform(-b, b, size=shape).astype(np.float32)
    if kind == "proj":    # MBConv projection (no activation after it, gate ~0.5 before it)
        b = np.sqrt(3.0 * 2.5 / int(np.prod(shape[1:])))
        return g.uniform(-b, b, size=shape).astype(np.float32)
    if kind == "img_head":   # CamEncode.image_head: the lift SUMS every frustum point of a BEV cell (tens near the camera), so the
        b = np.sqrt(3.0 * 0.05 / int(np.prod(shape[1:])))          # per-pixel features are kept small for an O(1) pooled map
        return g.uniform(-b, b, size=shape).astype(np.float32)
    if kind in ("conv", "lin", "head", "deconv"):
        if kind == "lin":
            fan_in = shape[1]
        elif kind == "deconv":  # ConvTranspose2d weight (Cin, Cout, k, k): each output sums Cin terms
            fan_in = shape[0]
        else:
            fan_in = int(np.prod(shape[1:]))
        gain = 4.0 if kind == "head" else 2.0
        b = np.sqrt(3.0 * gain / fan_in)
        w = g.uniform(-b, b, size=shape).astype(np.float32)
        if kind == "lin" and shape == (64, 10):
            # PFN input columns: abs x,y,z,intensity, cluster xyz, centre xyz.  Absolute x/y reach
            # +-140 m, so their weights are scaled down to keep pillar features O(1).
            w *= 4.0 * np.asarray([0.01, 0.02, 0.3, 1.0, 1.0, 1.0, 0.5, 2.0, 2.0, 0.02], dtype=np.float32)
        return w
    if kind == "bn_w":
        return g.uniform(0.7, 1.3, size=shape).astype(np.float32)
    if kind == "bn_b":
        return g.uniform(-0.1, 0.1, size=shape).astype(np.float32)
    if kind == "bn_m":
        return g.uniform(-0.1, 0.1, size=shape).astype(np.float32)
    if kind == "bn_v":
        return g.uniform(0.8, 1.2, size=shape).astype(np.float32)
    if kind == "bias":
        return g.uniform(-0.05, 0.05, size=shape).astype(np.float32)
    if kind == "cls_bias":
        return (g.uniform(-0.05, 0.05, size=shape) - 5.7).astype(np.float32)
    if kind == "obj_bias":
        return (g.uniform(-0.05, 0.05, size=shape) - 2.5).astype(np.float32)
    if kind == "gauss_w":
        k = shape[-1]
        c = k // 2
        x, y = np.mgrid[0 - c:k - c, 0 - c:k - c]
        # the value the reference stores in its checkpoints (where2comm_fuse.py:66-81, sigma=1)
        w = 1.0 / (2.0 * np.pi) * np.exp(-(np.square(x) + np.square(y)) / 2.0)
        return w.astype(np.float32).reshape(shape)
    if kind == "gauss_b":
        return np.zeros(shape, dtype=np.float32)
    raise KeyError(kind)


def synthetic_state_dict(spec, seed=0):
    sd = OrderedDict()
    for key, shape, kind in spec:
        sd[key] = torch.from_numpy(np.ascontiguousarray(synthetic_tensor(key, shape, kind, seed))).reshape(tuple(shape))
    return sd


# --------------------------------------------------------------------------
# synthetic point clouds and the model-input dictionary
# --------------------------------------------------------------------------

def agent_types_for(n_agents):
    """types cycle vehicle, vehicle, rsu, drone, ... (ego = vehicle 0)."""
    cyc = ("vehicle", "vehicle", "rsu", "drone")
    return [cyc[i % 4] for i in range(n_agents)]


def synthetic_cloud(agent_id, n_points=8192, lidar_range=None, seed=1234):
    """Uniform cloud inside the (ego-type) range; BASELINE.md §3 / SURVEY §8(d)."""
    r = DEFAULT_RANGE if lidar_range is None else lidar_range
    g = np.random.default_rng(seed + agent_id)
    pts = np.empty((n_points, 4), dtype=np.float32)
    pts[:, 0] = g.uniform(r[0], r[3], n_points)
    pts[:, 1] = g.uniform(r[1], r[4], n_points)
    pts[:, 2] = g.uniform(r[2], r[5], n_points)
    pts[:, 3] = g.uniform(0.0, 1.0, n_points)
    return pts


def clustered_cloud(agent_id, n_points=108000, lidar_range=None, seed=4321):
    """'dense' distribution: most points in compact clusters so
    that a few percent of the pillars overflow the 32-point cap."""
    r = DEFAULT_RANGE if lidar_range is None else lidar_range
    g = np.random.default_rng(seed + agent_id)
    n_bg = n_points // 3
    n_cl = n_points - n_bg
    n_centres = 400
    cx = g.uniform(r[0], r[3], n_centres)
    cy = g.uniform(r[1], r[4], n_centres)
    which = g.integers(0, n_centres, n_cl)
    pts = np.empty((n_points, 4), dtype=np.float32)
    pts[:n_cl, 0] = cx[which] + g.normal(0, 0.8, n_cl)
    pts[:n_cl, 1] = cy[which] + g.normal(0, 0.8, n_cl)
    pts[n_cl:, 0] = g.uniform(r[0], r[3], n_bg)
    pts[n_cl:, 1] = g.uniform(r[1], r[4], n_bg)
    pts[:, 2] = g.uniform(r[2], r[5], n_points)
    pts[:, 3] = g.uniform(0.0, 1.0, n_points)
    perm = g.permutation(n_points)
    return pts[perm]


def build_data_dict(voxelized, types, device="cpu", max_cav_num=15):
    """Assemble the dict ``model.forward`` receives (SURVEY §8b input contract;
    produced in the reference by intermediate_fusion_dataset.py:763-867).

    ``voxelized``: list (one per agent, in frame order [veh.., rsu.., drone..])
    of (voxels (M,32,4) f32, coords (M,3) i32 zyx, num_points (M,) i32).
    B is 1 (one collaborative frame).
    """
    order = {t: i for i, t in enumerate(AGENT_TYPES)}
    assert list(types) == sorted(types, key=lambda t: order[t]), "agents must be ordered veh, rsu, drone"
    dd = {}
    n_total = len(types)
    for t in AGENT_TYPES:
        idxs = [i for i, tt in enumerate(types) if tt == t]
        if not idxs:
            dd[t] = {"batch_merged_lidar_features_torch": None, "batch_merged_cam_inputs": None,
                     "record_len": torch.zeros(1, dtype=torch.int32), "batch_idxs": []}
            continue
        feats, coords, nums = [], [], []
        for k, i in enumerate(idxs):
            v, c, n = voxelized[i]
            feats.append(np.asarray(v, dtype=np.float32))
            # prepend the agent's index *within its type* (sp_voxel_preprocessor.py:163-170)
            coords.append(np.concatenate([np.full((c.shape[0], 1), k, dtype=np.int32),
                                          np.asarray(c, dtype=np.int32)], axis=1))
            nums.append(np.asarray(n, dtype=np.int32))
        dd[t] = {
            "batch_merged_lidar_features_torch": {
                "voxel_features": torch.from_numpy(np.concatenate(feats, 0)).to(device),
                "voxel_coords": torch.from_numpy(np.concatenate(coords, 0)).to(device),
                "voxel_num_points": torch.from_numpy(np.concatenate(nums, 0)).to(device),
            },
            "batch_merged_cam_inputs": None,
            "record_len": torch.tensor([len(idxs)], dtype=torch.int32),
            "batch_idxs": [0],
        }
    L = max_cav_num
    eye = torch.eye(4, dtype=torch.float32).view(1, 1, 1, 4, 4).repeat(1, L, L, 1, 1)
    dd["record_len"] = torch.tensor([n_total], dtype=torch.int32)
    dd["pairwise_t_matrix_collab"] = eye.clone().to(device)
    dd["img_pairwise_t_matrix_collab"] = eye.to(device)
    prior = torch.zeros(1, L, 3, dtype=torch.float32)
    for i, t in enumerate(types):
        prior[0, i, 2] = 0.0 if t == "vehicle" else 1.0
    dd["prior_encoding"] = prior.to(device)
    dd["spatial_correction_matrix"] = torch.eye(4, dtype=torch.float64).view(1, 1, 4, 4).repeat(1, L, 1, 1).to(device)
    return dd


def merge_frames(frames):
    """Batch several B = 1 dicts of build_data_dict into one B = len(frames) dict the way the reference's collate does
    (intermediate_fusion_dataset.py:763-867, merge_features_to_dict :1080-1121): per agent type the voxel tensors
    are concatenated and the agent index column counts the type's agents across the whole batch; ``record_len`` holds
    the per-sample counts and ``batch_idxs`` the samples in which the type is present."""
    B = len(frames)
    dd = {}
    for t in AGENT_TYPES:
        feats, coords, nums, rl, present, base = [], [], [], [], [], 0
        for b, f in enumerate(frames):
            d = f[t]
            k = int(d["record_len"][0]) if len(d["batch_idxs"]) else 0
            rl.append(k)
            if k == 0:
                continue
            present.append(b)
            lid = d["batch_merged_lidar_features_torch"]
            c = lid["voxel_coords"].clone()
            c[:, 0] += base
            feats.append(lid["voxel_features"]); coords.append(c); nums.append(lid["voxel_num_points"])
            base += k
        dd[t] = {"batch_merged_lidar_features_torch": None if not present else {
                     "voxel_features": torch.cat(feats, 0), "voxel_coords": torch.cat(coords, 0), "voxel_num_points": torch.cat(nums, 0)},
                 "batch_merged_cam_inputs": None, "record_len": torch.tensor(rl, dtype=torch.int32), "batch_idxs": present}
    dd["record_len"] = torch.cat([f["record_len"] for f in frames])
    for k in ("pairwise_t_matrix_collab", "img_pairwise_t_matrix_collab", "prior_encoding", "spatial_correction_matrix"):
        dd[k] = torch.cat([f[k] for f in frames], 0)
    return dd


def sort_types(types):
    order = {t: i for i, t in enumerate(AGENT_TYPES)}
    idx = sorted(range(len(types)), key=lambda i: (order[types[i]], i))
    return idx, [types[i] for i in idx]


def clone_hypes(h):
    return copy.deepcopy(h)


def build_data_dict_device(voxelized, types, device, max_cav_num=15):
    """Same layout as build_data_dict, from voxel tensors that already live on the device."""
    order = {t: i for i, t in enumerate(AGENT_TYPES)}
    assert list(types) == sorted(types, key=lambda t: order[t]), "agents must be ordered veh, rsu, drone"
    host = build_data_dict([(np.zeros((0, 32, 4), np.float32), np.zeros((0, 3), np.int32), np.zeros((0,), np.int32))
                            for _ in types], types, "cpu", max_cav_num)
    for t in AGENT_TYPES:
        idxs = [i for i, tt in enumerate(types) if tt == t]
        if not idxs:
            continue
        feats, coords, nums = [], [], []
        for k, i in enumerate(idxs):
            v, c, n = voxelized[i]
            feats.append(v)
            coords.append(torch.cat([torch.full((c.shape[0], 1), k, dtype=torch.int32, device=c.device), c.to(torch.int32)], 1))
            nums.append(n)
        host[t]["batch_merged_lidar_features_torch"] = {
            "voxel_features": torch.cat(feats, 0).contiguous(),
            "voxel_coords": torch.cat(coords, 0).contiguous(),
            "voxel_num_points": torch.cat(nums, 0).contiguous(),
        }
    for k in ("pairwise_t_matrix_collab", "img_pairwise_t_matrix_collab", "prior_encoding", "spatial_correction_matrix"):
        host[k] = host[k].to(device)
    return host


def data_dict_to(dd, device):
    """Recursive .to(device) that tolerates None members (train_utils.py:473-494 ``to_device``);
    per-type record_len stays on the host (it only drives host-side bookkeeping)."""
    if isinstance(dd, dict):
        return {k: (v if k in ("record_len", "batch_idxs") else data_dict_to(v, device)) for k, v in dd.items()}
    if isinstance(dd, torch.Tensor):
        return dd.to(device)
    return dd


# --------------------------------------------------------------------------
# CoBEVT (airv2x_intermediate_cobevt.yaml :148-293): same per-agent trunk, the trunk keys sit at
# the top level of model.args and the fusion is the fused-axial-attention encoder
# --------------------------------------------------------------------------

def default_hypes_cobevt(lidar_range=None, max_cav=(3, 2, 2), compression=0):
    hy = default_hypes(lidar_range, max_cav)
    a = hy["model"]["args"]
    mf = a.pop("modality_fusion")
    a.pop("where2com_fusion")
    a.pop("ego_type", None)
    a["base_bev_backbone"] = mf["base_bev_backbone"]
    a["shrink_header"] = mf["shrink_header"]
    a["compression"] = int(compression)
    a["fax_fusion"] = {"input_dim": 256, "mlp_dim": 256, "window_size": 4, "dim_head": 32, "drop_out": 0.1,
                       "depth": 3, "mask": True, "agent_size": int(sum(max_cav))}
    hy["model"]["core_method"] = "airv2x_cobevt"
    hy["name"] = "airv2x_intermediate_cobevt"
    return hy


def _relative_position_index(L, ws):
    coords = np.stack(np.meshgrid(np.arange(L), np.arange(ws), np.arange(ws), indexing="ij")).reshape(3, -1)
    rel = (coords[:, :, None] - coords[:, None, :]).transpose(1, 2, 0).copy()
    rel[:, :, 0] += L - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 2] += ws - 1
    rel[:, :, 0] *= (2 * ws - 1) * (2 * ws - 1)
    rel[:, :, 1] *= 2 * ws - 1
    return rel.sum(-1).astype(np.int64)


def compressor_param_spec(c, ratio, prefix="naive_compressor"):
    """NaiveCompressor(c, ratio) state_dict manifest (models/common_modules/naive_compress.py:10-36):
    encoder Conv3x3 c -> c/ratio + BN, decoder Conv3x3 c/ratio -> c + BN, Conv3x3 c -> c + BN (convs have biases)."""
    if not ratio:
        return []
    m = c // ratio
    spec = []
    pre = prefix + "." if prefix else ""
    for name, co, ci in ((f"{pre}encoder.0", m, c), (f"{pre}decoder.0", c, m), (f"{pre}decoder.3", c, c)):
        bn = name[:-1] + str(int(name[-1]) + 1)
        spec += [(name + ".weight", (co, ci, 3, 3), "conv"), (name + ".bias", (co,), "bias"),
                 (bn + ".weight", (co,), "bn_w"), (bn + ".bias", (co,), "bn_b"), (bn + ".running_mean", (co,), "bn_m"),
                 (bn + ".running_var", (co,), "bn_v"), (bn + ".num_batches_tracked", (), "count")]
    return spec


def fax_param_spec(fax, prefix=""):
    """SwapFusionEncoder(args) (swap_fusion_modules.py:233-275)."""
    C, M, ws, L = fax["input_dim"], fax["mlp_dim"], fax["window_size"], fax["agent_size"]
    nh = C // fax["dim_head"]
    T = L * ws * ws
    spec = []
    for i in range(fax["depth"]):
        for part in ("window", "grid"):
            p = f"{prefix}layers.{i}.{part}_attention"
            spec += [(p + ".norm.weight", (C,), "ln_w"), (p + ".norm.bias", (C,), "ln_b"),
                     (p + ".fn.relative_position_index", (T, T), f"relidx:{L}:{ws}"),
                     (p + ".fn.to_qkv.weight", (3 * C, C), "lin"),
                     (p + ".fn.to_out.0.weight", (C, C), "lin"),
                     (p + ".fn.relative_position_bias_table.weight", ((2 * L - 1) * (2 * ws - 1) ** 2, nh), "relbias")]
            p = f"{prefix}layers.{i}.{part}_ffd"
            spec += [(p + ".norm.weight", (C,), "ln_w"), (p + ".norm.bias", (C,), "ln_b"),
                     (p + ".fn.net.0.weight", (M, C), "lin"), (p + ".fn.net.0.bias", (M,), "bias"),
                     (p + ".fn.net.3.weight", (C, M), "lin"), (p + ".fn.net.3.bias", (C,), "bias")]
    spec += [(prefix + "mlp_head.2.weight", (C,), "ln_w"), (prefix + "mlp_head.2.bias", (C,), "ln_b"),
             (prefix + "mlp_head.3.weight", (C, C), "lin"), (prefix + "mlp_head.3.bias", (C,), "bias")]
    return spec


def cobevt_param_spec(args):
    """Ordered (key, shape, kind) manifest of Airv2xCoBEVT's state_dict (236 tensors at L = 7;
    checked against the reference's own state_dict by tools/gen_golden.py)."""
    w2c_like = {"collaborators": args["collaborators"], **{t: args[t] for t in AGENT_TYPES if t in args},     # per-type modalities / cam blocks
                "modality_fusion": {"base_bev_backbone": args["base_bev_backbone"], "shrink_header": args["shrink_header"]},
                "where2com_fusion": {"communication": {"gaussian_smooth": {"k_size": 5}}},
                "anchor_number": args["anchor_number"], "num_class": args["num_class"], "outC": args["outC"],
                "obj_head": args["obj_head"]}
    base = where2com_param_spec(w2c_like)
    trunk = [e for e in base if not e[0].startswith(("fusion_net.", "naive_compressor.", "cls_head", "reg_head", "obj_head"))]
    heads = [e for e in base if e[0].startswith(("cls_head", "reg_head", "obj_head"))]
    fax = args["fax_fusion"]
    spec = list(trunk) + compressor_param_spec(fax["input_dim"], args.get("compression", 0))
    spec += fax_param_spec(fax, "fusion_net.")
    return spec + heads


# --------------------------------------------------------------------------
# V2X-ViT (airv2x_intermediate_v2xvit.yaml :150-310): Where2Comm-style nesting (modality_fusion)
# plus the `transformer.encoder` block
# --------------------------------------------------------------------------

def default_hypes_v2xvit(lidar_range=None, max_cav=(5, 5, 5)):
    hy = default_hypes(lidar_range, max_cav)
    a = hy["model"]["args"]
    a.pop("where2com_fusion")
    a["transformer"] = {"encoder": {
        "num_blocks": 1, "depth": 3, "use_roi_mask": True, "use_RTE": True, "RTE_ratio": 2,
        "cav_att_config": {"dim": 256, "use_hetero": True, "use_RTE": True, "RTE_ratio": 2, "heads": 8, "dim_head": 32,
                           "dropout": 0.3},
        "pwindow_att_config": {"dim": 256, "heads": [16, 8, 4], "dim_head": [16, 32, 64], "dropout": 0.3,
                               "window_size": [2, 4, 4], "relative_pos_embedding": True, "fusion_method": "split_attn"},
        "feed_forward": {"mlp_dim": 256, "dropout": 0.3},
        "sttf": {"voxel_size": list(DEFAULT_VOXEL), "downsample_rate": 4},
    }}
    hy["model"]["core_method"] = "airv2x_v2xvit"
    hy["name"] = "airv2x_intermediate_v2xvit"
    return hy


def v2xvit_encoder_spec(enc, p="encoder"):
    """V2XTEncoder(args) under the key prefix ``p`` (v2xvit_basic.py:135-171)."""
    cav, pw = enc["cav_att_config"], enc["pwindow_att_config"]
    C, inner = cav["dim"], cav["heads"] * cav["dim_head"]
    spec = []
    spec += [(p + ".prior_feed.weight", (C, C + 3), "lin"), (p + ".prior_feed.bias", (C,), "bias")]
    for d in range(enc["depth"]):
        for nb in range(enc["num_blocks"]):
            q = f"{p}.layers.{d}.0.layers.{nb}"
            spec += [(q + ".0.norm.weight", (C,), "ln_w"), (q + ".0.norm.bias", (C,), "ln_b"),
                     (q + ".0.fn.relation_att", (4, cav["heads"], cav["dim_head"], cav["dim_head"]), "rel"),
                     (q + ".0.fn.relation_msg", (4, cav["heads"], cav["dim_head"], cav["dim_head"]), "rel")]
            for name in ("k", "q", "v"):
                for t in range(2):
                    spec += [(f"{q}.0.fn.{name}_linears.{t}.weight", (inner, C), "lin"),
                             (f"{q}.0.fn.{name}_linears.{t}.bias", (inner,), "bias")]
            for t in range(2):
                spec += [(f"{q}.0.fn.a_linears.{t}.weight", (C, inner), "lin"), (f"{q}.0.fn.a_linears.{t}.bias", (C,), "bias")]
            spec += [(q + ".1.norm.weight", (C,), "ln_w"), (q + ".1.norm.bias", (C,), "ln_b")]
            for i, (h, dh, ws) in enumerate(zip(pw["heads"], pw["dim_head"], pw["window_size"])):
                w = f"{q}.1.fn.pwmsa.{i}"
                spec += [(w + ".pos_embedding", (2 * ws - 1, 2 * ws - 1), "relbias"),
                         (w + ".to_qkv.weight", (3 * h * dh, C), "lin"),
                         (w + ".to_out.0.weight", (C, h * dh), "lin"), (w + ".to_out.0.bias", (C,), "bias")]
            sa = f"{q}.1.fn.split_attn"
            spec += [(sa + ".fc1.weight", (C, C), "lin"), (sa + ".bn1.weight", (C,), "ln_w"), (sa + ".bn1.bias", (C,), "ln_b"),
                     (sa + ".fc2.weight", (3 * C, C), "lin")]
        f = f"{p}.layers.{d}.1"
        spec += [(f + ".norm.weight", (C,), "ln_w"), (f + ".norm.bias", (C,), "ln_b"),
                 (f + ".fn.net.0.weight", (enc["feed_forward"]["mlp_dim"], C), "lin"),
                 (f + ".fn.net.0.bias", (enc["feed_forward"]["mlp_dim"],), "bias"),
                 (f + ".fn.net.3.weight", (C, enc["feed_forward"]["mlp_dim"]), "lin"), (f + ".fn.net.3.bias", (C,), "bias")]
    spec += [(p + ".rte.emb.emb.weight", (100, C), "rte_table"), (p + ".rte.emb.lin.weight", (C, C), "lin"),
             (p + ".rte.emb.lin.bias", (C,), "bias")]
    return spec


def v2xvit_param_spec(args):
    """Ordered (key, shape, kind) manifest of Airv2xV2XVit's state_dict (297 tensors; checked against the
    reference's own state_dict by tools/gen_golden.py)."""
    w2c_like = dict(args)
    w2c_like["where2com_fusion"] = {"communication": {"gaussian_smooth": {"k_size": 5}}}
    base = where2com_param_spec(w2c_like)
    trunk = [e for e in base if not e[0].startswith(("fusion_net.", "naive_compressor.", "cls_head", "reg_head", "obj_head"))]
    heads = [e for e in base if e[0].startswith(("cls_head", "reg_head", "obj_head"))]
    spec = list(trunk) + compressor_param_spec(256, model_compression(args)) + v2xvit_encoder_spec(args["transformer"]["encoder"], "fusion_net.encoder")
    return spec + heads


def model_compression(args):
    """NaiveCompressor ratio of the Where2Comm-nested models as the reference reads it (airv2x_v2xvit.py:42-44, airv2x_when2com.py:50-52):
    switched on by ``modality_fusion.compression > 0``, ratio taken from the TOP-LEVEL ``compression`` key (a KeyError in the reference when
    only the first is given -- the same here); 0 = no compressor."""
    if not args["modality_fusion"].get("compression", 0) > 0:
        return 0
    return int(args["compression"])


def se2_correction(yaw_deg, tx, ty):
    """4x4 float64 spatial_correction_matrix of an SE(2) motion (yaw ~ U(-10,10) deg, t ~ U(-8,8) m)."""
    c, s = np.cos(np.deg2rad(yaw_deg)), np.sin(np.deg2rad(yaw_deg))
    m = np.eye(4)
    m[0, 0], m[0, 1], m[1, 0], m[1, 1], m[0, 3], m[1, 3] = c, -s, s, c, tx, ty
    return m


# --------------------------------------------------------------------------
# When2com (airv2x_intermediate_when2com.yaml :144-290): the Where2Comm trunk + `when2com_fusion`
# --------------------------------------------------------------------------

def default_hypes_when2com(lidar_range=None, max_cav=(5, 5, 5), mode="softmax"):
    hy = default_hypes(lidar_range, max_cav)
    a = hy["model"]["args"]
    a.pop("where2com_fusion")
    g = a["vehicle"]["lidar"]["point_pillar_scatter"]["grid_size"]
    a["when2com_fusion"] = {"voxel_size": list(DEFAULT_VOXEL), "downsample_rate": 4, "num_iteration": 2, "in_channels": 256,
                            "query_size": 32, "key_size": 256, "mode": mode, "H": int(g[1]) // 2, "W": int(g[0]) // 2}
    hy["model"]["core_method"] = "airv2x_when2com"
    hy["name"] = "airv2x_intermediate_when2com"
    return hy


def when2com_pairwise(n, L):
    """(1,L,L,4,4) fp32 ``img_pairwise_t_matrix_collab``: row 0 (ego -> j) carries a small SE(2) motion for every
    non-ego agent.  The AirV2X dataset ships identities (proj_first); a non-trivial matrix exercises the warp."""
    t = torch.eye(4).view(1, 1, 1, 4, 4).repeat(1, L, L, 1, 1)
    for j in range(1, n):
        t[0, 0, j] = torch.from_numpy(se2_correction(3.0 * j, 1.1 * j, -0.7 * j)).float()
    return t


def when2com_fusion_spec(cfg, prefix=""):
    """When2comFusion(args) (when2com_modules/when2com.py:14-44): policy_net4 (5 x Conv3x3 + bias + BN + ReLU),
    two km_generator MLPs on the flattened (256, H/4, W/4) map, and the query -> key Linear of the attention."""
    spec = []
    chans = [(cfg["in_channels"], 512), (512, 256), (256, 256), (256, 256), (256, 256)]
    for i, (ci, co) in enumerate(chans, 1):
        p = f"{prefix}query_key_net.conv{i}.cbr_unit"
        spec += [(p + ".0.weight", (co, ci, 3, 3), "conv"), (p + ".0.bias", (co,), "bias")] + _bn(p + ".1", co)
    nf = 256 * (cfg["H"] // 4) * (cfg["W"] // 4)
    for net, out in (("key_net", cfg["key_size"]), ("query_net", cfg["query_size"])):
        for li, (ci, co) in zip((0, 2, 4), ((nf, 256), (256, 128), (128, out))):
            spec += [(f"{prefix}{net}.fc.{li}.weight", (co, ci), "lin"), (f"{prefix}{net}.fc.{li}.bias", (co,), "bias")]
    spec += [(prefix + "attention_net.linear.weight", (cfg["key_size"], cfg["query_size"]), "att_lin"),
             (prefix + "attention_net.linear.bias", (cfg["key_size"],), "bias")]
    return spec


def when2com_param_spec(args):
    """Ordered (key, shape, kind) manifest of Airv2xWhen2com's state_dict (checked against the reference's own
    state_dict by tools/gen_golden.py)."""
    w2c_like = dict(args)
    w2c_like["where2com_fusion"] = {"communication": {"gaussian_smooth": {"k_size": 5}}}
    base = where2com_param_spec(w2c_like)
    trunk = [e for e in base if not e[0].startswith(("fusion_net.", "naive_compressor.", "cls_head", "reg_head", "obj_head"))]
    heads = [e for e in base if e[0].startswith(("cls_head", "reg_head", "obj_head"))]
    return trunk + compressor_param_spec(256, model_compression(args)) + when2com_fusion_spec(args["when2com_fusion"], "fusion_net.") + heads


# --------------------------------------------------------------------------
# V2VNet (models/airv2x_v2vnet.py; no AirV2X YAML ships for it: the Where2Comm trunk + the `v2vfusion` block of
# hypes_yaml/opv2v/opv2v_v2vnet.yaml:91-102 re-sized to the AirV2X feature map)
# --------------------------------------------------------------------------

def default_hypes_v2vnet(lidar_range=None, max_cav=(5, 5, 5), agg="avg", num_iteration=2):
    hy = default_hypes(lidar_range, max_cav)
    a = hy["model"]["args"]
    a.pop("where2com_fusion")
    g = a["vehicle"]["lidar"]["point_pillar_scatter"]["grid_size"]
    a["v2vfusion"] = {"voxel_size": list(DEFAULT_VOXEL), "downsample_rate": 2, "num_iteration": num_iteration, "in_channels": 256,
                      "gru_flag": True, "agg_operator": agg,
                      "conv_gru": {"H": int(g[1]) // 2, "W": int(g[0]) // 2, "num_layers": 1, "kernel_size": [[3, 3]]}}
    a["backbone_fix"] = False
    hy["model"]["core_method"] = "airv2x_v2vnet"
    hy["name"] = "airv2x_intermediate_v2vnet"
    return hy


def v2vnet_pairwise(n, L):
    """(1,L,L,4,4) fp32 ``img_pairwise_t_matrix_collab`` with a small SE(2) motion on EVERY ordered pair (i != j): V2VNet
    warps every node's neighbours into that node's frame (v2v_fuse.py:142-146 uses row i of the matrix)."""
    t = torch.eye(4).view(1, 1, 1, 4, 4).repeat(1, L, L, 1, 1)
    for i in range(n):
        for j in range(n):
            if i != j:
                t[0, i, j] = torch.from_numpy(se2_correction(2.0 * (j - i), 0.9 * (j - i) + 0.3 * i, -0.6 * (j - i))).float()
    return t


def v2vnet_fusion_spec(cfg, prefix=""):
    """V2VNetFusion(args) (v2vnet_modules/v2v_fuse.py:19-47): msg_cnn, one ConvGRU cell (gates + candidate), mlp."""
    c = cfg["in_channels"]
    p = prefix + "conv_gru.cell_list.0"
    return [(prefix + "msg_cnn.weight", (c, 2 * c, 3, 3), "conv"), (prefix + "msg_cnn.bias", (c,), "bias"),
            (p + ".conv_gates.weight", (2 * c, 3 * c, 3, 3), "conv"), (p + ".conv_gates.bias", (2 * c,), "bias"),
            (p + ".conv_can.weight", (c, 3 * c, 3, 3), "conv"), (p + ".conv_can.bias", (c,), "bias"),
            (prefix + "mlp.weight", (c, c), "lin"), (prefix + "mlp.bias", (c,), "bias")]


def v2vnet_param_spec(args):
    """Ordered (key, shape, kind) manifest of Airv2xV2VNet's state_dict (checked against the reference's own state_dict
    by tools/gen_golden.py)."""
    w2c_like = dict(args)
    w2c_like["where2com_fusion"] = {"communication": {"gaussian_smooth": {"k_size": 5}}}
    base = where2com_param_spec(w2c_like)
    trunk = [e for e in base if not e[0].startswith(("fusion_net.", "naive_compressor.", "cls_head", "reg_head", "obj_head"))]
    heads = [e for e in base if e[0].startswith(("cls_head", "reg_head", "obj_head"))]
    return trunk + compressor_param_spec(256, model_compression(args)) + v2vnet_fusion_spec(args["v2vfusion"], "fusion_net.") + heads


# --------------------------------------------------------------------------
# Camera lift-splat inputs (hypes_yaml/airv2x/camera/det/airv2x_intermediate_where2com.yaml :52-95, :180-188)
# --------------------------------------------------------------------------

def cam_args(agent_type="vehicle", final_dim=(360, 640), xy=(-140.8, 140.8, -40.0, 40.0), img_features=64):
    """``args[agent_type]["cam"]`` of the shipped camera YAML (grid_conf / data_aug_conf anchors)."""
    z, dd, mode = {"vehicle": ([-10, 10, 20.0], [2, 50, 48], "LID"), "rsu": ([-30, 30, 60.0], [2, 50, 48], "LID"),
                   "drone": ([-150, -6, 144], [6, 150, 144], "UD")}[agent_type]
    return {"grid_conf": {"xbound": [xy[0], xy[1], 0.4], "ybound": [xy[2], xy[3], 0.4], "zbound": z, "ddiscr": dd, "mode": mode},
            "data_aug_conf": {"resize_lim": [0.65, 0.7], "final_dim": list(final_dim), "rot_lim": [-3.6, 3.6] if agent_type == "drone" else [0, 0],
                              "H": 720, "W": 1280,
                              "rand_flip": False, "bot_pct_lim": [0.0, 0.05]},
            "img_downsample": 8, "img_features": img_features, "bevout_feature": 64, "camera_encoder": "EfficientNet",
            "use_depth_gt": True, "depth_supervision": False}


def camera_rig(seed, B, N, final_dim=(360, 640), drone=False):
    """Seeded plausible camera tensors of ``batch_merged_cam_inputs``: rots / trans (camera -> ego), intrinsics of a
    1280x720 sensor, and the resize + crop of the image augmentation as post_rots / post_trans."""
    g = np.random.default_rng(int(seed))
    rots, trans, intr = np.zeros((B, N, 3, 3), np.float32), np.zeros((B, N, 3), np.float32), np.zeros((B, N, 3, 3), np.float32)
    post_rots, post_trans = np.zeros((B, N, 3, 3), np.float32), np.zeros((B, N, 3), np.float32)
    for b in range(B):
        for n in range(N):
            yaw = 2 * np.pi * n / N + g.uniform(-0.1, 0.1)
            # camera axes (x right, y down, z forward) -> ego (x forward, y left, z up); a drone camera looks down
            base = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], np.float64)
            if drone:
                base = np.array([[0, -1, 0], [-1, 0, 0], [0, 0, -1]], np.float64)
            c, s_ = np.cos(yaw), np.sin(yaw)
            rz = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]])
            rots[b, n] = (rz @ base).astype(np.float32)
            trans[b, n] = [g.uniform(-1, 1), g.uniform(-0.5, 0.5), 60.0 if drone else g.uniform(1.4, 2.0)]
            f = g.uniform(550, 650)
            intr[b, n] = [[f, 0, 640], [0, f, 360], [0, 0, 1]]
            r = g.uniform(0.65, 0.7)
            post_rots[b, n] = np.diag([r, r, 1.0]).astype(np.float32)
            # crop: centred in x; in y from the bottom edge (vehicle / RSU augmentation), centred for the nadir drone camera
            post_trans[b, n] = [-(1280 * r - final_dim[1]) / 2, -(720 * r - final_dim[0]) * (0.5 if drone else g.uniform(0.9, 1.0)), 0.0]
    t = lambda a: torch.from_numpy(a)
    return t(rots), t(trans), t(intr), t(post_rots), t(post_trans)


def lifted_features(seed, B, N, D, fH, fW, C, one_hot=True):
    """x of voxel_pooling, (B,N,D,fH,fW,C): image features x depth distribution.  one_hot = the shipped use_depth_gt
    path (every pixel's mass in ONE depth bin); else a dense softmax over depth."""
    g = np.random.default_rng(int(seed))
    feat = g.standard_normal((B, N, 1, fH, fW, C)).astype(np.float32)
    if one_hot:
        d = g.integers(0, D, (B, N, fH, fW))
        depth = np.zeros((B, N, D, fH, fW, 1), np.float32)
        np.put_along_axis(depth, d[:, :, None, :, :, None], 1.0, axis=2)
    else:
        lg = g.standard_normal((B, N, D, fH, fW, 1)).astype(np.float32) * 2
        e = np.exp(lg - lg.max(2, keepdims=True))
        depth = (e / e.sum(2, keepdims=True)).astype(np.float32)
    return torch.from_numpy(depth * feat)


# EfficientNet-B0 as the efficientnet_pytorch package lays it out (see oracle/camera_oracle.py for the restatement):
# (repeats, kernel, stride, expand, in, out); squeeze width = max(1, int(block input * 0.25))
EFFNET_B0_STAGES = ((1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
                    (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320))


def effnet_b0_blocks():
    """[(cin, cout, k, stride, expand, se_width, (pad_before, pad_after))] of the 16 MBConv blocks; the depthwise padding is
    TensorFlow 'same' computed from the package's NOMINAL 224-pixel input (Conv2dStaticSamePadding), whatever the real size."""
    rows, size = [], 112
    for rep, k, s, e, ci, co in EFFNET_B0_STAGES:
        for r in range(rep):
            st, cin = (s, ci) if r == 0 else (1, co)
            total = max((-(-size // st) - 1) * st + k - size, 0)
            rows.append((cin, co, k, st, e, max(1, int(cin * 0.25)), (total // 2, total - total // 2)))
            size = -(-size // st)
    return rows


def effnet_param_spec(prefix):
    spec = [(prefix + "_conv_stem.weight", (32, 3, 3, 3), "conv")] + _bn(prefix + "_bn0", 32)
    for i, (cin, cout, k, st, e, se, _) in enumerate(effnet_b0_blocks()):
        q, mid = f"{prefix}_blocks.{i}.", cin * e
        if e != 1:
            spec += [(q + "_expand_conv.weight", (mid, cin, 1, 1), "conv")] + _bn(q + "_bn0", mid)
        spec += [(q + "_depthwise_conv.weight", (mid, 1, k, k), "conv")] + _bn(q + "_bn1", mid)
        spec += [(q + "_se_reduce.weight", (se, mid, 1, 1), "se_w"), (q + "_se_reduce.bias", (se,), "bias"),
                 (q + "_se_expand.weight", (mid, se, 1, 1), "se_w"), (q + "_se_expand.bias", (mid,), "bias")]
        spec += [(q + "_project_conv.weight", (cout, mid, 1, 1), "proj")] + _bn(q + "_bn2", cout)
    spec += [(prefix + "_conv_head.weight", (1280, 320, 1, 1), "conv")] + _bn(prefix + "_bn1", 1280)
    spec += [(prefix + "_fc.weight", (1000, 1280), "lin"), (prefix + "_fc.bias", (1000,), "bias")]
    return spec


def up_param_spec(prefix, cin, cout):
    """lss_submodule.Up (:22-47): Conv3x3 + BN + ReLU twice, no conv bias."""
    return ([(prefix + "conv.0.weight", (cout, cin, 3, 3), "conv")] + _bn(prefix + "conv.1", cout)
            + [(prefix + "conv.3.weight", (cout, cout, 3, 3), "conv")] + _bn(prefix + "conv.4", cout))


def cam_depth_bins(cam):
    return int(cam["grid_conf"]["ddiscr"][2])


RESNET101_LAYERS = ((64, 3, 1), (128, 4, 2))       # (planes, bottlenecks, stride) of torchvision resnet101's layer1 / layer2: all CamEncode_Resnet101 keeps


def _bottleneck_spec(prefix, cin, planes, stride):
    """torchvision Bottleneck (expansion 4): 1x1 -> 3x3 (stride) -> 1x1, a 1x1 / stride downsample when the shape changes."""
    spec = ([(prefix + "conv1.weight", (planes, cin, 1, 1), "conv")] + _bn(prefix + "bn1", planes)
            + [(prefix + "conv2.weight", (planes, planes, 3, 3), "conv")] + _bn(prefix + "bn2", planes)
            + [(prefix + "conv3.weight", (4 * planes, planes, 1, 1), "conv")] + _bn(prefix + "bn3", 4 * planes))
    if stride != 1 or cin != 4 * planes:
        spec += [(prefix + "downsample.0.weight", (4 * planes, cin, 1, 1), "conv")] + _bn(prefix + "downsample.1", 4 * planes)
    return spec


def camencode_resnet101_param_spec(cam, prefix):
    """lss_submodule.CamEncode_Resnet101 (:191-224): conv1 / bn1 / layer1 / layer2 of a torchvision resnet101 (registration order of the
    module), depth_head (512 -> D, only without use_depth_gt), image_head (512 -> C)."""
    spec = [(prefix + "conv1.weight", (64, 3, 7, 7), "conv")] + _bn(prefix + "bn1", 64)
    cin = 64
    for li, (planes, nb, stride) in enumerate(RESNET101_LAYERS, 1):
        for bi in range(nb):
            spec += _bottleneck_spec(f"{prefix}layer{li}.{bi}.", cin, planes, stride if bi == 0 else 1)
            cin = 4 * planes
    if not cam["use_depth_gt"]:
        spec += [(prefix + "depth_head.weight", (cam_depth_bins(cam), 512, 1, 1), "head"), (prefix + "depth_head.bias", (cam_depth_bins(cam),), "bias")]
    spec += [(prefix + "image_head.weight", (cam["img_features"], 512, 1, 1), "img_head"), (prefix + "image_head.bias", (cam["img_features"],), "bias")]
    return spec


def camencode_param_spec(cam, prefix):
    """lss_submodule.CamEncode (:50-87) with the EfficientNet trunk, chain_channels 256; ``camera_encoder: Resnet101`` -> CamEncode_Resnet101."""
    if cam["camera_encoder"] == "Resnet101":
        return camencode_resnet101_param_spec(cam, prefix)
    if cam["camera_encoder"] != "EfficientNet":
        raise NotImplementedError(f"camera_encoder {cam['camera_encoder']!r}: EfficientNet or Resnet101 (airv2x_encoder.py:67-86)")
    spec = effnet_param_spec(prefix + "trunk.") + up_param_spec(prefix + "up1.", 320 + 112, 256)
    if cam["img_downsample"] == 8:
        spec += up_param_spec(prefix + "up2.", 256 + 40, 256)
    if not cam["use_depth_gt"]:
        spec += [(prefix + "depth_head.weight", (cam_depth_bins(cam), 256, 1, 1), "head"), (prefix + "depth_head.bias", (cam_depth_bins(cam),), "bias")]
    spec += [(prefix + "image_head.weight", (cam["img_features"], 256, 1, 1), "img_head"), (prefix + "image_head.bias", (cam["img_features"],), "bias")]
    return spec


def _basic_block_spec(prefix, cin, cout, stride):
    spec = ([(prefix + "conv1.weight", (cout, cin, 3, 3), "conv")] + _bn(prefix + "bn1", cout)
            + [(prefix + "conv2.weight", (cout, cout, 3, 3), "conv")] + _bn(prefix + "bn2", cout))
    if stride != 1 or cin != cout:
        spec += [(prefix + "downsample.0.weight", (cout, cin, 1, 1), "conv")] + _bn(prefix + "downsample.1", cout)
    return spec


def bevencode_param_spec(inC, outC, prefix):
    """lss_submodule.BevEncode (:312-333): 7x7/s2 stem, torchvision resnet18 layer1-3, Up(64+256 -> 256, x4), up2."""
    spec = [(prefix + "conv1.weight", (64, inC, 7, 7), "conv")] + _bn(prefix + "bn1", 64)
    cin = 64
    for li, c in enumerate((64, 128, 256)):
        spec += _basic_block_spec(f"{prefix}layer{li + 1}.0.", cin, c, 1 if li == 0 else 2)
        spec += _basic_block_spec(f"{prefix}layer{li + 1}.1.", c, c, 1)
        cin = c
    spec += up_param_spec(prefix + "up1.", 64 + 256, 256)
    spec += [(prefix + "up2.1.weight", (128, 256, 3, 3), "conv")] + _bn(prefix + "up2.2", 128)
    spec += [(prefix + "up2.4.weight", (outC, 128, 1, 1), "conv"), (prefix + "up2.4.bias", (outC,), "bias")]
    return spec


def lss_param_spec(cam, prefix):
    """LiftSplatShootEncoder (airv2x_encoder.py:31-91): camencode then bevencode (dx / bx / nx / frustum are plain attributes)."""
    return camencode_param_spec(cam, prefix + "camencode.") + bevencode_param_spec(cam["img_features"], cam["bevout_feature"], prefix + "bevencode.")


def multimodal_hypes(modalities=("cam", "lidar"), lidar_range=None, final_dim=(360, 640), use_depth_gt=True, max_cav=(5, 5, 5),
                     camera_encoder="EfficientNet"):
    """default_hypes with a camera encoder per agent type (``args[type]["cam"]`` = the shipped camera block,
    hypes_yaml/airv2x/camera/det/airv2x_intermediate_where2com.yaml:180-248) and ``modalities`` as given: ("cam",) is that
    YAML, ("cam", "lidar") is BASELINE configs[4] (no shipped YAML sets both)."""
    return add_camera_modalities(default_hypes(lidar_range, max_cav), modalities, final_dim, use_depth_gt, camera_encoder)


def add_camera_modalities(hy, modalities=("cam",), final_dim=(360, 640), use_depth_gt=True, camera_encoder="EfficientNet"):
    """Give every agent type of a model's hypes (default_hypes / default_hypes_cobevt / _v2xvit / _when2com / _v2vnet) the camera block
    of the shipped camera YAMLs (hypes_yaml/airv2x/camera/det/airv2x_intermediate_{where2com,cobevt,v2xvit,when2com}.yaml: the
    ``vehicle / rsu / drone`` entries are the same in all four) and set ``modalities``; the BEV grid follows the hypes' LiDAR range."""
    r = hy["preprocess"]["cav_lidar_range"]
    a = hy["model"]["args"]
    a["active_sensors"] = list(modalities)
    for t in AGENT_TYPES:
        a[t]["modalities"] = list(modalities)
        a[t]["cam"] = cam_args(t, final_dim, (r[0], r[3], r[1], r[4]))
        a[t]["cam"]["use_depth_gt"] = bool(use_depth_gt)
        a[t]["cam"]["camera_encoder"] = camera_encoder
    return hy


def camera_images(seed, B, N, final_dim=(360, 640), agent_type="vehicle"):
    """(B, N, 4, H, W) fp32: three normalised colour planes (smooth structure + noise) and a depth plane in metres that
    runs past both ends of the type's depth range, so that the out-of-range mask of bin_depths is exercised."""
    g = np.random.default_rng(int(seed))
    H, W = final_dim
    yy, xx = np.meshgrid(np.linspace(0, 1, H, dtype=np.float32), np.linspace(0, 1, W, dtype=np.float32), indexing="ij")
    img = np.empty((B, N, 4, H, W), np.float32)
    lo, hi = (6.0, 150.0) if agent_type == "drone" else (2.0, 50.0)
    for b in range(B):
        for n in range(N):
            for c in range(3):
                fx, fy, ph = g.uniform(1, 9), g.uniform(1, 9), g.uniform(0, 6.28)
                img[b, n, c] = np.sin(6.28 * (fx * xx + fy * yy) + ph) * g.uniform(0.5, 1.5) + g.standard_normal((H, W)).astype(np.float32) * 0.5
            d = lo - 1.5 + (hi - lo + 6.0) * (0.15 + 0.85 * yy[::-1] ** 2) * (1 + 0.1 * np.sin(9 * xx + n))
            img[b, n, 3] = d + g.uniform(-0.3, 0.3, (H, W)).astype(np.float32)
    return torch.from_numpy(img)


def cam_inputs_for(seed, n_agents, n_cams, final_dim=(360, 640), agent_type="vehicle"):
    """``batch_merged_cam_inputs`` of one agent type (intermediate_fusion_dataset.py:571-583, :766-781): B = the type's agents."""
    rots, trans, intr, post_rots, post_trans = camera_rig(seed, n_agents, n_cams, final_dim, drone=(agent_type == "drone"))
    return {"imgs": camera_images(seed + 7, n_agents, n_cams, final_dim, agent_type), "rots": rots, "trans": trans, "intrinsics": intr,
            "post_rots": post_rots, "post_trans": post_trans}


CAMS_PER_AGENT = {"vehicle": 4, "rsu": 4, "drone": 1}


def add_cameras(dd, types, seed=50, final_dim=(360, 640), cams_per_agent=None):
    """Fill ``batch_merged_cam_inputs`` of a build_data_dict() frame for every agent type present."""
    cpa = cams_per_agent or CAMS_PER_AGENT
    for ti, t in enumerate(AGENT_TYPES):
        k = sum(1 for tt in types if tt == t)
        if k:
            dd[t]["batch_merged_cam_inputs"] = cam_inputs_for(seed + 100 * ti, k, cpa[t], final_dim, t)
    return dd


# --------------------------------------------------------------------------
# sub-module harness (tests/test_submodules.py, tools/gen_golden.py submodules): one small configuration per
# reference sub-module of SURVEY 8b, inputs from seeded generators so that only outputs are stored as fixtures
# --------------------------------------------------------------------------

SUBMODULE_RANGE = [-6.4, -6.4, -3.0, 6.4, 6.4, 1.0]            # 32 x 32 pillars of 0.4 m


def seeded_uniform(seed, shape, lo=-1.0, hi=1.0):
    return np.random.default_rng(int(seed)).uniform(lo, hi, size=tuple(shape)).astype(np.float32)


def submodule_configs():
    return {
        "pillar_vfe": {"use_norm": True, "with_distance": False, "use_absolute_xyz": True, "num_filters": [64]},
        "scatter": {"num_features": 64, "grid_size": [32, 32, 1]},
        "backbone": {"layer_nums": [1, 1, 2], "layer_strides": [2, 2, 2], "num_filters": [32, 64, 64],
                     "upsample_strides": [1, 2, 4], "num_upsample_filter": [32, 32, 32]},
        # base_bev_backbone.py:87-121: a deblock that DOWN-samples (stride 0.5 -> Conv2d(2, stride 2)) and one more deblock on the concat
        "backbone_variant": {"layer_nums": [1, 1], "layer_strides": [1, 2], "num_filters": [32, 64],
                             "upsample_strides": [0.5, 1, 2], "num_upsample_filter": [32, 32, 0]},   # equal lengths are asserted (:25);
                             # the final deblock's width is sum(num_upsample_filter), so its own entry has to be 0
        "shrink": {"kernal_size": [1], "dim": [64], "stride": [1], "padding": [0], "input_dim": 96},
        "compressor": (64, 2),
        "where2comm": {"fully": False, "voxel_size": list(DEFAULT_VOXEL), "downsample_rate": 2, "in_channels": 64,
                       "multi_scale": True, "layer_nums": [1, 1, 2], "num_filters": [32, 64, 64],
                       "communication": {"round": 1, "threshold": 0.01, "gaussian_smooth": {"k_size": 5, "c_sigma": 1.0}}},
        "fax": {"input_dim": 256, "mlp_dim": 256, "window_size": 4, "dim_head": 32, "drop_out": 0.1, "depth": 2,
                "mask": True, "agent_size": 3},
        "v2xvit": default_hypes_v2xvit()["model"]["args"]["transformer"],
    }


def w2c_attn_configs():
    """Reduced configurations of the OPV2V-style Where2comm (where2comm_modules/where2comm_attn.py) and of the
    BaseBEVBackbone it drives; channel counts are the real ones (64 / 128 / 256), the map is 32 x 48 cells."""
    bb = {"layer_nums": [1, 1, 1], "layer_strides": [2, 2, 2], "num_filters": [64, 128, 256],
          "upsample_strides": [1, 2, 4], "num_upsample_filter": [32, 32, 32]}
    base = {"voxel_size": list(DEFAULT_VOXEL), "downsample_rate": 2, "multi_scale": True,
            "layer_nums": bb["layer_nums"], "num_filters": bb["num_filters"]}
    gauss = {"thre": 0.01, "gaussian_smooth": {"k_size": 5, "c_sigma": 1.0}}
    return {
        "backbone": bb,
        "ms_atten": dict(base, agg_operator={"mode": "ATTEN"}, communication=dict(gauss)),
        "ms_max": dict(base, agg_operator={"mode": "MAX"}, communication={"thre": 0.004}),
        "ms_atten2": dict(base, layer_nums=[1, 1], num_filters=[64, 128], agg_operator={"mode": "ATTEN"}, communication=dict(gauss)),   # two levels
        "ss_atten": dict(base, multi_scale=False, agg_operator={"mode": "ATTEN", "feature_dim": 256}, communication=dict(gauss)),
        "ss_max": dict(base, multi_scale=False, agg_operator={"mode": "MAX", "feature_dim": 64}),
        # the ResNet backbone variant (where2comm_attn.py:312-317: `backbone.resnet(x)` once, its three maps feed the levels)
        "resnet_backbone": {"layer_nums": [2, 1, 2], "layer_strides": [2, 2, 2], "num_filters": [64, 128, 256],
                            "upsample_strides": [1, 2, 4], "num_upsample_filter": [32, 32, 32]},
        # base_bev_backbone_resnet.py:57-110: a down-sampling deblock and the final deblock on the concatenation
        "resnet_backbone_variant": {"layer_nums": [1, 1], "layer_strides": [1, 2], "num_filters": [64, 128],
                                    "upsample_strides": [0.5, 1, 2], "num_upsample_filter": [32, 32, 0]},
    }


def w2c_attn_pairwise(record_len, L=5):
    """(B,L,L,4,4) fp32: row 0 of every sample (the ego's row, the only one where2comm_attn.py:364 reads) carries a
    different small SE(2) motion for every agent, the ego's own entry being the identity."""
    t = torch.eye(4).view(1, 1, 1, 4, 4).repeat(len(record_len), L, L, 1, 1)
    for b, n in enumerate(record_len):
        for j in range(1, n):
            t[b, 0, j] = torch.from_numpy(se2_correction(2.5 * j - 1.5 * b, 1.3 * j + 0.4 * b, -0.9 * j + 0.2 * b)).float()
    return t


def w2c_attn_features(seed, n, c, h, w, keep=0.35):
    """Pillar-canvas-like input: ``keep`` of the cells carry features, the rest are exactly zero."""
    x = seeded_uniform(seed, (n, c, h, w))
    occ = seeded_uniform(seed + 1000, (n, 1, h, w), 0.0, 1.0) < keep
    return (x * occ).astype(np.float32)


def w2c_attn_psm(seed, n, h, w, anchors=2):
    """Logits whose (smoothed) confidence straddles the thresholds of w2c_attn_configs on a band that moves with the agent."""
    psm = seeded_uniform(seed, (n, anchors, h, w), -9.0, -5.0)
    for a in range(n):
        psm[a, :, :, (5 * a) % w:(5 * a) % w + 8] += 3.5
    return psm


def loss_case(seed, B=2, H=12, W=20, A=2, C=7, pos_frac=0.02, empty_sample=None):
    """Seeded head maps + label dictionary for the loss tests: ~pos_frac of the anchors positive with a random class and
    regression target, one regression target NaN (the reference ignores those), optionally a sample without positives."""
    rng = np.random.default_rng(int(seed))
    psm = rng.normal(-2.0, 1.5, (B, A * C, H, W)).astype(np.float32)
    rm = rng.normal(0.0, 0.6, (B, A * 7, H, W)).astype(np.float32)
    obj = rng.normal(-1.0, 1.5, (B, A, H, W)).astype(np.float32)
    pos = (rng.uniform(size=(B, H, W, A)) < pos_frac).astype(np.float32)
    if empty_sample is not None:
        pos[empty_sample] = 0.0
    cls = np.where(pos > 0, rng.integers(1, C, size=pos.shape), 0).astype(np.int64)
    targets = (rng.normal(0.0, 0.5, (B, H, W, A * 7)) * np.repeat(pos, 7, axis=-1)).astype(np.float32)
    idx = np.argwhere(pos > 0)
    if len(idx):
        b, h, w, a = idx[0]
        targets[b, h, w, a * 7 + 2] = np.nan
    return {"psm": psm, "rm": rm, "obj": obj, "pos_equal_one": pos, "neg_equal_one": (1.0 - pos).astype(np.float32),
            "class_ids": cls, "targets": targets}


def submodule_psm(n=3, h=16, w=16):
    """Per-agent classification logits whose smoothed confidence straddles the 0.01 threshold: low everywhere,
    raised on a different part of the map for every agent."""
    psm = seeded_uniform(21, (n, 14, h, w), -9.0, -5.0)
    for a in range(n):
        psm[a, :, :, (3 * a) % w:(3 * a) % w + 6] += 3.5
    return psm
