"""ctypes binding of the C-ABI in include/airv2x_hip.h.

There is NO CPU fallback: if libairv2x_hip.so is missing and cannot be built the import of
this module raises, and every entry point raises on a non-zero status.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_uint64, c_void_p

# torch bundles its own libamdhip64; it MUST be the first HIP runtime mapped into the process so
# that libairv2x_hip.so binds to the same runtime instance (streams and device pointers are
# shared with torch).  Loading our library first gives two runtimes and "no ROCm-capable device".
import torch  # noqa: F401  (import order matters)

from . import build as _build

_LIB = None


class ConvDesc(Structure):
    """struct av2x_conv_desc (include/airv2x_hip.h)."""
    _fields_ = [(n, c_int32) for n in (
        "n", "h", "w", "cin", "in_ctot", "in_coff", "ho", "wo", "cout", "coutp", "out_ctot", "out_coff",
        "ks", "stride", "pad", "relu", "mode", "up", "tile", "sk_wgs", "act16")]


AV2X_CONV, AV2X_DECONV, AV2X_CONV_NCHW = 0, 1, 2

# name -> (restype, argtypes); the exported-symbol test walks this table against the header
SIGNATURES = {
    "av2x_version": (c_int32, []),
    "av2x_last_error": (c_char_p, []),
    "av2x_pillar_vfe_scatter": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "av2x_pillar_vfe": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p]),
    "av2x_pillar_vfe_scatter_dev": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "av2x_pillar_vfe_scatter_count": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p,
                                                c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "av2x_pillar_vfe_scatter_dev_count": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p,
                                                    c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "av2x_conv3x3s2_sparse": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                        c_int32, c_void_p]),
    "av2x_nonzero_slots_sum": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "av2x_voxelize_dummy_if_empty": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p]),
    "av2x_pillar_scatter": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "av2x_fill_zero": (c_int32, [c_void_p, c_uint64, c_void_p]),
    "av2x_linear_rows_workspace_bytes": (c_uint64, [c_int32, c_int32, c_int32]),
    "av2x_linear_rows": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                   c_uint64, c_void_p]),
    "av2x_linear_rows_backward": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                            c_void_p, c_void_p]),
    "av2x_unary_forward": (c_int32, [c_void_p, c_uint64, c_int32, c_void_p, c_void_p]),
    "av2x_unary_backward": (c_int32, [c_void_p, c_void_p, c_uint64, c_int32, c_void_p, c_void_p]),
    "av2x_add_act": (c_int32, [c_void_p, c_void_p, c_uint64, c_int32, c_void_p, c_void_p]),
    "av2x_gap_workspace_bytes": (c_uint64, [c_int32, c_int32, c_int32]),
    "av2x_gap": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p]),
    "av2x_channel_broadcast": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p]),
    "av2x_maxpool2d_backward": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                          c_void_p]),
    "av2x_dwconv2d_wgrad_workspace_bytes": (c_uint64, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "av2x_dwconv2d_wgrad": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                      c_void_p, c_void_p]),
    "av2x_lss_lift_pool_prob_backward": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                                   c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "av2x_softmax_channels_backward": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "av2x_resize_bilinear_backward_workspace_bytes": (c_uint64, [c_int32, c_int32, c_int32, c_int32]),
    "av2x_resize_bilinear_backward": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "av2x_lss_lift_pool_backward": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_int32, c_int32,
                                              c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p]),
    "av2x_gru_gate": (c_int32, [c_void_p, c_void_p, c_uint64, c_void_p, c_void_p]),
    "av2x_gru_gate_backward": (c_int32, [c_void_p, c_void_p, c_void_p, c_uint64, c_void_p, c_void_p, c_void_p]),
    "av2x_agent_argmax": (c_int32, [c_void_p, c_int32, c_uint64, c_void_p, c_void_p, c_void_p]),
    "av2x_agent_argmax_backward": (c_int32, [c_void_p, c_void_p, c_int32, c_uint64, c_void_p, c_void_p]),
    "av2x_when2com_fuse_backward_workspace_bytes": (c_uint64, [c_int32]),
    "av2x_when2com_fuse_backward": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_uint64, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_void_p]),
    "av2x_warp_affine_simple_backward": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "av2x_when2com_fuse": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_uint64, c_void_p, c_void_p, c_void_p]),
    "av2x_conv2d_wgrad_workspace_bytes": (c_uint64, [POINTER(ConvDesc)]),
    "av2x_conv2d_wgrad": (c_int32, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "av2x_pack_conv_weight": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "av2x_split3_koct": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "av2x_act_backward": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "av2x_bn_workspace_bytes": (c_uint64, [c_int64, c_int32]),
    "av2x_bn_stats": (c_int32, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "av2x_bn_finalize": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_float, c_int64, c_float, c_int32, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "av2x_bn_train_forward": (c_int32, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_float, c_float, c_int32, c_int32, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "av2x_affine_act": (c_int32, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "av2x_bn_backward": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p]),
    "av2x_pixel_attn_backward": (c_int32, [POINTER(c_void_p), c_int32, c_int32, c_int32, c_void_p, POINTER(c_void_p), c_void_p]),
    "av2x_pillar_train_workspace_bytes": (c_uint64, [c_int32]),
    "av2x_pillar_moments": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "av2x_pillar_vfe_backward": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                           c_void_p]),
    "av2x_pillar_vfe_backward_rows": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "av2x_pillar_gather": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "av2x_channel_sum_workspace_bytes": (c_uint64, [c_int64, c_int32]),
    "av2x_channel_sum": (c_int32, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    "av2x_generate_label": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_float, c_float, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "av2x_lss_pool_workspace_bytes": (c_uint64, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "av2x_lss_voxel_pool": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p]),
    "av2x_layernorm_backward_rows": (c_int32, [c_int64]),
    "av2x_layernorm_backward": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p, c_void_p, c_void_p]),
    "av2x_gelu": (c_int32, [c_void_p, c_void_p, c_void_p, c_uint64, c_void_p]),
    "av2x_scale_broadcast": (c_int32, [c_void_p, c_void_p, c_int32, c_uint64, c_float, c_void_p]),
    "av2x_dropout": (c_int32, [c_void_p, c_void_p, c_void_p, c_uint64, c_float, c_void_p]),
    "av2x_dropout_seeded": (c_int32, [c_void_p, c_void_p, c_void_p, c_uint64, c_float, c_uint64, c_void_p]),
    "av2x_fax_attention_backward_workspace_bytes": (c_uint64, [c_int32, c_int32, c_int32]),
    "av2x_fax_attention_backward": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                              c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "av2x_hgt_attention_backward": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "av2x_window_attention_backward_workspace_bytes": (c_uint64, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "av2x_window_attention_backward": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                                 c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "av2x_split_attn_backward_workspace_bytes": (c_uint64, [c_int32, c_int32]),
    "av2x_split_attn_sums": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "av2x_split_attn_backward": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "av2x_warp_affine_backward_workspace_bytes": (c_uint64, [c_int32, c_int32, c_int32, c_int32]),
    "av2x_warp_affine_backward": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "av2x_mean2": (c_int32, [c_void_p, c_void_p, c_void_p, c_uint64, c_void_p]),
    "av2x_cam_stem": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                c_void_p, c_void_p]),
    "av2x_dwconv2d": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "av2x_se_slabs": (c_int32, [c_int32]),
    "av2x_squeeze_excite_workspace_bytes": (c_uint64, [c_int32, c_int32, c_int32]),
    "av2x_squeeze_excite": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32,
                                      c_void_p]),
    "av2x_resize_bilinear": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                       c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p]),
    "av2x_maxpool2d": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "av2x_softmax_channels": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "av2x_lss_lift_pool": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_int32, c_int32,
                                     c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p]),
    "av2x_v2v_aggregate": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "av2x_points_in_boxes_cpu": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "av2x_points_in_boxes_gpu": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "av2x_roiaware_pool3d_workspace_bytes": (c_uint64, [c_int32, c_int32]),
    "av2x_roiaware_pool3d_forward": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                               c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "av2x_roiaware_pool3d_backward": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                                c_int32, c_void_p]),
    "av2x_bbox_overlaps": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int32]),
    "av2x_box_vote": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "av2x_conv2d": (c_int32, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "av2x_conv2d_res": (c_int32, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "av2x_conv2d_sk": (c_int32, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_uint64, c_void_p]),
    "av2x_conv2d_sk_workspace_bytes": (c_uint64, [c_int32, c_int32]),
    "av2x_wino_weight_bytes": (c_uint64, [c_int32, c_int32]),
    "av2x_wino_pack_weights": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "av2x_wino_x3_weight_bytes": (c_uint64, [c_int32, c_int32]),
    "av2x_wino_x3_pack_weights": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "av2x_wino4_x3_weight_bytes": (c_uint64, [c_int32, c_int32]),
    "av2x_wino4_x3_pack_weights": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "av2x_wino4_weight_bytes": (c_uint64, [c_int32, c_int32]),
    "av2x_wino4_pack_weights": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "av2x_eval_tp_fp": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_float, c_void_p, c_void_p, c_void_p,
                                  c_void_p]),
    "av2x_layernorm": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p]),
    "av2x_layernorm_stats": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p]),
    "av2x_conv2d_ln": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "av2x_fax_attention": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                     c_int32, c_int32, c_void_p]),
    "av2x_agent_mean": (c_int32, [c_void_p, c_void_p, c_int32, c_int64, c_void_p]),
    "av2x_layernorm_act": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_int32, c_void_p]),
    "av2x_warp_affine": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "av2x_warp_affine_simple": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "av2x_roi_mask": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "av2x_add_agent_vector": (c_int32, [c_void_p, c_void_p, c_int32, c_int64, c_int32, c_void_p]),
    "av2x_add_agent_vector_to": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_void_p]),
    "av2x_warp_affine_add": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "av2x_hgt_attention": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "av2x_hgt_attention_q": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                       c_void_p]),
    "av2x_window_attention": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                        c_int32, c_int32, c_void_p]),
    "av2x_split_attn_gap": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "av2x_split_attn_combine": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                          c_int32, c_void_p]),
    # bf16 activations (AMP mode of the V2X-ViT fusion; csrc/linear_bf16.hip, v2xvit.hip)
    "av2x_layernorm_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p]),
    "av2x_add_layernorm_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p]),
    "av2x_linear_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32,
                                   c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "av2x_ln_linear_bf16": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int32, c_int32,
                                      c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int64, c_void_p]),
    "av2x_hgt_attention_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                          c_void_p]),
    "av2x_window_attention_bf16": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                             c_int32, c_int32, c_void_p]),
    "av2x_combine_ln_linear_bf16": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                              c_float, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p,
                                              c_int32, c_int32, c_int64, c_void_p]),
    "av2x_window_attention_linear_bf16": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                                    c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "av2x_ln_qkv_window_attention_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "av2x_split_attn_gap_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "av2x_split_attn_combine_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                               c_int32, c_void_p]),
    "av2x_split_attn_combine_delta_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                                     c_int32, c_void_p]),
    "av2x_comm_mask": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32,
                                 c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "av2x_comm_mask_topk": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "av2x_apply_mask": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "av2x_mask_resize_bilinear": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "av2x_pixel_attn_fuse": (c_int32, [POINTER(c_void_p), c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "av2x_warp_fuse": (c_int32, [POINTER(c_void_p), c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "av2x_agent_max": (c_int32, [POINTER(c_void_p), c_int32, c_uint64, c_void_p, c_void_p]),
    "av2x_count_nonzero_where": (c_int32, [c_void_p, c_void_p, c_float, c_uint64, c_int32, c_void_p, c_void_p]),
    "av2x_pp_loss_workspace_bytes": (c_uint64, [c_int32, c_int32, c_int32]),
    "av2x_pp_loss": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                               c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "av2x_comm_rate": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "av2x_count_nonzero": (c_int32, [c_void_p, c_uint64, c_void_p, c_void_p]),
    "av2x_bf16_to_f32": (c_int32, [c_void_p, c_void_p, c_uint64, c_void_p]),
    "av2x_prepare_points_workspace_bytes": (c_uint64, [c_int32]),
    "av2x_prepare_points": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p,
                                      c_void_p]),
    "av2x_voxelize_workspace_bytes": (c_uint64, [c_int32, c_int32, c_int32, c_int32]),
    "av2x_voxelize": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p]),
    "av2x_prepare_voxelize": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32,
                                        c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "av2x_postprocess_workspace_bytes": (c_uint64, [c_int32, c_int32, c_int32, c_int32]),
    "av2x_postprocess_devt": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                        c_void_p, c_float, c_float, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p]),
    "av2x_postprocess": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                   c_void_p, c_float, c_float, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p]),
}


def lib_path():
    return _build.LIB


def load(build_if_missing=True):
    """Load (building in-tree first if needed and hipcc exists).  Raises if unavailable."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if build_if_missing:
        try:
            _build.build()
        except _build.HipccMissing as e:
            # no compiler on this box: a prebuilt library is the only option (the GPU box ships one with the snapshot)
            if not os.path.exists(path):
                raise RuntimeError(f"libairv2x_hip.so is missing and could not be built: {e}") from e
            if _build.needs_build():
                import warnings
                warnings.warn("libairv2x_hip.so is older than its sources and hipcc is not available: using the stale binary")
        # any other failure (compile / link error after a source edit) propagates: never run a stale binary silently
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: the HIP extension is required (no CPU fallback)")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.av2x_version() != 1:
        raise RuntimeError(f"ABI version mismatch: library {lib.av2x_version()}, binding 1")
    _LIB = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().av2x_last_error()
        raise RuntimeError(f"{what} failed: {msg.decode() if msg else 'unknown error'}")
