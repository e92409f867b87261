"""`sige_amd.hip` -- the MI355X native backend, bound over the C ABI of
libsige_hip.so (include/sige_hip.h).

This module is the equivalent of the reference's `sige.cuda` extension module
(sige/cuda/pybind_cuda.cpp:5-12): it exports the same five functions with the
same positional signatures on torch tensors

    gather, scatter, scatter_with_block_residual, scatter_gather, get_scatter_map

so `SIGEModule.load_runtime` (sige/nn/base.py:35-50) finds them under the
"cuda" key (torch-ROCm tensors report device.type == "cuda").  On top of that it
exports the MI355X-first extras the sige_amd.nn modules use: fused single-pass
scatter (tile tables), device reduce_mask, and the MFMA stacked-block conv.

There is NO fallback: if the shared library is missing or a call fails, a
RuntimeError is raised.  torch is used only for allocation and the stream.
"""
import ctypes
import os
import threading
from typing import Optional, Tuple

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SIGE_HIP_LIB", os.path.join(_PKG, "lib", "libsige_hip.so"))

ACT = {"identity": 0, "swish": 1}
# library extensions (include/sige_hip.h), accepted by the GauGAN helpers only
ACT_EXT = {"identity": 0, "relu": 2, "leaky": 3, "tanh": 4}

_c_int, _c_vp, _c_sz = ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t
_lib = None

_BC = [_c_vp, _c_int, _c_int, _c_int, _c_int]  # broadcastable operand: ptr + 4 dims

_SIGNATURES = {
    "sige_hip_version": (_c_int, []),
    "sige_hip_error_string": (ctypes.c_char_p, [_c_int]),
    "sige_hip_device_arch": (ctypes.c_char_p, []),
    "sige_hip_launch_count": (ctypes.c_int64, []),
    "sige_hip_last_launch_device": (_c_int, []),
    "sige_hip_preload": (_c_int, []),
    "sige_hip_gather_f32": (_c_int, [_c_vp] + [_c_int] * 6 + [_c_vp, _c_int] + _BC + _BC + [_c_int, _c_int, _c_vp, _c_vp]),
    "sige_hip_scatter_f32": (_c_int, [_c_vp, _c_vp] + [_c_int] * 10 + [_c_vp, _c_int] + _BC + [_c_vp, _c_vp]),
    "sige_hip_scatter_with_block_residual_f32": (
        _c_int, [_c_vp] * 4 + [_c_int] * 12 + [_c_vp, _c_int, _c_vp, _c_int, _c_vp, _c_vp]),
    "sige_hip_tile_table_i32": (_c_int, [_c_vp] + [_c_int] * 9 + [_c_vp, _c_vp]),
    "sige_hip_scatter_fused_f32": (_c_int, [_c_vp, _c_vp] + [_c_int] * 6 + [_c_vp, _c_int, _c_int, _c_int] + _BC + [_c_vp, _c_vp]),
    "sige_hip_scatter_with_block_residual_fused_f32": (
        _c_int, [_c_vp] * 4 + [_c_int] * 8 + [_c_vp, _c_int, _c_int, _c_int] * 2 + [_c_vp, _c_vp]),
    "sige_hip_scatter_map_i32": (_c_int, [_c_int] * 10 + [_c_vp, _c_int, _c_vp, _c_vp]),
    "sige_hip_scatter_gather_f32": (
        _c_int, [_c_vp, _c_vp] + [_c_int] * 8 + [_c_vp, _c_int, _c_vp] + _BC + _BC + [_c_int, _c_int, _c_vp, _c_vp]),
    "sige_hip_difference_mask_u8": (_c_int, [_c_vp, _c_vp] + [_c_int] * 3 + [ctypes.c_int64] * 3 + [ctypes.c_float, _c_vp, _c_vp]),
    "sige_hip_dilate_mask_u8": (_c_int, [_c_vp] + [_c_int] * 4 + [_c_vp, _c_vp]),
    "sige_hip_mask_pyramid_levels": (_c_int, [_c_int] * 4 + [_c_vp, _c_vp, _c_int]),
    "sige_hip_mask_pyramid_u8": (_c_int, [_c_vp] + [_c_int] * 6 + [ctypes.c_float] * 2 + [_c_vp, _c_sz, _c_vp, _c_vp]),
    "sige_hip_reduce_mask_capacity": (_c_int, [_c_int] * 6),
    "sige_hip_reduce_mask_i32": (_c_int, [_c_vp] + [_c_int] * 8 + [_c_vp, _c_int, _c_vp, _c_vp]),
    "sige_hip_block_conv_packed_size": (_c_sz, [_c_int] * 9),
    "sige_hip_block_conv_packed_size_f16c": (_c_sz, [_c_int] * 9),
    "sige_hip_block_conv_pack_f16c": (_c_int, [_c_vp] + [_c_int] * 4 + [_c_vp, _c_vp]),
    "sige_hip_block_conv_pack_f32": (_c_int, [_c_vp] + [_c_int] * 4 + [_c_vp, _c_vp]),
    "sige_hip_block_conv_f32": (_c_int, [_c_vp] + [_c_int] * 4 + [_c_vp, _c_vp] + [_c_int] * 5 + [_c_vp, _c_vp]),
    "sige_hip_gather_conv_f32": (
        _c_int, [_c_vp] + [_c_int] * 6 + [_c_vp, _c_int] + _BC + _BC + [_c_int, _c_vp, _c_vp] + [_c_int] * 5 + [_c_vp, _c_vp]),
    "sige_hip_scatter_gather_conv_f32": (
        _c_int, [_c_vp, _c_vp] + [_c_int] * 8 + [_c_vp, _c_int, _c_vp] + _BC + _BC + [_c_int, _c_vp, _c_vp]
        + [_c_int] * 5 + [_c_vp, _c_vp]),
    "sige_hip_gather_conv_nchw_f32": (
        _c_int, [_c_vp, _c_vp] + [_c_int] * 7 + [_c_vp, _c_int] + [_c_vp, _c_int, _c_int] * 2 + [_c_int, _c_vp, _c_vp]
        + [_c_int] * 7 + [_c_vp, _c_int, _c_int, _c_vp, _c_vp]),
    "sige_hip_group_norm_affine_workspace": (_c_sz, [_c_int] * 5),
    "sige_hip_group_norm_affine_f32": (_c_int, [_c_vp] + [_c_int] * 5 + [ctypes.c_float] + [_c_vp] * 6),
    "sige_hip_block_conv_direct_f32": (_c_int, [_c_vp] + [_c_int] * 4 + [_c_vp, _c_vp] + [_c_int] * 8 + [_c_vp, _c_vp]),
    "sige_hip_conv_pair_begin": (_c_int, []),
    "sige_hip_conv_pair_end": (_c_int, []),
    "sige_hip_conv_pairs_fused": (ctypes.c_int64, []),
    "sige_hip_attention_workspace": (_c_sz, [_c_int] * 3),
    "sige_hip_attention_f32": (_c_int, [_c_vp, _c_int, _c_int, _c_int, ctypes.c_float, _c_vp, _c_vp, _c_vp]),
    "sige_hip_copy_f32": (_c_int, [_c_vp, _c_vp, _c_sz, _c_vp]),
    # channels-last forms
    "sige_hip_block_conv_nhwc_f32": (_c_int, [_c_vp] + [_c_int] * 4 + [_c_vp, _c_vp] + [_c_int] * 5 + [_c_vp, _c_vp]),
    "sige_hip_gather_conv_nhwc_f32": (
        _c_int, [_c_vp, _c_vp] + [_c_int] * 7 + [_c_vp, _c_int] + [_c_vp, _c_int, _c_int] * 2 + [_c_int, _c_vp, _c_vp]
        + [_c_int] * 5 + [_c_int, _c_int, _c_int, _c_vp, _c_int, _c_int, _c_vp, _c_sz, _c_vp, _c_vp, _c_int, _c_int] + [_c_vp] * 6
        + [_c_vp, _c_vp]),
    "sige_hip_gather_conv_nhwc_v3_f32": (
        _c_int, [_c_vp, _c_vp] + [_c_int] * 7 + [_c_vp, _c_int] + [_c_vp, _c_int, _c_int] * 2 + [_c_int, _c_vp, _c_vp]
        + [_c_int] * 5 + [_c_int, _c_int, _c_int, _c_vp, _c_int, _c_int, _c_vp, _c_sz, _c_vp, _c_vp, _c_int, _c_int] + [_c_vp] * 6
        + [_c_vp, _c_int] + [_c_vp, _c_vp]),
    "sige_hip_scatter_gather_conv_scatter_nhwc_v3_f32": (
        _c_int, [_c_vp, _c_vp] + [_c_int] * 8 + [_c_vp, _c_int, _c_vp] + [_c_vp, _c_int, _c_int] * 2 + [_c_int, _c_vp, _c_vp]
        + [_c_int] * 3 + [_c_int, _c_int, _c_vp] + [_c_vp, _c_vp] + [_c_int] * 5 + [_c_vp] * 6 + [_c_vp, _c_int] + [_c_vp, _c_vp]),
    "sige_hip_conv_ksplit_hint": (_c_int, [_c_int] * 7),
    "sige_hip_block_conv_nhwc_f16c": (_c_int, [_c_vp] + [_c_int] * 4 + [_c_vp, _c_vp] + [_c_int] * 5 + [_c_vp, _c_vp]),
    "sige_hip_gather_conv_nhwc_f16c": (
        _c_int, [_c_vp, _c_vp] + [_c_int] * 7 + [_c_vp, _c_int] + [_c_vp, _c_int, _c_int] * 2 + [_c_int, _c_vp, _c_vp]
        + [_c_int] * 5 + [_c_int, _c_int, _c_int, _c_vp, _c_int, _c_int, _c_vp, _c_sz, _c_vp, _c_vp, _c_int, _c_int] + [_c_vp] * 6
        + [_c_vp, _c_vp]),
    "sige_hip_scatter_gather_conv_nhwc_f16c": (
        _c_int, [_c_vp, _c_vp] + [_c_int] * 8 + [_c_vp, _c_int, _c_vp] + [_c_vp, _c_int, _c_int] * 2 + [_c_int, _c_vp, _c_vp]
        + [_c_int] * 5 + [_c_vp, _c_vp]),
    "sige_hip_scatter_gather_conv_scatter_nhwc_f16c": (
        _c_int, [_c_vp, _c_vp] + [_c_int] * 8 + [_c_vp, _c_int, _c_vp] + [_c_vp, _c_int, _c_int] * 2 + [_c_int, _c_vp, _c_vp]
        + [_c_int] * 3 + [_c_int, _c_int, _c_vp] + [_c_vp, _c_vp] + [_c_int] * 5 + [_c_vp] * 6 + [_c_vp, _c_vp]),
    "sige_hip_scatter_gather_conv_nhwc_f32": (
        _c_int, [_c_vp, _c_vp] + [_c_int] * 8 + [_c_vp, _c_int, _c_vp] + [_c_vp, _c_int, _c_int] * 2 + [_c_int, _c_vp, _c_vp]
        + [_c_int] * 5 + [_c_vp, _c_vp]),
    "sige_hip_scatter_gather_conv_scatter_nhwc_f32": (
        _c_int, [_c_vp, _c_vp] + [_c_int] * 8 + [_c_vp, _c_int, _c_vp] + [_c_vp, _c_int, _c_int] * 2 + [_c_int, _c_vp, _c_vp]
        + [_c_int] * 3 + [_c_int, _c_int, _c_vp] + [_c_vp, _c_vp] + [_c_int] * 5 + [_c_vp] * 6 + [_c_vp, _c_vp]),
    "sige_hip_gather_nhwc_f32": (
        _c_int, [_c_vp] + [_c_int] * 6 + [_c_vp, _c_int] + [_c_vp, _c_int, _c_int] * 2 + [_c_int, _c_vp, _c_vp]),
    "sige_hip_scatter_gather_nhwc_f32": (
        _c_int, [_c_vp, _c_vp] + [_c_int] * 8 + [_c_vp, _c_int, _c_vp] + [_c_vp, _c_int, _c_int] * 2 + [_c_int, _c_vp, _c_vp]),
    "sige_hip_spade_modulate_nhwc_f32": (
        _c_int, [_c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_int] + [_c_vp, _c_int, _c_int] * 2 + [_c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_int]
        + [_c_int] * 6 + [_c_vp, _c_int, _c_int, ctypes.c_float, _c_vp, _c_vp]),
    "sige_hip_scatter_nhwc_f32": (
        _c_int, [_c_vp, _c_vp] + [_c_int] * 10 + [_c_vp, _c_vp, _c_int, _c_int, _c_int, _c_vp, _c_int, _c_vp, _c_vp]),
    "sige_hip_scatter_with_block_residual_nhwc_f32": (
        _c_int, [_c_vp] * 4 + [_c_int] * 12 + [_c_vp, _c_vp, _c_int, _c_int, _c_int] * 2 + [_c_int, _c_vp, _c_vp]),
    "sige_hip_group_norm_affine_nhwc_workspace": (_c_sz, [_c_int] * 5),
    "sige_hip_group_norm_affine_nhwc_f32": (_c_int, [_c_vp] + [_c_int] * 5 + [ctypes.c_float] + [_c_vp] * 6),
    "sige_hip_group_norm_affine_nhwc_bias_f32": (_c_int, [_c_vp] + [_c_int] * 5 + [ctypes.c_float] + [_c_vp] * 7),
    "sige_hip_conv3x3_small_cout_nhwc_f32": (
        _c_int, [_c_vp] + [_c_int] * 4 + [_c_vp, _c_int, _c_int] * 2 + [_c_int, _c_vp, _c_vp, _c_int, _c_vp, _c_vp]),
    "sige_hip_conv3x3_small_cin_nhwc_f32": (
        _c_int, [_c_vp] + [ctypes.c_int64] * 4 + [_c_int] * 4 + [_c_vp, _c_vp, _c_int, _c_vp, _c_vp]),
    "sige_hip_conv3x3_small_cin_tiles_nhwc_f32": (
        _c_int, [_c_vp] + [ctypes.c_int64] * 4 + [_c_int] * 4 + [_c_vp, _c_vp, _c_int, _c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp]),
    "sige_hip_attention_nhwc_f32": (_c_int, [_c_vp, _c_int, _c_int, _c_int, ctypes.c_float, _c_vp, _c_vp, _c_vp]),
    "sige_hip_attention_fused_workspace": (_c_sz, [_c_int] * 3),
    "sige_hip_attention_fused_nhwc_f32": (_c_int, [_c_vp, _c_int, _c_int, _c_int, ctypes.c_float, _c_vp, _c_vp, _c_vp]),
    # split fp16 operands (tile kernels)
    "sige_hip_block_conv_packed_size_f16x3": (_c_sz, [_c_int] * 9),
    "sige_hip_block_conv_pack_f16x3": (_c_int, [_c_vp] + [_c_int] * 4 + [_c_vp, _c_vp]),
    "sige_hip_block_conv_nhwc_f16x3": (_c_int, [_c_vp] + [_c_int] * 4 + [_c_vp, _c_vp] + [_c_int] * 5 + [_c_vp, _c_vp]),
    "sige_hip_gather_conv_nhwc_f16x3": (
        _c_int, [_c_vp, _c_vp] + [_c_int] * 7 + [_c_vp, _c_int] + [_c_vp, _c_int, _c_int] * 2 + [_c_int, _c_vp, _c_vp]
        + [_c_int] * 5 + [_c_int, _c_int, _c_int, _c_vp, _c_int, _c_int, _c_vp, _c_sz, _c_vp, _c_vp, _c_int, _c_int] + [_c_vp] * 6
        + [_c_vp, _c_vp]),
    "sige_hip_scatter_gather_conv_nhwc_f16x3": (
        _c_int, [_c_vp, _c_vp] + [_c_int] * 8 + [_c_vp, _c_int, _c_vp] + [_c_vp, _c_int, _c_int] * 2 + [_c_int, _c_vp, _c_vp]
        + [_c_int] * 5 + [_c_vp, _c_vp]),
    "sige_hip_scatter_gather_conv_scatter_nhwc_f16x3": (
        _c_int, [_c_vp, _c_vp] + [_c_int] * 8 + [_c_vp, _c_int, _c_vp] + [_c_vp, _c_int, _c_int] * 2 + [_c_int, _c_vp, _c_vp]
        + [_c_int] * 3 + [_c_int, _c_int, _c_vp] + [_c_vp, _c_vp] + [_c_int] * 5 + [_c_vp] * 6 + [_c_vp, _c_vp]),
    "sige_hip_affine_act_nhwc_f32": (_c_int, [_c_vp] + [_c_int] * 4 + [_c_vp, _c_vp, _c_int, _c_int, _c_vp, _c_vp]),
    # dense layers on the fp16 matrix cores (conv_wide)
    "sige_hip_wide_conv_supported": (_c_int, [_c_int] * 5),
    "sige_hip_wide_conv_packed_size": (_c_sz, [_c_int] * 5),
    "sige_hip_wide_conv_pack": (_c_int, [_c_vp] + [_c_int] * 6 + [_c_vp, _c_vp]),
    "sige_hip_wide_conv_workspace": (_c_sz, [_c_int] * 8),
    "sige_hip_wide_conv_nhwc": (
        _c_int, [_c_vp, _c_vp] + [_c_int] * 6 + [_c_vp, _c_vp, _c_int, _c_int] + [_c_vp, _c_int, _c_int, _c_vp, _c_int, _c_int, _c_int]
        + [_c_vp, _c_vp, _c_vp, _c_int] + [_c_vp] * 6 + [_c_vp, _c_sz, _c_vp, _c_vp, _c_vp]),
    "sige_hip_release_graph_tickets": (_c_int, []),
    "sige_hip_channel_stats_tiles": (_c_int, [_c_int, _c_int]),
    "sige_hip_channel_stats_nhwc_f32": (_c_int, [_c_vp] + [_c_int] * 4 + [_c_vp, _c_vp]),
    "sige_hip_group_norm_affine_from_stats_f32": (
        _c_int, [_c_vp, _c_int, _c_int, _c_int, _c_vp, _c_int, _c_int, _c_int, _c_int, _c_int, ctypes.c_float] + [_c_vp] * 6),
    "sige_hip_attention_tokens_supported": (_c_int, [_c_int] * 4),
    "sige_hip_attention_tokens_f32": (_c_int, [_c_vp] * 3 + [_c_int] * 5 + [ctypes.c_float, _c_vp, _c_vp]),
    # fp16-stored caches
    "sige_hip_gather_nhwc_f16": (
        _c_int, [_c_vp] + [_c_int] * 6 + [_c_vp, _c_int] + [_c_vp, _c_int, _c_int] * 2 + [_c_int, _c_vp, _c_vp]),
    "sige_hip_scatter_gather_nhwc_f16": (
        _c_int, [_c_vp, _c_vp] + [_c_int] * 8 + [_c_vp, _c_int, _c_vp] + [_c_vp, _c_int, _c_int] * 2 + [_c_int, _c_vp, _c_vp]),
    "sige_hip_scatter_nhwc_f16": (
        _c_int, [_c_vp, _c_vp] + [_c_int] * 10 + [_c_vp, _c_vp, _c_int, _c_int, _c_int, _c_vp, _c_int, _c_vp, _c_vp]),
    "sige_hip_scatter_with_block_residual_nhwc_f16": (
        _c_int, [_c_vp] * 4 + [_c_int] * 12 + [_c_vp, _c_vp, _c_int, _c_int, _c_int] * 2 + [_c_int, _c_vp, _c_vp]),
    "sige_hip_affine_act_nhwc_f16": (_c_int, [_c_vp] + [_c_int] * 4 + [_c_vp, _c_vp, _c_int, _c_int, _c_vp, _c_int, _c_vp]),
    "sige_hip_convert_f16_f32": (_c_int, [_c_vp, _c_vp, _c_sz, _c_vp]),
    "sige_hip_convert_f32_f16": (_c_int, [_c_vp, _c_vp, _c_sz, _c_vp]),
    "sige_hip_scatter_gather_conv_nhwc_c16": (
        _c_int, [_c_int, _c_vp, _c_vp] + [_c_int] * 8 + [_c_vp, _c_int, _c_vp] + [_c_vp, _c_int, _c_int] * 2 + [_c_int, _c_vp, _c_vp]
        + [_c_int] * 5 + [_c_vp, _c_vp]),
    "sige_hip_scatter_gather_conv_scatter_nhwc_c16": (
        _c_int, [_c_int, _c_vp, _c_vp] + [_c_int] * 8 + [_c_vp, _c_int, _c_vp] + [_c_vp, _c_int, _c_int] * 2 + [_c_int, _c_vp, _c_vp]
        + [_c_int] * 3 + [_c_int, _c_int, _c_vp, _c_int] + [_c_vp, _c_vp] + [_c_int] * 5 + [_c_vp] * 6 + [_c_vp, _c_vp]),
    "sige_hip_conv3x3_small_cout_act_nhwc_f32": (_c_int, [_c_vp] + [_c_int] * 5 + [ctypes.c_float, _c_vp, _c_vp, _c_int, _c_int, _c_vp, _c_vp]),
    "sige_hip_resize_nearest_nhwc_f32": (_c_int, [_c_vp] + [_c_int] * 6 + [_c_vp, _c_vp]),
    "sige_hip_act_split_nhwc_f32": (_c_int, [_c_vp, ctypes.c_int64, _c_int, _c_int, ctypes.c_int64, _c_int, ctypes.c_float, _c_vp, _c_vp]),
    "sige_hip_scatter_gather_split_nhwc_f32": (
        _c_int, [_c_vp, _c_vp] + [_c_int] * 8 + [_c_vp, _c_int, _c_vp, _c_int, ctypes.c_float, _c_int, ctypes.c_int64, _c_vp, _c_vp]),
    "sige_hip_spade_modulate_dense_nhwc_f32": (_c_int, [_c_vp, _c_vp, _c_vp, _c_int, _c_vp] + [_c_int] * 5 + [ctypes.c_float, _c_vp, _c_vp]),
    "sige_hip_block_conv_nhwc_keyed": (_c_int, [_c_int, _c_vp, _c_vp] + [_c_int] * 5 + [_c_vp, _c_vp] + [_c_int] * 5 + [_c_vp, _c_vp]),
    "sige_hip_tile_conv3_supported": (_c_int, [_c_int] * 3),
    "sige_hip_tile_conv3_nhwc_f32": (
        _c_int, [_c_int, _c_vp, _c_vp] + [_c_int] * 6 + [_c_vp, _c_int, _c_vp, _c_int, _c_int] + [_c_vp, _c_vp, _c_int, _c_int]
        + [_c_vp, _c_vp, _c_int] + [_c_int] * 5 + [_c_vp] + [_c_vp, _c_vp] + [_c_int] * 5 + [_c_vp, _c_vp, _c_int] + [_c_vp] * 6 + [_c_vp, _c_vp]),
    "sige_hip_tile_conv3_nhwc_f16c": (
        _c_int, [_c_int, _c_vp, _c_vp, _c_int] + [_c_int] * 6 + [_c_vp, _c_int, _c_vp, _c_int, _c_int] + [_c_vp, _c_vp, _c_int, _c_int]
        + [_c_vp, _c_vp, _c_int] + [_c_int] * 5 + [_c_vp, _c_int] + [_c_vp, _c_vp] + [_c_int] * 5 + [_c_vp, _c_vp, _c_int] + [_c_vp] * 6 + [_c_vp, _c_vp]),
    "sige_hip_gather_conv_nhwc_v3_f16c": (
        _c_int, [_c_vp, _c_vp] + [_c_int] * 7 + [_c_vp, _c_int] + [_c_vp, _c_int, _c_int] * 2 + [_c_int, _c_vp, _c_vp]
        + [_c_int] * 5 + [_c_int, _c_int, _c_int, _c_vp, _c_int, _c_int, _c_vp, _c_sz, _c_vp, _c_vp, _c_int, _c_int] + [_c_vp] * 6
        + [_c_vp, _c_int] + [_c_vp, _c_vp]),
    "sige_hip_scatter_gather_conv_scatter_nhwc_v3_f16c": (
        _c_int, [_c_vp, _c_vp, _c_int] + [_c_int] * 8 + [_c_vp, _c_int, _c_vp] + [_c_vp, _c_int, _c_int] * 2 + [_c_int, _c_vp, _c_vp]
        + [_c_int] * 3 + [_c_int, _c_int, _c_vp, _c_int] + [_c_vp, _c_vp] + [_c_int] * 5 + [_c_vp] * 6 + [_c_vp, _c_int] + [_c_vp, _c_vp]),
    "sige_hip_add_layer_norm_tokens_f32": (_c_int, [_c_vp] * 5 + [ctypes.c_int64, _c_int, ctypes.c_float, _c_vp, _c_vp, _c_vp]),
    "sige_hip_geglu_tokens_f32": (_c_int, [_c_vp, ctypes.c_int64, _c_int, _c_vp, _c_vp]),
    "sige_hip_add_bias_tokens_f32": (_c_int, [_c_vp, _c_vp, _c_vp, ctypes.c_int64, _c_int, _c_vp, _c_vp]),
    "sige_hip_set_edit_batch": (_c_int, [_c_int]),
    "sige_hip_get_edit_batch": (_c_int, []),
    # launch plans (csrc/plan.hip; host side: sige_amd/plan.py)
    "sige_hip_plan_create": (_c_vp, []),
    "sige_hip_plan_destroy": (_c_int, [_c_vp]),
    "sige_hip_plan_begin": (_c_int, [_c_vp, _c_int, _c_int]),
    "sige_hip_plan_end": (_c_int, [_c_vp]),
    "sige_hip_plan_recording": (_c_int, []),
    "sige_hip_plan_shape_bound": (_c_int, [_c_vp]),
    "sige_hip_plan_calls": (_c_int, [_c_vp, _c_int]),
    "sige_hip_plan_new_slots": (_c_int, [_c_vp, _c_int]),
    "sige_hip_plan_bind_ptr": (_c_int, [_c_vp, _c_vp, _c_int]),
    "sige_hip_plan_set_slot": (_c_int, [_c_vp, _c_int, _c_int]),
    "sige_hip_plan_bind_const": (_c_int, [_c_vp, _c_vp]),
    "sige_hip_plan_unbound": (_c_int, [_c_vp]),
    "sige_hip_plan_truncate": (_c_int, [_c_vp, _c_int, _c_int]),
    "sige_hip_plan_get_slots": (_c_int, [_c_vp, _c_vp, _c_int]),
    "sige_hip_plan_record_readback": (_c_int, [_c_vp, _c_vp, _c_int, _c_int]),
    "sige_hip_plan_run": (_c_int, [_c_vp, _c_int, _c_vp]),
}

EXPORTS = tuple(_SIGNATURES)  # every symbol include/sige_hip.h declares for the PRODUCT library

# measurement builds only (-DSIGE_HIP_TUNING: lib/libsige_hip_tuning.so and the probe builds): the one pair of knob entry points
_TUNING_SIGNATURES = {
    "sige_hip_tuning_set": (_c_int, [_c_int, _c_int]),
    "sige_hip_tuning_get": (_c_int, [_c_int]),
}
TUNING_LIB_PATH = os.path.join(_PKG, "lib", "libsige_hip_tuning.so")
# include/sige_hip.h: SIGE_HIP_TUNE_*
TUNE = {"conv_tile_mt": 0, "conv_tile_nb": 1, "conv_waves": 2, "conv_large_grid_nb1": 3, "conv_ksplit": 4, "conv_ksplit_second_pass": 5,
        "gather_one_tile_rows": 6, "scatter_gather_form": 7, "small_cout_scalar": 8, "wide_ksplit": 9, "attention_form": 10, "tile3_f16_tpw4_min": 11, "tile3_f16_pair_min": 12, "tile3_f16_sparse_min": 13}


# how many guarded entry-point calls had to switch HIP's current device to the tensor's ("switched") and how many found it current
# ("direct"); conv_pair begin / end count under "pair_switched".  A debug aid: tests/test_gpu_round4.py checks the guard with it.
GUARD_STATS = {"switched": 0, "direct": 0, "pair_switched": 0}


class _Guarded:
    """A C entry point that launches on the device of the tensor whose stream was just taken.

    The C ABI receives raw pointers and a stream handle and launches on HIP's *current* device
    (the reference's sige/cuda wrappers have no device guard either).  Every
    binding below evaluates `_stream(t)` as its last argument; that records `t`'s device, and
    the call is wrapped in `torch.cuda.device(...)` when it is not the current one -- a model on
    cuda:1 in a process whose current device is cuda:0 launches on cuda:1."""

    __slots__ = ("fn",)

    def __init__(self, fn):
        self.fn = fn

    def __call__(self, *args):
        rec = getattr(_plan_tls, "rec", None)
        if rec is not None:
            return self._recorded(rec, args)
        dev, _tls.pending_device = getattr(_tls, "pending_device", None), None
        if dev is not None and dev != (_raw_device() if _raw_device is not None else torch.cuda.current_device()):
            GUARD_STATS["switched"] += 1
            with torch.cuda.device(dev):
                return self.fn(*args)
        GUARD_STATS["direct"] += 1
        return self.fn(*args)

    def _recorded(self, rec, args):
        """While a launch plan records: the entry point's hook stores the call BEFORE the entry point validates it, so a call that
        returns an error status (a probe the wrapper then routes elsewhere) is taken out of the plan again."""
        L = lib()
        n0 = L.sige_hip_plan_calls(rec.handle, rec.section)
        dev, _tls.pending_device = getattr(_tls, "pending_device", None), None
        if dev is not None and dev != (_raw_device() if _raw_device is not None else torch.cuda.current_device()):
            GUARD_STATS["switched"] += 1
            with torch.cuda.device(dev):
                status = self.fn(*args)
        else:
            GUARD_STATS["direct"] += 1
            status = self.fn(*args)
        if status != 0 and n0 >= 0 and L.sige_hip_plan_calls(rec.handle, rec.section) > n0:
            L.sige_hip_plan_truncate(rec.handle, rec.section, n0)
        return status


class _Lib:
    pass


# per thread (one thread per GPU in one process is a supported set-up): the device of the tensor whose stream the binding
# being evaluated just took -- consumed by the _Guarded call that follows in the same thread
_tls = threading.local()


def _load(path):
    if not os.path.isfile(path):
        raise RuntimeError(
            "sige_amd: %s not found -- the HIP extension is not built "
            "(run `python -m sige_amd.build%s`); there is no CPU fallback." % (path, " --tuning" if "tuning" in os.path.basename(path) else ""))
    handle = ctypes.CDLL(path)
    table = _Lib()
    sigs = dict(_SIGNATURES)
    if hasattr(handle, "sige_hip_tuning_set"):
        sigs.update(_TUNING_SIGNATURES)
    for name, (res, args) in sigs.items():
        fn = getattr(handle, name)
        fn.restype, fn.argtypes = res, args
        guarded = args and args[-1] is _c_vp and res is _c_int and (not name.startswith("sige_hip_plan_") or name == "sige_hip_plan_run")
        setattr(table, name, _Guarded(fn) if guarded else fn)
    table.handle = handle
    table.path = path
    table.has_tuning = hasattr(handle, "sige_hip_tuning_set")
    table.preloaded = set()
    table.preloaded_units = {}  # device -> code objects sige_hip_preload touched there
    return table


_libs = {}


def lib():
    """Load libsige_hip.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        _lib = _libs.setdefault(LIB_PATH, None) or _load(LIB_PATH)
        _libs[LIB_PATH] = _lib
    return _lib


def use_library(path: Optional[str] = None):
    """Switch this process's bindings to another build of the library (None = the product build); returns the previous path.
    Objects a library owns (launch plans, the held conv of a pair) must not cross the switch."""
    global _lib
    prev = _lib.path if _lib is not None else LIB_PATH
    path = path or LIB_PATH
    if _libs.get(path) is None:
        _libs[path] = _load(path)
    _lib = _libs[path]
    return prev


class tuning_build:
    """`with hip.tuning_build():` -- run on lib/libsige_hip_tuning.so (the measurement build that exports sige_hip_tuning_set:
    include/sige_hip.h); every knob is reset to its default on exit and the product library comes back."""

    def __init__(self, path: Optional[str] = None):
        self.path = path or os.environ.get("SIGE_HIP_TUNING_LIB", TUNING_LIB_PATH)

    def __enter__(self):
        self.prev = use_library(self.path)
        return lib()

    def __exit__(self, *exc):
        try:
            tuning_reset()
        finally:
            use_library(self.prev)
        return False


def tuning_set(key, value: int):
    """sige_hip_tuning_set (measurement builds only): `key` a SIGE_HIP_TUNE_* number or its lower-case name in TUNE."""
    L = lib()
    if not L.has_tuning:
        raise RuntimeError("sige_amd.hip: dispatch knobs exist only in the measurement build -- `with hip.tuning_build(): ...` "
                           "(lib/libsige_hip_tuning.so, `python -m sige_amd.build --tuning`); the product library has none")
    _check(L.sige_hip_tuning_set(TUNE[key] if isinstance(key, str) else int(key), int(value)), "tuning_set(%s, %s)" % (key, value))


def tuning_get(key) -> int:
    L = lib()
    if not L.has_tuning:
        raise RuntimeError("sige_amd.hip: dispatch knobs exist only in the measurement build (hip.tuning_build())")
    return int(L.sige_hip_tuning_get(TUNE[key] if isinstance(key, str) else int(key)))


_TUNE_DEFAULTS = {"conv_large_grid_nb1": -1}


def tuning_reset():
    if lib().has_tuning:
        for k in TUNE:
            tuning_set(k, _TUNE_DEFAULTS.get(k, 0))


def preload(device: Optional[int] = None) -> int:
    """sige_hip_preload on `device` (default: the current one): load every code object of the library now instead of on the
    first launch that needs it.  Called by the first launch on a device (see _stream); returns the number touched."""
    L = lib()
    if device is None:
        device = _raw_device() if _raw_device is not None else torch.cuda.current_device()
    if device in L.preloaded:
        return 0
    L.preloaded.add(device)
    if os.environ.get("SIGE_HIP_NO_PRELOAD"):
        return 0
    if torch.cuda.is_current_stream_capturing():
        L.preloaded.discard(device)  # (loading is not capturable; the kernels of a capture were warmed up before it)
        return 0
    with torch.cuda.device(device):
        n = int(L.sige_hip_preload())
    if n < 0:
        raise RuntimeError("sige_amd.hip: sige_hip_preload failed: %s" % L.sige_hip_error_string(n).decode())
    L.preloaded_units[device] = n
    return n


def set_edit_batch(E: int):
    """Stacked edits (include/sige_hip.h: sige_hip_set_edit_batch; sige_amd/stacked.py): the tensors handed to the library from
    now on (this thread) are E images stacked along H."""
    _check(lib().sige_hip_set_edit_batch(int(E)), "set_edit_batch")


def get_edit_batch() -> int:
    return int(lib().sige_hip_get_edit_batch())


def last_launch_device() -> int:
    """HIP's current device at this library's most recent kernel launch (-1: none yet)."""
    return int(lib().sige_hip_last_launch_device())


def launch_count() -> int:
    """Kernel launches libsige_hip.so has issued in this process so far (bench.py: launches per forward)."""
    return int(lib().sige_hip_launch_count())


def available() -> bool:
    return os.path.isfile(LIB_PATH)


UNSUPPORTED = -2  # SIGE_HIP_EUNSUPPORTED
KSPLIT = True     # tools/conv_floor.py sets this to False to time the convs without the cross-workgroup K split


def _check(status: int, what: str):
    if status != 0:
        raise RuntimeError("sige_amd.hip.%s failed: %s" % (what, lib().sige_hip_error_string(status).decode()))


# (the raw-handle getters of torch._C are what torch.cuda.current_stream / current_device wrap; a sparse forward calls them
#  ~120 times, and the Stream-object round trip was 5 % of its host time)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream(t: torch.Tensor) -> int:
    """Stream handle for a launch on `t`'s device (and note that device for the guard, see _Guarded)."""
    idx = t.device.index
    _tls.pending_device = idx
    if _lib is not None and idx not in _lib.preloaded and idx is not None:
        preload(idx)
    if _raw_stream is not None and idx is not None:
        return _raw_stream(idx)
    return torch.cuda.current_stream(t.device).cuda_stream


def _req(t: torch.Tensor, dtype, name: str, dim: Optional[int] = 4):
    if not t.is_cuda:
        raise RuntimeError("sige_amd.hip: `%s` must live on the GPU (got %s)" % (name, t.device))
    if t.dtype != dtype:
        raise NotImplementedError("sige_amd.hip: `%s` must be %s (got %s)" % (name, dtype, t.dtype))
    if dim is not None and t.dim() != dim:
        raise NotImplementedError("sige_amd.hip: `%s` must have %d dims (got %d)" % (name, dim, t.dim()))
    return t if t.is_contiguous() else t.contiguous()


def _bc(t: Optional[torch.Tensor], name: str):
    """(ptr, B, C, H, W) of an optional broadcastable fp32 operand."""
    if t is None:
        return (None, 0, 0, 0, 0), None
    t = _req(t, torch.float32, name)
    return (t.data_ptr(), t.shape[0], t.shape[1], t.shape[2], t.shape[3]), t


def _act(name: str) -> int:
    try:
        return ACT[name]
    except KeyError:
        # the reference's native code hits __builtin_unreachable here (sige/common.cpp:17-23)
        raise ValueError("Unknown activation: [%s]!!!" % name)


# --------------------------------------------------------------------------
# Launch-plan recording (sige_amd/plan.py drives it; csrc/plan.hpp replays).  While a plan records on this thread:
#   * index lists come out of PERSISTENT buffers sized for every candidate tile (a later mask writes the same memory), bound
#     to a slot of the plan; tile tables built from them are bound to the same slot;
#   * tensors whose leading dimension is a tile count are allocated for the largest count and returned as a view;
#   * every tensor a recorded call of the mask section points at is kept alive by the plan (the forward section is recorded
#     under a hipGraph capture, whose memory pool does that).
# --------------------------------------------------------------------------
_plan_tls = threading.local()


def plan_recorder():
    """The sige_amd.plan._Recorder of this thread while a launch plan records, else None."""
    return getattr(_plan_tls, "rec", None)


def _plan_keep(*tensors):
    rec = plan_recorder()
    if rec is not None:
        rec.keep.extend(t for t in tensors if t is not None)


def base_ptr(t: torch.Tensor) -> int:
    """Address of `t`'s first element also when `t` is EMPTY (Tensor.data_ptr() of an empty view is 0; an index list of zero
    active tiles is still a view of its persistent buffer)."""
    return t.untyped_storage().data_ptr() + t.storage_offset() * t.element_size()


def _tile_capacity(idx: torch.Tensor) -> Optional[int]:
    """While a plan records: how many tiles the persistent buffer behind the index list `idx` can hold (None: not one of the
    plan's index lists -- e.g. the all-tiles list of a dense layer, whose count never changes)."""
    rec = plan_recorder()
    if rec is None:
        return None
    info = rec.idx_info.get(base_ptr(idx))
    return None if info is None else info[1]


def _empty_tiles_cl(B: int, idx: torch.Tensor, C: int, R: int, S: int, device) -> torch.Tensor:
    """Channels-last tile tensor [B * N, C, R, S] for the tiles of `idx`; while a plan records, backed by memory for every
    candidate tile (the same pointer serves any later mask)."""
    N = idx.shape[0]
    cap = _tile_capacity(idx)
    if cap is None or cap <= N:
        return _empty_cl((B * N, C, R, S), device)
    return _empty_cl((B * cap, C, R, S), device)[:B * N]


# --------------------------------------------------------------------------
# The five reference-signature entry points (sige/cuda/pybind_cuda.cpp:5-12)
# --------------------------------------------------------------------------
def gather(x, bSizeH, bSizeW, activeIndices, scale=None, shift=None, activationName="identity",
           activationFirst=False):
    x = _req(x, torch.float32, "x")
    idx = _req(activeIndices, torch.int32, "activeIndices", 2)
    (sa, s_keep), (ta, t_keep) = _bc(scale, "scale"), _bc(shift, "shift")
    B, C, H, W = x.shape
    N = idx.shape[0]
    out = torch.empty((B * N, C, bSizeH, bSizeW), dtype=torch.float32, device=x.device)
    _check(lib().sige_hip_gather_f32(x.data_ptr(), B, C, H, W, bSizeH, bSizeW, idx.data_ptr(), N, *sa, *ta,
                                     _act(activationName), int(bool(activationFirst)), out.data_ptr(),
                                     _stream(x)), "gather")
    return out


def scatter(x, y, offsetH, offsetW, strideH, strideW, activeIndices, residual=None):
    x = _req(x, torch.float32, "x")
    y = _req(y, torch.float32, "y")
    idx = _req(activeIndices, torch.int32, "activeIndices", 2)
    ra, r_keep = _bc(residual, "residual")
    B, C, H, W = y.shape
    out = torch.empty_like(y)
    _check(lib().sige_hip_scatter_f32(x.data_ptr(), y.data_ptr(), B, C, H, W, x.shape[2], x.shape[3],
                                      offsetH, offsetW, strideH, strideW, idx.data_ptr(), idx.shape[0], *ra,
                                      out.data_ptr(), _stream(y)), "scatter")
    return out


def scatter_with_block_residual(x0, y0, x1, y1, offsetH, offsetW, strideH, strideW, activeIndices0,
                                activeIndices1):
    x0, y0 = _req(x0, torch.float32, "x0"), _req(y0, torch.float32, "y0")
    x1, y1 = _req(x1, torch.float32, "x1"), _req(y1, torch.float32, "y1")
    i0 = _req(activeIndices0, torch.int32, "activeIndices0", 2)
    i1 = _req(activeIndices1, torch.int32, "activeIndices1", 2)
    if y1.shape != y0.shape:
        raise RuntimeError("scatter_with_block_residual: y1 %s must match y0 %s" % (tuple(y1.shape), tuple(y0.shape)))
    B, C, H, W = y0.shape
    out = torch.empty_like(y0)
    _check(lib().sige_hip_scatter_with_block_residual_f32(
        x0.data_ptr(), y0.data_ptr(), x1.data_ptr(), y1.data_ptr(), B, C, H, W,
        x0.shape[2], x0.shape[3], x1.shape[2], x1.shape[3], offsetH, offsetW, strideH, strideW,
        i0.data_ptr(), i0.shape[0], i1.data_ptr(), i1.shape[0], out.data_ptr(), _stream(y0)),
        "scatter_with_block_residual")
    return out


def get_scatter_map(H, W, bSizeH, bSizeW, kSizeH, kSizeW, offsetH, offsetW, strideH, strideW, activeIndices):
    idx = _req(activeIndices, torch.int32, "activeIndices", 2)
    out = torch.empty((H, W, 3), dtype=torch.int32, device=idx.device)
    _check(lib().sige_hip_scatter_map_i32(H, W, bSizeH, bSizeW, kSizeH, kSizeW, offsetH, offsetW, strideH, strideW,
                                          idx.data_ptr(), idx.shape[0], out.data_ptr(), _stream(idx)),
           "get_scatter_map")
    rec = plan_recorder()
    if rec is not None:  # (spade_modulate takes the tile count of a scatter MAP's list: looked up under the map's pointer)
        rec.bind_alias(out, idx)
        rec.keep.append(out)
    return out


def scatter_gather(x, y, bSizeH, bSizeW, activeIndices, scatterMap, scale=None, shift=None,
                   activationName="identity", activationFirst=False):
    x, y = _req(x, torch.float32, "x"), _req(y, torch.float32, "y")
    idx = _req(activeIndices, torch.int32, "activeIndices", 2)
    smap = _req(scatterMap, torch.int32, "scatterMap", 3)
    (sa, s_keep), (ta, t_keep) = _bc(scale, "scale"), _bc(shift, "shift")
    B, C, H, W = y.shape
    if x.shape[1] != C:
        raise RuntimeError("scatter_gather: channel mismatch x %d vs y %d" % (x.shape[1], C))
    N = idx.shape[0]
    out = torch.empty((B * N, C, bSizeH, bSizeW), dtype=torch.float32, device=y.device)
    _check(lib().sige_hip_scatter_gather_f32(x.data_ptr(), y.data_ptr(), B, C, H, W, x.shape[2], x.shape[3],
                                             bSizeH, bSizeW, idx.data_ptr(), N, smap.data_ptr(), *sa, *ta,
                                             _act(activationName), int(bool(activationFirst)), out.data_ptr(),
                                             _stream(y)), "scatter_gather")
    return out


# --------------------------------------------------------------------------
# MI355X-first extras
# --------------------------------------------------------------------------
def tile_table(activeIndices, offset: Tuple[int, int], stride: Tuple[int, int], out_tile: Tuple[int, int],
               out_res: Tuple[int, int]) -> torch.Tensor:
    """[gH,gW] int32: which tile of `activeIndices` covers each out_tile-sized cell
    of an out_res tensor (-1 = none).  Valid for reduce_mask index lists."""
    idx = _req(activeIndices, torch.int32, "activeIndices", 2)
    gH, gW = -(-out_res[0] // out_tile[0]), -(-out_res[1] // out_tile[1])
    table = torch.empty((gH, gW), dtype=torch.int32, device=idx.device)
    _check(lib().sige_hip_tile_table_i32(idx.data_ptr(), idx.shape[0], offset[0], offset[1], stride[0], stride[1],
                                         out_tile[0], out_tile[1], gH, gW, table.data_ptr(), _stream(idx)),
           "tile_table")
    rec = plan_recorder()
    if rec is not None:  # (calls that take a tile TABLE and a count look the count up under the table's pointer)
        rec.bind_alias(table, idx)
        rec.keep.append(table)
    return table


def scatter_fused(x, y, table, num_active: int, residual=None):
    x, y = _req(x, torch.float32, "x"), _req(y, torch.float32, "y")
    table = _req(table, torch.int32, "table", 2)
    ra, r_keep = _bc(residual, "residual")
    B, C, H, W = y.shape
    out = torch.empty_like(y)
    _check(lib().sige_hip_scatter_fused_f32(x.data_ptr(), y.data_ptr(), B, C, H, W, x.shape[2], x.shape[3],
                                            table.data_ptr(), table.shape[0], table.shape[1], num_active, *ra,
                                            out.data_ptr(), _stream(y)), "scatter_fused")
    return out


def scatter_with_block_residual_fused(x0, y0, x1, y1, table0, n0: int, table1, n1: int):
    x0, y0 = _req(x0, torch.float32, "x0"), _req(y0, torch.float32, "y0")
    x1, y1 = _req(x1, torch.float32, "x1"), _req(y1, torch.float32, "y1")
    table0, table1 = _req(table0, torch.int32, "table0", 2), _req(table1, torch.int32, "table1", 2)
    B, C, H, W = y0.shape
    out = torch.empty_like(y0)
    _check(lib().sige_hip_scatter_with_block_residual_fused_f32(
        x0.data_ptr(), y0.data_ptr(), x1.data_ptr(), y1.data_ptr(), B, C, H, W,
        x0.shape[2], x0.shape[3], x1.shape[2], x1.shape[3],
        table0.data_ptr(), table0.shape[0], table0.shape[1], n0,
        table1.data_ptr(), table1.shape[0], table1.shape[1], n1, out.data_ptr(), _stream(y0)),
        "scatter_with_block_residual_fused")
    return out


def reduce_mask(mask: torch.Tensor, block_size, stride, padding) -> torch.Tensor:
    """Device restatement of sige.utils.reduce_mask (sige/utils.py:8-37) for a CUDA
    bool/uint8 mask [H,W].  One D2H read of the active-tile count (the reference's
    torch.nonzero synchronises the same way)."""
    if mask.dim() != 2 or not mask.is_cuda:
        raise RuntimeError("sige_amd.hip.reduce_mask: expected a 2-D CUDA mask")
    if plan_recorder() is not None:  # (a launch plan records: persistent index list, recorded read-back)
        return reduce_mask_batch([(mask, block_size, stride, padding)])[0]
    m = mask if mask.dtype in (torch.bool, torch.uint8) else (mask != 0)
    m = m.contiguous()
    H, W = m.shape
    cap = lib().sige_hip_reduce_mask_capacity(H, W, stride[0], stride[1], padding[0], padding[1])
    buf = torch.empty((cap + 1, 2), dtype=torch.int32, device=m.device)  # last row holds the count
    count = buf[cap]
    _check(lib().sige_hip_reduce_mask_i32(m.data_ptr(), H, W, block_size[0], block_size[1], stride[0], stride[1],
                                          padding[0], padding[1], buf.data_ptr(), cap, count.data_ptr(),
                                          _stream(m)), "reduce_mask")
    n = int(count[0].item())
    return buf[:n].clone()


def reduce_mask_batch(requests) -> list:
    """`reduce_mask` for many (mask [H,W], block, stride, padding) requests with ONE device -> host read for all
    their counts (SIGEModel.set_masks: one request per distinct tile geometry and resolution of the network)."""
    if not requests:
        return []
    dev = requests[0][0].device
    counts = torch.empty(len(requests), dtype=torch.int32, device=dev)
    bufs = []
    for i, (mask, block, stride, padding) in enumerate(requests):
        if mask.dim() != 2 or not mask.is_cuda or mask.device != dev:
            raise RuntimeError("sige_amd.hip.reduce_mask_batch: expected 2-D masks on one GPU")
        m = mask if mask.dtype in (torch.bool, torch.uint8) else (mask != 0)
        m = m.contiguous()
        H, W = m.shape
        cap = lib().sige_hip_reduce_mask_capacity(H, W, stride[0], stride[1], padding[0], padding[1])
        buf = torch.empty((max(cap, 1), 2), dtype=torch.int32, device=dev)
        _check(lib().sige_hip_reduce_mask_i32(m.data_ptr(), H, W, block[0], block[1], stride[0], stride[1],
                                              padding[0], padding[1], buf.data_ptr(), cap,
                                              counts.data_ptr() + 4 * i, _stream(m)), "reduce_mask")
        bufs.append((buf, m))
    rec = plan_recorder()
    if rec is not None:
        # the plan replays these compaction launches under later masks: their outputs ARE the index lists (persistent, sized
        # for every candidate tile), and the count read-back is a recorded step that sets one slot per list
        first = rec.new_slots(len(requests))
        _check(lib().sige_hip_plan_record_readback(rec.handle, counts.data_ptr(), first, len(requests)), "plan_record_readback")
        ns = counts.cpu().tolist()
        out = []
        for i, ((buf, m), n) in enumerate(zip(bufs, ns)):
            rec.bind_index_list(buf, first + i, n)
            rec.keep.extend((buf, m))
            out.append(buf[:n])
        rec.keep.append(counts)
        return out
    ns = counts.cpu().tolist()  # the one synchronisation of the mask -> index pipeline
    return [buf[:n].clone() for (buf, _), n in zip(bufs, ns)]


def difference_mask(tensor1: torch.Tensor, tensor2: torch.Tensor, eps: float) -> torch.Tensor:
    """Device form of sige.utils.compute_difference_mask (sige/utils.py:74-85): bool [H,W]."""
    if tensor1.shape != tensor2.shape:
        raise RuntimeError("difference_mask: shapes differ")
    a, b = tensor1, tensor2
    if a.dim() == 4:
        assert a.shape[0] == 1
        a, b = a[0], b[0]
    if a.dim() == 2:
        a, b = a[None], b[None]
    if a.dim() != 3:
        raise NotImplementedError("Unknown mask dimension [%d]!!!" % tensor1.dim())
    if not (a.is_cuda and b.is_cuda and a.dtype == b.dtype == torch.float32):
        raise NotImplementedError("difference_mask: fp32 GPU tensors")
    if a.stride() != b.stride() or a.stride(1) != a.shape[2] * a.stride(2):  # (NCHW and channels-last both pass as they are)
        a, b = a.contiguous(), b.contiguous()
    C, H, W = a.shape
    out = torch.empty((H, W), dtype=torch.uint8, device=a.device)
    _check(lib().sige_hip_difference_mask_u8(a.data_ptr(), b.data_ptr(), C, H, W, a.stride(0), a.stride(1), a.stride(2),
                                             float(eps), out.data_ptr(), _stream(a)), "difference_mask")
    return out.view(torch.bool)


def _mask_u8(mask: torch.Tensor) -> torch.Tensor:
    if mask.dim() != 2 or not mask.is_cuda:
        raise RuntimeError("sige_amd.hip: expected a 2-D GPU mask")
    m = mask if mask.dtype in (torch.bool, torch.uint8) else (mask != 0)
    m = m.contiguous()
    return m.view(torch.uint8) if m.dtype == torch.bool else m


def dilate_mask(mask: torch.Tensor, dilation: Tuple[int, int]) -> torch.Tensor:
    """Device form of sige.utils.dilate_mask for a 2-D mask (sige/utils.py:57-61): a new bool [H,W]."""
    m = _mask_u8(mask)
    H, W = m.shape
    out = torch.empty_like(m)
    _check(lib().sige_hip_dilate_mask_u8(m.data_ptr(), H, W, max(0, dilation[0]), max(0, dilation[1]), out.data_ptr(),
                                         _stream(m)), "dilate_mask")
    _plan_keep(out, m)
    return out.view(torch.bool)


def mask_pyramid(mask: torch.Tensor, min_res: Tuple[int, int], dilation: Tuple[int, int], threshold: float, eps: float):
    """Device form of sige.utils.downsample_mask (sige/utils.py:88-118): {(h, w): bool [h,w]} -- ONE launch for all
    levels, no host synchronisation (the reference synchronises once per level for `level.max()`)."""
    m = _mask_u8(mask)
    H, W = m.shape
    # stacked edits (set_edit_batch): `mask` is E masks stacked along H; the levels are those of ONE image, E times as tall
    E = get_edit_batch()
    if H % E:
        raise RuntimeError("mask_pyramid: the mask's height is not a multiple of the edit batch %d" % E)
    Hp = H // E
    n = lib().sige_hip_mask_pyramid_levels(Hp, W, min_res[0], min_res[1], None, None, 0)
    hs, ws = (ctypes.c_int * n)(), (ctypes.c_int * n)()
    lib().sige_hip_mask_pyramid_levels(Hp, W, min_res[0], min_res[1], hs, ws, n)
    sizes = [(E * int(hs[i]), int(ws[i])) for i in range(n)]
    out = torch.empty(sum(h * w for h, w in sizes), dtype=torch.uint8, device=m.device)
    n_scratch = E * ((Hp // 2) * (W // 2) + (Hp // 4) * (W // 4) + (Hp * W + 3) // 4 + 8)
    scratch = torch.empty(n_scratch, dtype=torch.float32, device=m.device)
    _check(lib().sige_hip_mask_pyramid_u8(m.data_ptr(), H, W, min_res[0], min_res[1], max(0, dilation[0]), max(0, dilation[1]),
                                          float(threshold), float(eps), scratch.data_ptr(), n_scratch, out.data_ptr(),
                                          _stream(m)), "mask_pyramid")
    _plan_keep(out, scratch, m)
    pyramid, off = {}, 0
    for h, w in sizes:
        pyramid[(h, w)] = out[off:off + h * w].view(h, w).view(torch.bool)
        off += h * w
    return pyramid


def conv_packed_size(Cout, Cin, kH, kW, R, S, strH, strW, groups=1) -> int:
    return int(lib().sige_hip_block_conv_packed_size(Cout, Cin, kH, kW, R, S, strH, strW, groups))


class PackedWeights(torch.Tensor):
    """Opaque packed conv weights (fp32 storage); `.compute` says which matrix path they were laid out for (it follows
    the tensor through clone / detach / to: a copy read with the wrong kernel family would run past its end)."""
    compute = "f32"
    wshift = 0  # wide packs (compute "f16w" / "f16x3w"): the weights were stored as w * 2^wshift

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        out = super().__torch_function__(func, types, args, kwargs or {})
        if isinstance(out, PackedWeights):
            src = next((a for a in args if isinstance(a, PackedWeights)), None)
            if src is not None and src is not out:
                out.compute = src.__dict__.get("compute", "f32")
                out.wshift = src.__dict__.get("wshift", 0)
        return out


def conv_pack_weights(weight: torch.Tensor, R: int, S: int, stride: Tuple[int, int], compute: str = "f32") -> Optional[torch.Tensor]:
    """Re-lay a conv weight [Cout,Cin,k,k] for the MFMA block conv; None if the shape has no MFMA path.
    compute = "f16": fp16 operands on the 16x faster fp16 matrix path, fp32 accumulation (activations stay fp32 in HBM;
    BASELINE.json configs[4]).  Shapes without an f16 kernel (the stride-2 geometry) are packed for the fp32 path --
    the returned tensor's `.compute` tells which."""
    w = _req(weight.detach(), torch.float32, "weight")
    Cout, Cin, kH, kW = w.shape
    if compute not in ("f32", "f16", "f16x3"):
        raise ValueError("compute must be 'f32', 'f16' or 'f16x3'")
    n = 0
    if compute == "f16":
        n = int(lib().sige_hip_block_conv_packed_size_f16c(Cout, Cin, kH, kW, R, S, stride[0], stride[1], 1))
        if n == 0:
            compute = "f32"
    if compute == "f16x3":  # split fp16 operands: fp32-level results on the fp16 matrix cores
        n = int(lib().sige_hip_block_conv_packed_size_f16x3(Cout, Cin, kH, kW, R, S, stride[0], stride[1], 1))
        if n == 0:
            compute = "f32"
    if compute == "f32":
        n = conv_packed_size(Cout, Cin, kH, kW, R, S, stride[0], stride[1], 1)
    if n == 0:
        return None
    packed = torch.empty((n,), dtype=torch.float32, device=w.device).as_subclass(PackedWeights)
    packed.compute = compute
    fn = {"f16": lib().sige_hip_block_conv_pack_f16c, "f16x3": lib().sige_hip_block_conv_pack_f16x3,
          "f32": lib().sige_hip_block_conv_pack_f32}[compute]
    _check(fn(w.data_ptr(), Cout, Cin, kH, kW, packed.data_ptr(), _stream(w)), "conv_pack_weights")
    if compute in ("f32", "f16") and (kH, kW) == (3, 3) and (R, S) == (6, 6) and tuple(stride) == (1, 1) and Cin % 64 == 0 and Cout % 64 == 0:
        # the tile conv v3 (csrc/conv_tile3.hpp) reads the same weights in the dense-layer kernel's exact-fp32 order.  Packed HERE when
        # the router is on, whatever the first mask's tile count: the routing entry points decide per launch, in C, from the
        # count -- also when a launch plan replays this call under a larger mask later
        # (memory: a second fp32 copy of these weights -- 455 MB for the DDPM U-Net, of 288 GB.  The source tensor is only kept
        #  until the layout exists, and a shape without a v3 layout is remembered as such: ADVICE r5)
        packed._tile3_src = w
        if TILE3 is not False and TILE3_MIN_BLOCKS is not None:
            _tile3_pack_now(packed)
    return packed


def _conv_fn(name: str, packed):
    """The fp32, the f16-compute or the split-fp16 entry point, according to how `packed` was laid out."""
    return getattr(lib(), name + {"f16": "_f16c", "f16x3": "_f16x3"}.get(getattr(packed, "compute", "f32"), "_f32"))


# ---- dense layers on the fp16 matrix cores (csrc/conv_wide.hpp) ----
COMPUTE_DTYPES = ("f32", "f16", "f16x3")


_COMPUTE_ID = {"f32": 0, "f16": 1, "f16x3": 2}  # the `compute` argument of the "_c16" entry points (tile kernels: how `packed` is laid out)
_WIDE_PREC = {"f16": 0, "f16x3": 1, "f32": 2}  # the `prec` argument of sige_hip_wide_conv_* (conv_wide.hpp: WIDE_F16 / _X3 / _F32)


def wide_conv_supported(C1: int, C2: int, Cout: int, kernel: Tuple[int, int]) -> bool:
    return bool(lib().sige_hip_wide_conv_supported(C1, C2, Cout, kernel[0], kernel[1]))


def wide_conv_pack_weights(weight: torch.Tensor, compute: str) -> Optional[torch.Tensor]:
    """Weights [Cout,Cin,k,k] (k = 1 | 3) packed for the dense-layer conv on the fp16 matrix cores; compute "f16" (operands
    rounded to fp16) or "f16x3" (operands split into fp16 hi + lo, three products: fp32-level results).  None: no kernel
    for this shape.  For "f16x3" the weights are stored as w * 2^s with max |w| * 2^s in [2^13, 2^14) (one device -> host
    read of max |w| at pack time), so that the lo parts are normal fp16 numbers."""
    if compute not in _WIDE_PREC:
        raise ValueError("compute must be 'f16', 'f16x3' or 'f32'")
    w = _req(weight.detach(), torch.float32, "weight")
    Cout, Cin, kH, kW = w.shape
    prec = _WIDE_PREC[compute]
    x3 = compute == "f16x3"
    n = int(lib().sige_hip_wide_conv_packed_size(Cout, Cin, kH, kW, prec))
    if n == 0:
        return None
    wshift = 0
    if x3:
        import math

        m = float(w.abs().max())
        if m > 0 and math.isfinite(m):
            wshift = max(-40, min(40, 13 - math.frexp(m)[1] + 1))  # m = f * 2^e, f in [0.5, 1): m * 2^(14 - e) in [2^13, 2^14)
    packed = torch.empty((n,), dtype=torch.float32, device=w.device).as_subclass(PackedWeights)
    packed.compute = compute + "w"
    packed.wshift = wshift
    _check(lib().sige_hip_wide_conv_pack(w.data_ptr(), Cout, Cin, kH, kW, prec, wshift, packed.data_ptr(), _stream(w)),
           "wide_conv_pack_weights")
    return packed


def wide_conv_force_ksplit(ksplit: int = 0):
    """Benchmark knob: pin the cross-workgroup K split of the dense-layer conv (0 = automatic)."""
    tuning_set("wide_ksplit", ksplit)


def wide_conv_cl(x, x2, scale, shift, activationName: str, packed, bias, Cout: int, kernel: Tuple[int, int],
                 residual=None, out_affine: Optional[tuple] = None, twins=None, upsample2x: bool = False,
                 out: Optional[torch.Tensor] = None, stats: bool = False):
    """out_act(os * (conv(act(scale * cat(x, x2) + shift)) + bias + residual) + oh) over a whole channels-last tensor in one
    launch on the fp16 matrix cores (include/sige_hip.h: sige_hip_wide_conv_nhwc).  3x3 / padding 1 or 1x1, stride 1.
    `packed` from wide_conv_pack_weights.  None if the shape has no kernel.
    `stats`: the launch also leaves the per-channel statistics of its output (ChannelStats, attached to the returned tensor:
    channel_stats(out)) -- the GroupNorm of the output then needs no pass over it (group_norm_affine_from_stats)."""
    compute = getattr(packed, "compute", "f32")
    if compute not in ("f16w", "f16x3w", "f32w"):
        raise NotImplementedError("wide_conv_cl: weights must be packed with wide_conv_pack_weights")
    bias_keep = _vec(bias, "bias")
    x = _req_cl(x, "x")
    B, C1, H, W = x.shape
    if upsample2x:
        H, W = 2 * H, 2 * W
    C2 = 0
    if x2 is not None:
        x2 = _req_cl(x2, "x2")
        C2 = x2.shape[1]
        if tuple(x2.shape[2:]) != tuple(x.shape[2:]) or x2.shape[0] != B:
            raise RuntimeError("wide_conv_cl: x2 must match x in batch and resolution")
    if not wide_conv_supported(C1, C2, Cout, kernel):
        return None
    (sa, s_keep), (ta, t_keep) = _cvec(scale, "scale"), _cvec(shift, "shift")
    if (s_keep is None) != (t_keep is None):
        return None
    if s_keep is not None and (sa[1] != ta[1] or sa[2] != C1 + C2 or ta[2] != C1 + C2 or sa[1] not in (1, B)):
        return None
    if s_keep is None and activationName != "identity":
        return None
    if out is None:
        out = _empty_cl((B, Cout, H, W), x.device)
    elif tuple(out.shape) != (B, Cout, H, W) or not out.is_contiguous(memory_format=CL):
        raise RuntimeError("wide_conv_cl: `out` must be a channels-last [B,Cout,H,W] tensor")
    r = None
    if residual is not None:
        r = _req_cl(residual, "residual")
        if tuple(r.shape) != tuple(out.shape):
            raise RuntimeError("wide_conv_cl: residual %s != output %s" % (tuple(r.shape), tuple(out.shape)))
    if out_affine is not None:
        os_, oh_, oact = out_affine
        os_, oh_ = _req(os_.reshape(-1), torch.float32, "out_scale", 1), _req(oh_.reshape(-1), torch.float32, "out_shift", 1)
        if os_.numel() != Cout or oh_.numel() != Cout:
            raise RuntimeError("wide_conv_cl: out_affine must have one entry per output channel")
        oargs = (os_.data_ptr(), oh_.data_ptr(), _act(oact))
    else:
        os_ = oh_ = None
        oargs = (None, None, 0)
    targs, twin_keep = _twin_args(twins if twins else None, out, Cout, "wide_conv_cl")
    ws, ws_n = None, int(lib().sige_hip_wide_conv_workspace(B, H, W, C1, C2, Cout, kernel[0], kernel[1])) if KSPLIT else 0
    if ws_n:
        ws = torch.empty(ws_n, dtype=torch.float32, device=x.device)
    st = None
    if stats and out_affine is None:
        tiles = ((H + 7) // 8) * ((W + 7) // 8)
        st = torch.empty((B * tiles, Cout, 2), dtype=torch.float32, device=x.device)
    status = lib().sige_hip_wide_conv_nhwc(
        x.data_ptr(), None if x2 is None else x2.data_ptr(), B, C1, C2, H, W, int(bool(upsample2x)),
        sa[0], ta[0], sa[1] if s_keep is not None else 0, _act(activationName),
        packed.data_ptr(), _WIDE_PREC[compute[:-1]], int(getattr(packed, "wshift", 0)), _p(bias_keep), Cout, kernel[0], kernel[1],
        None if r is None else r.data_ptr(), *oargs, *targs, None if ws is None else ws.data_ptr(), ws_n, out.data_ptr(),
        None if st is None else st.data_ptr(), _stream(x))
    if status == UNSUPPORTED:
        return None
    _check(status, "wide_conv_cl")
    if st is not None:
        attach_channel_stats(out, ChannelStats(st, tiles, Cout, H * W))
    return out


class ChannelStats:
    """Per-channel (sum, sum of squares) of a tensor, as its producer left them: `data` [B * tiles, C, 2] partial sums per pixel
    block, `count` pixels per batch element.  Attached to the tensor OBJECT (attach_channel_stats) together with the tensor's
    version counter and address: any torch op makes a new object without them, an in-place op bumps the version, so stale
    statistics are never used (channel_stats returns None)."""

    __slots__ = ("data", "tiles", "channels", "count", "version", "ptr")

    def __init__(self, data, tiles, channels, count):
        self.data, self.tiles, self.channels, self.count = data, tiles, channels, count
        self.version = self.ptr = None


def attach_channel_stats(t: torch.Tensor, st: ChannelStats) -> torch.Tensor:
    st.version, st.ptr = t._version, t.data_ptr()
    t._sige_channel_stats = st
    return t


def channel_stats(t) -> Optional[ChannelStats]:
    st = getattr(t, "_sige_channel_stats", None)
    if st is None or st.version != t._version or st.ptr != t.data_ptr():
        return None
    return st


def channel_stats_cl(x: torch.Tensor) -> Optional[ChannelStats]:
    """Per-channel statistics of a channels-last tensor whose producer left none, in one pass (attached to `x` as well, so that
    a second consumer -- a skip connection -- does not read it again).  None if unsupported."""
    st = channel_stats(x)
    if st is not None:
        return st
    x = _req_cl(x, "x")
    B, C, H, W = x.shape
    tiles = int(lib().sige_hip_channel_stats_tiles(H, W))
    data = torch.empty((B * tiles, C, 2), dtype=torch.float32, device=x.device)
    status = lib().sige_hip_channel_stats_nhwc_f32(x.data_ptr(), B, C, H, W, data.data_ptr(), _stream(x))
    if status == UNSUPPORTED:
        return None
    _check(status, "channel_stats_cl")
    st = ChannelStats(data, tiles, C, H * W)
    attach_channel_stats(x, st)
    return st


def group_norm_affine_from_stats(parts, groups: int, eps: float, gamma=None, beta=None, channel_bias=None):
    """GroupNorm (scale, shift) [B,C,1,1] of torch.cat(tensors, 1) from the ChannelStats of one or two tensors, in one launch
    over the partial sums (include/sige_hip.h: sige_hip_group_norm_affine_from_stats_f32).  None if unsupported."""
    if len(parts) not in (1, 2):
        return None
    p1 = parts[0]
    p2 = parts[1] if len(parts) == 2 else None
    B = p1.data.shape[0] // p1.tiles
    if p2 is not None and p2.data.shape[0] // p2.tiles != B:
        return None
    C = p1.channels + (p2.channels if p2 is not None else 0)
    buf = torch.empty(2 * B * C, dtype=torch.float32, device=p1.data.device)
    scale, shift = buf[:B * C].view(B, C, 1, 1), buf[B * C:].view(B, C, 1, 1)
    gamma_keep, beta_keep, cb_keep = _vec(gamma, "gamma"), _vec(beta, "beta"), _vec(channel_bias, "channel_bias")
    if cb_keep is not None and cb_keep.numel() != C:
        raise RuntimeError("group_norm_affine_from_stats: channel_bias must have one entry per channel")
    status = lib().sige_hip_group_norm_affine_from_stats_f32(
        p1.data.data_ptr(), p1.tiles, p1.channels, p1.count,
        None if p2 is None else p2.data.data_ptr(), 0 if p2 is None else p2.tiles, 0 if p2 is None else p2.channels,
        0 if p2 is None else p2.count, B, groups, eps, _p(gamma_keep), _p(beta_keep), _p(cb_keep),
        scale.data_ptr(), shift.data_ptr(), _stream(p1.data))
    if status == UNSUPPORTED:
        return None
    _check(status, "group_norm_affine_from_stats")
    return scale, shift


def conv_force_tile(mt: int = 0, nb: int = 0):
    """Benchmark knob: pin the MFMA conv's output block (0, 0 = automatic)."""
    tuning_set("conv_tile_mt", mt)
    tuning_set("conv_tile_nb", nb)


def conv_large_grid_nb1(min_blocks: int = -1):
    """Plan policy: unsplit tile-conv launches that 32 x 64 output blocks would fill the chip with use 32 x 32 blocks (more
    workgroups per CU) from `min_blocks` such blocks on; -1 = the library's default (exact fp32: always), 0 = never
    (include/sige_hip.h: sige_hip_block_conv_large_grid_nb1)."""
    tuning_set("conv_large_grid_nb1", int(min_blocks))


def conv_force_waves(waves: int = 0):
    """Benchmark knob: 4 or 8 waves per workgroup for the channels-last stride-1 convs (0 = automatic)."""
    tuning_set("conv_waves", waves)


_pair_state = threading.local()  # .keep: operands of the launches of the open conv_pair() block; .depth: nesting


class conv_pair:
    """`with hip.conv_pair():` around the shortcut conv and the conv1 of a residual block (in that order, nothing else):
    the 1x1 is held back and launched inside the 3x3's kernel (sige_hip_conv_pair_begin / _end, include/sige_hip.h).
    Results do not depend on it; whatever cannot be paired is launched on its own, at the latest when the block ends.
    Per thread, like the C side; a nested block joins the outer one."""

    def __init__(self, like: Optional[torch.Tensor] = None):
        self.like = like  # (a tensor of the convs' device: the held conv may be launched when the block ends)

    def _on_device(self):
        """Device context for begin / end: both may launch a held conv (with another device's stream and pointers if
        HIP's current device is not the convs'), and neither takes a stream argument the _Guarded wrapper could key on."""
        import contextlib

        if self.like is not None and self.like.is_cuda:
            cur = _raw_device() if _raw_device is not None else torch.cuda.current_device()
            if self.like.device.index != cur:
                GUARD_STATS["pair_switched"] += 1
                return torch.cuda.device(self.like.device)
        return contextlib.nullcontext()

    def __enter__(self):
        depth = getattr(_pair_state, "depth", 0)
        if depth == 0:
            with self._on_device():
                _check(lib().sige_hip_conv_pair_begin(), "conv_pair_begin")
            _pair_state.keep = []
        _pair_state.depth = depth + 1
        return self

    def __exit__(self, *exc):
        _pair_state.depth -= 1
        if _pair_state.depth == 0:
            try:
                with self._on_device():
                    _check(lib().sige_hip_conv_pair_end(), "conv_pair_end")
            finally:
                _pair_state.keep = None
        return False


def conv_pairs_fused() -> int:
    return int(lib().sige_hip_conv_pairs_fused())


def conv_force_ksplit(ksplit: int = 0):
    """Benchmark knob: cross-workgroup K split of the channels-last launches with a workspace (0 = automatic)."""
    tuning_set("conv_ksplit", ksplit)


def gather_force_rows(one_tile_rows: bool = False):
    """Benchmark knob: the NCHW gather's one-tile row form always (True) instead of the grouped form where it applies."""
    tuning_set("gather_one_tile_rows", int(bool(one_tile_rows)))


def scatter_gather_force_elements(element_form=False):
    """Benchmark / test knob: the NCHW scatter_gather's element form always (True / 1), or its one-tile row form (2: never the
    grouped form), instead of the automatic choice (False / 0)."""
    tuning_set("scatter_gather_form", int(element_form))


def release_graph_tickets():
    """Every hipGraph captured so far on the current device has been destroyed: hand the K-split tickets of captured launches
    out again from the start (include/sige_hip.h: sige_hip_release_graph_tickets)."""
    _check(lib().sige_hip_release_graph_tickets(), "release_graph_tickets")


def conv_force_ksplit_pass(second_pass: bool = False):
    """Benchmark knob: finish K-split launches with a second launch (True) instead of inside the launch (default)."""
    tuning_set("conv_ksplit_second_pass", int(bool(second_pass)))


def _f32_packed(packed):
    if getattr(packed, "compute", "f32") != "f32":
        raise NotImplementedError("the f16-compute kernels are channels-last: pack with compute='f32' for NCHW tensors")


def block_conv(x, packed, bias, Cout: int, kernel: Tuple[int, int], stride: Tuple[int, int]):
    _f32_packed(packed)
    x = _req(x, torch.float32, "x")
    T, Cin, R, S = x.shape
    Ro, So = (R - kernel[0]) // stride[0] + 1, (S - kernel[1]) // stride[1] + 1
    out = torch.empty((T, Cout, Ro, So), dtype=torch.float32, device=x.device)
    bias_keep = _vec(bias, "bias")
    b = _p(bias_keep)
    _check(lib().sige_hip_block_conv_f32(x.data_ptr(), T, Cin, R, S, packed.data_ptr(), b, Cout, kernel[0], kernel[1],
                                         stride[0], stride[1], out.data_ptr(), _stream(x)), "block_conv")
    return out


def _channel_affine(t: Optional[torch.Tensor]) -> bool:
    return t is None or (t.dim() == 4 and t.shape[2] == 1 and t.shape[3] == 1)


def fusable_affine(scale, shift, activation_first: bool) -> bool:
    """True if gather_conv / scatter_gather_conv can absorb this gather."""
    return (not activation_first) and _channel_affine(scale) and _channel_affine(shift)


def gather_conv(x, block: Tuple[int, int], activeIndices, scale, shift, activationName: str,
                packed, bias, Cout: int, kernel: Tuple[int, int], stride: Tuple[int, int]):
    """gather(x, ...) followed by the stacked-block conv, in one kernel.  Returns None when
    the fused kernel cannot express the call (SIGE_HIP_EUNSUPPORTED: e.g. a per-batch affine
    whose tiles-per-workgroup straddle images); the caller then runs gather + block_conv."""
    _f32_packed(packed)
    x = _req(x, torch.float32, "x")
    idx = _req(activeIndices, torch.int32, "activeIndices", 2)
    (sa, s_keep), (ta, t_keep) = _bc(scale, "scale"), _bc(shift, "shift")
    B, C, H, W = x.shape
    N = idx.shape[0]
    Ro, So = (block[0] - kernel[0]) // stride[0] + 1, (block[1] - kernel[1]) // stride[1] + 1
    out = torch.empty((B * N, Cout, Ro, So), dtype=torch.float32, device=x.device)
    bias_keep = _vec(bias, "bias")
    b = _p(bias_keep)
    status = lib().sige_hip_gather_conv_f32(x.data_ptr(), B, C, H, W, block[0], block[1], idx.data_ptr(), N, *sa, *ta,
                                            _act(activationName), packed.data_ptr(), b, Cout, kernel[0], kernel[1],
                                            stride[0], stride[1], out.data_ptr(), _stream(x))
    if status == UNSUPPORTED:
        return None
    _check(status, "gather_conv")
    return out


def scatter_gather_conv(x, y, block: Tuple[int, int], activeIndices, scatterMap, scale, shift, activationName: str,
                        packed, bias, Cout: int, kernel: Tuple[int, int], stride: Tuple[int, int]):
    """scatter_gather(x, y, ...) followed by the stacked-block conv, in one kernel (None if
    the fused kernel cannot express the call, see gather_conv)."""
    _f32_packed(packed)
    x, y = _req(x, torch.float32, "x"), _req(y, torch.float32, "y")
    idx = _req(activeIndices, torch.int32, "activeIndices", 2)
    smap = _req(scatterMap, torch.int32, "scatterMap", 3)
    (sa, s_keep), (ta, t_keep) = _bc(scale, "scale"), _bc(shift, "shift")
    B, C, H, W = y.shape
    N = idx.shape[0]
    Ro, So = (block[0] - kernel[0]) // stride[0] + 1, (block[1] - kernel[1]) // stride[1] + 1
    out = torch.empty((B * N, Cout, Ro, So), dtype=torch.float32, device=y.device)
    bias_keep = _vec(bias, "bias")
    b = _p(bias_keep)
    status = lib().sige_hip_scatter_gather_conv_f32(
        x.data_ptr(), y.data_ptr(), B, C, H, W, x.shape[2], x.shape[3], block[0], block[1], idx.data_ptr(), N,
        smap.data_ptr(), *sa, *ta, _act(activationName), packed.data_ptr(), b, Cout, kernel[0], kernel[1],
        stride[0], stride[1], out.data_ptr(), _stream(y))
    if status == UNSUPPORTED:
        return None
    _check(status, "scatter_gather_conv")
    return out


_all_tiles_cache = {}


def all_tiles(H: int, W: int, out_tile: Tuple[int, int], stride: Tuple[int, int], offset: Tuple[int, int],
              device) -> torch.Tensor:
    """Index list [N,2] of EVERY tile of an H x W input (a dense layer seen as
    "all tiles active"): origins (o*s*i - off, o*s*j - off), one per o x o output cell."""
    key = (H, W, tuple(out_tile), tuple(stride), tuple(offset), str(device))
    idx = _all_tiles_cache.get(key)
    if idx is None:
        ph, pw = out_tile[0] * stride[0], out_tile[1] * stride[1]
        nh, nw = -(-H // ph), -(-W // pw)
        hh = torch.arange(nh, dtype=torch.int32) * ph - offset[0]
        ww = torch.arange(nw, dtype=torch.int32) * pw - offset[1]
        idx = torch.stack(torch.meshgrid(hh, ww, indexing="ij"), dim=-1).reshape(-1, 2).contiguous().to(device)
        _all_tiles_cache[key] = idx
    rec = plan_recorder()
    if rec is not None:  # (a count next to this list never changes with the mask: csrc/plan.hpp const_ptrs)
        rec.bind_constant(idx)
    return idx


def cat_fusable(B: int, C1: int, kernel: Tuple[int, int]) -> bool:
    """Can gather_conv_nchw take the channels from two tensors (a fused torch.cat)?  The
    split must fall on a channel-chunk boundary of one of the kernel's tile shapes
    (conv_mfma.hpp: 32 / 64 channels for 3x3, 128 / 256 for 1x1) and the batch must be 1."""
    return B == 1 and C1 % (128 if tuple(kernel) == (1, 1) else 32) == 0


def gather_conv_nchw(x, x2, block: Tuple[int, int], activeIndices, scale, shift, activationName: str,
                     packed, bias, Cout: int, kernel: Tuple[int, int], stride: Tuple[int, int],
                     offset: Tuple[int, int], out_res: Tuple[int, int], residual=None):
    """conv(act(cat(x, x2) * scale + shift)) + residual over the listed tiles, written
    straight into a fresh [B,Cout,Ho,Wo] tensor (pixels no tile covers are NOT written:
    pass the all-tiles list for a dense layer)."""
    _f32_packed(packed)
    x = _req(x, torch.float32, "x")
    B, C1, H, W = x.shape
    C2 = 0
    if x2 is not None:
        x2 = _req(x2, torch.float32, "x2")
        C2 = x2.shape[1]
    idx = _req(activeIndices, torch.int32, "activeIndices", 2)

    def cvec(t, name):
        if t is None:
            return (None, 0, 0), None
        t = _req(t, torch.float32, name)
        if t.shape[2] != 1 or t.shape[3] != 1:
            raise RuntimeError("gather_conv_nchw: `%s` must be [1|B, 1|C, 1, 1]" % name)
        return (t.data_ptr(), t.shape[0], t.shape[1]), t

    (sa, s_keep), (ta, t_keep) = cvec(scale, "scale"), cvec(shift, "shift")
    Ho, Wo = out_res
    out = torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device=x.device)
    r = None
    if residual is not None:
        residual = _req(residual, torch.float32, "residual")
        if tuple(residual.shape) != tuple(out.shape):
            raise RuntimeError("gather_conv_nchw: residual %s != output %s" % (tuple(residual.shape), tuple(out.shape)))
        r = residual.data_ptr()
    bias_keep = _vec(bias, "bias")
    b = _p(bias_keep)
    _check(lib().sige_hip_gather_conv_nchw_f32(
        x.data_ptr(), None if x2 is None else x2.data_ptr(), B, C1, C2, H, W, block[0], block[1], idx.data_ptr(),
        idx.shape[0], *sa, *ta, _act(activationName), packed.data_ptr(), b, Cout, kernel[0], kernel[1],
        stride[0], stride[1], offset[0], offset[1], r, Ho, Wo, out.data_ptr(), _stream(x)), "gather_conv_nchw")
    return out


def group_norm_affine(x, groups: int, eps: float, gamma=None, beta=None):
    """Per-channel (scale, shift) [B,C,1,1] with GroupNorm(x) == x*scale + shift."""
    x = _req(x, torch.float32, "x")
    B, C, H, W = x.shape
    n = int(lib().sige_hip_group_norm_affine_workspace(B, C, H, W, groups))
    if n == 0:
        raise RuntimeError("group_norm_affine: channels %d not divisible by groups %d" % (C, groups))
    buf = torch.empty(n + 2 * B * C, dtype=torch.float32, device=x.device)
    scale, shift = buf[n:n + B * C].view(B, C, 1, 1), buf[n + B * C:].view(B, C, 1, 1)
    gamma_keep = _vec(gamma, "gamma")
    ga = _p(gamma_keep)
    beta_keep = _vec(beta, "beta")
    be = _p(beta_keep)
    _check(lib().sige_hip_group_norm_affine_f32(x.data_ptr(), B, C, H, W, groups, eps, ga, be, buf.data_ptr(),
                                                scale.data_ptr(), shift.data_ptr(), _stream(x)), "group_norm_affine")
    return scale, shift


def attention_supported(C: int, HW: int) -> bool:
    return C % 16 == 0 and HW % 16 == 0 and HW <= 4096 and (16 * (HW + 4) + 64 * 260) * 4 <= 160 * 1024


def attention(qkv: torch.Tensor, scale: float) -> torch.Tensor:
    """softmax(scale * q^T k) applied to v for qkv [B,3C,H,W] (q, k, v stacked on the
    channel axis, one head): [B,C,H,W]."""
    qkv = _req(qkv, torch.float32, "qkv")
    B, C3, H, W = qkv.shape
    C, HW = C3 // 3, H * W
    n = int(lib().sige_hip_attention_workspace(B, C, HW))
    ws = torch.empty(n, dtype=torch.float32, device=qkv.device)
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=qkv.device)
    _check(lib().sige_hip_attention_f32(qkv.data_ptr(), B, C, HW, float(scale), ws.data_ptr(), out.data_ptr(),
                                        _stream(qkv)), "attention")
    return out


def block_conv_direct(x, weight, bias, stride: Tuple[int, int], groups: int = 1, dilation: Tuple[int, int] = (1, 1)):
    x = _req(x, torch.float32, "x")
    w = _req(weight.detach(), torch.float32, "weight")
    T, Cin, R, S = x.shape
    Cout, _, kH, kW = w.shape
    Ro = (R - (kH - 1) * dilation[0] - 1) // stride[0] + 1
    So = (S - (kW - 1) * dilation[1] - 1) // stride[1] + 1
    out = torch.empty((T, Cout, Ro, So), dtype=torch.float32, device=x.device)
    bias_keep = _vec(bias, "bias")
    b = _p(bias_keep)
    _check(lib().sige_hip_block_conv_direct_f32(x.data_ptr(), T, Cin, R, S, w.data_ptr(), b, Cout, kH, kW,
                                                stride[0], stride[1], dilation[0], dilation[1], groups, out.data_ptr(),
                                                _stream(x)),
           "block_conv_direct")
    return out


def copy_dense_(dst: torch.Tensor, src: torch.Tensor):
    """dst <- src for two fp32 tensors of one shape and one DENSE layout (plain or channels-last: equal strides, no gaps), as a
    library launch -- unlike Tensor.copy_ it is recorded by a launch plan (the refresh of a persistent Scatter output)."""
    if (dst.shape != src.shape or dst.stride() != src.stride() or not (dst.is_contiguous() or dst.is_contiguous(memory_format=CL))):
        raise RuntimeError("copy_dense_: two tensors of one shape and one dense layout")
    pair = (src.dtype, dst.dtype)
    if pair == (torch.float32, torch.float32):
        _check(lib().sige_hip_copy_f32(src.data_ptr(), dst.data_ptr(), src.numel(), _stream(src)), "copy")
    elif pair == (torch.float16, torch.float32) and src.numel() % 4 == 0:   # (an fp16-stored cache widened into a persistent output)
        _check(lib().sige_hip_convert_f16_f32(src.data_ptr(), dst.data_ptr(), src.numel(), _stream(src)), "convert_f16_f32")
    elif pair == (torch.float32, torch.float16) and src.numel() % 4 == 0:   # (a full-pass output rounded into the fp16 cache)
        _check(lib().sige_hip_convert_f32_f16(src.data_ptr(), dst.data_ptr(), src.numel(), _stream(src)), "convert_f32_f16")
    else:
        raise RuntimeError("copy_dense_: fp32 -> fp32, fp16 -> fp32 or fp32 -> fp16 (element count a multiple of 4)")
    return dst


def copy_(dst: torch.Tensor, src: torch.Tensor):
    assert dst.numel() == src.numel() and dst.is_contiguous() and src.is_contiguous()
    _check(lib().sige_hip_copy_f32(src.data_ptr(), dst.data_ptr(), src.numel(), _stream(src)), "copy")
    return dst


# --------------------------------------------------------------------------
# Channels-last (NHWC) forms.  Tensors keep their logical [B,C,H,W] / [T,C,R,S]
# shape and carry torch.channels_last strides; every function returns
# channels_last tensors.  Arithmetic is that of the NCHW functions above.
# --------------------------------------------------------------------------
CL = torch.channels_last


def is_cl(t: torch.Tensor) -> bool:
    """True if `t` is a 4-D tensor stored channels-last (and not also plain contiguous,
    which happens for C == 1 or H == W == 1: those go down the NCHW path)."""
    return t.dim() == 4 and t.is_contiguous(memory_format=CL) and not t.is_contiguous()


def cl_supported(*channel_counts: int) -> bool:
    return all(c % 4 == 0 for c in channel_counts)


def _req_cl(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("sige_amd.hip: `%s` must live on the GPU (got %s)" % (name, t.device))
    if t.dtype != torch.float32:
        raise NotImplementedError("sige_amd.hip: `%s` must be float32 (got %s)" % (name, t.dtype))
    if t.dim() != 4:
        raise NotImplementedError("sige_amd.hip: `%s` must have 4 dims (got %d)" % (name, t.dim()))
    return t if t.is_contiguous(memory_format=CL) else t.contiguous(memory_format=CL)


def _req_cl_cache(t: torch.Tensor, name: str) -> torch.Tensor:
    """A CACHED tensor: channels-last, fp32 or -- SIGEModel.set_cache_dtype("f16") -- fp16 storage (read through the "_f16" /
    "_c16" entry points of include/sige_hip.h; widened exactly when read)."""
    if t.dtype == torch.float16:
        if not t.is_cuda or t.dim() != 4:
            raise NotImplementedError("sige_amd.hip: `%s` (fp16 cache) must be a 4-D GPU tensor" % name)
        return t if t.is_contiguous(memory_format=CL) else t.contiguous(memory_format=CL)
    return _req_cl(t, name)


def _empty_cl(shape, device) -> torch.Tensor:
    return torch.empty(shape, dtype=torch.float32, device=device, memory_format=CL)


def _cvec(t: Optional[torch.Tensor], name: str):
    """(ptr, B, C) of an optional per-(batch, channel) vector [1|B, C, 1, 1]."""
    if t is None:
        return (None, 0, 0), None
    if t.dim() != 4 or t.shape[2] != 1 or t.shape[3] != 1:
        raise RuntimeError("sige_amd.hip: `%s` must be [1|B, C, 1, 1] for the channels-last path" % name)
    t = _req(t, torch.float32, name)
    return (t.data_ptr(), t.shape[0], t.shape[1]), t


def _vec(t: Optional[torch.Tensor], name: str) -> Optional[torch.Tensor]:
    """An optional 1-D fp32 vector (bias, gamma, ...) made contiguous.  The caller keeps the returned tensor
    alive until after the launch: `.contiguous()` of a strided view is a temporary whose memory the caching
    allocator may hand out again before the kernel has read it."""
    return None if t is None else _req(t.detach() if t.requires_grad else t, torch.float32, name, 1)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _twin_args(twins, like: torch.Tensor, Cout: int, name: str):
    """The six twin pointers of a full-tensor conv launch (include/sige_hip.h) from `twins` = up to two (buffer, scale,
    shift): buffer = channels-last tensor shaped like the launch's output, scale / shift = [Cout] (or [1,Cout,1,1]) fp32.
    Returns (args, keep-alive list)."""
    args, keep = [], []
    for k in range(2):
        if twins is not None and k < len(twins):
            buf, sc, sh = twins[k]
            if tuple(buf.shape) != tuple(like.shape) or not buf.is_contiguous(memory_format=CL) or buf.dtype != torch.float32:
                raise RuntimeError("%s: twin %d must be a channels-last fp32 tensor shaped like the output" % (name, k))
            sc, sh = _req(sc.reshape(-1), torch.float32, "twin_scale", 1), _req(sh.reshape(-1), torch.float32, "twin_shift", 1)
            if sc.numel() != Cout or sh.numel() != Cout:
                raise RuntimeError("%s: twin %d needs one scale / shift entry per output channel" % (name, k))
            args += [buf.data_ptr(), sc.data_ptr(), sh.data_ptr()]
            keep += [buf, sc, sh]
        else:
            args += [None, None, None]
    if twins is not None and len(twins) > 2:
        raise RuntimeError("%s: at most two twins per launch" % name)
    return args, keep


def tag_tiles(t: torch.Tensor, idx: torch.Tensor, B: int) -> torch.Tensor:
    """Remember on a tile tensor [B*N,C,R,S] which index list its N tiles belong to: a conv over the slab (block_conv_cl) then
    takes the keyed entry point, whose tile count a launch plan can follow (include/sige_hip.h: sige_hip_block_conv_nhwc_keyed)."""
    t._sige_count_key = (idx, int(B))
    return t


def block_conv_cl(x, packed, bias, Cout: int, kernel: Tuple[int, int], stride: Tuple[int, int]):
    bias_keep = _vec(bias, "bias")
    key = getattr(x, "_sige_count_key", None)
    x = _req_cl(x, "x")
    T, Cin, R, S = x.shape
    Ro, So = (R - kernel[0]) // stride[0] + 1, (S - kernel[1]) // stride[1] + 1
    if key is not None and key[0].shape[0] * key[1] == T and T > 0:
        idx, B = key
        out = _empty_tiles_cl(B, idx, Cout, Ro, So, x.device)
        status = lib().sige_hip_block_conv_nhwc_keyed(_COMPUTE_ID[getattr(packed, "compute", "f32")], x.data_ptr(), idx.data_ptr(), B,
                                                      idx.shape[0], Cin, R, S, packed.data_ptr(), _p(bias_keep), Cout,
                                                      kernel[0], kernel[1], stride[0], stride[1], out.data_ptr(), _stream(x))
        if status == UNSUPPORTED:
            return None
        _check(status, "block_conv_cl")
        return tag_tiles(out, idx, B)
    out = _empty_cl((T, Cout, Ro, So), x.device)
    status = _conv_fn("sige_hip_block_conv_nhwc", packed)(x.data_ptr(), T, Cin, R, S, packed.data_ptr(), _p(bias_keep), Cout,
                                                kernel[0], kernel[1], stride[0], stride[1], out.data_ptr(), _stream(x))
    if status == UNSUPPORTED:
        return None
    _check(status, "block_conv_cl")
    return out


# ---- tile conv v3 (csrc/conv_tile3.hpp): routing --------------------------------------------------------------------------
# The two calls a sparse forward makes for its 3x3 convs -- gather -> conv and scatter_gather -> conv -> scatter -- go through
# the ROUTING entry points (sige_hip_*_v3_f32): the same arguments plus the weights in the v3 layout and TILE3_MIN_BLOCKS; a launch
# whose v3 grid -- tile pairs x 64-channel output blocks -- has at least that many workgroups, and that is not about to share its
# launch with a held 1x1 shortcut (conv_pair()), runs on the v3 kernel, every other launch on conv_mfma.hpp as before.  The
# decision is taken in C from the tile count, so a launch plan replaying the call under another mask routes like the module path.
# Measured (tools/tile3_bench.py, tools/tile3_router_bench.py, kernel traces of the forward; profiles/r5h_tile3_bench.json,
# r5i_tile3_router.json, r5j_sequence_15pct_*.csv, r5_bench_detail.json: tile_conv3): launch by launch with warm operands v3 is
# 1.05-1.26x conv_mfma.hpp from ~150 workgroups on (gather + affine + SiLU -> tiles: 85-92 TFLOP/s = 0.54-0.58 of the fp32 MFMA peak
# at 834-1080 workgroups against 0.42-0.45) and 0.5-0.9x below; INSIDE the DDPM forward at a 15 % edit the same launches gain 14 %
# (affine + SiLU staging: the activation is computed once per 64 output channels instead of once per 32), 4 % (scatter_gather)
# and 0 % (raw gather): the forward moves by 1 % at 15 %, 2 % at 20 %, nothing at 1.2 % and 5 % (no launch reaches the threshold).
# TILE3_MIN_BLOCKS = None switches the router off (no v3 weights are packed); TILE3 = False does the same per call (tests, A/B);
# TILE3 = True sends every eligible call to sige_hip_tile_conv3_nhwc_f32 directly, whatever its grid (tests, tools/tile3_bench.py).
TILE3 = None
TILE3_MIN_BLOCKS = 512
# fp16 operands (round 6; conv_tile3.hpp Tile3Geo<2, WIDE_F16>): the same routing for compute dtype "f16" (sige_hip_*_v3_f16c)
TILE3_MIN_BLOCKS_F16 = 256


def _tile3_min_blocks(packed) -> int:
    return int(TILE3_MIN_BLOCKS_F16 if getattr(packed, "compute", "f32") == "f16" else TILE3_MIN_BLOCKS)


def _tile3_pack_now(packed):
    t3 = wide_conv_pack_weights(packed._tile3_src, "f16" if getattr(packed, "compute", "f32") == "f16" else "f32")
    packed.tile3 = False if t3 is None else t3  # (False: asked once, there is no v3 layout for this shape -- not asked again)
    packed._tile3_src = None                    # (the contiguous copy of a channels-last weight is not kept alive)
    return t3


def _tile3_packed(packed):
    """The v3 layout of `packed`'s weights (packing it on demand), or None when the router is off / the shape has no v3 kernel."""
    if TILE3 is False or (TILE3 is None and TILE3_MIN_BLOCKS is None):
        return None
    t3 = getattr(packed, "tile3", None)
    if t3 is False:
        return None
    if t3 is None:
        src = getattr(packed, "_tile3_src", None)
        if src is None or torch.cuda.is_current_stream_capturing():
            return None  # (packing is a launch of its own: never inside a capture -- the warm-up forwards come first)
        t3 = _tile3_pack_now(packed)
    return t3


def _tile3_route(packed, T: int, C1: int, C2: int, Cout: int, kernel, stride, block):
    """TILE3 = True only: the v3 weights if this call can run on the v3 kernel at all (the forced form of the tests and tools)."""
    if TILE3 is not True:
        return None
    if tuple(kernel) != (3, 3) or tuple(stride) != (1, 1) or tuple(block) != (6, 6) or C1 % 64 or C2 % 64 or Cout % 64:
        return None
    return _tile3_packed(packed)


def tile_conv3_cl(source: int, x, x2, B, C1, C2, H, W, up, idx, smap, rx_sx, scale, shift, activationName, t3, bias, Cout,
                  full, residual, bargs, out_affine, targs, out, f16: bool = False):
    """One launch of sige_hip_tile_conv3_nhwc_f32 / _f16c (include/sige_hip.h); `full` = None (tiles) | (offH, offW, Ho, Wo).
    UNSUPPORTED -> None."""
    bias_keep = _vec(bias, "bias")
    sc = sh = None
    affB = 0
    if scale is not None:
        (sa, s_keep), (ta, t_keep) = _cvec(scale, "scale"), _cvec(shift, "shift")
        if sa[2] != C1 + C2 or ta[2] != C1 + C2 or sa[1] != ta[1]:
            return None
        sc, sh, affB = sa[0], ta[0], sa[1]
    if out_affine is not None:
        os_, oh_, oact = out_affine
        os_, oh_ = _req(os_.reshape(-1), torch.float32, "out_scale", 1), _req(oh_.reshape(-1), torch.float32, "out_shift", 1)
        if os_.numel() != Cout or oh_.numel() != Cout:
            raise RuntimeError("tile_conv3_cl: out_affine must have one entry per output channel")
        oargs = (os_.data_ptr(), oh_.data_ptr(), _act(oact))
    else:
        oargs = (None, None, 0)
    fargs = (0, 0, 0, 0, 0) if full is None else (1, *full)
    if f16:  # (fp16 operands; x2 / residual may be fp16-stored caches)
        y16 = int(x2 is not None and x2.dtype == torch.float16)
        r16 = int(residual is not None and residual.dtype == torch.float16)
        status = lib().sige_hip_tile_conv3_nhwc_f16c(
            source, x.data_ptr(), None if x2 is None else x2.data_ptr(), y16, B, C1, C2, H, W, int(bool(up)), idx.data_ptr(), idx.shape[0],
            None if smap is None else smap.data_ptr(), rx_sx[0], rx_sx[1], sc, sh, affB, _act(activationName),
            t3.data_ptr(), _p(bias_keep), Cout, *fargs, None if residual is None else residual.data_ptr(), r16, *bargs, *oargs, *targs,
            out.data_ptr(), _stream(x))
    else:
        status = lib().sige_hip_tile_conv3_nhwc_f32(
            source, x.data_ptr(), None if x2 is None else x2.data_ptr(), B, C1, C2, H, W, int(bool(up)), idx.data_ptr(), idx.shape[0],
            None if smap is None else smap.data_ptr(), rx_sx[0], rx_sx[1], sc, sh, affB, _act(activationName),
            t3.data_ptr(), _p(bias_keep), Cout, *fargs, None if residual is None else residual.data_ptr(), *bargs, *oargs, *targs,
            out.data_ptr(), _stream(x))
    if status == UNSUPPORTED:
        return None
    _check(status, "tile_conv3_cl")
    return out


def gather_conv_cl(x, x2, block: Tuple[int, int], activeIndices, scale, shift, activationName: str,
                   packed, bias, Cout: int, kernel: Tuple[int, int], stride: Tuple[int, int],
                   full: Optional[dict] = None, out_affine: Optional[tuple] = None, upsample2x: bool = False,
                   out: Optional[torch.Tensor] = None, twins=None):
    """Channels-last gather -> conv.  `full` = dict(offset=(oh, ow), out_res=(Ho, Wo), residual=tensor|None)
    writes the output tiles straight into a [B,Cout,Ho,Wo] tensor (dense layers).  `twins` (full only): up to two
    (buffer, scale, shift): buffer = SiLU(scale * result + shift), the activated input of a consumer's conv1, written by
    the same launch.  None if unsupported."""
    bias_keep = _vec(bias, "bias")
    x = _req_cl(x, "x")
    B, C1, H, W = x.shape
    if upsample2x:  # `x` is the half-resolution tensor; the tiles index its x2 nearest upsampling
        H, W = 2 * H, 2 * W
    C2 = 0
    if x2 is not None:
        x2 = _req_cl(x2, "x2")
        C2 = x2.shape[1]
    idx = _req(activeIndices, torch.int32, "activeIndices", 2)
    (sa, s_keep), (ta, t_keep) = _cvec(scale, "scale"), _cvec(shift, "shift")
    N = idx.shape[0]
    if full is None:
        Ro, So = (block[0] - kernel[0]) // stride[0] + 1, (block[1] - kernel[1]) // stride[1] + 1
        out = _empty_tiles_cl(B, idx, Cout, Ro, So, x.device)
        fargs = (0, 0, 0, None, 0, 0)
    else:
        Ho, Wo = full["out_res"]
        if out is None:
            out = _empty_cl((B, Cout, Ho, Wo), x.device)
        elif tuple(out.shape) != (B, Cout, Ho, Wo) or not out.is_contiguous(memory_format=CL):
            raise RuntimeError("gather_conv_cl: `out` must be a channels-last [B,Cout,Ho,Wo] tensor")
        r = full.get("residual")
        if r is not None:
            r = _req_cl(r, "residual")
            if tuple(r.shape) != tuple(out.shape):
                raise RuntimeError("gather_conv_cl: residual %s != output %s" % (tuple(r.shape), tuple(out.shape)))
        fargs = (1, full["offset"][0], full["offset"][1], None if r is None else r.data_ptr(), Ho, Wo)
    t3 = _tile3_route(packed, B * N, C1, C2, Cout, kernel, stride, block)  # (TILE3 = True: the v3 kernel, forced)
    if t3 is not None and N > 0:
        if twins and full is None:
            raise RuntimeError("gather_conv_cl: twins need a full-tensor destination")
        targs3, twin_keep3 = _twin_args(twins if twins else None, out, Cout, "gather_conv_cl")
        got = tile_conv3_cl(1, x, x2, B, C1, C2, H, W, upsample2x, idx, None, (0, 0), scale, shift, activationName, t3, bias, Cout,
                            None if full is None else (full["offset"][0], full["offset"][1], Ho, Wo),
                            None if full is None else r, (None, None, 0, 0, 0, 0, 0), out_affine, targs3, out,
                            f16=getattr(packed, "compute", "f32") == "f16")
        if got is not None:
            return got if full is not None else tag_tiles(got, idx, B)
    # deep-K convs over a handful of tiles: workspace for the cross-workgroup K split
    ws, ws_n = None, 0
    ks = lib().sige_hip_conv_ksplit_hint(B * N, C1 + C2, Cout, kernel[0], kernel[1], stride[0], stride[1])
    if full is not None and N * ((block[0] - kernel[0]) // stride[0] + 1) * ((block[1] - kernel[1]) // stride[1] + 1) < Ho * Wo:
        ks = 1  # tiles written into a larger tensor: the second pass would need whole output copies
    cap = _tile_capacity(idx)
    if cap is not None and full is None and KSPLIT:
        # a launch plan records: the SAME call will run under other masks.  The launch splits K only while its 16 x 16 blocks
        # do not fill the chip (sige_hip_conv_ksplit_hint: < 224 blocks); give it a workspace for the largest tile count
        # that still splits, whatever this mask's count is (the entry point picks the factor from the workspace's capacity)
        px = ((block[0] - kernel[0]) // stride[0] + 1) * ((block[1] - kernel[1]) // stride[1] + 1)
        t_split = min(B * cap, max(1, 16 // px) * (224 // max(1, -(-Cout // 16)) + 1))
        ws_n = 8 * t_split * px * Cout
        ws = torch.empty(ws_n, dtype=torch.float32, device=x.device)
    elif ks > 1 and KSPLIT:
        ws_n = ks * out.numel()
        ws = torch.empty(ws_n, dtype=torch.float32, device=x.device)
    fargs = fargs + (None if ws is None else ws.data_ptr(), ws_n)
    # epilogue affine + activation of the consumer: (scale [*,Cout,1,1], shift, activation name)
    if out_affine is not None:
        os_, oh_, oact = out_affine
        os_, oh_ = _req(os_.reshape(-1), torch.float32, "out_scale", 1), _req(oh_.reshape(-1), torch.float32, "out_shift", 1)
        if os_.numel() != Cout or oh_.numel() != Cout:
            raise RuntimeError("gather_conv_cl: out_affine must have one entry per output channel")
        fargs = fargs + (os_.data_ptr(), oh_.data_ptr(), _act(oact))
    else:
        fargs = fargs + (None, None, 0)
    if twins and full is None:
        raise RuntimeError("gather_conv_cl: twins need a full-tensor destination")
    targs, twin_keep = _twin_args(twins if twins else None, out, Cout, "gather_conv_cl")
    fargs = fargs + (int(bool(upsample2x)), *targs)
    compute = getattr(packed, "compute", "f32")
    t3r = _tile3_packed(packed) if (TILE3 is None and compute in ("f32", "f16")) else None
    if t3r is not None:  # (the routing entry point: conv_mfma.hpp or the v3 kernel, decided in C from N)
        fn = lib().sige_hip_gather_conv_nhwc_v3_f16c if compute == "f16" else lib().sige_hip_gather_conv_nhwc_v3_f32
        status = fn(
            x.data_ptr(), None if x2 is None else x2.data_ptr(), B, C1, C2, H, W, block[0], block[1], idx.data_ptr(), N,
            *sa, *ta, _act(activationName), packed.data_ptr(), _p(bias_keep), Cout, kernel[0], kernel[1],
            stride[0], stride[1], *fargs, t3r.data_ptr(), _tile3_min_blocks(packed), out.data_ptr(), _stream(x))
    else:
        status = _conv_fn("sige_hip_gather_conv_nhwc", packed)(
            x.data_ptr(), None if x2 is None else x2.data_ptr(), B, C1, C2, H, W, block[0], block[1], idx.data_ptr(), N,
            *sa, *ta, _act(activationName), packed.data_ptr(), _p(bias_keep), Cout, kernel[0], kernel[1],
            stride[0], stride[1], *fargs, out.data_ptr(), _stream(x))
    if status == UNSUPPORTED:
        return None
    _check(status, "gather_conv_cl")
    keep = getattr(_pair_state, "keep", None)
    if keep is not None:  # (conv_pair(): a held launch reads these after this call has returned)
        keep.append((x, x2, idx, s_keep, t_keep, packed, bias_keep, out, ws, out_affine, full, twin_keep))
    return out


def scatter_gather_conv_cl(x, y, block: Tuple[int, int], activeIndices, scatterMap, scale, shift, activationName: str,
                           packed, bias, Cout: int, kernel: Tuple[int, int], stride: Tuple[int, int]):
    bias_keep = _vec(bias, "bias")
    x, y = _req_cl(x, "x"), _req_cl_cache(y, "y")
    idx = _req(activeIndices, torch.int32, "activeIndices", 2)
    smap = _req(scatterMap, torch.int32, "scatterMap", 3)
    (sa, s_keep), (ta, t_keep) = _cvec(scale, "scale"), _cvec(shift, "shift")
    B, C, H, W = y.shape
    N = idx.shape[0]
    Ro, So = (block[0] - kernel[0]) // stride[0] + 1, (block[1] - kernel[1]) // stride[1] + 1
    out = _empty_tiles_cl(B, idx, Cout, Ro, So, y.device)
    args = (x.data_ptr(), y.data_ptr(), B, C, H, W, x.shape[2], x.shape[3], block[0], block[1], idx.data_ptr(), N,
            smap.data_ptr(), *sa, *ta, _act(activationName), packed.data_ptr(), _p(bias_keep), Cout, kernel[0], kernel[1],
            stride[0], stride[1], out.data_ptr(), _stream(y))
    if y.dtype == torch.float16:  # (fp16-stored cache)
        status = lib().sige_hip_scatter_gather_conv_nhwc_c16(_COMPUTE_ID[getattr(packed, "compute", "f32")], *args)
    else:
        status = _conv_fn("sige_hip_scatter_gather_conv_nhwc", packed)(*args)
    if status == UNSUPPORTED:
        return None
    _check(status, "scatter_gather_conv_cl")
    return out


def scatter_gather_conv_scatter_cl(x, y, block, activeIndices, scatterMap, scale, shift, activationName: str,
                                   packed, bias, Cout: int, kernel, offset, out: torch.Tensor, residual=None,
                                   x1=None, table1=None, twins=None):
    """scatter_gather -> 3x3 conv -> Scatter (residual = a full tensor) or ScatterWithBlockResidual (residual = the
    cached shortcut tensor, x1 = the shortcut conv's tiles, table1 = their tile table) in one launch, written into
    `out` (a persistent buffer that already equals the cache outside this mask's tiles).  None if unsupported."""
    bias_keep = _vec(bias, "bias")
    x, y = _req_cl(x, "x"), _req_cl_cache(y, "y")
    idx = _req(activeIndices, torch.int32, "activeIndices", 2)
    smap = _req(scatterMap, torch.int32, "scatterMap", 3)
    (sa, s_keep), (ta, t_keep) = _cvec(scale, "scale"), _cvec(shift, "shift")
    B, C, H, W = y.shape
    if tuple(out.shape) != (B, Cout, H, W) or not out.is_contiguous(memory_format=CL):
        raise RuntimeError("scatter_gather_conv_scatter_cl: `out` must be a channels-last [B,Cout,H,W] tensor")
    # (an fp16 residual is the cached shortcut tensor of a fused ScatterWithBlockResidual; the cache dtype is one per model)
    r = None if residual is None else (_req_cl_cache(residual, "residual") if x1 is not None else _req_cl(residual, "residual"))
    if r is not None and r.dtype == torch.float16 and y.dtype != torch.float16:
        return None
    if r is not None and tuple(r.shape) != tuple(out.shape):
        raise RuntimeError("scatter_gather_conv_scatter_cl: residual must be full size")
    if x1 is not None:
        x1 = _req_cl(x1, "x1")
        t1 = _req(table1, torch.int32, "table1", 2)
        bargs = (x1.data_ptr(), t1.data_ptr(), t1.shape[0], t1.shape[1], x1.shape[0] // B, x1.shape[2], x1.shape[3])
    else:
        bargs = (None, None, 0, 0, 0, 0, 0)
    targs, twin_keep = _twin_args(twins if twins else None, out, Cout, "scatter_gather_conv_scatter_cl")
    t3 = _tile3_route(packed, B * idx.shape[0], C, 0, Cout, kernel, (1, 1), block)
    f16c = getattr(packed, "compute", "f32") == "f16"
    if (t3 is not None and idx.shape[0] > 0 and (f16c or (y.dtype == torch.float32 and (r is None or r.dtype == torch.float32)))
            and ((scale is None and shift is None and activationName == "identity") or (scale is not None and shift is not None))):
        got = tile_conv3_cl(2, x, y, B, C, 0, H, W, False, idx, smap, (x.shape[2], x.shape[3]), scale, shift, activationName, t3, bias, Cout,
                            (offset[0], offset[1], H, W), r, bargs, None, targs, out, f16=f16c)
        if got is not None:
            return got
    head = (x.data_ptr(), y.data_ptr(), B, C, H, W, x.shape[2], x.shape[3], block[0], block[1], idx.data_ptr(), idx.shape[0],
            smap.data_ptr(), *sa, *ta, _act(activationName), packed.data_ptr(), _p(bias_keep), Cout, kernel[0], kernel[1],
            offset[0], offset[1], None if r is None else r.data_ptr())
    t3r = _tile3_packed(packed) if (TILE3 is None and ((getattr(packed, "compute", "f32") == "f32" and y.dtype == torch.float32) or f16c)) else None
    if f16c and t3r is not None:  # (fp16 operands, fp32- or fp16-stored caches: routed in C like the fp32 pair)
        status = lib().sige_hip_scatter_gather_conv_scatter_nhwc_v3_f16c(
            head[0], head[1], int(y.dtype == torch.float16), *head[2:], int(r is not None and r.dtype == torch.float16), *bargs, *targs,
            t3r.data_ptr(), _tile3_min_blocks(packed), out.data_ptr(), _stream(y))
    elif y.dtype == torch.float16:  # (fp16-stored caches)
        status = lib().sige_hip_scatter_gather_conv_scatter_nhwc_c16(
            _COMPUTE_ID[getattr(packed, "compute", "f32")], *head, int(r is not None and r.dtype == torch.float16), *bargs, *targs,
            out.data_ptr(), _stream(y))
    elif t3r is not None:  # (the routing entry point: conv_mfma.hpp or the v3 kernel, decided in C from N)
        status = lib().sige_hip_scatter_gather_conv_scatter_nhwc_v3_f32(*head, *bargs, *targs, t3r.data_ptr(), _tile3_min_blocks(packed),
                                                                        out.data_ptr(), _stream(y))
    else:
        status = _conv_fn("sige_hip_scatter_gather_conv_scatter_nhwc", packed)(*head, *bargs, *targs, out.data_ptr(), _stream(y))
    if status == UNSUPPORTED:
        return None
    _check(status, "scatter_gather_conv_scatter_cl")
    return out


def gather_cl(x, bSizeH, bSizeW, activeIndices, scale=None, shift=None, activationName="identity"):
    x = _req_cl_cache(x, "x")
    idx = _req(activeIndices, torch.int32, "activeIndices", 2)
    (sa, s_keep), (ta, t_keep) = _cvec(scale, "scale"), _cvec(shift, "shift")
    B, C, H, W = x.shape
    N = idx.shape[0]
    out = _empty_tiles_cl(B, idx, C, bSizeH, bSizeW, x.device)
    fn = lib().sige_hip_gather_nhwc_f16 if x.dtype == torch.float16 else lib().sige_hip_gather_nhwc_f32
    _check(fn(x.data_ptr(), B, C, H, W, bSizeH, bSizeW, idx.data_ptr(), N, *sa, *ta, _act(activationName), out.data_ptr(), _stream(x)),
           "gather_cl")
    return tag_tiles(out, idx, B)


def scatter_gather_cl(x, y, bSizeH, bSizeW, activeIndices, scatterMap, scale=None, shift=None,
                      activationName="identity"):
    x, y = _req_cl(x, "x"), _req_cl_cache(y, "y")
    idx = _req(activeIndices, torch.int32, "activeIndices", 2)
    smap = _req(scatterMap, torch.int32, "scatterMap", 3)
    (sa, s_keep), (ta, t_keep) = _cvec(scale, "scale"), _cvec(shift, "shift")
    B, C, H, W = y.shape
    N = idx.shape[0]
    out = _empty_tiles_cl(B, idx, C, bSizeH, bSizeW, y.device)
    fn = lib().sige_hip_scatter_gather_nhwc_f16 if y.dtype == torch.float16 else lib().sige_hip_scatter_gather_nhwc_f32
    _check(fn(x.data_ptr(), y.data_ptr(), B, C, H, W, x.shape[2], x.shape[3], bSizeH, bSizeW, idx.data_ptr(), N, smap.data_ptr(),
              *sa, *ta, _act(activationName), out.data_ptr(), _stream(y)), "scatter_gather_cl")
    return tag_tiles(out, idx, B)


def spade_modulate_cl(x_full, x_tiles, map_x, scale, shift, gb_tiles, gb_full, map_g, activeIndices, block: Tuple[int, int],
                      slope: Optional[float] = None):
    """SPADE modulation of the tiles at `activeIndices` in one pass (include/sige_hip.h): leaky(n * (1 + gamma) + beta) with
    n = scale * X + shift, X gathered from `x_full` (x_tiles None) or scatter-gathered from (x_tiles, x_full) through map_x;
    gamma | beta scatter-gathered from (gb_tiles, gb_full) through map_g.  slope None = no activation.  -> [B*N,C,bH,bW]."""
    x_full = _req_cl(x_full, "x_full")
    gb_full = _req_cl(gb_full, "gb_full")
    gb_tiles = _req_cl(gb_tiles, "gb_tiles")
    idx = _req(activeIndices, torch.int32, "activeIndices", 2)
    map_g = _req(map_g, torch.int32, "map_g", 3)
    B, C, H, W = x_full.shape
    if gb_full.shape[1] != 2 * C or gb_tiles.shape[1] != 2 * C or tuple(gb_full.shape[2:]) != (H, W):
        raise RuntimeError("spade_modulate_cl: gamma|beta must have 2*C channels at the resolution of x")
    xt = (None, None, 0, 0, 0)
    if x_tiles is not None:
        x_tiles = _req_cl(x_tiles, "x_tiles")
        map_x = _req(map_x, torch.int32, "map_x", 3)
        xt = (x_tiles.data_ptr(), map_x.data_ptr(), x_tiles.shape[0] // B, x_tiles.shape[2], x_tiles.shape[3])
    (sa, s_keep), (ta, t_keep) = _cvec(scale, "scale"), _cvec(shift, "shift")
    N = idx.shape[0]
    out = _empty_tiles_cl(B, idx, C, block[0], block[1], x_full.device)
    _check(lib().sige_hip_spade_modulate_nhwc_f32(
        x_full.data_ptr(), xt[0], xt[1], xt[2], xt[3], xt[4], *sa, *ta,
        gb_tiles.data_ptr(), gb_full.data_ptr(), map_g.data_ptr(), gb_tiles.shape[0] // B, gb_tiles.shape[2], gb_tiles.shape[3],
        B, C, H, W, block[0], block[1], idx.data_ptr(), N, int(slope is not None), float(slope or 0.0), out.data_ptr(),
        _stream(x_full)), "spade_modulate_cl")
    return tag_tiles(out, idx, B)


# ---- token helpers of the SD spatial transformer (csrc/token_ops.hip) --------------------------------------------------------------
def _tok(t: torch.Tensor, name: str) -> torch.Tensor:
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 3):
        raise NotImplementedError("sige_amd.hip: `%s` must be fp32 GPU tokens [B,N,C]" % name)
    return t if t.is_contiguous() else t.contiguous()


def add_layer_norm_tokens(x, delta, bias, norm: torch.nn.LayerNorm, want_sum: bool = True):
    """(x + delta + bias, LayerNorm(x + delta + bias)) in one launch (delta None: just LayerNorm(x), the sum is x itself)."""
    x = _tok(x, "x")
    B, N, C = x.shape
    d = None if delta is None else _tok(delta, "delta")
    out = torch.empty_like(x)
    s = torch.empty_like(x) if (d is not None and want_sum) else None
    _check(lib().sige_hip_add_layer_norm_tokens_f32(x.data_ptr(), _p(d), _p(bias if d is not None else None), norm.weight.data_ptr(),
                                                    norm.bias.data_ptr(), B * N, C, float(norm.eps), _p(s), out.data_ptr(), _stream(x)),
           "add_layer_norm_tokens")
    return (x if d is None else s), out


def geglu_tokens(x):
    """a * gelu(gate) for x = [a | gate] along the last axis, one launch."""
    x = _tok(x, "x")
    B, N, C2 = x.shape
    out = torch.empty((B, N, C2 // 2), dtype=torch.float32, device=x.device)
    _check(lib().sige_hip_geglu_tokens_f32(x.data_ptr(), B * N, C2 // 2, out.data_ptr(), _stream(x)), "geglu_tokens")
    return out


def add_bias_tokens(x, delta, bias):
    """x + delta + bias in one launch."""
    x, delta = _tok(x, "x"), _tok(delta, "delta")
    out = torch.empty_like(x)
    _check(lib().sige_hip_add_bias_tokens_f32(x.data_ptr(), delta.data_ptr(), _p(bias), x.shape[0] * x.shape[1], x.shape[2], out.data_ptr(),
                                              _stream(x)), "add_bias_tokens")
    return out


# ---- GauGAN helpers (csrc/spade_ops.hip): the sparse forward of the SPADE generator without a torch kernel ----------------------
def resize_nearest_cl(x, size: Tuple[int, int]):
    """F.interpolate(x, size=size, mode="nearest") for an integer factor up or down, channels-last; None if unsupported."""
    x = _req_cl(x, "x")
    B, C, H, W = x.shape
    out = _empty_cl((B, C, int(size[0]), int(size[1])), x.device)
    status = lib().sige_hip_resize_nearest_nhwc_f32(x.data_ptr(), B, C, H, W, int(size[0]), int(size[1]), out.data_ptr(), _stream(x))
    if status == UNSUPPORTED:
        return None
    _check(status, "resize_nearest_cl")
    return out


def act_split_cl(x, parts: int, activationName: str = "relu", slope: float = 0.0):
    """tuple(act(x[:, k*C/parts:(k+1)*C/parts]) for k in range(parts)), each a dense channels-last tensor, in one pass."""
    x = _req_cl(x, "x")
    B, C, H, W = x.shape
    Cp = C // parts
    out = torch.empty((parts, B, H, W, Cp), dtype=torch.float32, device=x.device)
    _check(lib().sige_hip_act_split_nhwc_f32(x.data_ptr(), B * H * W, C, parts, B * H * W * Cp, ACT_EXT[activationName], float(slope),
                                             out.data_ptr(), _stream(x)), "act_split_cl")
    return tuple(out[k].permute(0, 3, 1, 2) for k in range(parts))


def scatter_gather_split_cl(x, y, bSizeH, bSizeW, activeIndices, scatterMap, parts: int, activationName: str = "relu", slope: float = 0.0):
    """scatter_gather_cl (no affine) + activation + a split into `parts` channel groups in one pass: a tuple of `parts` dense
    channels-last tile slabs [B*N, C/parts, bH, bW].  While a launch plan records, every part is backed by memory for every
    candidate tile, so the parts keep their addresses under later masks."""
    x, y = _req_cl(x, "x"), _req_cl(y, "y")
    idx = _req(activeIndices, torch.int32, "activeIndices", 2)
    smap = _req(scatterMap, torch.int32, "scatterMap", 3)
    B, C, H, W = y.shape
    N = idx.shape[0]
    Cp = C // parts
    cap = _tile_capacity(idx)
    cap = N if cap is None else max(cap, N)
    out = torch.empty((parts, B * cap, bSizeH, bSizeW, Cp), dtype=torch.float32, device=y.device)
    _check(lib().sige_hip_scatter_gather_split_nhwc_f32(
        x.data_ptr(), y.data_ptr(), B, C, H, W, x.shape[2], x.shape[3], bSizeH, bSizeW, idx.data_ptr(), N, smap.data_ptr(),
        ACT_EXT[activationName], float(slope), parts, B * cap * bSizeH * bSizeW * Cp, out.data_ptr(), _stream(y)), "scatter_gather_split_cl")
    return tuple(tag_tiles(out[k, :B * N].permute(0, 3, 1, 2), idx, B) for k in range(parts))


def spade_modulate_dense_cl(x, scale, shift, gb, slope: Optional[float] = None):
    """leaky((scale * x + shift) * (1 + gamma) + beta) on a full channels-last tensor; gb [B,2C,H,W] = gamma | beta."""
    x, gb = _req_cl(x, "x"), _req_cl(gb, "gb")
    B, C, H, W = x.shape
    if tuple(gb.shape) != (B, 2 * C, H, W):
        raise RuntimeError("spade_modulate_dense_cl: gamma|beta must be [B, 2C, H, W]")
    (sa, s_keep), (ta, t_keep) = _cvec(scale, "scale"), _cvec(shift, "shift")
    if s_keep is None or t_keep is None or sa[1] != ta[1] or sa[2] != C or ta[2] != C:
        raise RuntimeError("spade_modulate_dense_cl: scale and shift [1|B, C, 1, 1]")
    out = _empty_cl((B, C, H, W), x.device)
    _check(lib().sige_hip_spade_modulate_dense_nhwc_f32(x.data_ptr(), sa[0], ta[0], sa[1], gb.data_ptr(), B, C, H, W,
                                                        int(slope is not None), float(slope or 0.0), out.data_ptr(), _stream(x)),
           "spade_modulate_dense_cl")
    return out


def conv3x3_small_cout_act_cl(x, weight, bias, activationName: str = "identity", slope: float = 0.0, outActivationName: str = "identity"):
    """out_act(conv(act(x))) for a 3x3 / padding-1 conv with <= 4 output channels on a full channels-last tensor (GauGAN's
    tanh(conv_img(leaky_relu(x)))); None if unsupported."""
    x = _req_cl(x, "x")
    B, C, H, W = x.shape
    w = _req(weight.detach(), torch.float32, "weight")
    Cout = w.shape[0]
    if tuple(w.shape[1:]) != (C, 3, 3):
        return None
    bias_keep = _vec(bias, "bias")
    out = _empty_cl((B, Cout, H, W), x.device)
    status = lib().sige_hip_conv3x3_small_cout_act_nhwc_f32(x.data_ptr(), B, C, H, W, ACT_EXT[activationName], float(slope), w.data_ptr(),
                                                            _p(bias_keep), Cout, ACT_EXT[outActivationName], out.data_ptr(), _stream(x))
    if status == UNSUPPORTED:
        return None
    _check(status, "conv3x3_small_cout_act_cl")
    return out


def scatter_cl(x, y, offset, stride, activeIndices, table, residual=None, out: Optional[torch.Tensor] = None):
    """Channels-last scatter.  `out=None`: reference semantics (a fresh tensor, one pass).
    `out=buffer`: in-place form -- `buffer` already equals y outside this mask's tiles; only
    the covered pixels are written and `buffer` is returned."""
    x, y = _req_cl(x, "x"), _req_cl_cache(y, "y")
    idx = _req(activeIndices, torch.int32, "activeIndices", 2)
    table = _req(table, torch.int32, "table", 2)
    B, C, H, W = y.shape
    r = None
    if residual is not None:
        if tuple(residual.shape) != tuple(y.shape):
            raise RuntimeError("scatter_cl: the channels-last path takes a full-size residual")
        r = _req_cl(residual, "residual")
    in_place = out is not None
    if out is None:
        out = _empty_cl(tuple(y.shape), y.device)
    fn = lib().sige_hip_scatter_nhwc_f16 if y.dtype == torch.float16 else lib().sige_hip_scatter_nhwc_f32
    _check(fn(x.data_ptr(), y.data_ptr(), B, C, H, W, x.shape[2], x.shape[3], offset[0], offset[1], stride[0], stride[1],
              idx.data_ptr(), table.data_ptr(), table.shape[0], table.shape[1], idx.shape[0],
              None if r is None else r.data_ptr(), int(in_place), out.data_ptr(), _stream(y)), "scatter_cl")
    return out


def scatter_with_block_residual_cl(x0, y0, x1, y1, offset, stride, idx0, table0, idx1, table1,
                                   out: Optional[torch.Tensor] = None):
    x0, y0, x1, y1 = _req_cl(x0, "x0"), _req_cl_cache(y0, "y0"), _req_cl(x1, "x1"), _req_cl_cache(y1, "y1")
    if y0.dtype != y1.dtype:
        raise NotImplementedError("scatter_with_block_residual_cl: y0 and y1 must be stored in one dtype")
    i0, i1 = _req(idx0, torch.int32, "idx0", 2), _req(idx1, torch.int32, "idx1", 2)
    t0, t1 = _req(table0, torch.int32, "table0", 2), _req(table1, torch.int32, "table1", 2)
    B, C, H, W = y0.shape
    in_place = out is not None
    if out is None:
        out = _empty_cl(tuple(y0.shape), y0.device)
    fn = (lib().sige_hip_scatter_with_block_residual_nhwc_f16 if y0.dtype == torch.float16
          else lib().sige_hip_scatter_with_block_residual_nhwc_f32)
    _check(fn(
        x0.data_ptr(), y0.data_ptr(), x1.data_ptr(), y1.data_ptr(), B, C, H, W,
        x0.shape[2], x0.shape[3], x1.shape[2], x1.shape[3], offset[0], offset[1], stride[0], stride[1],
        i0.data_ptr(), t0.data_ptr(), t0.shape[0], t0.shape[1], i0.shape[0],
        i1.data_ptr(), t1.data_ptr(), t1.shape[0], t1.shape[1], i1.shape[0],
        int(in_place), out.data_ptr(), _stream(y0)), "scatter_with_block_residual_cl")
    return out


def group_norm_affine_cl(x, groups: int, eps: float, gamma=None, beta=None, channel_bias=None):
    """Channels-last GroupNorm statistics -> (scale, shift) [B,C,1,1]; None if the shape has no kernel.
    `channel_bias` [C]: the statistics are those of x + channel_bias, the affine is for x: GN(x + bias) == x * scale + shift."""
    x = _req_cl(x, "x")
    B, C, H, W = x.shape
    n = int(lib().sige_hip_group_norm_affine_nhwc_workspace(B, C, H, W, groups))
    if n == 0:
        return None
    buf = torch.empty(n + 2 * B * C, dtype=torch.float32, device=x.device)
    scale, shift = buf[n:n + B * C].view(B, C, 1, 1), buf[n + B * C:].view(B, C, 1, 1)
    gamma_keep = _vec(gamma, "gamma")
    ga = _p(gamma_keep)
    beta_keep = _vec(beta, "beta")
    be = _p(beta_keep)
    cb_keep = _vec(channel_bias, "channel_bias")
    if cb_keep is not None:
        if cb_keep.numel() != C:
            raise RuntimeError("group_norm_affine_cl: channel_bias must have one entry per channel")
        status = lib().sige_hip_group_norm_affine_nhwc_bias_f32(x.data_ptr(), B, C, H, W, groups, eps, ga, be, cb_keep.data_ptr(),
                                                                buf.data_ptr(), scale.data_ptr(), shift.data_ptr(), _stream(x))
    else:
        status = lib().sige_hip_group_norm_affine_nhwc_f32(x.data_ptr(), B, C, H, W, groups, eps, ga, be, buf.data_ptr(),
                                                           scale.data_ptr(), shift.data_ptr(), _stream(x))
    if status == UNSUPPORTED:
        return None
    _check(status, "group_norm_affine_cl")
    return scale, shift


def affine_act_cl(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, activationName: str,
                  out: Optional[torch.Tensor] = None):
    """act(scale * x + shift) over a whole channels-last tensor in one launch (scale / shift [1|B, C, 1, 1]); written into `out`
    (same shape and layout) when given.  None if unsupported."""
    x = _req_cl_cache(x, "x")
    B, C, H, W = x.shape
    (sa, s_keep), (ta, t_keep) = _cvec(scale, "scale"), _cvec(shift, "shift")
    if s_keep is None or t_keep is None or sa[1] != ta[1] or sa[2] != C or ta[2] != C or sa[1] not in (1, B) or C % 4:
        return None
    if out is None:  # (an fp16-stored cache gets an fp16 activated copy)
        out = torch.empty((B, C, H, W), dtype=x.dtype, device=x.device, memory_format=CL)
    elif tuple(out.shape) != (B, C, H, W) or not out.is_contiguous(memory_format=CL) or out.dtype not in (torch.float32, torch.float16):
        return None
    if x.dtype == torch.float16:
        status = lib().sige_hip_affine_act_nhwc_f16(x.data_ptr(), B, C, H, W, sa[0], ta[0], sa[1], _act(activationName), out.data_ptr(),
                                                    int(out.dtype == torch.float16), _stream(x))
    elif out.dtype != torch.float32:
        return None
    else:
        status = lib().sige_hip_affine_act_nhwc_f32(x.data_ptr(), B, C, H, W, sa[0], ta[0], sa[1], _act(activationName), out.data_ptr(),
                                                    _stream(x))
    if status == UNSUPPORTED:
        return None
    _check(status, "affine_act_cl")
    return out


# attention_cl: scores + softmax + values in ONE launch (csrc/attention_fused.hip) instead of two.  Off: measured on MI355X
# (profiles/r4h_attention_ab.json) the one-launch form is 18.8 us against 11.8 us at DDPM's 256 tokens x 512 channels (its chain
# loads -> scores -> softmax -> values -> partials -> ticket -> combine is serial in every workgroup; the two-launch form spreads
# each phase over 256 / 128 workgroups) and the forward 1.435 ms against 1.403 ms, with 6 launches fewer.
FUSED_ATTENTION = False


def attention_cl(qkv: torch.Tensor, scale: float):
    """Channels-last attention: qkv [B,3C,H,W] stored [B,H,W,3C] -> [B,C,H,W] channels-last; None if unsupported."""
    qkv = _req_cl(qkv, "qkv")
    B, C3, H, W = qkv.shape
    C, HW = C3 // 3, H * W
    out = _empty_cl((B, C, H, W), qkv.device)
    if FUSED_ATTENTION:  # one launch: the key slices of a query block are combined by the last one to finish
        n = int(lib().sige_hip_attention_fused_workspace(B, C, HW))
        if n:
            ws = torch.empty(n, dtype=torch.float32, device=qkv.device)
            status = lib().sige_hip_attention_fused_nhwc_f32(qkv.data_ptr(), B, C, HW, float(scale), ws.data_ptr(), out.data_ptr(),
                                                             _stream(qkv))
            if status != UNSUPPORTED:
                _check(status, "attention_fused_cl")
                return out
    ws = torch.empty(B * HW * HW, dtype=torch.float32, device=qkv.device)
    status = lib().sige_hip_attention_nhwc_f32(qkv.data_ptr(), B, C, HW, float(scale), ws.data_ptr(), out.data_ptr(),
                                               _stream(qkv))
    if status == UNSUPPORTED:
        return None
    _check(status, "attention_cl")
    return out


def attention_tokens(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: float) -> Optional[torch.Tensor]:
    """softmax(scale * q k^T) v per (batch, head) for token matrices q [B,Nq,C], k / v [B,Nk,C] (C = heads * d), in ONE launch
    with the heads as strides (include/sige_hip.h: sige_hip_attention_tokens_f32) -> [B,Nq,C].  None if unsupported (then the
    caller runs the reference's rearrange / bmm / softmax / bmm chain)."""
    if not (q.is_cuda and q.dtype == k.dtype == v.dtype == torch.float32 and q.dim() == k.dim() == v.dim() == 3):
        return None
    B, Nq, C = q.shape
    if tuple(k.shape) != tuple(v.shape) or k.shape[0] != B or k.shape[2] != C:
        return None
    Nk = k.shape[1]
    if not lib().sige_hip_attention_tokens_supported(Nq, Nk, C, heads):
        return None
    q, k, v = (t if t.is_contiguous() else t.contiguous() for t in (q, k, v))
    out = torch.empty((B, Nq, C), dtype=torch.float32, device=q.device)
    status = lib().sige_hip_attention_tokens_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), B, Nq, Nk, C, heads, float(scale),
                                                 out.data_ptr(), _stream(q))
    if status == UNSUPPORTED:
        return None
    _check(status, "attention_tokens")
    return out


def conv3x3_small_cout_cl(x, weight, bias, scale=None, shift=None, activationName="identity"):
    """conv(act(scale*x + shift)) for a 3x3 / padding-1 conv with <= 4 output channels on a full
    channels-last tensor (the U-Net's norm_out -> swish -> conv_out tail); None if unsupported."""
    x = _req_cl(x, "x")
    B, C, H, W = x.shape
    w = _req(weight.detach(), torch.float32, "weight")
    Cout = w.shape[0]
    if tuple(w.shape[1:]) != (C, 3, 3):
        return None
    bias_keep = _vec(bias, "bias")
    (sa, s_keep), (ta, t_keep) = _cvec(scale, "scale"), _cvec(shift, "shift")
    out = _empty_cl((B, Cout, H, W), x.device)
    status = lib().sige_hip_conv3x3_small_cout_nhwc_f32(x.data_ptr(), B, C, H, W, *sa, *ta, _act(activationName),
                                                        w.data_ptr(), _p(bias_keep), Cout, out.data_ptr(), _stream(x))
    if status == UNSUPPORTED:
        return None
    _check(status, "conv3x3_small_cout_cl")
    return out


def conv3x3_small_cout_force_scalar(on: bool):
    """Benchmarking / tests: route conv3x3_small_cout_cl through the scalar-weight kernel only."""
    tuning_set("small_cout_scalar", int(bool(on)))


def conv3x3_small_cin_cl(x, weight, bias, tiles=None, out=None):
    """conv(x) + bias for a 3x3 / padding-1 conv with <= 3 input channels and 32 / 64 / 128 output channels
    over a full image (the U-Net's conv_in); x in any dense layout, result channels-last.  `tiles` = (index list, (bH, bW)):
    only those windows, written into `out`.  None if unsupported."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        raise ValueError("x must be a 4-D fp32 GPU tensor")
    bias_keep = _vec(bias, "bias")
    B, C, H, W = x.shape
    w = _req(weight.detach(), torch.float32, "weight")
    Cout = w.shape[0]
    if tuple(w.shape[1:]) != (C, 3, 3) or C > 3 or Cout not in (32, 64, 128):
        return None
    sb, sc, sh, sw = x.stride()
    if tiles is not None:
        # (round 6) only the block[0] x block[1] windows at the index list are evaluated, in place in `out` (a persistent buffer the
        # caller owns: every other pixel keeps whatever it held -- nobody reads it: sige_hip_conv3x3_small_cin_tiles_nhwc_f32)
        idx, block = tiles
        idx = _req(idx, torch.int32, "activeIndices", 2)
        if out is None or tuple(out.shape) != (B, Cout, H, W) or not out.is_contiguous(memory_format=CL) or out.dtype != torch.float32:
            raise RuntimeError("conv3x3_small_cin_cl: the tile-list form writes into a channels-last fp32 [B,Cout,H,W] buffer")
        status = lib().sige_hip_conv3x3_small_cin_tiles_nhwc_f32(x.data_ptr(), sb, sc, sh, sw, B, C, H, W, w.data_ptr(), _p(bias_keep),
                                                                 Cout, idx.data_ptr(), idx.shape[0], block[0], block[1], out.data_ptr(),
                                                                 _stream(x))
    else:
        out = _empty_cl((B, Cout, H, W), x.device)
        status = lib().sige_hip_conv3x3_small_cin_nhwc_f32(x.data_ptr(), sb, sc, sh, sw, B, C, H, W, w.data_ptr(), _p(bias_keep),
                                                           Cout, out.data_ptr(), _stream(x))
    if status == UNSUPPORTED:
        return None
    _check(status, "conv3x3_small_cin_cl")
    return out
