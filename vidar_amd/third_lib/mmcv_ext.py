"""`mmcv._ext` surface for the two ops the reference binds with
`ext_loader.load_ext('_ext', ['ms_deform_attn_backward', 'ms_deform_attn_forward'])`
(bevformer/modules/multi_scale_deformable_attn_function.py:11-12; also spatial_cross_attention.py:26-27,
temporal_self_attention.py:20-21, encoder.py:20-21, vidar_decoder.py:20-21), same positional arguments
and the `im2col_step` keyword of its call sites (function.py:42-48, :74-84, :118-124, :150-160)."""
from __future__ import annotations

import ctypes

import torch

from .._lib import lib, check, ptr, stream_of


def _dims(value, loc):
    B, Nv, H, C = value.shape
    _, Nq, H2, L, P, two = loc.shape
    if H2 != H or two != 2:
        raise RuntimeError("inconsistent MSDA operand shapes")
    return B, Nv, H, C, Nq, L, P


def _ok(*ts):
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError("ms_deform_attn: CUDA tensors required (vidar_amd has no CPU path)")
        if not t.is_contiguous():
            raise RuntimeError("ms_deform_attn: tensors must be contiguous")


def _dtypes(floats, ints):
    """raw pointers cross the C ABI: anything but float32 operands / int64 shape tables would be reinterpreted
    silently (the reference's MultiScaleDeformableAttnFunction_fp16 casts to float32 before calling the op)"""
    for t in floats:
        if t.dtype != torch.float32:
            raise RuntimeError(f"ms_deform_attn: float32 tensors required, got {t.dtype}")
    for t in ints:
        if t.dtype != torch.int64:
            raise RuntimeError(f"ms_deform_attn: int64 spatial_shapes / level_start_index required, got {t.dtype}")


def ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                           attention_weights, im2col_step=64):
    """-> output [bs, num_queries, num_heads * channels]; `im2col_step` accepted, meaningless here."""
    _ok(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights)
    _dtypes((value, sampling_locations, attention_weights), (value_spatial_shapes, value_level_start_index))
    B, Nv, H, C, Nq, L, P = _dims(value, sampling_locations)
    out = torch.empty((B, Nq, H * C), dtype=torch.float32, device=value.device)
    check(lib().vidar_msda_fwd_f32(ptr(value), ptr(value_spatial_shapes), ptr(value_level_start_index),
                                   ptr(sampling_locations), ptr(attention_weights), ptr(out), B, Nv, H, C,
                                   Nq, L, P, stream_of(value)), "ms_deform_attn_forward")
    return out


def ms_deform_attn_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                            attention_weights, grad_output, grad_value, grad_sampling_loc, grad_attn_weight,
                            im2col_step=64):
    """Fills the three caller-allocated gradient buffers (the reference passes zeros_like buffers,
    function.py:146-148; this op overwrites them, which is the same result).  Returns None like mmcv."""
    _ok(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
        grad_output, grad_value, grad_sampling_loc, grad_attn_weight)
    _dtypes((value, sampling_locations, attention_weights, grad_output, grad_value, grad_sampling_loc,
             grad_attn_weight), (value_spatial_shapes, value_level_start_index))
    B, Nv, H, C, Nq, L, P = _dims(value, sampling_locations)
    from ..plugin.modules.multi_scale_deformable_attn_function import _bwd_workspace
    ws, nbytes = _bwd_workspace(value, B, Nv, H, Nq, L, P, None)
    check(lib().vidar_msda_bwd_f32(ptr(value), ptr(value_spatial_shapes), ptr(value_level_start_index),
                                   ptr(sampling_locations), ptr(attention_weights), ptr(grad_output),
                                   ptr(grad_value), ptr(grad_sampling_loc), ptr(grad_attn_weight), B, Nv, H, C,
                                   Nq, L, P, ptr(ws), ctypes.c_size_t(nbytes), stream_of(value)),
          "ms_deform_attn_backward")
