"""MultiScaleDeformableAttnFunction_fp32 / _fp16 -- same autograd surface as
projects/mmdet3d_plugin/bevformer/modules/multi_scale_deformable_attn_function.py:15-163,
backed by vidar_msda_{fwd,bwd}_f32 (gfx950 HIP) instead of mmcv._ext."""
from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ..._lib import lib, check, ptr, stream_of, TIMER


def msda_fwd_bytes(B, Nv, H, C, Nq, L, P):
    """ALGORITHMIC bytes: value + (loc, w) + out, each element once."""
    return 4 * (B * Nv * H * C + B * Nq * H * L * P * 3 + B * Nq * H * C)


def msda_bwd_bytes(B, Nv, H, C, Nq, L, P):
    return msda_fwd_bytes(B, Nv, H, C, Nq, L, P) + 4 * (B * Nq * H * C + B * Nv * H * C
                                                       + B * Nq * H * L * P * 3)


def _msda_forward(value, shapes, lsi, loc, w):
    B, Nv, H, C = value.shape
    _, Nq, H2, L, P, two = loc.shape
    if H2 != H or two != 2 or w.shape != (B, Nq, H, L, P):
        raise RuntimeError("inconsistent MSDA operand shapes")
    out = torch.empty((B, Nq, H * C), dtype=torch.float32, device=value.device)
    with TIMER.span(f"msda_fwd[L={L},P={P}]", msda_fwd_bytes(B, Nv, H, C, Nq, L, P)):
        check(lib().vidar_msda_fwd_f32(ptr(value), ptr(shapes), ptr(lsi), ptr(loc), ptr(w), ptr(out),
                                       B, Nv, H, C, Nq, L, P, stream_of(value)), "ms_deform_attn_forward")
    return out


# launches with at least this many (b, q, head, level, point) samples take the destination-binned
# scatter (5 launches, no atomic wall); smaller ones the one-launch atomic scatter.  Same results up to
# fp32 summation order; tests/test_msda_gpu.py runs every case under both.
BINNED_MIN_SAMPLES = 1 << 18


def _bwd_workspace(value, B, Nv, H, Nq, L, P, binned):
    import ctypes
    if binned is None:
        binned = B * Nq * H * L * P >= BINNED_MIN_SAMPLES
    if not binned:
        return None, 0
    f = lib().vidar_msda_bwd_workspace_bytes
    f.restype = ctypes.c_size_t
    n = int(f(B, Nv, H, Nq, L, P))
    if n == 0:
        return None, 0
    return torch.empty((n + 7) // 8, dtype=torch.int64, device=value.device), n


def _msda_backward(value, shapes, lsi, loc, w, grad_out, binned=None):
    import ctypes
    B, Nv, H, C = value.shape
    _, Nq, _, L, P, _ = loc.shape
    gv = torch.empty_like(value)
    gl = torch.empty_like(loc)
    gw = torch.empty_like(w)
    ws, nbytes = _bwd_workspace(value, B, Nv, H, Nq, L, P, binned)
    with TIMER.span(f"msda_bwd[L={L},P={P}]", msda_bwd_bytes(B, Nv, H, C, Nq, L, P)):
        check(lib().vidar_msda_bwd_f32(ptr(value), ptr(shapes), ptr(lsi), ptr(loc), ptr(w),
                                       ptr(grad_out), ptr(gv), ptr(gl), ptr(gw), B, Nv, H, C, Nq, L, P,
                                       ptr(ws), ctypes.c_size_t(nbytes), stream_of(value)),
              "ms_deform_attn_backward")
    return gv, gl, gw


def _prep(value, shapes, lsi, loc, w):
    if not value.is_cuda:
        raise RuntimeError("MultiScaleDeformableAttnFunction needs CUDA tensors (no CPU fallback)")
    f = lambda t: t.float().contiguous()
    i = lambda t: t.to(device=value.device, dtype=torch.int64).contiguous()
    return f(value), i(shapes), i(lsi), f(loc), f(w)


class MultiScaleDeformableAttnFunction_fp32(Function):
    """apply(value, value_spatial_shapes, value_level_start_index, sampling_locations,
    attention_weights, im2col_step) -> [B, Nq, H*C]; inputs are cast to fp32 like the reference's
    custom_fwd(cast_inputs=torch.float32) (function.py:92)."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step):
        ctx.im2col_step = im2col_step          # accepted, unused
        ctx.in_dtypes = (value.dtype, sampling_locations.dtype, attention_weights.dtype)
        v, s, l, loc, w = _prep(value, value_spatial_shapes, value_level_start_index,
                                sampling_locations, attention_weights)
        out = _msda_forward(v, s, l, loc, w)
        ctx.save_for_backward(v, s, l, loc, w)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        v, s, l, loc, w = ctx.saved_tensors
        gv, gl, gw = _msda_backward(v, s, l, loc, w, grad_output.float().contiguous())
        dv, dl, dw = ctx.in_dtypes
        return gv.to(dv), None, None, gl.to(dl), gw.to(dw), None


# the reference's fp16 variant differs only in the autocast decorator (function.py:15-87)
MultiScaleDeformableAttnFunction_fp16 = MultiScaleDeformableAttnFunction_fp32


def multi_scale_deformable_attn(value, spatial_shapes, level_start_index, sampling_locations,
                                attention_weights, im2col_step=64):
    return MultiScaleDeformableAttnFunction_fp32.apply(value, spatial_shapes, level_start_index,
                                                       sampling_locations, attention_weights,
                                                       im2col_step)
