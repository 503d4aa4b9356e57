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


# --------------------------------------------------------------------------------------------------
# fused operand preparation: the attention modules hand the raw outputs of their sampling_offsets /
# attention_weights Linear layers to the op (csrc/msda.hip, vidar_msda_fused_{fwd,bwd}_f32)
# --------------------------------------------------------------------------------------------------
def compose_operands(off_raw, logit_raw, ref, shapes, H, Qn, L, P, mode):
    """the reference's tensor program for (sampling_locations, attention_weights):
    temporal_self_attention.py:218-245 / vidar_decoder.py:463-490 (mode 0: one reference point per level,
    Qn BEV-queue entries folded into the batch) and spatial_cross_attention.py:359-383 (mode 1: point p
    belongs to pillar anchor p % Zn).  off_raw [bs, Nq, H*Qn*L*P*2], logit_raw [bs, Nq, H*Qn*L*P],
    ref [bs*Qn, Nq, R, 2]."""
    bs, Nq = off_raw.shape[:2]
    normalizer = torch.stack([shapes[..., 1], shapes[..., 0]], -1)
    if mode == 0:
        offsets = off_raw.view(bs, Nq, H, Qn, L, P, 2)
        weights = logit_raw.view(bs, Nq, H, Qn, L * P).softmax(-1)
        weights = weights.view(bs, Nq, H, Qn, L, P).permute(0, 3, 1, 2, 4, 5).reshape(bs * Qn, Nq, H, L, P).contiguous()
        offsets = offsets.permute(0, 3, 1, 2, 4, 5, 6).reshape(bs * Qn, Nq, H, L, P, 2)
        locations = ref[:, :, None, :, None, :] + offsets / normalizer[None, None, None, :, None, :]
        return locations, weights
    assert Qn == 1
    Zn = ref.shape[2]
    offsets = off_raw.view(bs, Nq, H, L, P, 2)
    weights = logit_raw.view(bs, Nq, H, L * P).softmax(-1).view(bs, Nq, H, L, P)
    offsets = offsets / normalizer[None, None, None, :, None, :]
    offsets = offsets.view(bs, Nq, H, L, P // Zn, Zn, 2)
    locations = (ref[:, :, None, None, None, :, :] + offsets).reshape(bs, Nq, H, L, P, 2)
    return locations, weights


class FusedDeformAttnFunction(Function):
    """apply(value [bs*Qn,Nv,H,C], shapes, lsi, off_raw, logit_raw, ref, Qn, L, P, mode, merge_queue=False)
    -> [bs*Qn, Nq, H*C], or with merge_queue the mean over the Qn queue entries [bs, Nq, H*C] (TemporalSelfAttention's
    `output.view(bs, Qn, Nq, C).mean(1)` inside the gather)"""

    @staticmethod
    def forward(ctx, value, shapes, lsi, off_raw, logit_raw, ref, Qn, L, P, mode, merge_queue=False):
        ctx.in_dtypes = (value.dtype, off_raw.dtype, logit_raw.dtype)
        f = lambda t: t.float().contiguous()
        i = lambda t: t.to(device=value.device, dtype=torch.int64).contiguous()
        value, off_raw, logit_raw, ref, shapes, lsi = f(value), f(off_raw), f(logit_raw), f(ref), i(shapes), i(lsi)
        Bq, Nv, H, C = value.shape
        bs, Nq = off_raw.shape[:2]
        R = ref.shape[2]
        if Bq != bs * Qn or ref.shape[:2] != (Bq, Nq) or off_raw.numel() != bs * Nq * H * Qn * L * P * 2 \
                or logit_raw.numel() != bs * Nq * H * Qn * L * P:
            raise RuntimeError("inconsistent fused MSDA operand shapes")
        loc = torch.empty((Bq, Nq, H, L, P, 2), dtype=torch.float32, device=value.device)
        w = torch.empty((Bq, Nq, H, L, P), dtype=torch.float32, device=value.device)
        merge = bool(merge_queue) and Qn > 1
        out = torch.empty((bs if merge else Bq, Nq, H * C), dtype=torch.float32, device=value.device)
        with TIMER.span(f"msda_fwd[L={L},P={P}]", msda_fwd_bytes(Bq, Nv, H, C, Nq, L, P)):
            check(lib().vidar_msda_fused_fwd_f32(ptr(value), ptr(shapes), ptr(lsi), ptr(off_raw), ptr(logit_raw),
                                                 ptr(ref), ptr(loc), ptr(w), ptr(out), bs, Qn, Nv, H, C, Nq, L, P,
                                                 R, mode, int(merge), stream_of(value)), "ms_deform_attn_forward (fused)")
        ctx.save_for_backward(value, shapes, lsi, loc, w)
        ctx.cfg = (bs, Qn, L, P, off_raw.shape, logit_raw.shape, merge)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        import ctypes
        value, shapes, lsi, loc, w = ctx.saved_tensors
        bs, Qn, L, P, off_shape, logit_shape, merge = ctx.cfg
        Bq, Nv, H, C = value.shape
        Nq = loc.shape[1]
        go = grad_output.float().contiguous()
        gv = torch.empty_like(value)
        g_off = torch.empty(off_shape, dtype=torch.float32, device=value.device)
        g_logit = torch.empty(logit_shape, dtype=torch.float32, device=value.device)
        ws, nbytes = _bwd_workspace(value, Bq, Nv, H, Nq, L, P, None)
        with TIMER.span(f"msda_bwd[L={L},P={P}]", msda_bwd_bytes(Bq, Nv, H, C, Nq, L, P)):
            check(lib().vidar_msda_fused_bwd_f32(ptr(value), ptr(shapes), ptr(lsi), ptr(loc), ptr(w), ptr(go),
                                                 ptr(gv), ptr(g_off), ptr(g_logit), bs, Qn, Nv, H, C, Nq, L, P,
                                                 int(merge), ptr(ws), ctypes.c_size_t(nbytes), stream_of(value)),
                  "ms_deform_attn_backward (fused)")
        dv, do, dl = ctx.in_dtypes
        return gv.to(dv), None, None, g_off.to(do), g_logit.to(dl), None, None, None, None, None, None


def fused_deform_attn(value, shapes, lsi, off_raw, logit_raw, ref, Qn, L, P, mode, im2col_step=64, merge_queue=False):
    """One entry for the three attention modules.  CUDA tensors: the fused HIP op.  Anything else (the
    oracle-routed CPU tests) or reference points that need a gradient: the reference's tensor program followed
    by MultiScaleDeformableAttnFunction_fp32.apply (which has no CPU implementation of its own).
    merge_queue: also the mean over the Qn queue entries (TemporalSelfAttention) -> [bs, Nq, H*C]."""
    if value.is_cuda and not ref.requires_grad:
        return FusedDeformAttnFunction.apply(value, shapes, lsi, off_raw, logit_raw, ref, Qn, L, P, mode, merge_queue)
    locations, weights = compose_operands(off_raw, logit_raw, ref, shapes, value.shape[2], Qn, L, P, mode)
    out = MultiScaleDeformableAttnFunction_fp32.apply(value, shapes, lsi, locations, weights, im2col_step)
    if merge_queue and Qn > 1:
        out = out.view(out.shape[0] // Qn, Qn, out.shape[1], out.shape[2]).mean(1)
    return out
