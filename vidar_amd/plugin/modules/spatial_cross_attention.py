"""SpatialCrossAttention + MSDeformableAttention3D -- registry names, kwargs, parameter names and
semantics of projects/mmdet3d_plugin/bevformer/modules/spatial_cross_attention.py:30-398.

Difference in *how*, not *what*: the reference compacts the visible queries of every camera with
python loops and one `nonzero()` (= host sync) per camera per layer (:136-152, :164-166).  Here the
compaction is a stable argsort on device + one gather / one index_add, and the per-camera index
table can be computed once per encoder pass and handed in through `sca_index` (encoder.py)."""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from ..bricks import Linear, drop_add_layernorm, constant_init, xavier_init
from ..registry import ATTENTION, build_attention
from ._attn_common import init_deformable_offsets
from .multi_scale_deformable_attn_function import MultiScaleDeformableAttnFunction_fp32, fused_deform_attn


def visible_query_index(bev_mask):
    """bev_mask [cams, bs, Q, D] bool -> (idx [cams, max_len] long, valid [cams, max_len] bool,
    count [bs, Q] float).  Like the reference the visible set is taken from batch item 0 (:137-139).
    One device->host read (max_len)."""
    vis = bev_mask[:, 0].sum(-1) > 0                       # [cams, Q]
    lens = vis.sum(1)
    max_len = int(lens.max())                              # the only sync
    order = torch.argsort((~vis).to(torch.int8), dim=1, stable=True)[:, :max_len]
    valid = torch.arange(max_len, device=vis.device)[None] < lens[:, None]
    count = (bev_mask.sum(-1) > 0).permute(1, 2, 0).sum(-1).clamp(min=1.0)
    return order, valid, count


class _ScaRows(torch.autograd.Function):
    """rebatch: [bs, Q, C] -> [bs, cams, max_len, C] (visible queries of every camera, zero padded);
    backward = sum over the cameras that see a query (vidar_sca_rows_f32 / vidar_sca_combine_f32)."""

    @staticmethod
    def forward(ctx, src, plan, with_count):
        from ..._lib import lib, check, ptr, stream_of
        idx, valid, count = plan.index
        bs, Q, C = src.shape
        N, S = idx.shape
        src = src.float().contiguous()
        dst = torch.empty((bs, N, S, C), device=src.device)
        check(lib().vidar_sca_rows_f32(ptr(src), ptr(idx), ptr(plan.valid_u8), ptr(count if with_count else None),
                                       ptr(dst), bs, N, S, plan.stride, Q, C, stream_of(src)), "sca_rows")
        ctx.plan, ctx.with_count = plan, with_count
        return dst

    @staticmethod
    def backward(ctx, g):
        return _ScaCombine.apply(g, ctx.plan, ctx.with_count), None, None


class _ScaCombine(torch.autograd.Function):
    """scatter-back: [bs, cams, max_len, C] -> [bs, Q, C] = sum over the cameras that see a query (/ count);
    backward = the rebatch gather (/ count)."""

    @staticmethod
    def forward(ctx, src, plan, with_count):
        from ..._lib import lib, check, ptr, stream_of
        idx, valid, count = plan.index
        bs, N, S, C = src.shape
        Q = plan.stride
        src = src.float().contiguous()
        dst = torch.empty((bs, Q, C), device=src.device)
        check(lib().vidar_sca_combine_f32(ptr(src), ptr(plan.slot_of), ptr(count if with_count else None), ptr(dst),
                                          bs, N, S, Q, C, stream_of(src)), "sca_combine")
        ctx.plan, ctx.with_count = plan, with_count
        return dst

    @staticmethod
    def backward(ctx, g):
        return _ScaRows.apply(g, ctx.plan, ctx.with_count), None, None


@ATTENTION.register_module()
class SpatialCrossAttention(nn.Module):
    def __init__(self, embed_dims=256, num_cams=6, pc_range=None, dropout=0.1, init_cfg=None,
                 batch_first=False,
                 deformable_attention=dict(type="MSDeformableAttention3D", embed_dims=256, num_levels=4),
                 **kwargs):
        super().__init__()
        self.init_cfg = init_cfg
        self.dropout = nn.Dropout(dropout)
        self.pc_range = pc_range
        self.fp16_enabled = False
        self.deformable_attention = build_attention(deformable_attention)
        self.embed_dims = embed_dims
        self.num_cams = num_cams
        self.output_proj = Linear(embed_dims, embed_dims)
        self.batch_first = batch_first
        self.init_weight()

    def init_weight(self):
        xavier_init(self.output_proj, distribution="uniform", bias=0.)

    def forward(self, query, key, value, residual=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, reference_points_cam=None,
                bev_mask=None, level_start_index=None, flag="encoder", sca_index=None, sca_plan=None,
                fuse_norm=None, **kwargs):
        if key is None:
            key = query
        if value is None:
            value = key
        inp_residual = query if residual is None else residual
        if query_pos is not None:
            query = query + query_pos
        bs, num_query, _ = query.size()
        D = reference_points_cam.size(3)
        num_cams, l, bs_, embed_dims = key.shape
        if sca_plan is not None and sca_plan.slot_of is not None and query.is_cuda and self.embed_dims % 4 == 0:
            # HIP rebatch: both directions of the gather / scatter-back are row gathers through the inverse index
            plan = sca_plan
            idx, valid, count = plan.index
            max_len = idx.shape[1]
            if plan.ref_re is None:                       # once per frame, shared by the six layers
                cams = torch.arange(self.num_cams, device=idx.device)
                plan.ref_re = reference_points_cam.permute(1, 0, 2, 3, 4)[
                    torch.arange(bs, device=idx.device)[:, None, None], cams[None, :, None], idx[None]] \
                    * valid[None, :, :, None, None].to(query.dtype)
            q_re = _ScaRows.apply(query, plan, False)
            key = key.permute(2, 0, 1, 3).reshape(bs * self.num_cams, l, self.embed_dims)
            value = value.permute(2, 0, 1, 3).reshape(bs * self.num_cams, l, self.embed_dims)
            out = self.deformable_attention(
                query=q_re.view(bs * self.num_cams, max_len, self.embed_dims), key=key, value=value,
                reference_points=plan.ref_re.view(bs * self.num_cams, max_len, D, 2),
                spatial_shapes=spatial_shapes, level_start_index=level_start_index,
            ).view(bs, self.num_cams, max_len, self.embed_dims)
            slots = _ScaCombine.apply(out, plan, True).to(query.dtype)      # / (#cameras seeing the query)
            slots = self.output_proj(slots)
            if fuse_norm is not None:
                return drop_add_layernorm(slots, inp_residual, fuse_norm, self.dropout.p, self.training)
            return self.dropout(slots) + inp_residual
        idx, valid, count = sca_index if sca_index is not None else visible_query_index(bev_mask)
        max_len = idx.shape[1]
        vmask = valid[None, :, :, None].to(query.dtype)
        # [bs, cams, max_len, C] / [bs, cams, max_len, D, 2]; padded slots are zero like the reference
        q_re = query[:, idx] * vmask
        ref_re = reference_points_cam.permute(1, 0, 2, 3, 4)[torch.arange(bs, device=idx.device)[:, None, None],
                                                                torch.arange(self.num_cams, device=idx.device)[None, :, None],
                                                                idx[None]] * vmask[..., None]
        key = key.permute(2, 0, 1, 3).reshape(bs * self.num_cams, l, self.embed_dims)
        value = value.permute(2, 0, 1, 3).reshape(bs * self.num_cams, l, self.embed_dims)
        out = self.deformable_attention(
            query=q_re.view(bs * self.num_cams, max_len, self.embed_dims), key=key, value=value,
            reference_points=ref_re.view(bs * self.num_cams, max_len, D, 2),
            spatial_shapes=spatial_shapes, level_start_index=level_start_index,
        ).view(bs, self.num_cams, max_len, self.embed_dims)
        out = out * vmask
        slots = torch.zeros_like(query)
        flat_idx = idx.reshape(-1)
        for j in range(bs):
            slots[j].index_add_(0, flat_idx, out[j].reshape(-1, self.embed_dims).to(slots.dtype))
        slots = slots / count[..., None]
        slots = self.output_proj(slots)
        if fuse_norm is not None:
            return drop_add_layernorm(slots, inp_residual, fuse_norm, self.dropout.p, self.training)
        return self.dropout(slots) + inp_residual


@ATTENTION.register_module()
class MSDeformableAttention3D(nn.Module):
    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=8, im2col_step=64,
                 dropout=0.1, batch_first=True, norm_cfg=None, init_cfg=None):
        super().__init__()
        if embed_dims % num_heads != 0:
            raise ValueError(f"embed_dims must be divisible by num_heads, but got {embed_dims} and {num_heads}")
        self.norm_cfg = norm_cfg
        self.batch_first = batch_first
        self.output_proj = None
        self.fp16_enabled = False
        self.im2col_step = im2col_step
        self.embed_dims = embed_dims
        self.num_levels = num_levels
        self.num_heads = num_heads
        self.num_points = num_points
        self.sampling_offsets = Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        init_deformable_offsets(self.sampling_offsets, self.num_heads, self.num_levels, self.num_points)
        constant_init(self.attention_weights, val=0., bias=0.)
        xavier_init(self.value_proj, distribution="uniform", bias=0.)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None,
                key_padding_mask=None, reference_points=None, spatial_shapes=None,
                level_start_index=None, **kwargs):
        if value is None:
            value = query
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query = query.permute(1, 0, 2)
            value = value.permute(1, 0, 2)
        bs, num_query, _ = query.shape
        bs, num_value, _ = value.shape
        H, L, P = self.num_heads, self.num_levels, self.num_points
        value = self.value_proj(value)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        value = value.view(bs, num_value, H, -1)
        if reference_points.shape[-1] != 2:
            raise ValueError("Last dim of reference_points must be 2, but get "
                             f"{reference_points.shape[-1]} instead.")
        # each pillar anchor (num_Z_anchors of them) owns P / num_Z_anchors sampling points; the softmax, the
        # offset normalisation and the anchor add (:359-383) happen inside the op
        assert P % reference_points.shape[2] == 0
        out = fused_deform_attn(value, spatial_shapes, level_start_index, self.sampling_offsets(query),
                                self.attention_weights(query), reference_points, 1, L, P, 1, self.im2col_step)
        out = out.to(query.dtype)
        if not self.batch_first:
            out = out.permute(1, 0, 2)
        return out
