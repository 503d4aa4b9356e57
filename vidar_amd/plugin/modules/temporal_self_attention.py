"""TemporalSelfAttention -- same registry name, kwargs, parameter names and forward semantics as
projects/mmdet3d_plugin/bevformer/modules/temporal_self_attention.py:24-271; the bilinear gather
is the gfx950 MSDA kernel, the four Linears are GEMMs."""
from __future__ import annotations

import torch
import torch.nn as nn

from ..bricks import Linear, drop_add_layernorm, constant_init, xavier_init
from ..registry import ATTENTION
from ._attn_common import init_deformable_offsets
from .multi_scale_deformable_attn_function import MultiScaleDeformableAttnFunction_fp32, fused_deform_attn


@ATTENTION.register_module()
class TemporalSelfAttention(nn.Module):
    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, num_bev_queue=2,
                 im2col_step=64, dropout=0.1, batch_first=True, norm_cfg=None, init_cfg=None):
        super().__init__()
        if embed_dims % num_heads != 0:
            raise ValueError(f"embed_dims must be divisible by num_heads, but got {embed_dims} and {num_heads}")
        self.norm_cfg = norm_cfg
        self.dropout = nn.Dropout(dropout)
        self.batch_first = batch_first
        self.fp16_enabled = False
        self.im2col_step = im2col_step
        self.embed_dims = embed_dims
        self.num_levels = num_levels
        self.num_heads = num_heads
        self.num_points = num_points
        self.num_bev_queue = num_bev_queue
        self.sampling_offsets = Linear(embed_dims * num_bev_queue,
                                          num_bev_queue * num_heads * num_levels * num_points * 2)
        self.attention_weights = Linear(embed_dims * num_bev_queue,
                                           num_bev_queue * num_heads * num_levels * num_points)
        self.value_proj = Linear(embed_dims, embed_dims)
        self.output_proj = Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        init_deformable_offsets(self.sampling_offsets, self.num_heads,
                                self.num_levels * self.num_bev_queue, self.num_points)
        constant_init(self.attention_weights, val=0., bias=0.)
        xavier_init(self.value_proj, distribution="uniform", bias=0.)
        xavier_init(self.output_proj, distribution="uniform", bias=0.)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None,
                key_padding_mask=None, reference_points=None, spatial_shapes=None,
                level_start_index=None, flag="decoder", fuse_norm=None, **kwargs):
        if value is None:
            assert self.batch_first
            bs, len_bev, c = query.shape
            value = torch.stack([query, query], 1).reshape(bs * 2, len_bev, c)
        if identity is None:
            identity = query
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query = query.permute(1, 0, 2)
            value = value.permute(1, 0, 2)
        bs, num_query, embed_dims = query.shape
        _, num_value, _ = value.shape
        assert self.num_bev_queue == 2
        H, Qn, L, P = self.num_heads, self.num_bev_queue, self.num_levels, self.num_points

        head = kwargs.get("value_head")                     # the caller's `value[:bs]` of a value shared by several layers
        query = torch.cat([value[:bs] if head is None else head, query], -1)   # (sic) first bs rows, see SURVEY App. D.1
        value = self.value_proj(value)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        value = value.reshape(bs * Qn, num_value, H, -1)

        off_raw = self.sampling_offsets(query)
        logit_raw = self.attention_weights(query)
        if reference_points.shape[-1] == 2:
            # softmax / offset normalisation / reference add / queue permutes (:218-245) happen inside the op
            # ... and so does the mean over the (prev, cur) pair (:256-264): the op returns [bs, Nq, C]
            out = fused_deform_attn(value, spatial_shapes, level_start_index, off_raw, logit_raw,
                                    reference_points, Qn, L, P, 0, self.im2col_step, merge_queue=True)
        elif reference_points.shape[-1] == 4:
            offsets = off_raw.view(bs, num_query, H, Qn, L, P, 2)
            weights = logit_raw.view(bs, num_query, H, Qn, L * P).softmax(-1)
            weights = weights.view(bs, num_query, H, Qn, L, P).permute(0, 3, 1, 2, 4, 5) \
                .reshape(bs * Qn, num_query, H, L, P).contiguous()
            offsets = offsets.permute(0, 3, 1, 2, 4, 5, 6).reshape(bs * Qn, num_query, H, L, P, 2)
            locations = reference_points[:, :, None, :, None, :2] \
                + offsets / P * reference_points[:, :, None, :, None, 2:] * 0.5
            out = MultiScaleDeformableAttnFunction_fp32.apply(value, spatial_shapes, level_start_index,
                                                              locations, weights, self.im2col_step)
            out = out.view(bs, Qn, num_query, embed_dims).mean(1)           # mean over the (prev, cur) pair
        else:
            raise ValueError("Last dim of reference_points must be 2 or 4, but get "
                             f"{reference_points.shape[-1]} instead.")
        out = out.to(identity.dtype)
        out = self.output_proj(out)
        if not self.batch_first:
            out = out.permute(1, 0, 2)
        if fuse_norm is not None:
            return drop_add_layernorm(out, identity, fuse_norm, self.dropout.p, self.training)
        return self.dropout(out) + identity
