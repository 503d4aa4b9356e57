"""Future-BEV decoder: PredictionDecoder / PredictionTransformerLayer /
PredictionMSDeformableAttention -- names, kwargs, parameter names and semantics of
projects/mmdet3d_plugin/bevformer/modules/vidar_decoder.py:25-516."""
from __future__ import annotations

import torch

from ..utils.host import const_tensor
import torch.nn as nn

from ..bricks import Linear, can_fuse_norm, drop_add_layernorm, constant_init, xavier_init
from ..registry import ATTENTION, TRANSFORMER_LAYER, TRANSFORMER_LAYER_SEQUENCE
from ._attn_common import init_deformable_offsets
from .custom_base_transformer_layer import MyCustomBaseTransformerLayer
from .encoder import TransformerLayerSequence
from .multi_scale_deformable_attn_function import MultiScaleDeformableAttnFunction_fp32, fused_deform_attn
from .ray_operations.latent_rendering import LatentRendering


@TRANSFORMER_LAYER_SEQUENCE.register_module()
class PredictionDecoder(TransformerLayerSequence):
    def __init__(self, *args, return_intermediate=False, keep_idx=(2,), **kwargs):
        super().__init__(*args, **kwargs)
        self.return_intermediate = return_intermediate
        self.keep_idx = keep_idx
        for lid, layer in enumerate(self.layers):
            if lid not in self.keep_idx and getattr(layer, "latent_render", None) is not None:
                del layer.latent_render
                layer.operation_order = ("self_attn", "norm", "cross_attn", "norm", "ffn", "norm")

    def forward(self, bev_query, prev_feats, *args, tgt_points=None, ref_points=None, bev_h=None,
                bev_w=None, bev_pos=None, **kwargs):
        output = bev_query
        intermediate = []
        for layer in self.layers:
            output = layer(bev_query, prev_feats, *args, bev_pos=bev_pos, tgt_points=tgt_points,
                           ref_points=ref_points, bev_h=bev_h, bev_w=bev_w, **kwargs)
            bev_query = output
            if self.return_intermediate:
                intermediate.append(output)
        return torch.stack(intermediate) if self.return_intermediate else output


@TRANSFORMER_LAYER.register_module()
class PredictionTransformerLayer(MyCustomBaseTransformerLayer):
    def __init__(self, attn_cfgs, feedforward_channels, ffn_dropout=0.0, operation_order=None,
                 act_cfg=dict(type="ReLU", inplace=True), norm_cfg=dict(type="LN"), ffn_num_fcs=2,
                 latent_render=None, **kwargs):
        super().__init__(attn_cfgs=attn_cfgs, feedforward_channels=feedforward_channels,
                         ffn_dropout=ffn_dropout, operation_order=operation_order, act_cfg=act_cfg,
                         norm_cfg=norm_cfg, ffn_num_fcs=ffn_num_fcs, **kwargs)
        self.fp16_enabled = False
        if latent_render is not None:
            self.latent_render = LatentRendering(**latent_render)

    def forward(self, query, prev_feats=None, bev_pos=None, query_pos=None, key_pos=None,
                attn_masks=None, query_key_padding_mask=None, key_padding_mask=None, tgt_points=None,
                ref_points=None, bev_h=None, bev_w=None, **kwargs):
        norm_index = attn_index = ffn_index = 0
        identity = query
        tgt_points = tgt_points.unsqueeze(2)
        if ref_points.dim() != 4:
            ref_points = ref_points.unsqueeze(2)
        bs, num_frames, prev_tokens, prev_dims = prev_feats.shape
        assert prev_tokens == bev_h * bev_w
        dev = query.device
        self_shapes = const_tensor([[bev_h, bev_w]], dev, torch.int64)
        self_lsi = const_tensor([0], dev, torch.int64)
        cross_shapes = const_tensor([[bev_h, bev_w]] * num_frames, dev, torch.int64)
        cross_lsi = torch.cat((cross_shapes.new_zeros((1,)), cross_shapes.prod(1).cumsum(0)[:-1]))
        prev_feats = prev_feats.reshape(bs, num_frames * prev_tokens, prev_dims)
        ops = self.operation_order
        skip = False
        for k, layer in enumerate(ops):
            if skip:                      # this norm was fused into the block in front of it
                skip = False
                continue
            fuse = None
            if not self.pre_norm and layer in ("self_attn", "cross_attn", "ffn") and k + 1 < len(ops) \
                    and ops[k + 1] == "norm" and can_fuse_norm(self.norms[norm_index], query):
                fuse = self.norms[norm_index]
                norm_index += 1
                skip = True
            if layer == "self_attn":
                query = self.attentions[attn_index](
                    query, None, None, identity if self.pre_norm else None, query_pos=bev_pos,
                    key_pos=bev_pos, key_padding_mask=query_key_padding_mask,
                    reference_points=tgt_points, spatial_shapes=self_shapes,
                    level_start_index=self_lsi, fuse_norm=fuse, **kwargs)
                attn_index += 1
                identity = query
            elif layer == "norm":
                query = self.norms[norm_index](query)
                norm_index += 1
            elif layer == "cross_attn":
                query = self.attentions[attn_index](
                    query, prev_feats, prev_feats, identity if self.pre_norm else None,
                    query_pos=bev_pos, reference_points=ref_points, key_padding_mask=key_padding_mask,
                    spatial_shapes=cross_shapes, level_start_index=cross_lsi, fuse_norm=fuse, **kwargs)
                attn_index += 1
                identity = query
            elif layer == "latent_render":
                b, n, c = query.shape
                query = self.latent_render(query.view(b, bev_h, bev_w, c)).view(b, n, c)
            elif layer == "ffn":
                query = self.ffns[ffn_index](query, identity if self.pre_norm else None, fuse_norm=fuse)
                ffn_index += 1
        return query


@ATTENTION.register_module()
class PredictionMSDeformableAttention(nn.Module):
    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64,
                 dropout=0.1, batch_first=True, norm_cfg=None, init_cfg=None):
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
        self.sampling_offsets = Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = Linear(embed_dims, embed_dims)
        self.output_proj = Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        init_deformable_offsets(self.sampling_offsets, self.num_heads, self.num_levels, self.num_points)
        constant_init(self.attention_weights, val=0., bias=0.)
        xavier_init(self.value_proj, distribution="uniform", bias=0.)
        xavier_init(self.output_proj, distribution="uniform", bias=0.)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None,
                key_padding_mask=None, reference_points=None, spatial_shapes=None,
                level_start_index=None, flag="decoder", fuse_norm=None, **kwargs):
        if value is None:
            value = query
        if identity is None:
            identity = query
        if query_pos is not None:
            query = query + query_pos
        bs, num_query, _ = query.shape
        _, num_value, _ = value.shape
        H, L, P = self.num_heads, self.num_levels, self.num_points
        value = self.value_proj(value)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        value = value.view(bs, num_value, H, -1)
        if reference_points.shape[-1] == 2:
            out = fused_deform_attn(value, spatial_shapes, level_start_index, self.sampling_offsets(query),
                                    self.attention_weights(query), reference_points, 1, L, P, 0, self.im2col_step)
        elif reference_points.shape[-1] == 4:
            offsets = self.sampling_offsets(query).view(bs, num_query, H, L, P, 2)
            weights = self.attention_weights(query).view(bs, num_query, H, L * P).softmax(-1) \
                .view(bs, num_query, H, L, P)
            locations = reference_points[:, :, None, :, None, :2] \
                + offsets / P * reference_points[:, :, None, :, None, 2:] * 0.5
            out = MultiScaleDeformableAttnFunction_fp32.apply(value, spatial_shapes, level_start_index,
                                                              locations, weights, self.im2col_step)
        else:
            raise ValueError("Last dim of reference_points must be 2 or 4, but get "
                             f"{reference_points.shape[-1]} instead.")
        out = self.output_proj(out.to(identity.dtype))
        if not self.batch_first:
            out = out.permute(1, 0, 2)
        if fuse_norm is not None:
            return drop_add_layernorm(out, identity, fuse_norm, self.dropout.p, self.training)
        return self.dropout(out) + identity


@ATTENTION.register_module()
class CustomMSDeformableAttention(PredictionMSDeformableAttention):
    """Deformable-DETR attention of the detection decoder (bevformer/modules/decoder.py:132-345): the same op with
    sequence-first tensors by default (`batch_first=False`) and the assertion that the key length matches the
    level shapes.  Registered for config compatibility (fine-tuning configs); ViDAR pre-training deletes the
    detection decoder (detectors/vidar.py:105-107)."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64, dropout=0.1,
                 batch_first=False, norm_cfg=None, init_cfg=None):
        super().__init__(embed_dims=embed_dims, num_heads=num_heads, num_levels=num_levels, num_points=num_points,
                         im2col_step=im2col_step, dropout=dropout, batch_first=batch_first, norm_cfg=norm_cfg,
                         init_cfg=init_cfg)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, level_start_index=None, flag="decoder", fuse_norm=None,
                **kwargs):
        # `fuse_norm` (the layer's offer to fuse the following LayerNorm) must not reach the parent here: it would
        # normalise BEFORE the identity is added below.  The norm is applied after the add instead.
        if value is None:
            value = query
        if identity is None:
            identity = query
        if query_pos is not None:
            query = query + query_pos
            query_pos = None
        if not self.batch_first:                     # (num_query, bs, C) -> (bs, num_query, C)
            query = query.permute(1, 0, 2)
            value = value.permute(1, 0, 2)
        assert int((spatial_shapes[:, 0] * spatial_shapes[:, 1]).sum()) == value.shape[1]
        batch_first, self.batch_first = self.batch_first, True
        try:
            out = super().forward(query, key, value, identity=torch.zeros_like(query), query_pos=None,
                                  key_padding_mask=key_padding_mask, reference_points=reference_points,
                                  spatial_shapes=spatial_shapes, level_start_index=level_start_index, **kwargs)
        finally:
            self.batch_first = batch_first
        # the parent returned dropout(output_proj(attn)) + 0: add the identity in the caller's layout
        if not self.batch_first:
            out = out.permute(1, 0, 2)
        out = out + identity
        return fuse_norm(out) if fuse_norm is not None else out
