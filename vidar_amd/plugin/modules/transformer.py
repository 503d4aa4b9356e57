"""PerceptionTransformer -- names / kwargs / parameter names / semantics of
projects/mmdet3d_plugin/bevformer/modules/transformer.py:24-195 (get_bev_features path).  The
detection decoder of the config is never built: ViDAR deletes it right after construction
(detectors/vidar.py:105-107)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from ..utils.host import const_tensor, to_device_async

from ..bricks import rotate_nearest, xavier_init
from ..registry import TRANSFORMER, build_transformer_layer_sequence
from .spatial_cross_attention import MSDeformableAttention3D
from .temporal_self_attention import TemporalSelfAttention


@TRANSFORMER.register_module()
class PerceptionTransformer(nn.Module):
    def __init__(self, num_feature_levels=4, num_cams=6, two_stage_num_proposals=300, encoder=None,
                 decoder=None, embed_dims=256, rotate_prev_bev=True, use_shift=True, use_can_bus=True,
                 can_bus_norm=True, use_cams_embeds=True, rotate_center=[100, 100], **kwargs):
        super().__init__()
        self.encoder = build_transformer_layer_sequence(encoder)
        self.decoder = None                 # config keeps a detection decoder ViDAR never uses
        self.embed_dims = embed_dims
        self.num_feature_levels = num_feature_levels
        self.num_cams = num_cams
        self.fp16_enabled = False
        self.rotate_prev_bev = rotate_prev_bev
        self.use_shift = use_shift
        self.use_can_bus = use_can_bus
        self.can_bus_norm = can_bus_norm
        self.use_cams_embeds = use_cams_embeds
        self.two_stage_num_proposals = two_stage_num_proposals
        self.level_embeds = nn.Parameter(torch.Tensor(num_feature_levels, embed_dims))
        self.cams_embeds = nn.Parameter(torch.Tensor(num_cams, embed_dims))
        # detection-branch leftover (transformer.py:73): unused on this path, kept so that released
        # checkpoints load with strict=True; frozen so that DDP never waits for its gradient
        self.reference_points = nn.Linear(embed_dims, 3).requires_grad_(False)
        self.can_bus_mlp = nn.Sequential(nn.Linear(18, embed_dims // 2), nn.ReLU(inplace=True),
                                         nn.Linear(embed_dims // 2, embed_dims), nn.ReLU(inplace=True))
        if can_bus_norm:
            self.can_bus_mlp.add_module("norm", nn.LayerNorm(embed_dims))
        self.rotate_center = rotate_center
        self.init_weights()

    def init_weights(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, (MSDeformableAttention3D, TemporalSelfAttention)):
                m.init_weights()
        nn.init.normal_(self.level_embeds)
        nn.init.normal_(self.cams_embeds)
        if hasattr(self, "reference_points"):         # ViDARBEVFormerHead drops it (vidar_bevformer_head.py:20-23)
            xavier_init(self.reference_points, distribution="uniform", bias=0.)
        xavier_init(self.can_bus_mlp, distribution="uniform", bias=0.)

    def get_bev_features(self, mlvl_feats, bev_queries, bev_h, bev_w, grid_length=[0.512, 0.512],
                         bev_pos=None, prev_bev=None, **kwargs):
        img_metas = kwargs["img_metas"]
        bs = mlvl_feats[0].size(0)
        bev_queries = bev_queries.unsqueeze(1).repeat(1, bs, 1)
        bev_pos = bev_pos.flatten(2).permute(2, 0, 1)

        # ego-motion since the previous frame, in BEV-grid units (host math, as the reference)
        delta_global = np.array([m["can_bus"][:3] for m in img_metas])
        rot = np.array([m["lidar2global_rotation"] for m in img_metas])
        delta_lidar = np.array([np.linalg.inv(rot[i]) @ delta_global[i] for i in range(bs)])
        shift_y = delta_lidar[:, 1] / grid_length[0] / bev_h * self.use_shift
        shift_x = delta_lidar[:, 0] / grid_length[1] / bev_w * self.use_shift
        shift = to_device_async(np.array([shift_x, shift_y]).T, bev_queries.device, bev_queries.dtype)

        if prev_bev is not None:
            if prev_bev.shape[1] == bev_h * bev_w:
                prev_bev = prev_bev.permute(1, 0, 2)
            if self.rotate_prev_bev:
                rotated = []
                for i in range(bs):
                    angle = img_metas[i]["can_bus"][-1]
                    t = prev_bev[:, i].reshape(bev_h, bev_w, -1).permute(2, 0, 1)
                    t = rotate_nearest(t, angle, center=self.rotate_center)
                    rotated.append(t.permute(1, 2, 0).reshape(bev_h * bev_w, -1))
                prev_bev = torch.stack(rotated, 1)

        can_bus = to_device_async(np.array([m["can_bus"] for m in img_metas]), bev_queries.device, bev_queries.dtype)
        can_bus = self.can_bus_mlp(can_bus)[None, :, :]
        bev_queries = bev_queries + can_bus * self.use_can_bus

        feat_flatten, spatial_shapes = [], []
        for lvl, feat in enumerate(mlvl_feats):
            _, num_cam, c, h, w = feat.shape
            spatial_shapes.append((h, w))
            feat = feat.flatten(3).permute(1, 0, 3, 2)          # [cam, bs, hw, c]
            if self.use_cams_embeds:
                feat = feat + self.cams_embeds[:, None, None, :].to(feat.dtype)
            feat = feat + self.level_embeds[None, None, lvl:lvl + 1, :].to(feat.dtype)
            feat_flatten.append(feat)
        feat_flatten = torch.cat(feat_flatten, 2)
        sizes = [h * w for h, w in spatial_shapes]
        level_start_index = const_tensor([sum(sizes[:i]) for i in range(len(sizes))], bev_pos.device, torch.long)
        spatial_shapes = const_tensor(spatial_shapes, bev_pos.device, torch.long)
        feat_flatten = feat_flatten.permute(0, 2, 1, 3)        # [cam, sum(hw), bs, c]
        return self.encoder(bev_queries, feat_flatten, feat_flatten, bev_h=bev_h, bev_w=bev_w,
                            bev_pos=bev_pos, spatial_shapes=spatial_shapes,
                            level_start_index=level_start_index, prev_bev=prev_bev, shift=shift,
                            **kwargs)
