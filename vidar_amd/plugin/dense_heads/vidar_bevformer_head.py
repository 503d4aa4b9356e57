"""ViDARBEVFormerHead -- the BEV-encoder front half of BEVFormerHead that ViDAR keeps
(dense_heads/vidar_bevformer_head.py:14-61 on top of bevformer_head.py:40-165): bev_embedding,
positional_encoding and the PerceptionTransformer.  Detection branches / query embedding / bbox
coder / losses of the config are accepted and ignored: ViDAR deletes them at construction
(detectors/vidar.py:103-107)."""
from __future__ import annotations

import torch
import torch.nn as nn

from ..registry import HEADS, build_positional_encoding, build_transformer


@HEADS.register_module()
class ViDARBEVFormerHead(nn.Module):
    def __init__(self, *args, with_box_refine=False, as_two_stage=False, transformer=None,
                 bbox_coder=None, num_cls_fcs=2, code_weights=None, bev_h=30, bev_w=30,
                 positional_encoding=None, num_query=100, num_classes=10, in_channels=256, **kwargs):
        super().__init__()
        self.bev_h, self.bev_w = bev_h, bev_w
        self.fp16_enabled = False
        self.as_two_stage = as_two_stage
        self.pc_range = (bbox_coder or {}).get("pc_range", [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0])
        self.real_w = self.pc_range[3] - self.pc_range[0]
        self.real_h = self.pc_range[4] - self.pc_range[1]
        self.num_query = num_query
        self.positional_encoding = build_positional_encoding(positional_encoding)
        self.transformer = build_transformer(transformer)
        self.embed_dims = self.transformer.embed_dims
        cw = code_weights if code_weights is not None else [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2]
        self.code_weights = nn.Parameter(torch.tensor(cw), requires_grad=False)
        self.bev_embedding = nn.Embedding(bev_h * bev_w, self.embed_dims)
        self._drop_reference_points()

    def _drop_reference_points(self):
        # vidar_bevformer_head.py:20-23 deletes this Linear inside init_weights(); released checkpoints
        # therefore carry no pts_bbox_head.transformer.reference_points.* keys.  Dropped at construction
        # too, so the state_dict layout (strict loading) does not depend on init_weights() being called.
        if hasattr(self.transformer, "reference_points"):
            del self.transformer.reference_points

    def init_weights(self):
        self.transformer.init_weights()
        self._drop_reference_points()

    def forward(self, mlvl_feats, img_metas, prev_bev=None, only_bev=False, return_intermediate=False):
        assert only_bev
        bs = mlvl_feats[0].shape[0]
        dtype = mlvl_feats[0].dtype
        bev_queries = self.bev_embedding.weight.to(dtype)
        bev_mask = torch.zeros((bs, self.bev_h, self.bev_w), device=bev_queries.device).to(dtype)
        bev_pos = self.positional_encoding(bev_mask).to(dtype)
        return self.transformer.get_bev_features(
            mlvl_feats, bev_queries, self.bev_h, self.bev_w,
            grid_length=(self.real_h / self.bev_h, self.real_w / self.bev_w), bev_pos=bev_pos,
            img_metas=img_metas, prev_bev=prev_bev, return_intermediate=return_intermediate)
