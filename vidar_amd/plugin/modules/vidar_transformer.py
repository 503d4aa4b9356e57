"""PredictionTransformer (projects/mmdet3d_plugin/bevformer/modules/vidar_transformer.py:27-113)."""
from __future__ import annotations

import torch.nn as nn

from ..registry import TRANSFORMER, build_transformer_layer_sequence


@TRANSFORMER.register_module()
class PredictionTransformer(nn.Module):
    def __init__(self, decoder=None, embed_dims=256, **kwargs):
        super().__init__()
        self.decoder = build_transformer_layer_sequence(decoder)
        self.embed_dims = embed_dims

    def init_weights(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if m is not self and hasattr(m, "init_weights") and hasattr(m, "sampling_offsets"):
                m.init_weights()

    def get_bev_features(self, prev_feats, bev_queries, tgt_points, ref_points, bev_pos, bev_h, bev_w,
                         **kwargs):
        bev_pos = bev_pos.flatten(2).permute(0, 2, 1).contiguous()
        return self.decoder(bev_queries, prev_feats, tgt_points=tgt_points, ref_points=ref_points,
                            bev_h=bev_h, bev_w=bev_w, bev_pos=bev_pos, **kwargs)

    def forward(self, prev_feats, bev_queries, tgt_points, ref_points, bev_h, bev_w, bev_pos, **kwargs):
        return self.get_bev_features(prev_feats=prev_feats, bev_queries=bev_queries,
                                     tgt_points=tgt_points, ref_points=ref_points, bev_h=bev_h,
                                     bev_w=bev_w, bev_pos=bev_pos, **kwargs)
