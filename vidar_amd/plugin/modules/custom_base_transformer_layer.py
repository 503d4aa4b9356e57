"""MyCustomBaseTransformerLayer -- construction protocol of
projects/mmdet3d_plugin/bevformer/modules/custom_base_transformer_layer.py:71-170 (attn_cfgs /
ffn_cfgs / operation_order / norm_cfg, deprecated feedforward_channels|ffn_dropout|ffn_num_fcs
kwargs folded into ffn_cfgs), module attribute names `attentions`, `ffns`, `norms`."""
from __future__ import annotations

import copy

import torch.nn as nn

from ..bricks import build_norm_layer
from ..registry import build_attention, build_feedforward_network


class MyCustomBaseTransformerLayer(nn.Module):
    def __init__(self, attn_cfgs=None,
                 ffn_cfgs=dict(type="FFN", embed_dims=256, feedforward_channels=1024, num_fcs=2,
                               ffn_drop=0., act_cfg=dict(type="ReLU", inplace=True)),
                 operation_order=None, norm_cfg=dict(type="LN"), init_cfg=None, batch_first=True,
                 **kwargs):
        super().__init__()
        ffn_cfgs = copy.deepcopy(dict(ffn_cfgs))
        for old, new in dict(feedforward_channels="feedforward_channels", ffn_dropout="ffn_drop",
                             ffn_num_fcs="num_fcs").items():
            if old in kwargs:
                ffn_cfgs[new] = kwargs[old]
        if "act_cfg" in kwargs:
            ffn_cfgs["act_cfg"] = kwargs["act_cfg"]
        self.batch_first = batch_first
        need = {"self_attn", "norm", "ffn", "cross_attn"}
        assert set(operation_order) & need == need, \
            f"The operation_order of {self.__class__.__name__} should contain all four operation types"
        num_attn = operation_order.count("self_attn") + operation_order.count("cross_attn")
        if isinstance(attn_cfgs, dict):
            attn_cfgs = [copy.deepcopy(attn_cfgs) for _ in range(num_attn)]
        else:
            attn_cfgs = [copy.deepcopy(dict(c)) for c in attn_cfgs]
            assert num_attn == len(attn_cfgs)
        self.num_attn = num_attn
        self.operation_order = tuple(operation_order)
        self.norm_cfg = norm_cfg
        self.pre_norm = operation_order[0] == "norm"
        self.attentions = nn.ModuleList()
        index = 0
        for name in operation_order:
            if name in ("self_attn", "cross_attn"):
                if "batch_first" in attn_cfgs[index]:
                    assert self.batch_first == attn_cfgs[index]["batch_first"]
                else:
                    attn_cfgs[index]["batch_first"] = self.batch_first
                attn = build_attention(attn_cfgs[index])
                attn.operation_name = name
                self.attentions.append(attn)
                index += 1
        self.embed_dims = self.attentions[0].embed_dims
        self.ffns = nn.ModuleList()
        for _ in range(operation_order.count("ffn")):
            cfg = copy.deepcopy(ffn_cfgs)
            cfg.setdefault("type", "FFN")
            cfg["embed_dims"] = self.embed_dims
            self.ffns.append(build_feedforward_network(cfg))
        self.norms = nn.ModuleList()
        for _ in range(operation_order.count("norm")):
            self.norms.append(build_norm_layer(norm_cfg, self.embed_dims)[1])
