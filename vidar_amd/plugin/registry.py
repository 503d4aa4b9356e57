"""Minimal stand-in for the mmcv/mmdet registries the reference populates at plugin import
(projects/mmdet3d_plugin/__init__.py:1-12).  Same registered type strings, same
`dict(type=..., **kwargs)` build protocol, so the released config dictionaries build unchanged."""
from __future__ import annotations

import copy


class Registry:
    def __init__(self, name):
        self.name = name
        self._mods = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            key = name or cls.__name__
            if key in self._mods and not force and self._mods[key] is not cls:
                raise KeyError(f"{key} is already registered in {self.name}")
            self._mods[key] = cls
            return cls
        if module is not None:
            return deco(module)
        return deco

    def get(self, key):
        return self._mods.get(key)

    def __contains__(self, key):
        return key in self._mods

    def build(self, cfg, **default_args):
        if cfg is None:
            return None
        if not isinstance(cfg, dict) or "type" not in cfg:
            raise TypeError(f"{self.name}: cfg must be a dict with a 'type' key, got {cfg!r}")
        args = copy.deepcopy(dict(cfg))
        for k, v in default_args.items():
            args.setdefault(k, v)
        typ = args.pop("type")
        cls = typ if isinstance(typ, type) else self._mods.get(typ)
        if cls is None:
            raise KeyError(f"{typ} is not in the {self.name} registry")
        return cls(**args)


ATTENTION = Registry("attention")
FEEDFORWARD_NETWORK = Registry("feed-forward network")
POSITIONAL_ENCODING = Registry("position encoding")
TRANSFORMER_LAYER = Registry("transformerLayer")
TRANSFORMER_LAYER_SEQUENCE = Registry("transformer-layers sequence")
TRANSFORMER = Registry("Transformer")
HEADS = Registry("head")
DETECTORS = Registry("detector")
BACKBONES = Registry("backbone")
NECKS = Registry("neck")
LOSSES = Registry("loss")


def build_attention(cfg, **kw): return ATTENTION.build(cfg, **kw)
def build_feedforward_network(cfg, **kw): return FEEDFORWARD_NETWORK.build(cfg, **kw)
def build_positional_encoding(cfg, **kw): return POSITIONAL_ENCODING.build(cfg, **kw)
def build_transformer_layer(cfg, **kw): return TRANSFORMER_LAYER.build(cfg, **kw)
def build_transformer_layer_sequence(cfg, **kw): return TRANSFORMER_LAYER_SEQUENCE.build(cfg, **kw)
def build_transformer(cfg, **kw): return TRANSFORMER.build(cfg, **kw)
def build_head(cfg, **kw): return HEADS.build(cfg, **kw)
def build_detector(cfg, **kw): return DETECTORS.build(cfg, **kw)
def build_backbone(cfg, **kw): return BACKBONES.build(cfg, **kw)
def build_neck(cfg, **kw): return NECKS.build(cfg, **kw)
