"""Checkpoint save / resume in the dictionary layout mmcv's CheckpointHook writes and
`runner.resume` / `load_checkpoint` read (apis/mmdet_train.py:195-198; tools/train.py:239-249):
`{'meta': {...}, 'state_dict': OrderedDict, 'optimizer': {...}}`, DDP 'module.' prefixes stripped.
Module attribute names equal the reference's, so released `.pth` files map
key for key; `load_checkpoint` reports what did not match instead of failing silently."""
from __future__ import annotations

import time
from collections import OrderedDict

import torch


def _unwrap(model):
    return model.module if hasattr(model, "module") and isinstance(model.module, torch.nn.Module) else model


def save_checkpoint(model, path, optimizer=None, meta=None):
    sd = OrderedDict((k, v.detach().cpu()) for k, v in _unwrap(model).state_dict().items())
    ckpt = dict(meta=dict(time=time.asctime(), **(meta or {})), state_dict=sd)
    if optimizer is not None:
        ckpt["optimizer"] = optimizer.state_dict()
    torch.save(ckpt, path)
    return path


def load_checkpoint(model, path, map_location="cpu", strict=False, revise_keys=(("module.", ""),)):
    """-> (checkpoint dict, missing_keys, unexpected_keys)"""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    sd = ckpt.get("state_dict", ckpt)
    out = OrderedDict()
    for k, v in sd.items():
        for old, new in revise_keys:
            if k.startswith(old):
                k = new + k[len(old):]
        out[k] = v
    res = _unwrap(model).load_state_dict(out, strict=strict)
    return ckpt, list(res.missing_keys), list(res.unexpected_keys)


def resume(model, optimizer, path, map_location="cpu"):
    """restore weights + optimizer + (epoch, iter) like `runner.resume`."""
    ckpt, missing, unexpected = load_checkpoint(model, path, map_location, strict=True)
    if optimizer is not None and "optimizer" in ckpt:
        optimizer.load_state_dict(ckpt["optimizer"])
    meta = ckpt.get("meta", {})
    return int(meta.get("epoch", 0)), int(meta.get("iter", 0))
