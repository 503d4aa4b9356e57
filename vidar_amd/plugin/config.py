"""A ~100-line stand-in for mmcv.Config: plain-Python config files with `_base_` inheritance,
attribute access and dotted `--cfg-options` overrides (tools/train.py:67-76, :105-107)."""
from __future__ import annotations

import copy
import os
import runpy
from pathlib import Path


class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return ConfigDict({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    if isinstance(x, tuple):
        return tuple(_wrap(v) for v in x)
    return x


def _merge(base, new):
    out = copy.deepcopy(base)
    for k, v in new.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get("_delete_", False):
            out[k] = _merge(out[k], v)
        else:
            v = copy.deepcopy(v)
            if isinstance(v, dict):
                v.pop("_delete_", None)
            out[k] = v
    return out


def _load(path: Path):
    ns = runpy.run_path(str(path))
    cfg = {k: v for k, v in ns.items() if not k.startswith("__") and not callable(v)
           and type(v).__name__ != "module"}
    bases = cfg.pop("_base_", [])
    if isinstance(bases, str):
        bases = [bases]
    merged = {}
    for b in bases:
        bp = (path.parent / b).resolve()
        if not bp.exists():           # released configs reference _base_ files that may be absent
            continue
        merged = _merge(merged, _load(bp))
    return _merge(merged, cfg)


class Config:
    def __init__(self, d, filename=None):
        object.__setattr__(self, "_cfg", _wrap(d))
        object.__setattr__(self, "filename", filename)

    @staticmethod
    def fromfile(path):
        return Config(_load(Path(os.fspath(path)).resolve()), filename=str(path))

    def __getattr__(self, k):
        return getattr(self._cfg, k)

    def __getitem__(self, k):
        return self._cfg[k]

    def get(self, k, default=None):
        return self._cfg.get(k, default)

    def merge_from_dict(self, options):
        for key, v in options.items():
            d = self._cfg
            parts = key.split(".")
            for p in parts[:-1]:
                d = d[int(p)] if isinstance(d, (list, tuple)) else d.setdefault(p, ConfigDict())
            d[parts[-1]] = _wrap(v)
