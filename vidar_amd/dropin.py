"""Drop-in installation: make the HIP shims answer to the names the reference binds.

The reference reaches its native code in three ways:
  * `torch.utils.cpp_extension.load("dvxlr" | "dvxlr_v2" | "dvr", sources=[...])` at import time
    (bevformer/utils/e2e_predictor_utils.py:86-90, :118-121; third_lib/dvr is built the same way),
  * `import chamferdist` (utils/e2e_predictor_utils.py:163-183, detectors/vidar.py eval path),
  * `mmcv.utils.ext_loader.load_ext('_ext', ['ms_deform_attn_backward', 'ms_deform_attn_forward'])`.
`install()` registers `dvr`, `dvxlr`, `dvxlr_v2`, `chamferdist` (and `chamferdist._C`) as top-level modules
and, with `patch_loaders=True`, answers the two loader calls with the shims, so the reference's wrapper
files run unchanged on top of libvidar_hip.so.  `uninstall()` undoes it."""
from __future__ import annotations

import importlib
import sys

NAMES = ("dvr", "dvxlr", "dvxlr_v2")
_saved = {}


def shim(name):
    if name in NAMES:
        return importlib.import_module(f"vidar_amd.third_lib.{name}")
    if name in ("chamferdist", "chamferdist._C", "chamferdist.chamfer"):
        return importlib.import_module("vidar_amd.third_lib." + name)
    if name == "_ext":
        return importlib.import_module("vidar_amd.third_lib.mmcv_ext")
    raise KeyError(name)


def load(name, sources=None, **kw):
    """stand-in for torch.utils.cpp_extension.load: the named shim, nothing is compiled"""
    if name not in NAMES:
        raise RuntimeError(f"vidar_amd.dropin.load: no HIP shim for extension {name!r} (have {NAMES})")
    return shim(name)


def load_ext(name, funcs):
    """stand-in for mmcv.utils.ext_loader.load_ext"""
    mod = shim(name)
    for f in funcs:
        if not hasattr(mod, f):
            raise AttributeError(f"{f} is not provided by the HIP shim of {name}")
    return mod


def install(patch_loaders=True):
    for n in NAMES + ("chamferdist", "chamferdist._C", "chamferdist.chamfer"):
        _saved.setdefault(("mod", n), sys.modules.get(n))
        sys.modules[n] = shim(n)
    if patch_loaders:
        import torch.utils.cpp_extension as E
        _saved.setdefault(("load",), E.load)
        E.load = load
        ext_loader = sys.modules.get("mmcv.utils.ext_loader") or getattr(sys.modules.get("mmcv.utils"), "ext_loader", None)
        if ext_loader is not None:
            _saved.setdefault(("load_ext", ext_loader), ext_loader.load_ext)
            ext_loader.load_ext = load_ext


def uninstall():
    for key, val in list(_saved.items()):
        if key[0] == "mod":
            if val is None:
                sys.modules.pop(key[1], None)
            else:
                sys.modules[key[1]] = val
        elif key[0] == "load":
            import torch.utils.cpp_extension as E
            E.load = val
        elif key[0] == "load_ext":
            key[1].load_ext = val
    _saved.clear()
