"""Import pieces of the PYTHON reference (/root/reference) in this container to generate golden
vectors.  mmcv / mmdet / mmdet3d are not installable here, so their few symbols the imported files
touch at import time are stubbed (registries -> no-op decorators, Linear -> nn.Linear,
BaseModule -> nn.Module).  None of the arithmetic under test lives in the stubs.
Used only by the tests/golden/make_*.py generators."""
import importlib.util
import sys
import types
from pathlib import Path

import torch
import torch.nn as nn

REF = Path("/root/reference")
PLUGIN = REF / "projects/mmdet3d_plugin"


class _Registry:
    def register_module(self, *a, **k):
        return lambda cls: cls


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    if "mmcv" in sys.modules and getattr(sys.modules["mmcv"], "_vidar_stub", False):
        return
    ident = lambda *a, **k: (lambda f: f)

    class BaseModule(nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()

    _mod("mmcv", _vidar_stub=True)
    _mod("mmcv.cnn", Linear=nn.Linear, bias_init_with_prob=lambda p: 0.0,
         xavier_init=lambda *a, **k: None, constant_init=lambda *a, **k: None)
    _mod("mmcv.cnn.bricks")
    _mod("mmcv.cnn.bricks.registry", ATTENTION=_Registry(), TRANSFORMER_LAYER=_Registry(),
         TRANSFORMER_LAYER_SEQUENCE=_Registry())
    _mod("mmcv.cnn.bricks.transformer", build_positional_encoding=lambda cfg: None)
    _mod("mmcv.runner", force_fp32=ident, auto_fp16=ident)
    _mod("mmcv.runner.base_module", BaseModule=BaseModule, ModuleList=nn.ModuleList,
         Sequential=nn.Sequential)
    _mod("mmdet")
    _mod("mmdet.models", HEADS=_Registry(), build_loss=lambda cfg: None)
    _mod("mmdet.models.utils", build_transformer=lambda cfg: None)
    _mod("mmdet3d")
    _mod("mmdet3d.models")

    def chamfer_distance(src, dst, src_weight=1.0, dst_weight=1.0, criterion_mode="l2",
                         reduction="mean"):
        # [3P] mmdet3d v0.17.1 formula (dense expand, mse, mean) -- see oracle/chamfer.py
        d = ((src.unsqueeze(2) - dst.unsqueeze(1)) ** 2).sum(-1)
        d1, i1 = d.min(2)
        d2, i2 = d.min(1)
        return (d1 * src_weight).mean(1).mean(), (d2 * dst_weight).mean(1).mean(), i1, i2
    _mod("mmdet3d.models.losses", chamfer_distance=chamfer_distance)

    class _CD(nn.Module):
        pass
    _mod("chamferdist", ChamferDistance=_CD)


def load_file(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def latent_rendering_module():
    """reference modules/ray_operations/latent_rendering.py with its hard-coded device='cuda'
    default (:14) switched to 'cpu'."""
    install_stubs()
    m = load_file("ref_latent_rendering", PLUGIN / "bevformer/modules/ray_operations/latent_rendering.py")
    d = list(m.get_bev_grids.__defaults__)
    d[d.index("cuda")] = "cpu"
    m.get_bev_grids.__defaults__ = tuple(d)
    return m


def head_modules():
    """reference dense_heads/vidar_head_base.py + utils/e2e_predictor_utils.py.  The import-time
    JIT of dvxlr/dvxlr_v2 (e2e_predictor_utils.py:86-90,118-121) is turned into a no-op."""
    install_stubs()
    import torch.utils.cpp_extension as ce
    real = ce.load
    ce.load = lambda *a, **k: None
    try:
        pkg = _mod("refvidar"); pkg.__path__ = []
        up = _mod("refvidar.utils"); up.__path__ = []
        dp = _mod("refvidar.dense_heads"); dp.__path__ = []
        e2e = load_file("refvidar.utils.e2e_predictor_utils",
                        PLUGIN / "bevformer/utils/e2e_predictor_utils.py")
        up.e2e_predictor_utils = e2e
        for fn in (e2e.get_bev_grids, e2e.get_bev_grids_3d):
            d = list(fn.__defaults__); d[d.index("cuda")] = "cpu"; fn.__defaults__ = tuple(d)
        head = load_file("refvidar.dense_heads.vidar_head_base",
                         PLUGIN / "bevformer/dense_heads/vidar_head_base.py")
    finally:
        ce.load = real
    return head, e2e
