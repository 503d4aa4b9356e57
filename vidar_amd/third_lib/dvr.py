"""`dvr` (third_lib/dvr/dvr.cpp:65-69).  Unknown loss / phase names raise ValueError; the
reference prints and calls exit(1) (dvr.cu:362-365, :669-672)."""
from __future__ import annotations

import torch

from .._lib import lib, check, ptr, stream_of
from ._common import check_input, ray_dims

MAX_D = 1446  # dvr.cu:9
_LOSS = {"l1": 0, "l2": 1, "absrel": 2, "bce": 0}  # dvr.cu:658-668 ("bce" -> L1)
_PHASE = {"test": 0, "train": 1}                   # dvr.cu:357-361


def render_forward(sigma, origin, points, tindex, grid, phase_name):
    """-> [pred_dist, gt_dist].  `grid` is accepted and ignored like in the reference kernel."""
    for x, nm in ((sigma, "sigma"), (origin, "origin"), (points, "points"), (tindex, "tindex")):
        check_input(x, nm)
    if phase_name not in _PHASE:
        raise ValueError(f"UNKNOWN PHASE NAME: {phase_name}")
    N, M, T, TO, Z, Y, X = ray_dims(sigma, origin, points, tindex)
    pred = torch.empty((N, M), device=sigma.device); gt = torch.empty((N, M), device=sigma.device)
    check(lib().vidar_dvr_render_forward_f32(ptr(sigma), ptr(origin), ptr(points), ptr(tindex),
                                             ptr(pred), ptr(gt), N, M, T, TO, Z, Y, X,
                                             _PHASE[phase_name], stream_of(sigma)),
          "dvr.render_forward")
    return [pred, gt]


def render(sigma, origin, points, tindex, loss_name):
    """-> [pred_dist, gt_dist, grad_sigma]"""
    for x, nm in ((sigma, "sigma"), (origin, "origin"), (points, "points"), (tindex, "tindex")):
        check_input(x, nm)
    if loss_name not in _LOSS:
        raise ValueError(f"UNKNOWN LOSS TYPE: {loss_name}")
    N, M, T, TO, Z, Y, X = ray_dims(sigma, origin, points, tindex)
    pred = torch.empty((N, M), device=sigma.device); gt = torch.empty((N, M), device=sigma.device)
    grad = torch.empty_like(sigma)
    check(lib().vidar_dvr_render_f32(ptr(sigma), ptr(origin), ptr(points), ptr(tindex), ptr(pred),
                                     ptr(gt), ptr(grad), N, M, T, TO, Z, Y, X, _LOSS[loss_name],
                                     stream_of(sigma)), "dvr.render")
    return [pred, gt, grad]


def init(points, tindex, grid):
    from .dvxlr import init as _init
    return _init(points, tindex, grid)
