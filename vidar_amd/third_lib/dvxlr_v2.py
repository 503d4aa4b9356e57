"""`dvxlr_v2` (third_lib/dvxlr/dvxlr_v2.cpp:66-70; loaded at e2e_predictor_utils.py:118-121)."""
from __future__ import annotations

import torch

from .._lib import lib, check, ptr, stream_of, workspace
from ._common import check_input, ray_dims

MAX_D = 1026


def render_v2(sigma, origin, points, tindex, sigma_regul):
    """-> [pred_dist, gt_dist, dd_dsigma, indices, ray_pred, indicator]"""
    for x, nm in ((sigma, "sigma"), (origin, "origin"), (points, "points"), (tindex, "tindex"),
                  (sigma_regul, "sigma_regul")):
        check_input(x, nm)
    N, M, T, TO, Z, Y, X = ray_dims(sigma, origin, points, tindex)
    if sigma_regul.shape != sigma.shape:
        raise RuntimeError("sigma_regul must have sigma's shape")
    dev = sigma.device
    pred = torch.empty((N, M), device=dev); gt = torch.empty((N, M), device=dev)
    dd = torch.empty((N, M, MAX_D), device=dev); idx = torch.empty((N, M, MAX_D, 3), device=dev)
    rp = torch.empty((N, M, MAX_D), device=dev); ind = torch.empty((N, M, MAX_D), device=dev)
    check(lib().vidar_dvxlr2_render_f32(ptr(sigma), ptr(origin), ptr(points), ptr(tindex),
                                        ptr(sigma_regul), ptr(pred), ptr(gt), ptr(dd), ptr(idx),
                                        ptr(rp), ptr(ind), N, M, T, TO, Z, Y, X, stream_of(sigma)),
          "dvxlr_v2.render_v2")
    return [pred, gt, dd, idx, rp, ind]


def get_grad_sigma_v2(elementwise_mult, indices, tindex, sigma_shape, indicator, grad_ray_pred):
    """-> [grad_sigma, grad_sigma_regul]"""
    for x, nm in ((elementwise_mult, "elementwise_mult"), (indices, "indices"), (tindex, "tindex"),
                  (indicator, "indicator"), (grad_ray_pred, "grad_ray_pred")):
        check_input(x, nm)
    N, T, Z, Y, X = sigma_shape.shape
    M, L = elementwise_mult.shape[1], elementwise_mult.shape[2]
    dev = elementwise_mult.device
    g = torch.empty((N, T, Z, Y, X), device=dev); g2 = torch.empty((N, T, Z, Y, X), device=dev)
    ws, wsp, wsn = workspace(lib().vidar_dvxlr_get_grad_sigma_workspace_bytes, N, T, Z, Y, X, 2, like=g)
    check(lib().vidar_dvxlr2_get_grad_sigma_f32(ptr(elementwise_mult), ptr(indices), ptr(tindex),
                                                ptr(indicator), ptr(grad_ray_pred), ptr(g), ptr(g2),
                                                N, M, L, T, Z, Y, X, wsp, wsn, stream_of(g)),
          "dvxlr_v2.get_grad_sigma_v2")
    return [g, g2]
