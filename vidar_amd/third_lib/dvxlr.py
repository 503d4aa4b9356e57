"""`dvxlr` -- same callables as the reference's JIT-built extension
(third_lib/dvxlr/dvxlr.cpp:61-65; loaded at bevformer/utils/e2e_predictor_utils.py:86-90)."""
from __future__ import annotations

import torch

from .._lib import lib, check, ptr, stream_of, workspace
from ._common import check_input, ray_dims

MAX_D = 1026  # dvxlr.cu:10


def render(sigma, origin, points, tindex):
    """-> [pred_dist [N,M], gt_dist [N,M], dd_dsigma [N,M,1026], indices [N,M,1026,3]]"""
    for x, nm in ((sigma, "sigma"), (origin, "origin"), (points, "points"), (tindex, "tindex")):
        check_input(x, nm)
    N, M, T, TO, Z, Y, X = ray_dims(sigma, origin, points, tindex)
    dev = sigma.device
    pred = torch.empty((N, M), device=dev); gt = torch.empty((N, M), device=dev)
    dd = torch.empty((N, M, MAX_D), device=dev); idx = torch.empty((N, M, MAX_D, 3), device=dev)
    check(lib().vidar_dvxlr_render_f32(ptr(sigma), ptr(origin), ptr(points), ptr(tindex), ptr(pred),
                                       ptr(gt), ptr(dd), ptr(idx), N, M, T, TO, Z, Y, X,
                                       stream_of(sigma)), "dvxlr.render")
    return [pred, gt, dd, idx]


def get_grad_sigma(elementwise_mult, indices, tindex, sigma_shape):
    """`sigma_shape` is a tensor shaped like sigma (the reference passes sigma itself and uses
    zeros_like, dvxlr.cu:138).  -> [grad_sigma]"""
    for x, nm in ((elementwise_mult, "elementwise_mult"), (indices, "indices"), (tindex, "tindex")):
        check_input(x, nm)
    N, T, Z, Y, X = sigma_shape.shape
    M, L = elementwise_mult.shape[1], elementwise_mult.shape[2]
    grad = torch.empty((N, T, Z, Y, X), device=elementwise_mult.device)
    ws, wsp, wsn = workspace(lib().vidar_dvxlr_get_grad_sigma_workspace_bytes, N, T, Z, Y, X, 1, like=grad)
    check(lib().vidar_dvxlr_get_grad_sigma_f32(ptr(elementwise_mult), ptr(indices), ptr(tindex),
                                               ptr(grad), N, M, L, T, Z, Y, X, wsp, wsn,
                                               stream_of(grad)), "dvxlr.get_grad_sigma")
    return [grad]


def init(points, tindex, grid):
    """grid = [T, Z, Y, X] -> occupancy [N,T,Z,Y,X]"""
    check_input(points, "points"); check_input(tindex, "tindex")
    T, Z, Y, X = (int(g) for g in grid)
    N, M = points.shape[:2]
    occ = torch.empty((N, T, Z, Y, X), device=points.device)
    check(lib().vidar_dvr_init_f32(ptr(points), ptr(tindex), ptr(occ), N, M, T, Z, Y, X,
                                   stream_of(points)), "dvxlr.init")
    return occ
