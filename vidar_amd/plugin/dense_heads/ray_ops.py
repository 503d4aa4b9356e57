"""Autograd wrappers of the fused head ray-march kernels (csrc/ray_march.hip).  They replace the
tensor programs inside ViDARHeadBase (dense_heads/vidar_head_base.py:420-509, :586-592, :697-731,
:754-773); per-ray results come back and the (cheap) weighting / reductions stay in torch."""
from __future__ import annotations

import ctypes

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ..._lib import lib, check, ptr, stream_of, workspace, TIMER

K_SAMPLES = 512


def _f(t):
    return t.float().contiguous()


def _dims(sigma, pts):
    F_, Z, Y, X = sigma.shape
    return F_, pts.shape[0], Z, Y, X


class _RayCE(Function):
    @staticmethod
    def forward(ctx, sigma, origin, gt, tindex, step, K):
        sigma, origin, gt, tindex = _f(sigma), _f(origin), _f(gt), _f(tindex)
        F_, R, Z, Y, X = _dims(sigma, gt)
        ce = torch.empty(R, device=sigma.device); lse = torch.empty_like(ce); valid = torch.empty_like(ce)
        with TIMER.span("ray_ce_fwd", 4 * (sigma.numel() + R * 7)):
          check(lib().vidar_ray_ce_fwd_f32(ptr(sigma), ptr(origin), ptr(gt), ptr(tindex), ptr(ce),
                                         ptr(lse), ptr(valid), F_, R, Z, Y, X, K,
                                         ctypes.c_float(step), stream_of(sigma)), "ray_ce_fwd")
        ctx.save_for_backward(sigma, origin, gt, tindex, lse)
        ctx.cfg = (step, K)
        ctx.mark_non_differentiable(valid)
        return ce, valid

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_ce, _grad_valid):
        sigma, origin, gt, tindex, lse = ctx.saved_tensors
        step, K = ctx.cfg
        F_, R, Z, Y, X = _dims(sigma, gt)
        g = torch.empty_like(sigma)
        ws, wsp, wsn = workspace(lib().vidar_ray_bwd_workspace_bytes, F_, Z, Y, X, like=sigma)
        with TIMER.span("ray_ce_bwd", 4 * (2 * sigma.numel() + R * 6)):
          check(lib().vidar_ray_ce_bwd_f32(ptr(sigma), ptr(origin), ptr(gt), ptr(tindex), ptr(lse),
                                         ptr(_f(grad_ce)), ptr(g), F_, R, Z, Y, X, K,
                                         ctypes.c_float(step), wsp, wsn, stream_of(sigma)), "ray_ce_bwd")
        return g, None, None, None, None, None


class _RayGumbel(Function):
    @staticmethod
    def forward(ctx, sigma, origin, pts, tindex, noise, step, K):
        sigma, origin, pts, tindex, noise = _f(sigma), _f(origin), _f(pts), _f(tindex), _f(noise)
        F_, R, Z, Y, X = _dims(sigma, pts)
        if noise.shape != (R, K):
            raise RuntimeError(f"noise must be [{R},{K}]")
        dist = torch.empty(R, device=sigma.device); aux = torch.empty((R, 3), device=sigma.device)
        with TIMER.span("ray_gumbel_fwd", 4 * (sigma.numel() + R * (K + 8))):
          check(lib().vidar_ray_gumbel_fwd_f32(ptr(sigma), ptr(origin), ptr(pts), ptr(tindex),
                                             ptr(noise), ptr(dist), ptr(aux), F_, R, Z, Y, X, K,
                                             ctypes.c_float(step), stream_of(sigma)), "ray_gumbel_fwd")
        ctx.save_for_backward(sigma, origin, pts, tindex, aux)
        ctx.cfg = (step, K)
        return dist

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_dist):
        sigma, origin, pts, tindex, aux = ctx.saved_tensors
        step, K = ctx.cfg
        F_, R, Z, Y, X = _dims(sigma, pts)
        g = torch.empty_like(sigma)
        ws, wsp, wsn = workspace(lib().vidar_ray_bwd_workspace_bytes, F_, Z, Y, X, like=sigma)
        with TIMER.span("ray_gumbel_bwd", 4 * (2 * sigma.numel() + R * 8)):
          check(lib().vidar_ray_gumbel_bwd_f32(ptr(sigma), ptr(origin), ptr(pts), ptr(tindex), ptr(aux),
                                             ptr(_f(grad_dist)), ptr(g), F_, R, Z, Y, X, K,
                                             ctypes.c_float(step), wsp, wsn, stream_of(sigma)), "ray_gumbel_bwd")
        return g, None, None, None, None, None, None


def ray_ce(sigma, origin, gt, tindex, step=1.0, K=K_SAMPLES):
    """sigma [F,Z,Y,X], origin [F,3], gt [R,3], tindex [R] -> (ce [R], valid [R])"""
    return _RayCE.apply(sigma, origin, gt, tindex, float(step), int(K))


def gumbel_noise(R, K=K_SAMPLES, device="cuda", generator=None):
    """F.gumbel_softmax's noise: -log(Exponential(1)) (torch/nn/functional.py gumbel_softmax)."""
    e = torch.empty((R, K), device=device, dtype=torch.float32).exponential_(generator=generator)
    return -e.log()


def ray_gumbel(sigma, origin, pts, tindex, noise=None, step=1.0, K=K_SAMPLES):
    """-> differentiable rendered distance [R] of _custom_gumbel_softmax_distance."""
    if noise is None:
        noise = gumbel_noise(pts.shape[0], K, sigma.device)
    return _RayGumbel.apply(sigma, origin, pts, tindex, noise, float(step), int(K))


def ray_argmax(sigma, origin, pts, tindex, step=1.0, K=K_SAMPLES):
    """test-time decode -> (pred_dist [R], gt_dist [R]) in voxel units, no grad."""
    sigma, origin, pts, tindex = _f(sigma.detach()), _f(origin), _f(pts), _f(tindex)
    F_, R, Z, Y, X = _dims(sigma, pts)
    pred = torch.empty(R, device=sigma.device); gt = torch.empty_like(pred)
    check(lib().vidar_ray_argmax_f32(ptr(sigma), ptr(origin), ptr(pts), ptr(tindex), ptr(pred),
                                     ptr(gt), F_, R, Z, Y, X, int(K), ctypes.c_float(step),
                                     stream_of(sigma)), "ray_argmax")
    return pred, gt
