"""LatentRendering -- registered ATTENTION module with the reference's constructor kwargs and
parameter names (`unsup_raymarching_head.*`, `lora_a.*`, `lora_b.*`), forward semantics of
projects/mmdet3d_plugin/bevformer/modules/ray_operations/latent_rendering.py:37-162.
The ray-march (":96-150") runs in two fused gfx950 kernels + adjoints (csrc/latent_render.hip);
the three Linear layers stay GEMMs (rocBLAS/hipBLASLt through torch)."""
from __future__ import annotations

import ctypes

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ...bricks import Linear
from ...registry import ATTENTION
from ...._lib import lib, check, ptr, stream_of, workspace, TIMER

_ACT = {"sigmoid": 0, "exp": 1}


def _step(grid_step, H, W):
    # grid_step / (min(h, w) // 2) in Python double, then cast to the f32 tensor dtype (:102-104)
    return float(np.float32(grid_step / (min(H, W) // 2)))


class _PathProb(Function):
    @staticmethod
    def forward(ctx, occ, grid_num, grid_step, act):
        bs, H, W, Z = occ.shape
        occ = occ.float().contiguous()
        prob = torch.empty_like(occ)
        step = _step(grid_step, H, W)
        with TIMER.span("lr_prob_fwd", 4 * occ.numel() * 2):
          check(lib().vidar_latent_render_prob_fwd_f32(ptr(occ), ptr(prob), bs, H, W, Z, grid_num,
                                                     ctypes.c_float(step), act, stream_of(occ)),
              "latent_render_prob_fwd")
        ctx.save_for_backward(occ)
        ctx.cfg = (grid_num, step, act)
        return prob

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_prob):
        (occ,) = ctx.saved_tensors
        grid_num, step, act = ctx.cfg
        bs, H, W, Z = occ.shape
        g = torch.empty_like(occ)
        ws, wsp, wsn = workspace(lib().vidar_latent_render_bwd_workspace_bytes, bs, H, W, Z, 1, like=occ)
        with TIMER.span("lr_prob_bwd", 4 * occ.numel() * 3):
          check(lib().vidar_latent_render_prob_bwd_f32(ptr(occ), ptr(grad_prob.float().contiguous()),
                                                     ptr(g), bs, H, W, Z, grid_num,
                                                     ctypes.c_float(step), act, wsp, wsn, stream_of(occ)),
              "latent_render_prob_bwd")
        return g, None, None, None


class _RayGather(Function):
    @staticmethod
    def forward(ctx, prob, a, grid_num, grid_step, eps):
        bs, H, W, Z = prob.shape
        prob = prob.float().contiguous(); a = a.float().contiguous()
        feat = torch.empty_like(prob); msum = torch.empty_like(prob)
        step = _step(grid_step, H, W)
        with TIMER.span("lr_gather_fwd", 4 * prob.numel() * 4):
          check(lib().vidar_latent_render_gather_fwd_f32(ptr(prob), ptr(a), ptr(feat), ptr(msum), bs, H,
                                                       W, Z, grid_num, ctypes.c_float(step),
                                                       ctypes.c_float(eps), stream_of(prob)),
              "latent_render_gather_fwd")
        ctx.save_for_backward(prob, a, feat, msum)
        ctx.cfg = (grid_num, step, eps)
        return feat

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_feat):
        prob, a, feat, msum = ctx.saved_tensors
        grid_num, step, eps = ctx.cfg
        bs, H, W, Z = prob.shape
        gp = torch.empty_like(prob); ga = torch.empty_like(a)
        ws, wsp, wsn = workspace(lib().vidar_latent_render_bwd_workspace_bytes, bs, H, W, Z, 2, like=prob)
        with TIMER.span("lr_gather_bwd", 4 * prob.numel() * 7):
          check(lib().vidar_latent_render_gather_bwd_f32(ptr(prob), ptr(a), ptr(feat), ptr(msum),
                                                       ptr(grad_feat.float().contiguous()), ptr(gp),
                                                       ptr(ga), bs, H, W, Z, grid_num,
                                                       ctypes.c_float(step), ctypes.c_float(eps),
                                                       wsp, wsn, stream_of(prob)), "latent_render_gather_bwd")
        return gp, ga, None, None, None


def latent_render_path_prob(occ, grid_num, grid_step, act="sigmoid"):
    """occ [bs,H,W,16] logits -> path probability [bs,H,W,16] (stage 1)."""
    return _PathProb.apply(occ, int(grid_num), float(grid_step), _ACT[act])


def latent_render_gather(prob, a, grid_num, grid_step, eps=1e-3):
    """prob, a [bs,H,W,16] -> ray-aggregated feature [bs,H,W,16] (stage 2)."""
    return _RayGather.apply(prob, a, int(grid_num), float(grid_step), float(eps))


@ATTENTION.register_module()
class LatentRendering(nn.Module):
    def __init__(self, embed_dims=256, num_pred_fcs=2, pred_height=1, grid_num=128, grid_step=0.5,
                 reduction=16, act="exp", viz_response=False, init_cfg=None):
        super().__init__()
        if act not in _ACT:
            raise NotImplementedError("Only support exp or sigmoid activation_fn for now.")
        self.embed_dims = embed_dims
        self.num_pred_fcs = num_pred_fcs
        self.grid_num = grid_num
        self.grid_step = grid_step
        self.viz_response = viz_response
        self.act = act
        branch = []
        for _ in range(num_pred_fcs):
            branch += [Linear(embed_dims, embed_dims), nn.LayerNorm(embed_dims), nn.ReLU(inplace=True)]
        branch.append(Linear(embed_dims, pred_height))
        self.unsup_raymarching_head = nn.Sequential(*branch)
        self.pred_height = pred_height
        self.lora_a = Linear(embed_dims, embed_dims // reduction)
        self.lora_b = Linear(embed_dims // reduction, embed_dims)
        if pred_height != 16 or embed_dims // reduction != pred_height:
            # the reference's view(bs, pred_height, -1, ...) (:151-154) allows more LoRA channels per
            # height bin; every released config uses 16/16 and the kernels are specialised for it
            raise NotImplementedError("LatentRendering HIP kernels need pred_height == "
                                      "embed_dims // reduction == 16")

    def forward(self, embed, eps=1e-3, **kwargs):
        bs, bev_h, bev_w, _ = embed.shape
        occ = self.unsup_raymarching_head(embed)                       # [bs,h,w,16]
        prob = latent_render_path_prob(occ, self.grid_num, self.grid_step, self.act)
        feat = latent_render_gather(prob, self.lora_a(embed), self.grid_num, self.grid_step, eps)
        out = self.lora_b(feat)                                        # [bs,h,w,C]
        shape = out.shape
        out = out.view(bs, bev_h, bev_w, self.pred_height, -1) * prob.view(bs, bev_h, bev_w,
                                                                           self.pred_height, 1)
        return out.view(shape)
