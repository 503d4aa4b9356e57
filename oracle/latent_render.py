"""CPU oracle for LatentRendering -- TEST INFRASTRUCTURE.

Restates projects/mmdet3d_plugin/bevformer/modules/ray_operations/latent_rendering.py:79-162
(forward) with the device parametrised, as plain torch ops (so autograd provides the backward).
Pinned against the reference module itself, imported in the build container with mmcv stubbed
(tests/golden/make_latent_render_golden.py -> tests/golden/latent_render_*.npz)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def bev_grids(H, W, bs, dtype=torch.float32):
    """latent_rendering.py:14-34 (offset 0.5): cell centres in [0,1], (x, y) order."""
    ys = torch.linspace(0.5, H - 0.5, H, dtype=dtype)
    xs = torch.linspace(0.5, W - 0.5, W, dtype=dtype)
    ry, rx = torch.meshgrid(ys, xs, indexing="ij")
    g = torch.stack((rx.reshape(-1) / W, ry.reshape(-1) / H), -1)
    return g[None].repeat(bs, 1, 1)


def ray_geometry(H, W, bs, grid_num, grid_step):
    grids = bev_grids(H, W, bs)
    r = grids - 0.5
    rn = torch.nan_to_num(r / torch.sqrt((r ** 2).sum(-1, keepdim=True)))
    step = grid_step / (min(H, W) // 2)
    steps = torch.from_numpy(np.arange(0, grid_num) + 0.5).to(rn.dtype) * step
    prev = 0.5 + rn.view(bs, -1, 1, 2) * steps.view(1, 1, -1, 1)
    path = torch.cat([prev, grids.view(bs, H * W, 1, 2)], 2) * 2 - 1       # [bs,Q,G+1,2]
    return rn, path


def path_prob(occ, grid_num, grid_step, act="sigmoid"):
    """stage 1 (:96-129). occ [bs,H,W,Z] (channel-last logits) -> [bs,H,W,Z]"""
    bs, H, W, Z = occ.shape
    rn, path = ray_geometry(H, W, bs, grid_num, grid_step)
    s = F.grid_sample(occ.permute(0, 3, 1, 2).contiguous(), path, align_corners=False)
    s = s.permute(0, 2, 3, 1)                                                # [bs,Q,G+1,Z]
    length = torch.sqrt((path ** 2).sum(-1, keepdim=True))
    valid = length < length[..., -1:, :]
    if act == "exp":
        p = 1 - torch.exp(-F.relu(s))
    elif act == "sigmoid":
        p = torch.sigmoid(s)
    else:
        raise NotImplementedError(act)
    trans = torch.cumprod(1 - p * valid, dim=2)
    return (trans[..., -1, :] * p[..., -1, :]).view(bs, H, W, Z)


def gather(prob, a, grid_num, grid_step, eps=1e-3):
    """stage 2 (:131-150). prob, a [bs,H,W,Z] -> feat [bs,H,W,Z] (Z = lora channels = heights)"""
    bs, H, W, Z = prob.shape
    rn, path = ray_geometry(H, W, bs, grid_num, grid_step)
    length = torch.sqrt((path ** 2).sum(-1, keepdim=True))
    path = path[..., :-1, :].contiguous()
    av = F.grid_sample(a.permute(0, 3, 1, 2).contiguous(), path, align_corners=False)   # [bs,Z,Q,G]
    bound = torch.minimum(1 / rn[..., 0:1].abs(), 1 / rn[..., 1:2].abs())
    valid = length[..., :-1, :] < bound.view(bs, -1, 1, 1)
    m = F.grid_sample(prob.permute(0, 3, 1, 2).contiguous(), path, align_corners=False)
    m = m * valid.view(bs, 1, H * W, grid_num)
    m = m / (m.sum(-1, keepdim=True) + eps)
    return (av * m).sum(-1).permute(0, 2, 1).reshape(bs, H, W, Z)


def forward(embed, w_occ, b_occ, w_a, b_a, w_b, b_b, grid_num, grid_step, act="sigmoid", eps=1e-3):
    """Whole LatentRendering.forward for num_pred_fcs=0 (the released configs):
    unsup_raymarching_head = Linear(C->Z), lora_a = Linear(C->C/r), lora_b = Linear(C/r->C)."""
    bs, H, W, C = embed.shape
    Z = w_occ.shape[0]
    occ = F.linear(embed, w_occ, b_occ)
    prob = path_prob(occ, grid_num, grid_step, act)
    feat = gather(prob, F.linear(embed, w_a, b_a), grid_num, grid_step, eps)
    out = F.linear(feat, w_b, b_b)
    return (out.view(bs, H, W, Z, -1) * prob.view(bs, H, W, Z, 1)).view(bs, H, W, C)
