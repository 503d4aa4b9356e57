"""CPU oracle for the ViDAR head's ray-march / loss math -- TEST INFRASTRUCTURE.

Restates (torch CPU, per-ray outputs kept in input order so they can be compared 1:1 with the
fused kernels) projects/mmdet3d_plugin/bevformer/dense_heads/vidar_head_base.py:
  grid_features   <- _get_grid_features :420-509 (single batch item, single level)
  ce_per_ray      <- F.cross_entropy(label 0) :586-592
  gumbel_distance <- _custom_gumbel_softmax_distance :754-773 (noise injected instead of sampled)
  argmax_decode   <- get_point_cloud_prediction :697-731
  rendered_points <- get_rendered_pcds :344-389
Pinned against the reference's own methods (imported with mmcv/mmdet stubbed) by
tests/golden/make_head_golden.py -> tests/golden/head_*.npz."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def _waypoints(origin, pts, num, step, with_end=True):
    r = pts - origin
    rn = r / torch.sqrt((r ** 2).sum(-1, keepdim=True))
    rg = torch.from_numpy(np.arange(0, num) + 0.5).to(pts.dtype) * step
    g = origin.view(-1, 1, 3) + rn.view(-1, 1, 3) * rg.view(1, -1, 1)
    if with_end:
        g = torch.cat([pts.view(-1, 1, 3), g], 1)
    length = torch.sqrt(((g - origin.view(-1, 1, 3)) ** 2).sum(-1))
    return g, length


def _normalise(g, Z, Y, X):
    g = g.clone()
    g[..., 0] = g[..., 0] / X
    g[..., 1] = g[..., 1] / Y
    g[..., 2] = g[..., 2] / Z
    return g * 2 - 1


def grid_features(sigma, origin, pts, tindex, num=512, step=1.0):
    """sigma [F,Z,Y,X], origin [F,3], pts [R,3], tindex [R].
    -> feat [R, num+1] (-inf masked), length [R, num+1], keep [R] bool (ray contributes)"""
    Fn, Z, Y, X = sigma.shape
    R = pts.shape[0]
    feat = torch.zeros(R, num + 1, dtype=sigma.dtype)
    length = torch.zeros(R, num + 1, dtype=sigma.dtype)
    keep = torch.zeros(R, dtype=torch.bool)
    for f in range(Fn):
        sel = (tindex == f).nonzero().squeeze(-1)
        if sel.numel() == 0:
            continue
        g, ln = _waypoints(origin[f:f + 1], pts[sel], num, step)
        gn = _normalise(g, Z, Y, X)
        mask = ((gn <= -1.) | (gn >= 1)).any(-1)
        ok = ((gn[:, 0] > -1.) & (gn[:, 0] < 1.)).all(-1)
        s = F.grid_sample(sigma[f].view(1, 1, Z, Y, X), gn.view(1, 1, *gn.shape)).view(gn.shape[0], -1)
        s = s + mask.float().masked_fill(mask, float("-inf"))
        idx = sel[ok]
        feat = feat.index_put((idx,), s[ok])
        length = length.index_put((idx,), ln[ok])
        keep[idx] = True
    return feat, length, keep


def ce_per_ray(feat):
    """cross entropy with target class 0 over the num+1 samples of each ray."""
    return torch.logsumexp(feat, -1) - feat[:, 0]


def gumbel_distance(feat, length, noise):
    """feat/length [R,K] (end-point sample already removed), noise [R,K] gumbel(0,1)."""
    idx = torch.softmax(feat + noise, -1).max(-1)[1]
    hard = F.one_hot(idx, feat.shape[-1]).to(feat.dtype)
    pd = (hard * length).sum(-1).detach()
    e = torch.exp(feat - feat.max(-1, keepdim=True)[0])
    pn = (e * (length > pd.unsqueeze(-1)).float()).sum(-1) / e.sum(-1)
    pn = 1 - pn.detach() + pn
    return pn * pd


def argmax_decode(sigma, origin, pts, tindex, num=512, step=1.0):
    Fn, Z, Y, X = sigma.shape
    R = pts.shape[0]
    pred = torch.zeros(R); gt = torch.zeros(R)
    for f in range(Fn):
        sel = (tindex == f).nonzero().squeeze(-1)
        if sel.numel() == 0:
            continue
        gt[sel] = torch.sqrt(((pts[sel] - origin[f:f + 1]) ** 2).sum(-1))
        g, ln = _waypoints(origin[f:f + 1], pts[sel], num, step, with_end=False)
        gn = _normalise(g, Z, Y, X)
        s = F.grid_sample(sigma[f].view(1, 1, Z, Y, X), gn.view(1, 1, *gn.shape)).view(gn.shape[0], -1)
        s = s.masked_fill(s == 0, float("-inf"))
        pred[sel] = torch.gather(ln, 1, s.max(1)[1].view(-1, 1)).squeeze(-1)
    return pred, gt


def rendered_points(origin_f, pts, dist):
    r = pts - origin_f.view(1, 3)
    return origin_f.view(1, 3) + r / torch.sqrt((r ** 2).sum(1, keepdim=True)) * dist.view(-1, 1)
