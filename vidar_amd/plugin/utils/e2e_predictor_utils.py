"""Host helpers with the reference's names and semantics
(projects/mmdet3d_plugin/bevformer/utils/e2e_predictor_utils.py): coordinate <-> grid maps, BEV /
voxel reference grids, the autograd wrappers around the dvxlr extensions and the chamfer metric.
Nothing is JIT-compiled at import: the extensions are the prebuilt gfx950 library."""
from __future__ import annotations

import torch

from ...third_lib import dvxlr, dvxlr_v2
from ...third_lib.chamferdist import ChamferDistance


def _span(pc_range, axis):
    return pc_range[axis + 3] - pc_range[axis]


def bev_grids_to_coordinates(ref_grids, pc_range):
    """[0,1] BEV grid positions -> metric x/y (reference :8-14)."""
    out = ref_grids.clone()
    out[..., 0:1] = out[..., 0:1] * _span(pc_range, 0) + pc_range[0]
    out[..., 1:2] = out[..., 1:2] * _span(pc_range, 1) + pc_range[1]
    return out


def bev_coords_to_grids(ref_coords, bev_h, bev_w, pc_range):
    """metric x/y -> [-1,1] grid positions + mask of points away from the half-cell border (:16-34)."""
    g = ref_coords.clone()
    g[..., 0] = (g[..., 0] - pc_range[0]) / _span(pc_range, 0)
    g[..., 1] = (g[..., 1] - pc_range[1]) / _span(pc_range, 1)
    g = g * 2 - 1.
    lo_x, hi_x = 0.5 / bev_w * 2 - 1, (bev_w - 0.5) / bev_w * 2 - 1
    lo_y, hi_y = 0.5 / bev_h * 2 - 1, (bev_h - 0.5) / bev_h * 2 - 1
    valid = ((g[..., 0:1] > lo_x) & (g[..., 0:1] < hi_x) & (g[..., 1:2] > lo_y) & (g[..., 1:2] < hi_y))
    return g, valid


def coords_to_voxel_grids(ref_coords, bev_h, bev_w, pillar_num, pc_range):
    """metric xyz -> continuous voxel coordinates (:36-45)."""
    g = ref_coords.clone()
    g[..., 0] = (g[..., 0] - pc_range[0]) / _span(pc_range, 0) * bev_w
    g[..., 1] = (g[..., 1] - pc_range[1]) / _span(pc_range, 1) * bev_h
    g[..., 2] = (g[..., 2] - pc_range[2]) / _span(pc_range, 2) * pillar_num
    return g


def get_bev_grids(H, W, bs=1, device="cuda", dtype=torch.float, offset=0.5):
    """cell centres (x, y) in [0,1], row-major, [bs, H*W, 2] (:48-68)."""
    ys = torch.linspace(offset, H - (1 - offset), H, dtype=dtype, device=device)
    xs = torch.linspace(offset, W - (1 - offset), W, dtype=dtype, device=device)
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    ref = torch.stack((gx.reshape(-1) / W, gy.reshape(-1) / H), -1)
    return ref[None].repeat(bs, 1, 1)


def get_bev_grids_3d(H, W, Z, bs=1, device="cuda", dtype=torch.float):
    """voxel centres (x, y, z) in [0,1], [bs, Z, H*W, 3] (:71-83)."""
    zs = (torch.linspace(0.5, Z - 0.5, Z, dtype=dtype, device=device) / Z).view(Z, 1, 1).expand(Z, H, W)
    xs = (torch.linspace(0.5, W - 0.5, W, dtype=dtype, device=device) / W).view(1, 1, W).expand(Z, H, W)
    ys = (torch.linspace(0.5, H - 0.5, H, dtype=dtype, device=device) / H).view(1, H, 1).expand(Z, H, W)
    ref = torch.stack((xs, ys, zs), -1).reshape(Z, H * W, 3)
    return ref[None].repeat(bs, 1, 1, 1)


class DifferentiableVoxelRenderingLayer(torch.autograd.Function):
    """(:91-112) forward = dvxlr.render, backward = dvxlr.get_grad_sigma of gradpred*dd_dsigma with
    NaNs zeroed."""

    @staticmethod
    def forward(ctx, sigma, origin, points, tindex):
        pred_dist, gt_dist, dd_dsigma, indices = dvxlr.render(sigma, origin, points, tindex)
        ctx.save_for_backward(dd_dsigma, indices, tindex, sigma)
        return pred_dist, gt_dist

    @staticmethod
    def backward(ctx, gradpred, gradgt):
        dd_dsigma, indices, tindex, sigma = ctx.saved_tensors
        em = torch.nan_to_num(gradpred[..., None] * dd_dsigma, nan=0.0, posinf=float("inf"),
                              neginf=float("-inf"))
        return dvxlr.get_grad_sigma(em.contiguous(), indices, tindex, sigma)[0], None, None, None


DifferentiableVoxelRendering = DifferentiableVoxelRenderingLayer.apply


class DifferentiableVoxelRenderingLayerV2(torch.autograd.Function):
    """(:122-140)"""

    @staticmethod
    def forward(ctx, sigma, origin, points, tindex, sigma_regul):
        pred, gt, dd, idx, ray_pred, indicator = dvxlr_v2.render_v2(sigma, origin, points, tindex,
                                                                    sigma_regul)
        ctx.save_for_backward(dd, idx, tindex, sigma, indicator)
        ctx.mark_non_differentiable(indicator)
        return pred, gt, ray_pred, indicator

    @staticmethod
    def backward(ctx, gradpred, gradgt, grad_ray_pred, grad_indicator):
        dd, idx, tindex, sigma, indicator = ctx.saved_tensors
        em = (gradpred[..., None] * dd).contiguous()
        g, g_regul = dvxlr_v2.get_grad_sigma_v2(em, idx, tindex, sigma, indicator,
                                                grad_ray_pred.contiguous())
        return g, None, None, None, g_regul


DifferentiableVoxelRenderingV2 = DifferentiableVoxelRenderingLayerV2.apply


def get_inside_mask(points, point_cloud_range):
    """closed-box membership of [..., 3] points (:146-160)."""
    from .host import const_tensor
    lo = const_tensor(point_cloud_range[:3], points.device, points.dtype)
    hi = const_tensor(point_cloud_range[3:], points.device, points.dtype)
    return ((points[..., :3] >= lo) & (points[..., :3] <= hi)).all(-1)


chamfer_distance = ChamferDistance()


def compute_chamfer_distance(pred_pcd, gt_pcd):
    """bidirectional squared-L2 chamfer, each direction averaged over its cloud (:165-170)."""
    fwd, bwd, _ = chamfer_distance(pred_pcd[None, ...], gt_pcd[None, ...], bidirectional=True,
                                   reduction="sum")
    return (fwd / pred_pcd.shape[0] + bwd / gt_pcd.shape[0]) / 2.0


def compute_chamfer_distance_inner(pred_pcd, gt_pcd, pc_range):
    """(:173-183) both clouds cropped to pc_range; python float 0.0 if either crop is empty."""
    pm = get_inside_mask(pred_pcd, pc_range)
    gm = get_inside_mask(gt_pcd, pc_range)
    if pm.sum() == 0 or gm.sum() == 0:
        return 0.0
    return compute_chamfer_distance(pred_pcd[pm], gt_pcd[gm])
