"""4d-occ ray-error metrics (L1 / AbsRel along GT rays) with the names and semantics of
projects/mmdet3d_plugin/bevformer/utils/eval_utils.py:8-225 (host-side evaluation code of the
reference, numpy).  The per-ray python loops of `_clamp` (:79-173) are vectorised over rays (six
passes over the sorted plane hits instead of a loop per ray); the nearest-neighbour match in
spherical coordinates (:199-203) runs on the gfx950 KNN kernel through chamferdist."""
from __future__ import annotations

import numpy as np
import torch

PC_RANGE = [-70.0, -70.0, -4.5, 70.0, 70.0, 4.5]     # eval_utils.py:8 (fixed, not the model's range)
MAX_VALUE = 1e8


def inside_volume(xyz):
    lo = np.array(PC_RANGE[:3]) - 0.02
    hi = np.array(PC_RANGE[3:]) + 0.02
    return np.logical_and(xyz >= lo, xyz <= hi).all(-1)


def _first_hit(start, direction, dist_to_planes):
    """first plane crossing (in ascending distance, distance >= -1e-4) whose point lies in the
    volume; -> (found [N] bool, point [N,3], distance [N])"""
    order = np.argsort(dist_to_planes, axis=0)
    n = start.shape[0]
    found = np.zeros(n, bool)
    point = np.full((n, 3), np.inf)
    dist = np.full(n, np.inf)
    cols = np.arange(n)
    for i in range(order.shape[0]):
        t = dist_to_planes[order[i], cols]
        inter = start + t[:, None] * direction
        hit = (~found) & (t + 1e-4 >= 0.0) & inside_volume(inter)
        point[hit] = inter[hit]
        dist[hit] = t[hit]
        found |= hit
    return found, point, dist


def _plane_distances(p, d, sign):
    out = []
    for a in range(3):
        da = sign * d[:, a]
        near0 = np.isclose(d[:, a], 0.0)
        with np.errstate(divide="ignore", invalid="ignore"):
            out.append(np.where(near0, MAX_VALUE, (PC_RANGE[a] - p[:, a]) / da))
            out.append(np.where(near0, MAX_VALUE, (PC_RANGE[a + 3] - p[:, a]) / da))
    return np.stack(out)


def _clamp(points, origin):
    """clip rays origin->points to PC_RANGE (:79-173). -> (new_origin [N,3], points [N,3]);
    rays that miss the volume become points at infinity."""
    points = points.astype(float).copy()
    origin = origin.reshape(1, 3).astype(float)
    new_origin = np.zeros_like(points) + origin
    v = points - origin
    if (v == 0).all(1).any():
        raise RuntimeError("Origin and the end point should not be identical")
    l = np.sqrt((v ** 2).sum(1))
    d = v / l[:, None]
    intersects = np.ones(len(points), bool)
    if not inside_volume(origin)[0]:
        found, hit, t = _first_hit(np.repeat(origin, len(points), 0), d, _plane_distances(
            np.repeat(origin, len(points), 0), d, 1.0))
        intersects = found
        new_origin[found] = hit[found]
        beyond = found & (t > l)
        points[beyond] = new_origin[beyond]
    outside = intersects & ~inside_volume(points)
    if outside.any():
        found, hit, _ = _first_hit(points[outside], -d[outside], _plane_distances(points[outside], d[outside], -1.0))
        assert found.all()
        points[outside] = hit
    new_origin[~intersects] = np.inf
    points[~intersects] = np.inf
    return new_origin, points


def clamp(pcd_, org_, return_invalid_mask=False):
    pcd = pcd_.cpu().numpy().copy() if torch.is_tensor(pcd_) else np.array(pcd_, copy=True)
    org = org_.cpu().numpy().copy() if torch.is_tensor(org_) else np.array(org_, copy=True)
    inner = ((np.array(PC_RANGE[:3]) <= pcd) & (pcd <= np.array(PC_RANGE[3:]))).all(1)
    origins = np.zeros_like(pcd) + org.reshape(1, 3)
    if (~inner).any():
        o, p = _clamp(pcd[~inner], org.reshape(1, 3))
        pcd[~inner] = p.astype(float)
        origins[~inner] = o
    invalid = np.isinf(pcd).all(1) | np.isnan(pcd).all(1)
    if return_invalid_mask:
        return origins, pcd, invalid
    return origins[~invalid], pcd[~invalid]


def spherical_projection(pcd):
    x, y, z = pcd.T
    return np.arctan2(x, y), np.arctan2(z, y), np.sqrt(x * x + y * y + z * z)


def compute_ray_errors(pred_pcd, gt_pcd, origin, device, return_interpolated_pcd=False, savename="",
                       chamfer=None):
    """-> (l1_error, absrel_error) averaged over GT rays (:185-225).  `chamfer` defaults to the HIP
    ChamferDistance; tests inject a CPU implementation."""
    if chamfer is None:
        from ...third_lib.chamferdist import ChamferDistance
        chamfer = ChamferDistance()
    theta_hat, phi_hat, d_hat = spherical_projection(pred_pcd - origin[None, :])
    theta, phi, d = spherical_projection(gt_pcd - origin[None, :])
    mh, m = d_hat > 1e-2, d > 1e-2
    theta_hat, phi_hat, d_hat, pred_pcd = theta_hat[mh], phi_hat[mh], d_hat[mh], pred_pcd[mh]
    theta, phi, d, gt_pcd = theta[m], phi[m], d[m], gt_pcd[m]
    count = theta.shape[0]
    ps = np.stack([theta_hat, phi_hat, np.ones_like(theta_hat)], 1)
    gs = np.stack([theta, phi, np.ones_like(theta)], 1)
    _, info = chamfer(torch.from_numpy(ps[None]).float().to(device), torch.from_numpy(gs[None]).float().to(device),
                      reverse=True, reduction="mean")
    pred_idx = info[1].cpu().numpy()
    v = gt_pcd - origin[None, :]
    unit = v / np.sqrt((v ** 2).sum(1, keepdims=True))
    interp = origin[None, :] + d_hat[pred_idx].T * unit
    if return_interpolated_pcd:
        return interp
    g_origin, g_pcd, invalid = clamp(gt_pcd, origin, return_invalid_mask=True)
    _, p_pcd, _ = clamp(interp, origin, return_invalid_mask=True)
    keep = ~invalid
    g_pcd, p_pcd, g_origin = g_pcd[keep], p_pcd[keep], g_origin[keep]
    dc = np.sqrt(((g_pcd - g_origin) ** 2).sum(1))
    valid = dc > 0.01
    eucl = np.sqrt(((g_pcd[valid] - p_pcd[valid]) ** 2).sum(1))
    return eucl.sum() / count, (eucl / dc[valid]).sum() / count
