"""Deterministic synthetic inputs for the hot path: LiDAR-like ray sets in voxel
coordinates, occupancy volumes, multi-scale deformable-attention operands.  numpy only."""
from __future__ import annotations

import numpy as np

PC_RANGE = (-51.2, -51.2, -5.0, 51.2, 51.2, 3.0)   # config vidar_1_8_nusc_3future.py:10


def metric_to_voxel(xyz, bev_h=200, bev_w=200, pillar=16, pc_range=PC_RANGE):
    """e2e_predictor_utils.coords_to_voxel_grids (utils/e2e_predictor_utils.py:36-45)."""
    out = np.array(xyz, dtype=np.float32, copy=True)
    out[..., 0] = (out[..., 0] - pc_range[0]) / (pc_range[3] - pc_range[0]) * bev_w
    out[..., 1] = (out[..., 1] - pc_range[1]) / (pc_range[4] - pc_range[1]) * bev_h
    out[..., 2] = (out[..., 2] - pc_range[2]) / (pc_range[5] - pc_range[2]) * pillar
    return out


def lidar_points(rng: np.random.Generator, P: int, max_range=70.0):
    """P metric end points of a 32-beam spinning LiDAR at the origin."""
    az = rng.uniform(0.0, 2 * np.pi, P)
    beams = np.deg2rad(np.linspace(-30.0, 10.0, 32))
    el = beams[rng.integers(0, 32, P)]
    rg = np.minimum(rng.lognormal(np.log(15.0), 0.7, P), max_range)
    xyz = np.stack([rg * np.cos(el) * np.cos(az), rg * np.cos(el) * np.sin(az), rg * np.sin(el)], -1)
    return xyz.astype(np.float32)


def ray_set(seed=0, N=1, T=1, rays_per_frame=30000, grid=(16, 200, 200), pad=0, origin_jitter=0.0,
            sigma_T=None):
    """Inputs of dvr/dvxlr: sigma [N,T,Z,Y,X], origin [N,T,3], points [N,M,3], tindex [N,M]
    (all float32, voxel units), M = T*rays_per_frame + pad; padded rays have tindex -1 / NaN points."""
    rng = np.random.default_rng(seed)
    Z, Y, X = grid
    sT = T if sigma_T is None else sigma_T
    sigma = (np.log1p(np.exp(rng.standard_normal((N, sT, Z, Y, X)))) * 0.1).astype(np.float32)
    origin_m = np.zeros((N, T, 3), np.float32)
    if origin_jitter:
        origin_m[..., :2] = rng.uniform(-origin_jitter, origin_jitter, (N, T, 2))
    pts, tix = [], []
    for n in range(N):
        p_n, t_n = [], []
        for t in range(T):
            p = lidar_points(rng, rays_per_frame) + origin_m[n, t]
            p_n.append(metric_to_voxel(p, Y, X, Z))
            t_n.append(np.full(rays_per_frame, t, np.float32))
        if pad:
            p_n.append(np.full((pad, 3), np.nan, np.float32))
            t_n.append(np.full(pad, -1, np.float32))
        pts.append(np.concatenate(p_n)); tix.append(np.concatenate(t_n))
    points = np.stack(pts); tindex = np.stack(tix)
    # shuffle rays so frames interleave (as after voxel sub-sampling) -- deterministic
    perm = rng.permutation(points.shape[1])
    points = np.ascontiguousarray(points[:, perm]); tindex = np.ascontiguousarray(tindex[:, perm])
    origin = metric_to_voxel(origin_m, Y, X, Z)
    return sigma, origin, points, tindex
