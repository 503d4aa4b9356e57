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


# --------------------------------------------------------------------------------------------------
# whole-sample generator for the training step: img_metas with consistent rigid
# transforms, LiDAR-like GT clouds with the frame index in the last column, FPN feature pyramids.
# --------------------------------------------------------------------------------------------------
FPN_SHAPES_NUSC = [(116, 200), (58, 100), (29, 50), (15, 25)]      # 928x1600 input, strides 8..64
CAM_YAWS_DEG = (0.0, 55.0, -55.0, 110.0, -110.0, 180.0, 27.0, -27.0)   # nuScenes-like 6 (+2 for OpenScene's 8)


def _pose(x, y, yaw):
    c, s = np.cos(yaw), np.sin(yaw)
    T = np.eye(4)
    T[:2, :2] = [[c, -s], [s, c]]
    T[:2, 3] = [x, y]
    return T


def camera_matrices(img_hw=(928, 1600), focal=1266.0, centre=(800.0, 450.0), yaws_deg=CAM_YAWS_DEG):
    """lidar2img [cams,4,4] of pin-hole cameras looking along the given yaws."""
    K = np.eye(4)
    K[0, 0] = K[1, 1] = focal
    K[0, 2], K[1, 2] = centre
    out = []
    for d in yaws_deg:
        a = np.deg2rad(d)
        R = np.array([[np.sin(a), -np.cos(a), 0.0], [0.0, 0.0, -1.0], [np.cos(a), np.sin(a), 0.0]])
        C = np.array([0.5 * np.cos(a), 0.5 * np.sin(a), -0.3])
        E = np.eye(4)
        E[:3, :3] = R
        E[:3, 3] = -R @ C
        out.append(K @ E)
    return np.stack(out)


def make_sample(seed=0, queue_length=4, future_frames=2, rays_per_frame=30000, img_hw=(928, 1600),
                num_cams=6, first_has_prev=False):
    """-> (img_metas: list[T] of dict, gt_points [sum P, 5] float32).  T = queue_length + 1 image
    frames; GT clouds exist for T + future_frames frames (dataset convention
    datasets/nuscenes_vidar_dataset_v1.py:57-70, :199-200)."""
    rng = np.random.default_rng(seed)
    T = queue_length + 1
    n_all = T + future_frames
    xs, ys, yaws = [0.0], [0.0], [rng.uniform(-np.pi, np.pi)]
    for _ in range(1, n_all):
        xs.append(xs[-1] + rng.normal(0, 2.0)); ys.append(ys[-1] + rng.normal(0, 2.0))
        yaws.append(yaws[-1] + np.deg2rad(rng.normal(0, 2.0)))
    poses = [_pose(x, y, a) for x, y, a in zip(xs, ys, yaws)]
    ref = T - 1
    inv = np.linalg.inv
    cur2ref = [(inv(poses[ref]) @ poses[k]).T for k in range(n_all)]     # row-vector convention
    ref2cur = [(inv(poses[k]) @ poses[ref]).T for k in range(n_all)]
    lidar2img = camera_matrices(img_hw, centre=(img_hw[1] / 2.0, img_hw[0] / 2.0 - 14.0))[:num_cams]
    assert lidar2img.shape[0] == num_cams, "extend CAM_YAWS_DEG for more cameras"
    metas = []
    for k in range(T):
        can_bus = np.zeros(18)
        if k > 0:
            can_bus[:3] = [xs[k] - xs[k - 1], ys[k] - ys[k - 1], 0.0]
            can_bus[-1] = np.rad2deg(yaws[k] - yaws[k - 1])
        can_bus[-2] = yaws[k]
        m = dict(lidar2img=[lidar2img[c] for c in range(num_cams)],
                 img_shape=[(img_hw[0], img_hw[1], 3)] * num_cams, can_bus=can_bus,
                 lidar2global_rotation=poses[k][:3, :3].copy(),
                 prev_bev_exists=bool(k > 0 or first_has_prev),
                 ref_lidar_to_cur_lidar=ref2cur[k], aug_param=None,
                 sample_idx=f"synthetic{seed:06d}", scene_token=f"scene{seed:06d}")
        metas.append(m)
    fut = [ref + j for j in range(future_frames + 1)]
    metas[ref].update(
        future2ref_lidar_transform=[cur2ref[k] for k in fut],
        ref2future_lidar_transform=[ref2cur[k] for k in fut],
        total_cur2ref_lidar_transform=cur2ref, total_ref2cur_lidar_transform=ref2cur,
        future_can_bus=[np.concatenate([[xs[k] - xs[ref], ys[k] - ys[ref], 0.0], np.zeros(14),
                                        [np.rad2deg(yaws[k] - yaws[ref])]]) for k in fut])
    pts = []
    for k in range(n_all):
        p = lidar_points(rng, rays_per_frame)
        pts.append(np.concatenate([p, rng.uniform(0, 1, (rays_per_frame, 1)).astype(np.float32),
                                   np.full((rays_per_frame, 1), k, np.float32)], 1))
    return metas, np.concatenate(pts).astype(np.float32)


def fpn_features(seed, T, num_cams=6, channels=256, shapes=FPN_SHAPES_NUSC, device="cpu", bs=1):
    import torch
    g = torch.Generator(device="cpu").manual_seed(seed)
    return [torch.randn(bs, T, num_cams, channels, h, w, generator=g).to(device) for h, w in shapes]


def msda_operands(seed, B, shapes, Nq, H=8, C=32, P=4, spread=0.05, device="cpu"):
    """Synthetic operands of the deformable-attention op at a given shape: value ~ N(0,1), sampling
    locations = a random reference point per query +- U(-spread, spread) clipped to [-0.1, 1.1] (exercises the zero
    padding), softmaxed weights.  -> value [B,Nv,H,C], shapes [L,2] i64, level_start_index [L] i64,
    loc [B,Nq,H,L,P,2], w [B,Nq,H,L,P]."""
    import torch
    g = torch.Generator().manual_seed(seed)
    L = len(shapes)
    Nv = sum(int(h) * int(w) for h, w in shapes)
    value = torch.randn(B, Nv, H, C, generator=g)
    ref = torch.rand(B, Nq, 1, 1, 1, 2, generator=g) * 1.2 - 0.1
    loc = (ref + (torch.rand(B, Nq, H, L, P, 2, generator=g) * 2 - 1) * spread).clamp(-0.1, 1.1).contiguous()
    w = torch.softmax(torch.randn(B, Nq, H, L * P, generator=g), -1).view(B, Nq, H, L, P).contiguous()
    sh = torch.tensor(shapes, dtype=torch.int64)
    sizes = torch.tensor([int(h) * int(w_) for h, w_ in shapes])
    lsi = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)[:-1]])
    return tuple(t.to(device) for t in (value, sh, lsi, loc, w))


def msda_operands_coherent(seed, B, shapes, Nq, H=8, C=32, P=4, px=2.0, device="cpu"):
    """Operands whose queries are spatially COHERENT, like the ones the model produces (neighbouring BEV queries
    project next to each other): query q sits `px` level-0 pixels right of query q-1 on a sqrt(Nq)-wide raster, every
    (head, level, point) adds its own fixed offset of up to +-1 % of the image.  Same layout as msda_operands."""
    import torch
    value, sh, lsi, _, w = msda_operands(seed, B, shapes, Nq, H=H, C=C, P=P)
    L = len(shapes)
    side = int(Nq ** 0.5)
    q = torch.arange(Nq)
    step = px / shapes[0][1]                                   # in normalised image units
    base = torch.stack([(q % side) * step + 0.1, (q // side) * step * 0.5 + 0.1], -1)   # [Nq, 2]
    g = torch.Generator().manual_seed(seed + 1)
    off = (torch.rand(1, 1, H, L, P, 2, generator=g) - 0.5) * 0.02
    loc = (base[None, :, None, None, None, :] + off).expand(B, Nq, H, L, P, 2).contiguous().clamp(0.0, 0.999)
    return tuple(t.to(device) for t in (value, sh, lsi, loc, w))


def dense_rays(Fn, Z, Y, X, device="cpu"):
    """end points of the dense-loss rays: the voxel centres of a (Y/4, X/4, Z/4) sub-grid per frame
    (dense_heads/vidar_head_base.py:606-630) -> (pts [Fn*n, 3] in voxel units, frame index [Fn*n])"""
    import torch
    from .plugin.utils.e2e_predictor_utils import get_bev_grids_3d
    v = get_bev_grids_3d(Y // 4, X // 4, Z // 4, bs=1, device=device)
    v = (v * v.new_tensor([X, Y, Z])).view(-1, 3)
    pts = torch.cat([v for _ in range(Fn)], 0)
    tix = torch.cat([torch.full((v.shape[0],), float(f), device=device) for f in range(Fn)], 0)
    return pts, tix
