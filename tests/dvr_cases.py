"""Shared ray-set cases + the compact golden format for the dvr family."""
import numpy as np

from vidar_amd.synthetic import ray_set

L = 1026


def case(name):
    if name == "two_frames":      # jittered origins, padded rays, a few adversarial rays
        sigma, origin, points, tindex = ray_set(seed=1, N=2, T=2, rays_per_frame=250, pad=37,
                                                origin_jitter=20.0)
        o = origin[0, 0]
        points[0, :6] = [[250, 100, 8], [100.5, 100.5, 30], [10, 10, 1], [o[0], o[1], o[2] + 3],
                         [o[0] + 40, o[1], o[2]], [o[0], o[1] - 25, o[2]]]
        tindex[0, :6] = 0
        origin[1, 1] = [-5.5, 230.25, 7.0]          # frame whose origin lies outside the volume
    elif name == "static_sigma":  # T == 1 sigma shared by 3 ray frames (dvxlr.cu:197)
        sigma, origin, points, tindex = ray_set(seed=2, N=1, T=3, rays_per_frame=150, pad=5,
                                                origin_jitter=5.0, sigma_T=1)
    elif name == "small_grid":
        sigma, origin, points, tindex = ray_set(seed=3, N=1, T=1, rays_per_frame=300,
                                                grid=(4, 24, 20), origin_jitter=30.0)
    elif name == "empty":
        sigma, origin, points, tindex = ray_set(seed=4, N=1, T=1, rays_per_frame=0, pad=0)
    elif name == "all_padded":
        sigma, origin, points, tindex = ray_set(seed=5, N=1, T=1, rays_per_frame=0, pad=70)
    else:
        raise KeyError(name)
    return sigma, origin, points, tindex


CASES = ["two_frames", "static_sigma", "small_grid", "empty", "all_padded"]


def compact(dd, idx, extra=()):
    """[N,M,L(,3)] padded rows -> (count [N,M], concatenated live prefixes)."""
    live = (idx != 0).any(-1) | (dd != 0)
    # live prefix length = last live slot + 1 (voxel (0,0,0) with dd==0 can only be a tail slot
    # or a genuine sample at the volume corner; rows are compared in full elsewhere)
    cnt = np.where(live.any(-1), L - np.argmax(live[..., ::-1], -1), 0).astype(np.int32)
    m = np.arange(L)[None, None, :] < cnt[..., None]
    out = [cnt, dd[m], idx[m].astype(np.int16)]
    for e in extra:
        out.append(e[m])
    return out


def expand(cnt, dd_c, idx_c, fill_extra=()):
    N, M = cnt.shape
    m = np.arange(L)[None, None, :] < cnt[..., None]
    dd = np.zeros((N, M, L), np.float32); dd[m] = dd_c
    idx = np.zeros((N, M, L, 3), np.float32); idx[m] = idx_c.astype(np.float32)
    extras = []
    for vals, fill in fill_extra:
        e = np.full((N, M, L), fill, np.float32); e[m] = vals
        extras.append(e)
    return (dd, idx, *extras)
