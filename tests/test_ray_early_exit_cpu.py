"""The invariant the staged early exit of csrc/ray_march.hip (VIDAR_RAY_EARLY_EXIT) rests on, checked in the kernels'
own fp32 arithmetic: along a ray the waypoints inside the open volume form ONE run of consecutive k, so after a pass
(a block of 32 or 64 consecutive waypoints) without a live waypoint that follows a pass with one, no later waypoint
is live.  Also reports how much of the 512-waypoint loop the exit removes at the recipe's geometry."""
import numpy as np
import pytest

K = 512
f32 = np.float32


def live_mask(origin, pts, dims, step):
    """!Tri.masked of make_tri for every (ray, k): the operation order of load_ray / waypoint / make_tri, in fp32."""
    X, Y, Z = (f32(d) for d in dims)
    o, p = origin.astype(f32), pts.astype(f32)
    r = p - o
    n = np.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1] + r[:, 2] * r[:, 2], dtype=f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        d = r / n[:, None]
    dist = (np.arange(K, dtype=f32) + f32(0.5)) * f32(step)                    # (k + 0.5f) * step
    s = o[:, None, :] + d[:, None, :] * dist[None, :, None]                    # o + d * dist, one rounding each
    with np.errstate(invalid="ignore"):
        g = s / np.array([X, Y, Z], f32) * f32(2) - f32(1)
        out = (g <= -1) | (g >= 1) | np.isnan(g)
    return ~out.any(-1)


def rays(seed, n, dims, spread):
    rng = np.random.default_rng(seed)
    X, Y, Z = dims
    origin = np.stack([rng.uniform(-spread, X + spread, n), rng.uniform(-spread, Y + spread, n),
                       rng.uniform(-spread / 4, Z + spread / 4, n)], -1)
    pts = np.stack([rng.uniform(-60, X + 60, n), rng.uniform(-60, Y + 60, n), rng.uniform(-8, Z + 8, n)], -1)
    # grazing and axis-aligned directions, end points on faces, zero-length rays
    q = n // 8
    pts[:q, 2] = origin[:q, 2]                                   # horizontal
    pts[q:2 * q, 0] = origin[q:2 * q, 0]                         # in a y-z plane
    pts[2 * q:3 * q] = np.round(pts[2 * q:3 * q])                # integer end points
    origin[3 * q:4 * q] = np.round(origin[3 * q:4 * q] * 2) / 2  # half-integer origins
    pts[4 * q:4 * q + 8] = origin[4 * q:4 * q + 8]               # zero length -> NaN direction -> all masked
    return origin.astype(f32), pts.astype(f32)


@pytest.mark.parametrize("dims,step,spread", [((200, 200, 16), 1.0, 0.0), ((200, 200, 16), 1.0, 40.0),
                                              ((24, 24, 4), 1.0, 6.0), ((200, 200, 16), 0.5, 20.0),
                                              ((50, 50, 16), 2.0, 10.0)])
def test_live_waypoints_form_one_run(dims, step, spread):
    origin, pts = rays(3, 40000, dims, spread)
    live = live_mask(origin, pts, dims, step)
    edges = np.diff(np.concatenate([np.zeros((len(live), 1), np.int8), live.astype(np.int8),
                                    np.zeros((len(live), 1), np.int8)], 1), axis=1)
    assert ((edges == 1).sum(1) <= 1).all()                      # at most one rising edge per ray: one run
    for block in (32, 64):                                       # the backward / forward pass sizes
        b = live.reshape(len(live), K // block, block).any(-1)
        seen = np.maximum.accumulate(b, 1)
        stop = seen & ~b                                         # passes after which the kernels leave the loop
        first_stop = np.where(stop.any(1), stop.argmax(1), K // block)
        later = np.arange(K // block)[None, :] > first_stop[:, None]
        assert not (b & later).any()                             # nothing live is ever skipped


def test_share_of_the_loop_the_exit_removes_at_the_recipe_geometry():
    """origin near the centre of the 200 x 200 x 16 volume, end points up to 70 voxels out: the run is short"""
    rng = np.random.default_rng(0)
    n = 30000
    origin = np.tile(np.array([[100.0, 100.0, 6.0]], f32), (n, 1)) + rng.normal(0, 0.5, (n, 3)).astype(f32)
    ang = rng.uniform(0, 2 * np.pi, n)
    rad = rng.uniform(3, 70, n)
    pts = np.stack([100 + rad * np.cos(ang), 100 + rad * np.sin(ang), rng.uniform(0.5, 15.5, n)], -1).astype(f32)
    live = live_mask(origin, pts, (200, 200, 16), 1.0)
    assert 30 < live.sum(1).mean() < 200
    b = live.reshape(n, K // 32, 32).any(-1)
    executed = np.minimum(b.sum(1) + 1, K // 32)                 # live passes + the one that detects the end
    assert executed.mean() < 0.5 * (K // 32)                     # more than half of the passes disappear
