"""CPU: adversarial / randomised ray sets -- oracle/dvr_oracle.c vs the reference kernels compiled
for the host (oracle/_ref).  Integer-aligned origins and end points, axis-parallel rays, rays that
start outside, graze corners or have zero length (NaN direction), tiny and non-cubic volumes."""
import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from oracle import dvr as O

ts = torch.from_numpy


def edge_case():
    Z, Y, X = 4, 9, 7
    rng = np.random.default_rng(0)
    sigma = rng.uniform(0, 1, (1, 2, Z, Y, X)).astype(np.float32)
    origin = np.array([[[3.0, 4.0, 2.0], [3.5, 4.25, 1.75]]], np.float32)     # integer-aligned / generic
    pts = [[6.0, 4.0, 2.0], [3.0, 8.0, 2.0], [3.0, 4.0, 0.0], [0.0, 0.0, 0.0], [7.0, 9.0, 4.0],   # axis / corners
           [3.0, 4.0, 2.0],                                                     # zero length
           [3.0 + 1e-6, 4.0, 2.0], [-5.0, 4.0, 2.0], [50.0, 60.0, 2.5], [3.0, 4.0, 40.0],
           [2.999999, 3.999999, 1.999999], [6.5, 0.5, 3.5], [0.5, 8.5, 0.5]]
    pts = np.array(pts + rng.uniform(-3, 12, (40, 3)).tolist(), np.float32)[None]
    tindex = np.tile(np.array([0, 1], np.float32), pts.shape[1] // 2 + 1)[None, :pts.shape[1]].copy()
    tindex[0, 5] = 0          # the zero-length ray uses the integer origin
    return sigma, origin, np.ascontiguousarray(pts), np.ascontiguousarray(tindex)


def same(a, b):
    return np.array_equal(a, b, equal_nan=True)


def test_edge_rays_match_reference_build(ref_modules):
    sigma, origin, points, tindex = edge_case()
    r = [x.numpy() for x in ref_modules("ref_dvxlr").render(ts(sigma), ts(origin), ts(points), ts(tindex))]
    o = O.dvxlr_render(sigma, origin, points, tindex)
    for a, b, nm in zip(r, o, ["pred", "gt", "dd", "idx"]):
        assert same(a, b), nm
    rf = ref_modules("ref_dvr").render_forward(ts(sigma), ts(origin), ts(points), ts(tindex), [2, 4, 9, 7], "test")
    of = O.render_forward(sigma, origin, points, tindex, "test")
    assert same(rf[0].numpy(), of[0]) and same(rf[1].numpy(), of[1])
    rr = ref_modules("ref_dvr").render(ts(sigma), ts(origin), ts(points), ts(tindex), "l2")
    orr = O.render(sigma, origin, points, tindex, "l2")
    assert same(rr[0].numpy(), orr[0]) and same(rr[1].numpy(), orr[1])


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 10_000), Z=st.integers(1, 6), Y=st.integers(1, 12), X=st.integers(1, 12),
       snap=st.booleans())
def test_random_volumes_match_reference_build(seed, Z, Y, X, snap):
    from oracle import build_ref
    if not build_ref.so_path("ref_dvxlr_v2").exists():
        pytest.skip("oracle/_ref not built")
    ref = build_ref.load("ref_dvxlr_v2")
    rng = np.random.default_rng(seed)
    sigma = rng.uniform(0, 2, (1, 1, Z, Y, X)).astype(np.float32)
    regul = rng.standard_normal(sigma.shape).astype(np.float32)
    origin = rng.uniform(-1, [X + 1, Y + 1, Z + 1], (1, 1, 3)).astype(np.float32)
    pts = rng.uniform(-4, [X + 4, Y + 4, Z + 4], (1, 24, 3)).astype(np.float32)
    if snap:                       # integer / half-integer coordinates: ties in the traversal order
        origin = np.round(origin * 2) / 2
        pts = np.round(pts * 2) / 2
        pts[0][(pts[0] == origin[0, 0]).all(1)] += 1.0
    tindex = np.where(rng.uniform(size=(1, 24)) < 0.1, -1.0, 0.0).astype(np.float32)
    r = [x.numpy() for x in ref.render_v2(ts(sigma), ts(origin), ts(pts), ts(tindex), ts(regul))]
    o = O.dvxlr_render(sigma, origin, pts, tindex, regul)
    for a, b, nm in zip(r, o, ["pred", "gt", "dd", "idx", "ray_pred", "indicator"]):
        assert same(a, b), (nm, seed, Z, Y, X, snap)
