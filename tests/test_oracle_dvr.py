"""CPU: oracle/dvr_oracle.c  vs  the reference's own kernels compiled for the host (oracle/_ref)
and vs the committed golden fixtures (tests/golden/dvr_family_*.npz, made by
tests/golden/make_dvr_golden.py from oracle/_ref)."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import dvr as O
from dvr_cases import CASES, case, expand

GOLD = Path(__file__).parent / "golden"
ts = torch.from_numpy


@pytest.mark.parametrize("name", CASES)
def test_dvxlr_render_matches_reference_build(name, ref_modules):
    ref = ref_modules("ref_dvxlr")
    sigma, origin, points, tindex = case(name)
    r = [x.numpy() for x in ref.render(ts(sigma), ts(origin), ts(points), ts(tindex))]
    o = O.dvxlr_render(sigma, origin, points, tindex)
    for a, b, nm in zip(r, o, ["pred_dist", "gt_dist", "dd_dsigma", "indices"]):
        assert a.shape == b.shape, nm
        assert np.array_equal(a, b), f"{nm} differs (bit-exact expected)"


@pytest.mark.parametrize("name", CASES)
def test_dvxlr_v2_matches_reference_build(name, ref_modules):
    ref = ref_modules("ref_dvxlr_v2")
    sigma, origin, points, tindex = case(name)
    regul = np.random.default_rng(7).standard_normal(sigma.shape).astype(np.float32)
    r = [x.numpy() for x in ref.render_v2(ts(sigma), ts(origin), ts(points), ts(tindex), ts(regul))]
    o = O.dvxlr_render(sigma, origin, points, tindex, regul)
    for a, b, nm in zip(r, o, ["pred", "gt", "dd", "idx", "ray_pred", "indicator"]):
        assert np.array_equal(a, b), nm
    # scatter
    rng = np.random.default_rng(8)
    gp = rng.standard_normal(o[0].shape).astype(np.float32)
    em = gp[..., None] * o[2]
    grp = rng.standard_normal(o[4].shape).astype(np.float32)
    if em.size == 0:
        return
    rg = ref.get_grad_sigma_v2(ts(em), ts(o[3]), ts(tindex), ts(sigma), ts(o[5]), ts(grp))
    og = O.dvxlr_get_grad_sigma(em, o[3], tindex, sigma.shape, o[5], grp)
    for a, b in zip(rg, og):
        a = a.numpy()
        assert np.allclose(a, b, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(a).max()))


@pytest.mark.parametrize("name", CASES)
def test_dvxlr_get_grad_sigma_matches_reference_build(name, ref_modules):
    ref = ref_modules("ref_dvxlr")
    sigma, origin, points, tindex = case(name)
    pred, gt, dd, idx = O.dvxlr_render(sigma, origin, points, tindex)
    if dd.size == 0:
        return
    em = np.random.default_rng(9).standard_normal(pred.shape).astype(np.float32)[..., None] * dd
    rg = ref.get_grad_sigma(ts(em), ts(idx), ts(tindex), ts(sigma))[0].numpy()
    og = O.dvxlr_get_grad_sigma(em, idx, tindex, sigma.shape)
    # fp32 atomicAdd order (reference) vs fp64 sequential (oracle)
    assert np.allclose(rg, og, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(rg).max()))


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("phase", ["train", "test"])
def test_dvr_render_forward_matches_reference_build(name, phase, ref_modules):
    ref = ref_modules("ref_dvr")
    sigma, origin, points, tindex = case(name)
    grid = list(sigma.shape[1:])
    r = ref.render_forward(ts(sigma), ts(origin), ts(points), ts(tindex), grid, phase)
    o = O.render_forward(sigma, origin, points, tindex, phase)
    for a, b in zip(r, o):
        assert np.array_equal(a.numpy(), b)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("loss", ["l1", "l2", "absrel", "bce"])
def test_dvr_render_matches_reference_build(name, loss, ref_modules):
    ref = ref_modules("ref_dvr")
    sigma, origin, points, tindex = case(name)
    r = ref.render(ts(sigma), ts(origin), ts(points), ts(tindex), loss)
    o = O.render(sigma, origin, points, tindex, loss)
    assert np.array_equal(r[0].numpy(), o[0]) and np.array_equal(r[1].numpy(), o[1])
    rg = r[2].numpy()
    assert np.allclose(rg, o[2], rtol=1e-4, atol=1e-5 * max(1.0, np.abs(rg).max()))


@pytest.mark.parametrize("name", CASES)
def test_init_matches_reference_build(name, ref_modules):
    ref = ref_modules("ref_dvr")
    sigma, origin, points, tindex = case(name)
    grid = [3, *sigma.shape[2:]]
    r = ref.init(ts(points), ts(tindex), grid).numpy()
    assert np.array_equal(r, O.init(points, tindex, grid))


# ---- golden fixtures: run everywhere (GPU box included), no reference needed ------------------
@pytest.mark.parametrize("name", ["two_frames", "static_sigma", "small_grid"])
def test_oracle_matches_golden(name):
    g = np.load(GOLD / f"dvr_family_{name}.npz")
    sigma, origin, points, tindex = case(name)
    regul = np.random.default_rng(7).standard_normal(sigma.shape).astype(np.float32)
    o = O.dvxlr_render(sigma, origin, points, tindex, regul)
    dd, idx, rp, ind = expand(g["count"], g["dd"], g["idx"], [(g["ray_pred"], 0.0), (g["indicator"], -1.0)])
    assert np.array_equal(o[0], g["pred"]) and np.array_equal(o[1], g["gt"])
    assert np.array_equal(o[3], idx), "voxel index lists must be bit-exact"
    assert np.array_equal(o[2], dd) and np.array_equal(o[4], rp) and np.array_equal(o[5], ind)
    f = O.render_forward(sigma, origin, points, tindex, "train")
    assert np.array_equal(f[0], g["fwd_pred"]) and np.array_equal(f[1], g["fwd_gt"])
    r = O.render(sigma, origin, points, tindex, "l2")
    assert np.array_equal(r[0], g["dvr_pred"]) and np.array_equal(r[1], g["dvr_gt"])
    assert np.allclose(r[2].sum(axis=(2, 3)), g["dvr_grad_zsum"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("kw", [dict(T=1, rays_per_frame=30000),
                                dict(T=10, rays_per_frame=27000, origin_jitter=30 * 0.512)],
                         ids=["1x30k", "c4_10x27k"])
def test_oracle_matches_reference_build_at_baseline_size(kw, ref_modules):
    """the ray sets of tests/test_fullsize_parity_gpu.py (one 30 000-ray frame; the OpenScene stress
    shape: 10 frames, 270 000 rays, origins up to 30 voxels off centre): the C restatement the GPU is
    compared with is bit-exact with the reference's own dvxlr kernel (host build) at that size too."""
    from vidar_amd.synthetic import ray_set
    ref = ref_modules("ref_dvxlr")
    sigma, origin, points, tindex = ray_set(seed=21, N=1, **kw)
    r = ref.render(ts(sigma), ts(origin), ts(points), ts(tindex))
    o = O.dvxlr_render(sigma, origin, points, tindex)
    for a, b, nm in zip(r, o, ["pred_dist", "gt_dist", "dd_dsigma", "indices"]):
        assert torch.equal(a, ts(b)), f"{nm} differs (bit-exact expected)"
    ref2 = ref_modules("ref_dvr")
    rf = ref2.render_forward(ts(sigma), ts(origin), ts(points), ts(tindex), list(sigma.shape[1:]), "train")
    of = O.render_forward(sigma, origin, points, tindex, "train")
    for a, b in zip(rf, of):
        assert torch.equal(a, ts(b))
