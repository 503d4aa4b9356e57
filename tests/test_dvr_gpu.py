"""GPU parity: HIP dvr / dvxlr / dvxlr_v2 (through the C ABI) vs the CPU oracle and the golden
fixtures.  Index lists, pred/gt masks: bit-exact.  Values: fp64 summation-order tolerance."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import dvr as O
from dvr_cases import CASES, case, expand

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


PAD_MODE = {"plain": 0, "ranked": 0, "ranked-prefill": 1, "step-parallel": 1}


@pytest.fixture(autouse=True, params=list(PAD_MODE))
def march_order(request):
    """every test runs under each launch variant: the lane-per-ray kernels plain, with the rays of each workgroup
    ranked by estimated length (what large launches do), and with the device-fill-first padding of dvxlr.render
    (mode 1); and the step-parallel traversal (csrc/dvr_par.h: what launches of up to 98 304 rays use by default)
    -- results must not depend on it."""
    from vidar_amd._lib import lib
    prev = lib().vidar_dvr_set_sort_min_waves(1 << 30 if request.param == "plain" else 0)
    prev_pad = lib().vidar_dvxlr_set_pad_mode(PAD_MODE[request.param])
    prev_trav = lib().vidar_dvr_set_traversal(1 if request.param == "step-parallel" else 0)
    yield request.param
    lib().vidar_dvr_set_sort_min_waves(prev)
    lib().vidar_dvxlr_set_pad_mode(prev_pad)
    lib().vidar_dvr_set_traversal(prev_trav)


def dev(*arrs):
    return [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in arrs]


def close(a, b, rtol=2e-5, atol_rel=2e-6):
    a = a.cpu().numpy() if isinstance(a, torch.Tensor) else a
    scale = max(1.0, float(np.abs(b).max())) if b.size else 1.0
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol_rel * scale)


@pytest.mark.parametrize("name", CASES)
def test_dvxlr_render(name):
    from vidar_amd.third_lib import dvxlr
    sigma, origin, points, tindex = case(name)
    o = O.dvxlr_render(sigma, origin, points, tindex)
    pred, gt, dd, idx = dvxlr.render(*dev(sigma, origin, points, tindex))
    torch.cuda.synchronize()
    assert np.array_equal(idx.cpu().numpy(), o[3]), "voxel index lists must be bit-exact"
    assert np.array_equal(pred.cpu().numpy() < 0, o[0] < 0)
    assert np.array_equal(gt.cpu().numpy(), o[1]), "gt_dist is pure traversal arithmetic: bit-exact"
    close(pred, o[0]); close(dd, o[2])


@pytest.mark.parametrize("name", CASES)
def test_dvxlr_v2_render_and_scatter(name):
    from vidar_amd.third_lib import dvxlr_v2
    sigma, origin, points, tindex = case(name)
    regul = np.random.default_rng(7).standard_normal(sigma.shape).astype(np.float32)
    o = O.dvxlr_render(sigma, origin, points, tindex, regul)
    d = dev(sigma, origin, points, tindex, regul)
    pred, gt, dd, idx, rp, ind = dvxlr_v2.render_v2(*d)
    assert np.array_equal(idx.cpu().numpy(), o[3])
    assert np.array_equal(ind.cpu().numpy(), o[5])
    assert np.array_equal(rp.cpu().numpy(), o[4])
    assert np.array_equal(gt.cpu().numpy(), o[1])
    close(pred, o[0]); close(dd, o[2])
    if dd.numel() == 0:
        return
    rng = np.random.default_rng(8)
    em = rng.standard_normal(o[0].shape).astype(np.float32)[..., None] * o[2]
    grp = rng.standard_normal(o[4].shape).astype(np.float32)
    og = O.dvxlr_get_grad_sigma(em, o[3], tindex, sigma.shape, o[5], grp)
    g, g2 = dvxlr_v2.get_grad_sigma_v2(*dev(em, o[3], tindex), d[0], *dev(o[5], grp))
    close(g, og[0], rtol=1e-4, atol_rel=1e-5); close(g2, og[1], rtol=1e-4, atol_rel=1e-5)


@pytest.mark.parametrize("name", CASES)
def test_dvxlr_get_grad_sigma(name):
    from vidar_amd.third_lib import dvxlr
    sigma, origin, points, tindex = case(name)
    pred, gt, dd, idx = O.dvxlr_render(sigma, origin, points, tindex)
    if dd.size == 0:
        return
    em = np.random.default_rng(9).standard_normal(pred.shape).astype(np.float32)[..., None] * dd
    og = O.dvxlr_get_grad_sigma(em, idx, tindex, sigma.shape)
    g = dvxlr.get_grad_sigma(*dev(em, idx, tindex, sigma))[0]
    close(g, og, rtol=1e-4, atol_rel=1e-5)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("phase", ["train", "test"])
def test_dvr_render_forward(name, phase):
    from vidar_amd.third_lib import dvr
    sigma, origin, points, tindex = case(name)
    o = O.render_forward(sigma, origin, points, tindex, phase)
    pred, gt = dvr.render_forward(*dev(sigma, origin, points, tindex), list(sigma.shape[1:]), phase)
    assert np.array_equal(gt.cpu().numpy(), o[1])
    close(pred, o[0])


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("loss", ["l1", "l2", "absrel", "bce"])
def test_dvr_render(name, loss):
    from vidar_amd.third_lib import dvr
    sigma, origin, points, tindex = case(name)
    o = O.render(sigma, origin, points, tindex, loss)
    pred, gt, grad = dvr.render(*dev(sigma, origin, points, tindex), loss)
    assert np.array_equal(gt.cpu().numpy(), o[1])
    close(pred, o[0])
    if loss in ("l1", "bce", "absrel"):
        # sign(pred-gt) can flip where |pred-gt| is at rounding level; exclude those rays' voxels
        # by comparing with a tolerance on the aggregate instead
        close(grad.sum(), np.float32(o[2].sum(dtype=np.float64)), rtol=1e-3, atol_rel=1e-3)
    close(grad, o[2], rtol=1e-3, atol_rel=1e-4)


@pytest.mark.parametrize("name", CASES)
def test_init(name):
    from vidar_amd.third_lib import dvr, dvxlr
    sigma, origin, points, tindex = case(name)
    grid = [3, *sigma.shape[2:]]
    ref = O.init(points, tindex, grid)
    for m in (dvr, dvxlr):
        occ = m.init(*dev(points, tindex), grid)
        assert np.array_equal(occ.cpu().numpy(), ref)


@pytest.mark.parametrize("name", ["two_frames", "static_sigma", "small_grid"])
def test_golden_from_reference_kernels(name):
    """HIP vs fixtures produced by the reference's own kernels (compiled for host)."""
    from vidar_amd.third_lib import dvxlr_v2
    g = np.load(GOLD / f"dvr_family_{name}.npz")
    sigma, origin, points, tindex = case(name)
    regul = np.random.default_rng(7).standard_normal(sigma.shape).astype(np.float32)
    pred, gt, dd, idx, rp, ind = dvxlr_v2.render_v2(*dev(sigma, origin, points, tindex, regul))
    gdd, gidx, grp, gind = expand(g["count"], g["dd"], g["idx"],
                                  [(g["ray_pred"], 0.0), (g["indicator"], -1.0)])
    assert np.array_equal(idx.cpu().numpy(), gidx)
    assert np.array_equal(ind.cpu().numpy(), gind) and np.array_equal(rp.cpu().numpy(), grp)
    assert np.array_equal(gt.cpu().numpy(), g["gt"])
    close(pred, g["pred"]); close(dd, gdd)


def test_errors():
    from vidar_amd.third_lib import dvr, dvxlr
    sigma, origin, points, tindex = case("small_grid")
    d = dev(sigma, origin, points, tindex)
    with pytest.raises(RuntimeError):
        dvxlr.render(d[0].cpu(), *d[1:])                 # CHECK_CUDA
    with pytest.raises(RuntimeError):
        dvxlr.render(d[0].transpose(3, 4), *d[1:])       # CHECK_CONTIGUOUS
    with pytest.raises(ValueError):
        dvr.render(*d, "huber")
    with pytest.raises(ValueError):
        dvr.render_forward(*d, [1, 4, 24, 20], "val")


def test_full_size_properties():
    """BASELINE-size ray set (30k rays, 16x200x200): properties that need no oracle."""
    from vidar_amd.third_lib import dvxlr
    from vidar_amd.synthetic import ray_set
    sigma, origin, points, tindex = ray_set(seed=11, N=1, T=1, rays_per_frame=30000)
    d = dev(sigma, origin, points, tindex)
    pred, gt, dd, idx = dvxlr.render(*d)
    hit = pred >= 0
    assert hit.float().mean() > 0.9
    live = (idx != 0).any(-1)
    cnt = live.sum(-1)
    assert int(cnt.max()) <= 420                     # ray cannot cross more voxels than X+Y+Z
    # consecutive live samples are distinct voxels that differ by <=1 per axis (merged path)
    step = (idx[..., 1:, :] - idx[..., :-1, :]).abs()
    both = live[..., 1:] & live[..., :-1]
    assert float(step[both].max()) <= 1.0 and bool((step[both].sum(-1) > 0).all())
    assert bool((pred[hit] > 0).all()) and bool((pred[hit] < 500).all())
    assert bool((dd <= 1e-6).all())                  # more density can only shorten the ray
    # linearity of the scatter: get_grad_sigma(a*em) == a*get_grad_sigma(em)
    em = dd * 0.5
    g1 = dvxlr.get_grad_sigma(em, idx, d[3], d[0])[0]
    g2 = dvxlr.get_grad_sigma(em * 2, idx, d[3], d[0])[0]
    torch.testing.assert_close(g2, g1 * 2, rtol=1e-4, atol=1e-5)
    assert abs(float(g1.sum()) - float(em.double().sum())) <= 1e-3 * float(em.double().abs().sum())


def _nan_equal(a, b):
    return np.array_equal(a, b, equal_nan=True)


def test_adversarial_rays_bit_exact():
    """integer-aligned origins / end points, axis-parallel, outside starts, zero length (NaN dir)."""
    from test_oracle_dvr_edge import edge_case
    from vidar_amd.third_lib import dvxlr, dvr
    sigma, origin, points, tindex = edge_case()
    o = O.dvxlr_render(sigma, origin, points, tindex)
    pred, gt, dd, idx = dvxlr.render(*dev(sigma, origin, points, tindex))
    assert _nan_equal(idx.cpu().numpy(), o[3]) and _nan_equal(gt.cpu().numpy(), o[1])
    ok = np.isfinite(o[0])
    close(pred.cpu().numpy()[ok], o[0][ok])
    assert np.array_equal(np.isnan(pred.cpu().numpy()), np.isnan(o[0]))
    ddh = dd.cpu().numpy()
    assert np.array_equal(np.isnan(ddh), np.isnan(o[2]))       # NaN rays poison their rows like the reference
    fin = np.isfinite(o[2])
    close(ddh[fin], o[2][fin])
    of = O.render_forward(sigma, origin, points, tindex, "test")
    pf, gf = dvr.render_forward(*dev(sigma, origin, points, tindex), [2, 4, 9, 7], "test")
    assert _nan_equal(gf.cpu().numpy(), of[1])
    close(pf.cpu().numpy()[ok], of[0][ok])


@pytest.mark.parametrize("seed", range(12))
def test_random_snapped_volumes_bit_exact(seed):
    """half-integer coordinates produce exact ties in the traversal order; small odd volumes."""
    from vidar_amd.third_lib import dvxlr_v2
    rng = np.random.default_rng(seed)
    Z, Y, X = rng.integers(1, 7), rng.integers(1, 13), rng.integers(1, 13)
    sigma = rng.uniform(0, 2, (1, 1, Z, Y, X)).astype(np.float32)
    regul = rng.standard_normal(sigma.shape).astype(np.float32)
    origin = (np.round(rng.uniform(-1, [X + 1, Y + 1, Z + 1], (1, 1, 3)) * 2) / 2).astype(np.float32)
    pts = (np.round(rng.uniform(-4, [X + 4, Y + 4, Z + 4], (1, 40, 3)) * 2) / 2).astype(np.float32)
    pts[0][(pts[0] == origin[0, 0]).all(1)] += 1.0
    tindex = np.where(rng.uniform(size=(1, 40)) < 0.1, -1.0, 0.0).astype(np.float32)
    o = O.dvxlr_render(sigma, origin, pts, tindex, regul)
    g = dvxlr_v2.render_v2(*dev(sigma, origin, pts, tindex, regul))
    assert np.array_equal(g[3].cpu().numpy(), o[3]) and np.array_equal(g[5].cpu().numpy(), o[5])
    assert np.array_equal(g[1].cpu().numpy(), o[1]) and np.array_equal(g[4].cpu().numpy(), o[4])
    close(g[0], o[0]); close(g[2], o[2])


def test_ranked_march_is_bitwise_invisible_at_full_size():
    """BASELINE size (5 frames x 30k rays): ranked and plain launches give identical bytes, every
    output byte is written by the call (poisoned buffers), counts agree with render_forward."""
    from vidar_amd._lib import lib
    from vidar_amd.synthetic import ray_set
    from vidar_amd.third_lib import dvr, dvxlr, dvxlr_v2
    sigma, origin, points, tindex = dev(*ray_set(seed=3, N=1, T=5, rays_per_frame=30000, pad=111))
    outs = {}
    for mode, thr in (("plain", 1 << 30), ("ranked", 0)):
        lib().vidar_dvr_set_sort_min_waves(thr)
        torch.empty(3 * 1024 ** 3 // 4, device="cuda").fill_(float("nan"))   # poison the allocator's blocks
        torch.cuda.synchronize()
        outs[mode] = (dvxlr.render(sigma, origin, points, tindex),
                      dvxlr_v2.render_v2(sigma, origin, points, tindex, sigma),
                      dvr.render_forward(sigma, origin, points, tindex, [5, 16, 200, 200], "train"),
                      dvr.render(sigma, origin, points, tindex, "l2")[:2])
    for a, b in zip(outs["plain"], outs["ranked"]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    pred, gt, dd, idx = outs["ranked"][0]
    assert torch.isfinite(dd).all() and torch.isfinite(idx).all()
    valid = tindex >= 0
    assert bool((pred[~valid] == -1).all()) and bool((dd[~valid] == 0).all())
