"""vidar_amd/csrc/dvr_march.h (the per-lane traversal + integrator of every dvr / dvxlr kernel)
compiled for the host and compared with the oracle: bit-exact voxel lists, counts and dt, values to
fp32 round-off.  Lets the traversal logic be checked without a GPU; the GPU tests remain the parity
tests proper (tests/test_dvr_gpu.py)."""
import ctypes
import subprocess
from pathlib import Path

import numpy as np
import pytest

from oracle import dvr as O
from dvr_cases import CASES, case
from test_oracle_dvr_edge import edge_case as adversarial_case

ROOT = Path(__file__).resolve().parents[1]
L = 1026


# the header as shipped: a sample's density is consumed one commit later (same arithmetic in the same order, see
# dvr_march.h; on the GPU the load gets a whole traversal step to arrive: dvxlr.render 0.254 -> 0.225 ms at 30 000 rays)
# "step_parallel": the same tests with every traversal in its step-parallel form (csrc/dvr_par.h: independent per-axis
# chains, comparison-only merge, rounded-path chains) -- the arithmetic of the round-5 dvr kernels
BUILDS = {"default": [], "step_parallel": ["-DVIDAR_MARCH_PAR"]}


def _build(tmp_path_factory, flags):
    so = tmp_path_factory.mktemp("march_host") / "libmarch_host.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", *flags,
                    f"-I{ROOT / 'vidar_amd' / 'csrc'}", str(ROOT / "tests" / "march_host.cpp"),
                    "-o", str(so)], check=True)
    return ctypes.CDLL(str(so))


@pytest.fixture(scope="module", params=list(BUILDS), ids=list(BUILDS))
def host(request, tmp_path_factory):
    return _build(tmp_path_factory, BUILDS[request.param])


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _prep(sigma, origin, points, tindex):
    f = lambda a: np.ascontiguousarray(a, np.float32)
    sigma, origin, points, tindex = f(sigma), f(origin), f(points), f(tindex)
    N, T, Z, Y, X = sigma.shape
    return sigma, origin, points, tindex, (N, points.shape[1], T, origin.shape[1], Z, Y, X)


def snapped_case(seed):
    """half-integer coordinates produce exact ties in the traversal order; small odd volumes."""
    rng = np.random.default_rng(seed)
    Z, Y, X = rng.integers(1, 7), rng.integers(1, 13), rng.integers(1, 13)
    sigma = rng.uniform(0, 2, (1, 1, Z, Y, X)).astype(np.float32)
    origin = (np.round(rng.uniform(-1, [X + 1, Y + 1, Z + 1], (1, 1, 3)) * 2) / 2).astype(np.float32)
    pts = (np.round(rng.uniform(-4, [X + 4, Y + 4, Z + 4], (1, 40, 3)) * 2) / 2).astype(np.float32)
    pts[0][(pts[0] == origin[0, 0]).all(1)] += 1.0
    tindex = np.where(rng.uniform(size=(1, 40)) < 0.1, -1.0, 0.0).astype(np.float32)
    return sigma, origin, pts, tindex


def _cases():
    out = [(n, lambda n=n: case(n)) for n in CASES]
    out.append(("adversarial", adversarial_case))
    out += [(f"snapped{s}", lambda s=s: snapped_case(s)) for s in range(100, 124)]
    return out


@pytest.mark.parametrize("name,make", _cases(), ids=[c[0] for c in _cases()])
@pytest.mark.parametrize("v2", [False, True])
def test_dvxlr_rows(host, name, make, v2):
    sigma, origin, points, tindex, dims = _prep(*make())
    N, M = dims[0], dims[1]
    reg = (np.random.default_rng(7).random(sigma.shape, dtype=np.float32)) if v2 else None
    ref = O.dvxlr_render(sigma, origin, points, tindex, reg)
    pred = np.empty((N, M), np.float32); gt = np.empty((N, M), np.float32)
    poison = lambda *shape: np.full(shape, 12345.0, np.float32)     # the call owns every byte
    dd = poison(N, M, L); idx = poison(N, M, L, 3); rp = poison(N, M, L); ind = poison(N, M, L)
    est = np.zeros((N, M), np.int32)
    rc = host.host_dvxlr_render(_p(sigma), _p(reg) if v2 else None, _p(origin), _p(points), _p(tindex),
                                _p(pred), _p(gt), _p(dd), _p(idx), _p(rp), _p(ind), _p(est), *dims)
    assert rc == 0
    np.testing.assert_array_equal(idx, ref[3])                     # voxel lists: bit-exact
    np.testing.assert_allclose(dd, ref[2], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(pred, ref[0], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(gt, ref[1])
    if v2:
        np.testing.assert_array_equal(rp, ref[4])
        np.testing.assert_array_equal(ind, ref[5])
    if M:
        # the ranking key tracks the true number of samples (only a heuristic, but it should be a
        # useful one): rank correlation on the valid rays
        cnt = (ref[3] != 0).any(-1).sum(-1).ravel().astype(np.float64)
        e = est.ravel().astype(np.float64)
        ok = cnt > 0
        if ok.sum() > 100 and cnt[ok].std() > 0 and e[ok].std() > 0:
            assert np.corrcoef(cnt[ok], e[ok])[0, 1] > 0.8


@pytest.mark.parametrize("name,make", _cases(), ids=[c[0] for c in _cases()])
@pytest.mark.parametrize("phase", ["test", "train"])
def test_dvr_render_forward(host, name, make, phase):
    sigma, origin, points, tindex, dims = _prep(*make())
    N, M = dims[0], dims[1]
    ref = O.render_forward(sigma, origin, points, tindex, phase)
    pred = np.empty((N, M), np.float32); gt = np.empty((N, M), np.float32)
    assert host.host_dvr_render_forward(_p(sigma), _p(origin), _p(points), _p(tindex), _p(pred), _p(gt),
                                        *dims, O.PHASE[phase]) == 0
    np.testing.assert_allclose(pred, ref[0], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(gt, ref[1])


@pytest.mark.parametrize("name,make", _cases(), ids=[c[0] for c in _cases()])
@pytest.mark.parametrize("loss", ["l1", "l2", "absrel"])
def test_dvr_render(host, name, make, loss):
    sigma, origin, points, tindex, dims = _prep(*make())
    N, M = dims[0], dims[1]
    ref = O.render(sigma, origin, points, tindex, loss)
    pred = np.empty((N, M), np.float32); gt = np.empty((N, M), np.float32)
    grad = np.empty(sigma.shape, np.float64)
    assert host.host_dvr_render(_p(sigma), _p(origin), _p(points), _p(tindex), _p(pred), _p(gt), _p(grad),
                                *dims, O.LOSS[loss]) == 0
    np.testing.assert_allclose(pred, ref[0], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(gt, ref[1])
    scale = max(1.0, float(np.abs(ref[2]).max()))
    np.testing.assert_allclose(grad, ref[2], rtol=1e-4, atol=1e-4 * scale)


def test_step_parallel_traversal_full_size_frame(tmp_path_factory):
    """One BASELINE frame (30 000 rays from the grid centre) and a jittered-origin multi-frame set through the
    step-parallel traversal (csrc/dvr_par.h) on the host: voxel lists and gt_dist bit-equal to the oracle, and the
    rays really take the parallel form (the sequential march is only the fallback for irregular rays)."""
    from vidar_amd.synthetic import ray_set
    host = _build(tmp_path_factory, BUILDS["step_parallel"])
    for kw in (dict(seed=0, N=1, T=1, rays_per_frame=30000),
               dict(seed=9, N=1, T=10, rays_per_frame=1500, origin_jitter=30.0)):
        sigma, origin, points, tindex, dims = _prep(*ray_set(**kw))
        N, M = dims[0], dims[1]
        ref = O.dvxlr_render(sigma, origin, points, tindex, None)
        pred = np.empty((N, M), np.float32); gt = np.empty((N, M), np.float32)
        dd = np.full((N, M, L), 7.0, np.float32); idx = np.full((N, M, L, 3), 7.0, np.float32)
        est = np.zeros((N, M), np.int32)
        before = ctypes.c_int(0); r0 = host.host_par_regular_count(ctypes.byref(before))
        assert host.host_dvxlr_render(_p(sigma), None, _p(origin), _p(points), _p(tindex), _p(pred), _p(gt), _p(dd),
                                      _p(idx), None, None, _p(est), *dims) == 0
        after = ctypes.c_int(0); r1 = host.host_par_regular_count(ctypes.byref(after))
        np.testing.assert_array_equal(idx, ref[3])
        np.testing.assert_array_equal(gt, ref[1])
        np.testing.assert_allclose(dd, ref[2], rtol=2e-5, atol=1e-6)
        assert (r1 - r0) >= 0.99 * (after.value - before.value) > 0


def test_round_clamp_equals_library_round(tmp_path_factory):
    """csrc/dvr_par.h par_round_clamp (trunc + exact remainder test) == clamp((int)round(p)) on half-way values, their
    fp64 neighbours, negatives and random positions."""
    host = _build(tmp_path_factory, BUILDS["step_parallel"])
    rng = np.random.default_rng(0)
    k = np.arange(-40, 440, dtype=np.float64)
    half = np.concatenate([k + 0.5, np.nextafter(k + 0.5, np.inf), np.nextafter(k + 0.5, -np.inf), k,
                           np.nextafter(k, np.inf), np.nextafter(k, -np.inf),
                           np.array([0.49999999999999994, -0.49999999999999994, -0.5, -0.0, 0.0, 1e9, -1e9])])
    p = np.ascontiguousarray(np.concatenate([half, rng.uniform(-30, 430, 200000)]))
    for size in (1, 16, 200):
        assert host.host_round_clamp_mismatches(_p(p), len(p), size) == 0
