"""numpy front-end of oracle/dvr_oracle.c (CPU restatement of third_lib/dvr, dvxlr, dvxlr_v2).

TEST INFRASTRUCTURE.  Function names mirror the reference's pybind surface
(third_lib/dvr/dvr.cpp:65-69, third_lib/dvxlr/dvxlr.cpp:61-65, third_lib/dvxlr/dvxlr_v2.cpp:66-70).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None
DVR_MAX_D = 1446
DVXLR_MAX_D = 1026
LOSS = {"l1": 0, "bce": 0, "l2": 1, "absrel": 2}   # dvr.cu:658-672
PHASE = {"test": 0, "train": 1}                    # dvr.cu:357-366


def lib():
    global _LIB
    if _LIB is None:
        so = _HERE / "libvidar_oracle.so"
        if not so.exists():
            subprocess.run(["make", "-C", str(_HERE)], check=True)
        _LIB = ctypes.CDLL(str(so))
    return _LIB


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def set_threads(n: int):
    """OpenMP threads of the C restatement.  The environment variable only counts before libgomp starts, so the
    runtime the library is linked against is told directly as well."""
    os.environ["OMP_NUM_THREADS"] = str(n)
    try:
        import ctypes
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:
        pass


def _dims(sigma, origin, points):
    N, T, Z, Y, X = sigma.shape
    return N, points.shape[1], T, origin.shape[1], Z, Y, X


def render_forward(sigma, origin, points, tindex, phase_name="train"):
    sigma, ps = _f(sigma); origin, po = _f(origin); points, pp = _f(points); tindex, pt = _f(tindex)
    N, M, T, TO, Z, Y, X = _dims(sigma, origin, points)
    pred = np.empty((N, M), np.float32); gt = np.empty((N, M), np.float32)
    rc = lib().oracle_dvr_render_forward(ps, po, pp, pt, pred.ctypes.data_as(ctypes.c_void_p),
                                         gt.ctypes.data_as(ctypes.c_void_p), N, M, T, TO, Z, Y, X,
                                         PHASE[phase_name])
    assert rc == 0
    return pred, gt


def render(sigma, origin, points, tindex, loss_name="l1"):
    sigma, ps = _f(sigma); origin, po = _f(origin); points, pp = _f(points); tindex, pt = _f(tindex)
    N, M, T, TO, Z, Y, X = _dims(sigma, origin, points)
    pred = np.empty((N, M), np.float32); gt = np.empty((N, M), np.float32)
    grad = np.empty_like(sigma)
    rc = lib().oracle_dvr_render(ps, po, pp, pt, pred.ctypes.data_as(ctypes.c_void_p),
                                 gt.ctypes.data_as(ctypes.c_void_p),
                                 grad.ctypes.data_as(ctypes.c_void_p), N, M, T, TO, Z, Y, X,
                                 LOSS[loss_name])
    assert rc == 0
    return pred, gt, grad


def init(points, tindex, grid):
    points, pp = _f(points); tindex, pt = _f(tindex)
    T, Z, Y, X = grid
    N, M = points.shape[:2]
    occ = np.empty((N, T, Z, Y, X), np.float32)
    rc = lib().oracle_dvr_init(pp, pt, occ.ctypes.data_as(ctypes.c_void_p), N, M, T, Z, Y, X)
    assert rc == 0
    return occ


def dvxlr_render(sigma, origin, points, tindex, sigma_regul=None):
    """dvxlr.render (sigma_regul None) -> 4 arrays; dvxlr_v2.render_v2 -> 6 arrays."""
    sigma, ps = _f(sigma); origin, po = _f(origin); points, pp = _f(points); tindex, pt = _f(tindex)
    N, M, T, TO, Z, Y, X = _dims(sigma, origin, points)
    L = DVXLR_MAX_D
    pred = np.empty((N, M), np.float32); gt = np.empty((N, M), np.float32)
    dd = np.empty((N, M, L), np.float32); idx = np.empty((N, M, L, 3), np.float32)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    if sigma_regul is None:
        rc = lib().oracle_dvxlr_render(ps, None, po, pp, pt, vp(pred), vp(gt), vp(dd), vp(idx), None,
                                       None, N, M, T, TO, Z, Y, X)
        assert rc == 0
        return pred, gt, dd, idx
    sigma_regul, pr = _f(sigma_regul)
    rp = np.empty((N, M, L), np.float32); ind = np.empty((N, M, L), np.float32)
    rc = lib().oracle_dvxlr_render(ps, pr, po, pp, pt, vp(pred), vp(gt), vp(dd), vp(idx), vp(rp),
                                   vp(ind), N, M, T, TO, Z, Y, X)
    assert rc == 0
    return pred, gt, dd, idx, rp, ind


def dvxlr_get_grad_sigma(em, indices, tindex, sigma_shape, indicator=None, grad_ray_pred=None):
    em, pe = _f(em); indices, pi = _f(indices); tindex, pt = _f(tindex)
    N, T, Z, Y, X = sigma_shape
    M, L = em.shape[1], em.shape[2]
    g = np.empty(sigma_shape, np.float32)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    if indicator is None:
        rc = lib().oracle_dvxlr_get_grad_sigma(pe, pi, pt, None, None, vp(g), None, N, M, L, T, Z, Y, X)
        assert rc == 0
        return g
    indicator, pin = _f(indicator); grad_ray_pred, pg = _f(grad_ray_pred)
    g2 = np.empty(sigma_shape, np.float32)
    rc = lib().oracle_dvxlr_get_grad_sigma(pe, pi, pt, pin, pg, vp(g), vp(g2), N, M, L, T, Z, Y, X)
    assert rc == 0
    return g, g2
