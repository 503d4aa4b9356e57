"""CPU oracle for the chamfer / KNN path -- TEST INFRASTRUCTURE.

knn_points_idx / knn_points_backward restate third_lib/chamfer_dist/chamferdist/chamferdist/
knn_cpu.cpp:7-58 and :64-106 in numpy float32 (elementwise numpy never fuses mul+add, so
((dx*dx+dy*dy)+dz*dz) rounds exactly like the gcc build of the reference).  Pinned against the
reference's own ext.cpp+knn_cpu.cpp compiled unmodified (oracle/_ref/ref_chamferdist_C) in
tests/test_oracle_chamfer.py.

chamfer_distance_mmdet3d restates mmdet3d v0.17.1 `mmdet3d.models.losses.chamfer_distance`
(call site dense_heads/vidar_head_base.py:654).  mmdet3d is NOT vendored in /root/reference and
not installable here: its nearest-neighbour part (per-point min d^2 and arg-min, both directions) is
checked against the reference's own knn_cpu.cpp build (tests/test_oracle_chamfer.py::
test_training_chamfer_formula_rests_on_the_reference_knn); the `.mean(1).mean()` normalisation stays recalled.
compute_chamfer_distance{,_inner}: bevformer/utils/e2e_predictor_utils.py:163-183.
"""
from __future__ import annotations

import numpy as np


def knn_points_idx(p1, p2, lengths1=None, lengths2=None, chunk=2048):
    p1 = np.asarray(p1, np.float32); p2 = np.asarray(p2, np.float32)
    N, P1, _ = p1.shape
    P2 = p2.shape[1]
    l1 = np.full(N, P1) if lengths1 is None else np.asarray(lengths1)
    l2 = np.full(N, P2) if lengths2 is None else np.asarray(lengths2)
    idx = np.zeros((N, P1, 1), np.int64); dist = np.zeros((N, P1, 1), np.float32)
    for n in range(N):
        a, b = p1[n, :l1[n]], p2[n, :l2[n]]
        if len(a) == 0 or len(b) == 0:
            continue
        for s in range(0, len(a), chunk):
            q = a[s:s + chunk]
            dx = q[:, None, 0] - b[None, :, 0]
            dy = q[:, None, 1] - b[None, :, 1]
            dz = q[:, None, 2] - b[None, :, 2]
            d = dx * dx
            d = d + dy * dy
            d = d + dz * dz
            j = np.argmin(d, 1)                      # first minimum == strict '<' scan
            idx[n, s:s + len(q), 0] = j
            dist[n, s:s + len(q), 0] = d[np.arange(len(q)), j]
    return idx, dist


def knn_points_backward(p1, p2, lengths1, lengths2, idx, grad_dists):
    p1 = np.asarray(p1, np.float32); p2 = np.asarray(p2, np.float32)
    g1 = np.zeros_like(p1); g2 = np.zeros_like(p2)
    for n in range(p1.shape[0]):
        if lengths2[n] == 0:
            continue
        for i in range(int(lengths1[n])):
            j = idx[n, i, 0]
            diff = np.float32(2.0) * np.float32(grad_dists[n, i, 0]) * (p1[n, i] - p2[n, j])
            g1[n, i] += diff
            g2[n, j] += np.float32(-1.0) * diff
    return g1, g2


def chamfer_distance_mmdet3d(src, dst, src_weight=1.0, dst_weight=1.0):
    """[3P, parity unpinned] criterion 'l2' (= mse), reduction 'mean'.  src [B,N,3], dst [B,M,3]."""
    src = np.asarray(src, np.float32); dst = np.asarray(dst, np.float32)
    d = ((src[:, :, None, :] - dst[:, None, :, :]) ** 2).sum(-1)
    d1, i1 = d.min(2), d.argmin(2)
    d2, i2 = d.min(1), d.argmin(1)
    loss_src = (d1 * src_weight).mean(1).mean()
    loss_dst = (d2 * dst_weight).mean(1).mean()
    return loss_src, loss_dst, i1, i2


def get_inside_mask(p, r):
    return ((r[0] <= p[..., 0]) & (p[..., 0] <= r[3]) & (r[1] <= p[..., 1]) & (p[..., 1] <= r[4]) &
            (r[2] <= p[..., 2]) & (p[..., 2] <= r[5]))


def compute_chamfer_distance(pred, gt):
    _, d1 = knn_points_idx(pred[None], gt[None])
    _, d2 = knn_points_idx(gt[None], pred[None])
    a = np.float32(d1.sum(dtype=np.float32)); b = np.float32(d2.sum(dtype=np.float32))
    return (a / pred.shape[0] + b / gt.shape[0]) / 2.0


def compute_chamfer_distance_inner(pred, gt, pc_range):
    pm, gm = get_inside_mask(pred, pc_range), get_inside_mask(gt, pc_range)
    if pm.sum() == 0 or gm.sum() == 0:
        return 0.0
    return compute_chamfer_distance(pred[pm], gt[gm])
