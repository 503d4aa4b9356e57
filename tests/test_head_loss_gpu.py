"""GPU parity of the whole head loss (ray-march CE + dense gumbel render + chamfer, fused kernels)
against the loss values and gradients the REFERENCE's ViDARHeadBase.loss produced on the same
inputs and the same gumbel noise (tests/golden/head_small.npz).  mmdet3d's chamfer_distance inside
that run is the documented-formula stub (third party, unpinned)."""
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
G = np.load(Path(__file__).parent / "golden" / "head_small.npz")
Fn, Z, Y, X = 2, 8, 20, 24


def make_head():
    from vidar_amd.plugin.dense_heads.vidar_head_base import ViDARHeadBase
    h = ViDARHeadBase.__new__(ViDARHeadBase)
    nn.Module.__init__(h)
    h.ray_grid_num, h.ray_grid_step = 512, 1.0
    h.use_ce_loss, h.use_dist_loss, h.use_dense_loss, h.dense_loss_weight = True, False, True, 1.0
    h.loss_weight = G["loss_weight"]
    h.eval_within_grid = False
    noise = torch.from_numpy(G["noise"][0]).cuda()
    h.gumbel_noise_fn = lambda R, K: noise
    return h


def test_loss_and_gradient_match_reference():
    h = make_head()
    bev = torch.from_numpy(G["bev_preds"]).cuda().requires_grad_(True)
    gt = torch.from_numpy(G["gt_points"]).cuda()
    origin = torch.from_numpy(G["origin_pts"]).cuda()
    out = h.loss(dict(next_bev_preds=bev, valid_frames=[0, 1]), [gt], 0, Y, X, list(G["pc_range"]),
                 Fn, batched_origin_points=origin.clone())
    np.testing.assert_allclose(float(out["regularization.loss"]), float(G["loss_ce"]), rtol=1e-4)
    np.testing.assert_allclose(float(out["loss.dense_voxel"]), float(G["loss_dense"]), rtol=1e-3,
                               atol=1e-6)    # CD within 1e-3
    total = out["regularization.loss"] + 2.0 * out["loss.dense_voxel"]
    g, = torch.autograd.grad(total, bev)
    ref = torch.from_numpy(G["grad_bev_preds"])
    torch.testing.assert_close(g.cpu(), ref, rtol=2e-3, atol=2e-6 * max(1.0, float(ref.abs().max()) * 1e3))


def test_decode_matches_reference():
    h = make_head()
    bev = torch.from_numpy(G["bev_preds"]).cuda()
    gt = torch.from_numpy(G["gt_points"]).cuda()
    origin = torch.from_numpy(G["origin_pts"]).cuda()
    d = h.get_point_cloud_prediction(dict(next_bev_preds=bev, valid_frames=[0, 1]), [gt], 0, Y, X,
                                     list(G["pc_range"]), batched_origin_points=origin.clone())
    for f in range(Fn):
        torch.testing.assert_close(d["pred_pcds"][0][f].cpu(), torch.from_numpy(G[f"pred_pcd{f}"]),
                                   rtol=1e-5, atol=1e-4)
        torch.testing.assert_close(d["gt_pcds"][0][f].cpu(), torch.from_numpy(G[f"gt_pcd{f}"]),
                                   rtol=1e-5, atol=1e-4)
    # chamfer metric on the decoded clouds through the chamferdist mirror == oracle value
    from oracle import chamfer as C
    from vidar_amd.plugin.utils.e2e_predictor_utils import compute_chamfer_distance_inner
    pc_range = list(G["pc_range"])
    cd = compute_chamfer_distance_inner(d["pred_pcds"][0][0], d["gt_pcds"][0][0], pc_range)
    ref = C.compute_chamfer_distance_inner(G["pred_pcd0"], G["gt_pcd0"], pc_range)
    assert abs(float(cd) - float(ref)) <= 1e-3
