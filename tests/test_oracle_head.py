"""CPU: oracle/head.py restatement vs golden vectors from the reference's own ViDARHeadBase
methods (tests/golden/make_head_golden.py)."""
from pathlib import Path

import numpy as np
import torch

from oracle import head as H

G = np.load(Path(__file__).parent / "golden" / "head_small.npz")
Fn, Z, Y, X = 2, 8, 20, 24


def tensors():
    t = {k: torch.from_numpy(G[k]) for k in G.files}
    sigma = t["bev_preds"][:, 0, 0].permute(0, 2, 1).contiguous().view(Fn, Z, Y, X)
    return t, sigma


def ref_order(tindex, keep):
    return torch.cat([((tindex == f) & keep).nonzero().squeeze(-1) for f in range(Fn)])


def test_grid_features_and_ce():
    t, sigma = tensors()
    feat, length, keep = H.grid_features(sigma, t["origin_grids"][0], t["gt_grids"][0], t["gt_tindex"][0])
    order = ref_order(t["gt_tindex"][0], keep)
    assert order.numel() == t["feat"].shape[1]
    torch.testing.assert_close(feat[order], t["feat"][0], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(length[order], t["length"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(H.ce_per_ray(feat[order]), t["ce"][0], rtol=1e-5, atol=1e-5)
    w = t["weight"][0]
    loss = (H.ce_per_ray(feat[order]) * w).sum() / torch.clamp(w.sum(), min=1)
    torch.testing.assert_close(loss, t["loss_ce"], rtol=1e-5, atol=1e-6)


def test_argmax_decode():
    t, sigma = tensors()
    pred, gt = H.argmax_decode(sigma, t["origin_grids"][0], t["gt_grids"][0], t["gt_tindex"][0])
    scale = (G["pc_range"][3] - G["pc_range"][0]) / X
    for f in range(Fn):
        sel = (t["gt_tindex"][0] == f) & (gt > 0)
        pts = t["gt_points"][t["gt_points"][:, -1] == f][:, :3]
        o = t["origin_pts"][0, f]
        got = H.rendered_points(o, pts, (pred * scale)[t["gt_tindex"][0] == f])
        torch.testing.assert_close(got, t[f"pred_pcd{f}"], rtol=1e-5, atol=1e-5)
