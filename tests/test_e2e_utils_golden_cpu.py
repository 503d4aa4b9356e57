"""CPU: vidar_amd/plugin/utils/e2e_predictor_utils.py against a golden produced by the reference's
own e2e_predictor_utils.py (tests/golden/make_e2e_utils_golden.py): grid / coordinate helpers, chamfer
wrappers and the DifferentiableVoxelRenderingLayer{,V2} autograd wrappers.  The wrappers call the dvxlr
kernels; here those are routed to the CPU oracle (the reference side ran its kernels' host build)."""
import sys
from contextlib import contextmanager
from pathlib import Path

import numpy as np
import pytest
import torch

GOLD = Path(__file__).parent / "golden"
sys.path.insert(0, str(GOLD))
PC = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD / "e2e_utils.npz")


def test_grid_helpers_match_reference(gold):
    from make_e2e_utils_golden import inputs
    from vidar_amd.plugin.utils import e2e_predictor_utils as U
    grids, coords, pts, pred = inputs()
    np.testing.assert_allclose(U.bev_grids_to_coordinates(grids.clone(), PC).numpy(), gold["grids_to_coords"], rtol=0, atol=1e-5)
    g2, m2 = U.bev_coords_to_grids(coords[..., :2].clone(), 20, 24, PC)
    np.testing.assert_allclose(g2.numpy(), gold["coords_to_grids"], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(m2.numpy(), gold["coords_to_grids_mask"])
    np.testing.assert_allclose(U.coords_to_voxel_grids(coords.clone(), 20, 24, 16, PC).numpy(),
                               gold["coords_to_voxel_grids"], rtol=0, atol=1e-5)
    for off in (0.5, 0.0):
        np.testing.assert_allclose(U.get_bev_grids(5, 7, bs=2, device="cpu", offset=off).numpy(), gold[f"bev_grids_{off}"],
                                   rtol=0, atol=1e-7)
    np.testing.assert_allclose(U.get_bev_grids_3d(4, 6, 3, bs=2, device="cpu").numpy(), gold["bev_grids_3d"], rtol=0, atol=1e-7)
    np.testing.assert_array_equal(U.get_inside_mask(pts, PC).numpy(), gold["inside_mask"])


def test_chamfer_wrappers_match_reference(gold):
    from make_e2e_utils_golden import inputs
    from oracle import cpu_ops
    from vidar_amd.plugin.utils import e2e_predictor_utils as U
    _, _, pts, pred = inputs()
    with cpu_ops.patched():
        np.testing.assert_allclose(float(U.compute_chamfer_distance(pred, pts)), gold["cd"], rtol=1e-5)
        np.testing.assert_allclose(float(U.compute_chamfer_distance_inner(pred, pts, PC)), gold["cd_inner"], rtol=1e-5)
        assert float(U.compute_chamfer_distance_inner(pred + 1000.0, pts, PC)) == float(gold["cd_inner_empty"]) == 0.0


@contextmanager
def dvxlr_on_oracle():
    """route vidar_amd.third_lib.{dvxlr,dvxlr_v2} to oracle/dvr.py (CPU tensors in / out)"""
    from oracle import dvr as O
    from vidar_amd.third_lib import dvxlr, dvxlr_v2
    t = torch.from_numpy
    n = lambda x: x.detach().numpy()
    saved = (dvxlr.render, dvxlr.get_grad_sigma, dvxlr_v2.render_v2, dvxlr_v2.get_grad_sigma_v2)
    dvxlr.render = lambda s, o, p, ti: [t(a) for a in O.dvxlr_render(n(s), n(o), n(p), n(ti))]
    dvxlr.get_grad_sigma = lambda em, idx, ti, s: [t(O.dvxlr_get_grad_sigma(n(em), n(idx), n(ti), tuple(s.shape)))]
    dvxlr_v2.render_v2 = lambda s, o, p, ti, r: [t(a) for a in O.dvxlr_render(n(s), n(o), n(p), n(ti), n(r))]
    dvxlr_v2.get_grad_sigma_v2 = lambda em, idx, ti, s, ind, grp: [
        t(a) for a in O.dvxlr_get_grad_sigma(n(em), n(idx), n(ti), tuple(s.shape), n(ind), n(grp))]
    try:
        yield
    finally:
        dvxlr.render, dvxlr.get_grad_sigma, dvxlr_v2.render_v2, dvxlr_v2.get_grad_sigma_v2 = saved


def test_differentiable_voxel_rendering_layers_match_reference(gold):
    from make_e2e_utils_golden import ray_case
    from vidar_amd.plugin.utils import e2e_predictor_utils as U
    sigma, origin, points, tindex = ray_case()
    with dvxlr_on_oracle():
        s = sigma.clone().requires_grad_(True)
        p, g = U.DifferentiableVoxelRendering(s, origin, points, tindex)
        w = torch.from_numpy(gold["l1_w"])
        (p * w).sum().backward()
        np.testing.assert_allclose(p.detach().numpy(), gold["l1_pred"], rtol=2e-5, atol=1e-4)
        np.testing.assert_array_equal(g.detach().numpy(), gold["l1_gt"])
        scale = np.abs(gold["l1_grad"]).max()
        np.testing.assert_allclose(s.grad.numpy(), gold["l1_grad"], rtol=1e-4, atol=1e-5 * scale)

        s2 = sigma.clone().requires_grad_(True)
        reg = torch.from_numpy(gold["l2_reg"]).clone().requires_grad_(True)
        p2, g2, rp, ind = U.DifferentiableVoxelRenderingV2(s2, origin, points, tindex, reg)
        wr = torch.from_numpy(gold["l2_wr"])
        ((p2 * w).sum() + (rp * wr * (ind >= 0)).sum()).backward()
        np.testing.assert_allclose(p2.detach().numpy(), gold["l2_pred"], rtol=2e-5, atol=1e-4)
        np.testing.assert_array_equal(rp.detach().numpy(), gold["l2_ray_pred"])
        np.testing.assert_array_equal(ind.numpy(), gold["l2_indicator"])
        np.testing.assert_allclose(s2.grad.numpy(), gold["l2_grad"], rtol=1e-4, atol=1e-5 * np.abs(gold["l2_grad"]).max())
        np.testing.assert_allclose(reg.grad.numpy(), gold["l2_grad_reg"], rtol=1e-5, atol=1e-6)
