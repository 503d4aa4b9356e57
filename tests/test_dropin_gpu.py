"""GPU: the call patterns of the reference's wrapper files (pinned on CPU by tests/test_dropin_cpu.py,
which executes those files unchanged; they cannot travel to the GPU box) on the HIP shims, under the
reference's top-level names after `vidar_amd.dropin.install()`:
  * DifferentiableVoxelRenderingLayer{,V2} (utils/e2e_predictor_utils.py:91-143) vs the golden the
    reference's own wrapper text produced on its kernels' host build (tests/golden/e2e_utils.npz),
  * compute_chamfer_distance{,_inner} (:163-183) vs the same golden,
  * ext_module.ms_deform_attn_{forward,backward} exactly as function.py:118-124, :146-160 calls them."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"
sys.path.insert(0, str(GOLD))
PC = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]


@pytest.fixture
def names():
    from vidar_amd import dropin as D
    D.install(patch_loaders=False)
    yield
    D.uninstall()


def test_voxel_rendering_layers_on_hip_match_reference_golden(names):
    import dvxlr
    import dvxlr_v2
    from make_e2e_utils_golden import ray_case
    gold = np.load(GOLD / "e2e_utils.npz")
    sigma, origin, points, tindex = (t.cuda() for t in ray_case())
    c = lambda k: torch.from_numpy(gold[k]).cuda()

    class Layer(torch.autograd.Function):              # call pattern of e2e_predictor_utils.py:91-112
        @staticmethod
        def forward(ctx, sigma, origin, points, tindex):
            pred_dist, gt_dist, dd_dsigma, indices = dvxlr.render(sigma, origin, points, tindex)
            ctx.save_for_backward(dd_dsigma, indices, tindex, sigma)
            return pred_dist, gt_dist

        @staticmethod
        def backward(ctx, gradpred, gradgt):
            dd_dsigma, indices, tindex, sigma_shape = ctx.saved_tensors
            em = gradpred[..., None] * dd_dsigma
            em[torch.isnan(em)] = 0.0
            return dvxlr.get_grad_sigma(em, indices, tindex, sigma_shape)[0], None, None, None

    class LayerV2(torch.autograd.Function):            # :122-141
        @staticmethod
        def forward(ctx, sigma, origin, points, tindex, sigma_regul):
            pred_dist, gt_dist, dd_dsigma, indices, ray_pred, indicator = dvxlr_v2.render_v2(
                sigma, origin, points, tindex, sigma_regul)
            ctx.save_for_backward(dd_dsigma, indices, tindex, sigma, indicator)
            return pred_dist, gt_dist, ray_pred, indicator

        @staticmethod
        def backward(ctx, gradpred, gradgt, grad_ray_pred, grad_indicator):
            dd_dsigma, indices, tindex, sigma_shape, indicator = ctx.saved_tensors
            gs, gr = dvxlr_v2.get_grad_sigma_v2(gradpred[..., None] * dd_dsigma, indices, tindex, sigma_shape,
                                                indicator, grad_ray_pred)
            return gs, None, None, None, gr

    s = sigma.clone().requires_grad_(True)
    p, g = Layer.apply(s, origin, points, tindex)
    (p * c("l1_w")).sum().backward()
    np.testing.assert_allclose(p.detach().cpu().numpy(), gold["l1_pred"], rtol=2e-5, atol=1e-4)
    np.testing.assert_array_equal(g.detach().cpu().numpy(), gold["l1_gt"])
    np.testing.assert_allclose(s.grad.cpu().numpy(), gold["l1_grad"], rtol=1e-4, atol=1e-5 * np.abs(gold["l1_grad"]).max())
    s2 = sigma.clone().requires_grad_(True)
    reg = c("l2_reg").clone().requires_grad_(True)
    p2, g2, rp, ind = LayerV2.apply(s2, origin, points, tindex, reg)
    ((p2 * c("l1_w")).sum() + (rp * c("l2_wr") * (ind.detach() >= 0)).sum()).backward()
    np.testing.assert_allclose(p2.detach().cpu().numpy(), gold["l2_pred"], rtol=2e-5, atol=1e-4)
    np.testing.assert_array_equal(rp.detach().cpu().numpy(), gold["l2_ray_pred"])
    np.testing.assert_array_equal(ind.detach().cpu().numpy(), gold["l2_indicator"])
    np.testing.assert_allclose(s2.grad.cpu().numpy(), gold["l2_grad"], rtol=1e-4, atol=1e-5 * np.abs(gold["l2_grad"]).max())
    np.testing.assert_allclose(reg.grad.cpu().numpy(), gold["l2_grad_reg"], rtol=1e-5, atol=1e-6)


def test_chamfer_wrappers_on_hip_match_reference_golden(names):
    from chamferdist import ChamferDistance
    from make_e2e_utils_golden import inputs
    gold = np.load(GOLD / "e2e_utils.npz")
    _, _, pts, pred = inputs()
    pts, pred = pts.cuda(), pred.cuda()
    cd_fn = ChamferDistance()

    def compute_chamfer_distance(pred_pcd, gt_pcd):    # :163-170
        loss_src, loss_dst, _ = cd_fn(pred_pcd[None, ...], gt_pcd[None, ...], bidirectional=True, reduction="sum")
        return (loss_src / pred_pcd.shape[0] + loss_dst / gt_pcd.shape[0]) / 2.0

    np.testing.assert_allclose(float(compute_chamfer_distance(pred, pts)), gold["cd"], rtol=1e-5)
    from vidar_amd.plugin.utils import e2e_predictor_utils as U
    np.testing.assert_allclose(float(U.compute_chamfer_distance_inner(pred, pts, PC)), gold["cd_inner"], rtol=1e-5)


def test_mmcv_ext_call_pattern_on_hip():
    from oracle import msda as M
    from vidar_amd import dropin as D
    ext_module = D.load_ext("_ext", ["ms_deform_attn_backward", "ms_deform_attn_forward"])
    shapes = [(12, 20), (6, 10), (3, 5)]
    value, sh, loc, w = M.make_case(4, 2, shapes, 333, P=4)
    lsi = M.level_start_index(shapes)
    gout = torch.randn(2, 333, 256, generator=torch.Generator().manual_seed(5))
    v64, l64, w64 = (t.double().requires_grad_(True) for t in (value, loc, w))
    ref = M.msda_gather(v64, sh, l64, w64)
    gref = torch.autograd.grad((ref * gout.double()).sum(), [v64, l64, w64])
    value, sh, lsi, loc, w, gout = (t.cuda() for t in (value, sh, lsi, loc, w, gout))
    output = ext_module.ms_deform_attn_forward(value, sh, lsi, loc, w, im2col_step=64)        # function.py:118-124
    torch.testing.assert_close(output.cpu().double(), ref.detach(), rtol=1e-4, atol=1e-5)
    grad_value = torch.zeros_like(value)                                                        # :146-148
    grad_sampling_loc = torch.zeros_like(loc)
    grad_attn_weight = torch.zeros_like(w)
    assert ext_module.ms_deform_attn_backward(value, sh, lsi, loc, w, gout.contiguous(), grad_value,
                                              grad_sampling_loc, grad_attn_weight, im2col_step=64) is None
    for g, r in zip((grad_value, grad_sampling_loc, grad_attn_weight), gref):
        torch.testing.assert_close(g.cpu().double(), r, rtol=2e-4, atol=2e-5 * max(1.0, float(r.abs().max())))
    with pytest.raises(RuntimeError):
        ext_module.ms_deform_attn_forward(value.cpu(), sh, lsi, loc, w, im2col_step=64)
