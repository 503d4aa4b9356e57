"""GPU parity: fused head ray-march kernels vs golden vectors from the reference's ViDARHeadBase
(tests/golden/head_small.npz) and vs the torch-CPU oracle on a second random case.
Tolerance 1e-4 relative (fp32 logsumexp over 513 terms, tree vs sequential order)."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import head as H
from test_oracle_head import ref_order, tensors, Fn, Z, Y, X

pytestmark = pytest.mark.gpu


from vidar_amd.synthetic import dense_rays  # noqa: E402,F401  (shared with tools/kbench.py)


def test_ray_ce_matches_reference_golden():
    from vidar_amd.plugin.dense_heads.ray_ops import ray_ce
    t, sigma = tensors()
    sg = sigma.cuda().requires_grad_(True)
    og, gg, ti = t["origin_grids"][0].cuda(), t["gt_grids"][0].cuda(), t["gt_tindex"][0].cuda()
    ce, valid = ray_ce(sg, og, gg, ti)
    order = ref_order(t["gt_tindex"][0], valid.cpu() > 0)
    assert order.numel() == t["ce"].shape[1], "kept-ray set must equal the reference's"
    torch.testing.assert_close(ce.detach().cpu()[order], t["ce"][0], rtol=1e-4, atol=1e-4)
    lw = torch.from_numpy(np.asarray(t["loss_weight"])).float().view(-1).cuda()
    w = lw[ti.clamp(min=0).long()] * valid
    loss = (ce * w).sum() / torch.clamp(w.sum(), min=1)
    torch.testing.assert_close(loss.detach().cpu(), t["loss_ce"], rtol=1e-4, atol=1e-5)
    # gradient vs oracle autograd
    s2 = sigma.clone().requires_grad_(True)
    feat, length, keep = H.grid_features(s2, t["origin_grids"][0], t["gt_grids"][0], t["gt_tindex"][0])
    w_ref = torch.from_numpy(np.asarray(t["loss_weight"])).float().view(-1)[t["gt_tindex"][0].clamp(min=0).long()] * keep
    l_ref = (H.ce_per_ray(feat[keep]) * w_ref[keep]).sum() / torch.clamp(w_ref.sum(), min=1)
    g_ref, = torch.autograd.grad(l_ref, s2)
    g, = torch.autograd.grad(loss, sg)
    torch.testing.assert_close(g.cpu(), g_ref, rtol=2e-4, atol=2e-6)


def test_ray_gumbel_matches_oracle_with_reference_noise():
    from vidar_amd.plugin.dense_heads.ray_ops import ray_gumbel
    t, sigma = tensors()
    pts, tix = dense_rays(Fn, Z, Y, X)
    noise = t["noise"][0]                                 # the noise the reference run consumed
    s2 = sigma.clone().requires_grad_(True)
    feat, length, keep = H.grid_features(s2, t["origin_grids"][0], pts, tix)
    assert bool(keep.all())
    d_ref = H.gumbel_distance(feat[:, 1:], length[:, 1:], noise)
    gout = torch.randn(d_ref.shape, generator=torch.Generator().manual_seed(3))
    g_ref, = torch.autograd.grad((d_ref * gout).sum(), s2)
    sg = sigma.cuda().requires_grad_(True)
    d = ray_gumbel(sg, t["origin_grids"][0].cuda(), pts.cuda(), tix.cuda(), noise.cuda())
    torch.testing.assert_close(d.detach().cpu(), d_ref.detach(), rtol=1e-5, atol=1e-5)
    g, = torch.autograd.grad((d * gout.cuda()).sum(), sg)
    torch.testing.assert_close(g.cpu(), g_ref, rtol=3e-4, atol=3e-5)


def test_ray_argmax_matches_oracle():
    from vidar_amd.plugin.dense_heads.ray_ops import ray_argmax
    t, sigma = tensors()
    sigma = sigma.clone(); sigma[0, :, :3] = 0.0          # exact zeros must be masked like outside
    pred_ref, gt_ref = H.argmax_decode(sigma, t["origin_grids"][0], t["gt_grids"][0], t["gt_tindex"][0])
    pred, gt = ray_argmax(sigma.cuda(), t["origin_grids"][0].cuda(), t["gt_grids"][0].cuda(),
                          t["gt_tindex"][0].cuda())
    sel = t["gt_tindex"][0] >= 0
    torch.testing.assert_close(gt.cpu()[sel], gt_ref[sel], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(pred.cpu()[sel], pred_ref[sel], rtol=1e-5, atol=1e-5)


def test_random_volume_16x200x200():
    """BASELINE volume size, 4000 LiDAR-like rays: CE / validity vs oracle."""
    from vidar_amd.plugin.dense_heads.ray_ops import ray_ce
    from vidar_amd.synthetic import ray_set
    sig, origin, points, tindex = ray_set(seed=21, N=1, T=2, rays_per_frame=2000, pad=9, origin_jitter=8.0)
    sigma = torch.randn(2, 16, 200, 200, generator=torch.Generator().manual_seed(1))
    o, p, ti = torch.from_numpy(origin[0]), torch.from_numpy(points[0]), torch.from_numpy(tindex[0])
    feat, length, keep = H.grid_features(sigma, o, torch.nan_to_num(p, nan=-1e4), ti)
    ce, valid = ray_ce(sigma.cuda(), o.cuda(), p.cuda(), ti.cuda())
    assert torch.equal(valid.cpu() > 0, keep)
    torch.testing.assert_close(ce.cpu()[keep], H.ce_per_ray(feat[keep]), rtol=1e-4, atol=1e-4)


def test_full_size_30k_rays_ce_forward_backward():
    """BASELINE frame: 30 000 GT rays through a 16x200x200 logit volume: kept-ray set, per-ray CE and
    d loss / d sigma against the oracle (trilinear grid_sample + logsumexp over 513 samples)."""
    from vidar_amd.plugin.dense_heads.ray_ops import ray_ce
    from vidar_amd.synthetic import ray_set
    sig, origin, points, tindex = ray_set(seed=33, N=1, T=1, rays_per_frame=30000)
    sigma = torch.randn(1, 16, 200, 200, generator=torch.Generator().manual_seed(2))
    o, p, ti = torch.from_numpy(origin[0]), torch.from_numpy(points[0]), torch.from_numpy(tindex[0])
    s2 = sigma.clone().requires_grad_(True)
    feat, length, keep = H.grid_features(s2, o, torch.nan_to_num(p, nan=-1e4), ti)
    ce_ref = H.ce_per_ray(feat[keep])
    wts = torch.rand(int(keep.sum()), generator=torch.Generator().manual_seed(3))
    g_ref, = torch.autograd.grad((ce_ref * wts).sum(), s2)
    sg = sigma.cuda().requires_grad_(True)
    ce, valid = ray_ce(sg, o.cuda(), p.cuda(), ti.cuda())
    assert torch.equal(valid.cpu() > 0, keep)
    torch.testing.assert_close(ce.detach().cpu()[keep], ce_ref.detach(), rtol=1e-4, atol=1e-4)
    g, = torch.autograd.grad((ce[keep.cuda()] * wts.cuda()).sum(), sg)
    torch.testing.assert_close(g.cpu(), g_ref, rtol=3e-4, atol=3e-5 * float(g_ref.abs().max()))
