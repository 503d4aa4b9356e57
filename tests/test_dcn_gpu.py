"""GPU parity: DCNv2 (HIP im2col/col2im + GEMM) vs the grid_sample oracle, forward and all five
gradients; tolerance 1e-4 (fp32 sums over C*9 products).  Oracle is 'parity unpinned' (mmcv)."""
import os

import pytest
import torch

from oracle import dcn as D

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,C,Cout,H,W,stride", [(2, 8, 6, 9, 11, 1), (1, 16, 16, 12, 10, 2), (1, 3, 4, 5, 5, 1),
                                                 (2, 40, 8, 58, 100, 1),       # stage-3 feature map size
                                                 (1, 20, 4, 6, 2, 1)])         # narrowest image of the pair-load kernels
def test_dcn_fwd_bwd(N, C, Cout, H, W, stride):
    from vidar_amd.plugin.backbones import modulated_deform_conv2d
    g = torch.Generator().manual_seed(N * 100 + C)
    Ho = (H + 2 - 3) // stride + 1; Wo = (W + 2 - 3) // stride + 1
    x = torch.randn(N, C, H, W, generator=g, dtype=torch.float64)
    off = torch.randn(N, 18, Ho, Wo, generator=g, dtype=torch.float64) * 1.5     # leaves the image at borders
    mask = torch.rand(N, 9, Ho, Wo, generator=g, dtype=torch.float64)
    wgt = torch.randn(Cout, C, 3, 3, generator=g, dtype=torch.float64) * 0.2
    bias = torch.randn(Cout, generator=g, dtype=torch.float64)
    leaves = [t.clone().requires_grad_(True) for t in (x, off, mask, wgt, bias)]
    ref = D.modulated_deform_conv2d(*leaves, stride=stride, padding=1)
    gout = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    gref = torch.autograd.grad((ref * gout).sum(), leaves)
    dl = [t.float().cuda().requires_grad_(True) for t in (x, off, mask, wgt, bias)]
    out = modulated_deform_conv2d(*dl, stride=stride, padding=1)
    torch.testing.assert_close(out.cpu().double(), ref.detach(), rtol=1e-4, atol=1e-4)
    got = torch.autograd.grad((out * gout.float().cuda()).sum(), dl)
    for a, b, nm in zip(got, gref, ["x", "offset", "mask", "weight", "bias"]):
        torch.testing.assert_close(a.cpu().double(), b, rtol=2e-4, atol=2e-4 * max(1.0, float(b.abs().max())),
                                   msg=lambda m: nm + ": " + m)


def test_one_column_image_equals_the_same_image_padded_with_a_zero_column():
    """W == 1 takes the scalar-load kernels (the pair-load ones need two columns; the grid_sample oracle cannot
    express a one-column image): zero padding makes it equal to the W == 2 image with an empty second column."""
    from vidar_amd.plugin.backbones import modulated_deform_conv2d
    g = torch.Generator().manual_seed(11)
    x1 = torch.randn(2, 6, 5, 1, generator=g).cuda().requires_grad_(True)
    off = (torch.randn(2, 18, 5, 1, generator=g) * 0.8).cuda().requires_grad_(True)
    mask = torch.rand(2, 9, 5, 1, generator=g).cuda().requires_grad_(True)
    wgt = torch.randn(4, 6, 3, 3, generator=g).cuda()
    gout = torch.randn(2, 4, 5, 1, generator=g).cuda()
    out1 = modulated_deform_conv2d(x1, off, mask, wgt, None, stride=1, padding=1)
    g1 = torch.autograd.grad((out1 * gout).sum(), [x1, off, mask])
    x2 = torch.cat([x1.detach(), torch.zeros_like(x1)], -1).requires_grad_(True)               # [.., 5, 2]
    pad = lambda t, v: torch.cat([t.detach(), torch.full_like(t, v)], -1).requires_grad_(True)  # second output column unused
    off2, mask2 = pad(off, 0.0), pad(mask, 0.0)
    out2 = modulated_deform_conv2d(x2, off2, mask2, wgt, None, stride=1, padding=1)
    torch.testing.assert_close(out2[..., :1], out1, rtol=1e-5, atol=1e-5)
    g2 = torch.autograd.grad((out2[..., :1] * gout).sum(), [x2, off2, mask2])
    for a, b in zip(g1, g2):
        torch.testing.assert_close(a, b[..., :1], rtol=1e-4, atol=1e-5)


def test_im2col_far_and_nan_offsets_contribute_nothing():
    """a tap sampled outside the image, or at a NaN location, is zero padding (not NaN x 0)"""
    from vidar_amd.plugin.backbones import modulated_deform_conv2d
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 17, 7, 9, generator=g).cuda()
    off = torch.zeros(1, 18, 7, 9); off[:, :6] = 50.0; off[:, 6:8, 2:4] = float("nan")
    mask = torch.ones(1, 9, 7, 9)
    wgt = torch.randn(5, 17, 3, 3, generator=g)
    out = modulated_deform_conv2d(x, off.cuda(), mask.cuda(), wgt.cuda(), None, stride=1, padding=1)
    assert torch.isfinite(out).all()
    keep = torch.ones(9); keep[:3] = 0                      # taps 0-2 left the image everywhere
    ref = torch.nn.functional.conv2d(x, (wgt * keep.view(1, 1, 3, 3)).cuda(), padding=1)
    ok = torch.ones(7, 9, dtype=torch.bool); ok[2:4] = False       # rows whose 4th tap is NaN lose that tap too
    torch.testing.assert_close(out[..., ok.cuda()], ref[..., ok.cuda()], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("N,C,H,W,stride", [(2, 19, 9, 11, 1), (1, 32, 29, 50, 1), (2, 5, 12, 10, 2)])
def test_col2im_gather_equals_atomic_scatter(N, C, H, W, stride):
    """both grad_x strategies of vidar_dcn_col2im_f32 (workspace / reverse-map gather vs atomics)"""
    from vidar_amd.plugin.backbones import dcn_col2im
    g = torch.Generator().manual_seed(C)
    Ho = (H + 2 - 3) // stride + 1; Wo = (W + 2 - 3) // stride + 1
    x = torch.randn(N, C, H, W, generator=g).cuda()
    off = (torch.randn(N, 18, Ho, Wo, generator=g) * 2.0).cuda()
    mask = torch.rand(N, 9, Ho, Wo, generator=g).cuda()
    gcols = torch.randn(N, C * 9, Ho * Wo, generator=g).cuda()
    a = dcn_col2im(gcols, x, off, mask, 3, 3, stride, 1, 1, Ho, Wo, gather=True)
    b = dcn_col2im(gcols, x, off, mask, 3, 3, stride, 1, 1, Ho, Wo, gather=False)
    for u, v, nm in zip(a, b, ["grad_x", "grad_offset", "grad_mask"]):
        torch.testing.assert_close(u, v, rtol=1e-4, atol=1e-5 * max(1.0, float(v.abs().max())), msg=lambda m: nm + m)


def test_zero_offsets_equal_plain_convolution():
    from vidar_amd.plugin.backbones import ModulatedDeformConv2dPack
    m = ModulatedDeformConv2dPack(8, 8, 3, padding=1, bias=False).cuda()   # conv_offset is zero-init
    x = torch.randn(2, 8, 14, 13, device="cuda")
    ref = torch.nn.functional.conv2d(x, m.weight, padding=1) * 0.5         # mask = sigmoid(0)
    torch.testing.assert_close(m(x), ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("shape,res", [((2, 8, 6, 8), True), ((1, 5, 7, 5), True), ((3, 4, 29, 50), False),
                                       ((2, 3, 58, 100), True), ((1, 2, 116, 200), False)])
def test_fused_frozen_bn_epilogue(shape, res):
    """FrozenBN(x, residual, relu) == relu(batch_norm_eval(x) + residual), values and gradients."""
    from vidar_amd.plugin.backbones import FrozenBN
    import torch.nn.functional as F
    torch.manual_seed(0)
    bn = FrozenBN(shape[1]).cuda()
    bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2)
    x = torch.randn(*shape, device="cuda", requires_grad=True)
    r = torch.randn(*shape, device="cuda", requires_grad=True) if res else None
    y = bn(x, residual=r, relu=True)
    ref = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
    ref = F.relu(ref + r if res else ref)
    torch.testing.assert_close(y, ref, rtol=1e-5, atol=1e-5)
    g = torch.randn_like(y)
    leaves = [x, r] if res else [x]
    a = torch.autograd.grad(y, leaves, g, retain_graph=True)
    b = torch.autograd.grad(ref, leaves, g)
    for u, v in zip(a, b):
        torch.testing.assert_close(u, v, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("stride,bias", [(1, False), (2, False), (1, True)])
def test_conv1x1_as_batched_gemm_equals_conv2d(stride, bias):
    """backbones.Conv1x1 (1x1 convolutions of the ResNet bottlenecks as torch GEMMs) against F.conv2d"""
    import torch.nn.functional as F
    from vidar_amd.plugin.backbones import Conv1x1
    torch.manual_seed(0)
    m = Conv1x1(24, 40, 1, stride=stride, bias=bias).cuda()
    x = torch.randn(3, 24, 17, 22, device="cuda", requires_grad=True)
    y = m(x)
    ref = F.conv2d(x, m.weight, m.bias, stride=stride)
    torch.testing.assert_close(y, ref, rtol=1e-4, atol=1e-4)
    g = torch.randn_like(ref)
    params = [x, m.weight] + ([m.bias] if bias else [])
    a = torch.autograd.grad(y, params, g, retain_graph=True)
    b = torch.autograd.grad(ref, params, g)
    for u, v in zip(a, b):
        torch.testing.assert_close(u, v, rtol=1e-4, atol=1e-4 * max(1.0, float(v.abs().max())))
    assert sorted(m.state_dict()) == (["bias", "weight"] if bias else ["weight"]) and m.weight.shape == (40, 24, 1, 1)


@pytest.mark.parametrize("shape", [(2, 3, 8, 8), (1, 4, 7, 12), (1, 2, 1, 4), (3, 64, 58, 100), (2, 64, 464, 800)])
def test_fused_stem_pool_matches_two_kernel_path(shape):
    """vidar_stem_bn_relu_pool_f32 == max_pool2d(FrozenBN(x, relu=True), 3, 2, 1), bit for bit"""
    from vidar_amd.plugin.backbones import FrozenBN, stem_bn_relu_pool
    import torch.nn.functional as F
    torch.manual_seed(0)
    bn = FrozenBN(shape[1]).cuda()
    bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2)
    x = torch.randn(*shape, device="cuda")
    with torch.no_grad():
        ref = F.max_pool2d(bn(x, relu=True), 3, stride=2, padding=1)
        out = stem_bn_relu_pool(x, bn)
    assert out is not None and out.shape == ref.shape
    assert torch.equal(out, ref)
    assert stem_bn_relu_pool(x.requires_grad_(), bn) is None           # gradients needed -> the caller's two-kernel path


@pytest.mark.parametrize("N,C,H,W,std", [(2, 40, 58, 100, 1.5), (1, 33, 29, 50, 4.0), (1, 16, 9, 70, 8.0), (3, 5, 7, 3, 1.0)])
def test_col2im_lds_window_gather_equals_global_gather(N, C, H, W, std):
    """the two grad_x gathers of 3x3 / stride 1 layers (vidar_dcn_set_variant): per-tap LDS window of grad_cols + global
    loads for sources beyond the 4-pixel halo (std 4 / 8: most of them) against 4-byte global gathers everywhere; NaN and
    far-outside offsets included (their samples have no entries)"""
    from vidar_amd._lib import lib
    from vidar_amd.plugin.backbones import dcn_col2im
    g = torch.Generator().manual_seed(C + W)
    x = torch.randn(N, C, H, W, generator=g).cuda()
    off = torch.randn(N, 18, H, W, generator=g) * std
    off[0, 4, 0, :] = float("nan"); off[0, 7, -1, -1] = 1e4; off[0, 0, :, 0] = -3.0 * std
    off = off.cuda()
    mask = torch.rand(N, 9, H, W, generator=g).cuda()
    gcols = torch.randn(N, C * 9, H * W, generator=g).cuda()
    outs = []
    for variant in (1, 0):
        prev = lib().vidar_dcn_set_variant(variant)
        try:
            outs.append(dcn_col2im(gcols, x, off, mask, 3, 3, 1, 1, 1, H, W, gather=True)[0])
        finally:
            lib().vidar_dcn_set_variant(prev)
    assert torch.isfinite(outs[0]).all()
    torch.testing.assert_close(outs[0], outs[1], rtol=1e-4, atol=1e-5 * max(1.0, float(outs[1].abs().max())))
