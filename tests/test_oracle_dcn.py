"""CPU: the DCNv2 oracle's SAMPLER (bilinear taps, zero padding, base position = out * stride - pad + tap * dilation,
modulation mask) against the reference's in-tree deformable kernels compiled for the host (oracle/_ref/ref_dcnv3:
`dcnv3_im2col_gpu_kernel` / `dcnv3_col2im_gpu_kernel_gm`, ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh:217-276, :776-839).

With one group, offset_scale 1 and per-tap identity weights a modulated deformable convolution v2 IS the DCNv3 core:
out[c] = sum_t mask_t * bilinear(x[c], base_t + offset_t).  What stays recalled from mmcv (third party, not vendored) is
only the channel order of `offset` ((dy, dx) per tap, taps row-major) -- the tap weights are an ordinary einsum, and
tests/test_dcn_gpu.py::test_zero_offsets_equal_plain_convolution ties the zero-offset case to F.conv2d."""
import pytest
import torch

from oracle import dcn as D


@pytest.mark.parametrize("N,C,H,W,stride,pad,dil", [(2, 5, 9, 11, 1, 1, 1), (1, 4, 10, 13, 2, 1, 1), (1, 3, 8, 8, 1, 2, 2)])
def test_dcnv2_sampler_matches_reference_dcnv3_kernels(N, C, H, W, stride, pad, dil, ref_modules):
    ref = ref_modules("ref_dcnv3")
    k = 3
    K = k * k
    Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    g = torch.Generator().manual_seed(N * 100 + H)
    x = torch.randn(N, C, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    offset = (torch.randn(N, 2 * K, Ho, Wo, generator=g, dtype=torch.float64) * 1.5).requires_grad_(True)
    mask = torch.rand(N, K, Ho, Wo, generator=g, dtype=torch.float64, requires_grad=True)
    weight = torch.zeros(C, C, k, k, dtype=torch.float64)
    weight[torch.arange(C), torch.arange(C)] = 1.0                       # identity on every tap
    out = D.modulated_deform_conv2d(x, offset, mask, weight, None, stride, pad, dil)
    gout = torch.randn(out.shape, generator=g, dtype=torch.float64)
    gx, go, gm = torch.autograd.grad((out * gout).sum(), [x, offset, mask])

    # the same operands in the reference kernels' layout: channel-last input, point p = i_w * kh + j_h holding
    # (offset_w, offset_h) (cuh:247-253); oracle tap t = i_h * kw + j_w holding (dy, dx)
    t_of_p = [j * k + i for i in range(k) for j in range(k)]             # p = i*k + j  ->  t = j*k + i
    o = offset.detach().view(N, K, 2, Ho, Wo)[:, t_of_p]                 # [N, P, (dy,dx), Ho, Wo]
    off3 = torch.stack([o[:, :, 1], o[:, :, 0]], 2).permute(0, 3, 4, 1, 2).reshape(N, Ho, Wo, K * 2).contiguous()
    m3 = mask.detach()[:, t_of_p].permute(0, 2, 3, 1).contiguous()
    x3 = x.detach().permute(0, 2, 3, 1).contiguous()
    r = ref.im2col(x3, off3, m3, k, k, stride, pad, dil, 1, C, 1.0)
    torch.testing.assert_close(out.detach(), r.permute(0, 3, 1, 2), rtol=1e-11, atol=1e-11)
    ri, ro, rm = ref.col2im(gout.permute(0, 2, 3, 1).contiguous(), x3, off3, m3, k, k, stride, pad, dil, 1, C, 1.0)
    torch.testing.assert_close(gx, ri.permute(0, 3, 1, 2), rtol=1e-10, atol=1e-11)
    inv = [t_of_p.index(t) for t in range(K)]
    torch.testing.assert_close(gm, rm.permute(0, 3, 1, 2)[:, inv], rtol=1e-10, atol=1e-11)
    ro = ro.view(N, Ho, Wo, K, 2).permute(0, 3, 4, 1, 2)[:, inv]          # [N, t, (w,h), Ho, Wo]
    torch.testing.assert_close(go.view(N, K, 2, Ho, Wo), torch.stack([ro[:, :, 1], ro[:, :, 0]], 2), rtol=1e-9, atol=1e-10)
