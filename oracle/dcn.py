"""CPU oracle for modulated deformable convolution v2 -- TEST INFRASTRUCTURE.

mmcv-full 1.4.0 (`mmcv.ops.modulated_deform_conv2d`) is third party and neither vendored in
/root/reference nor installable here; this restates its published semantics with F.grid_sample
(align_corners=True maps pixel centres exactly; zero padding == the op's per-corner bounds checks).
offset [N,2K,Ho,Wo] holds (dy,dx) per tap, mask [N,K,Ho,Wo].
PINNED (round 5) as far as reference-held code reaches: the SAMPLER (bilinear taps, zero padding, base
position, stride / padding / dilation, modulation, and all three gradients) equals the reference's
in-tree deformable kernels `dcnv3_im2col_gpu_kernel` / `dcnv3_col2im_gpu_kernel_gm`
(ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh) compiled for the host -- tests/test_oracle_dcn.py.  What
stays recalled from mmcv is the channel order of `offset` only."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def modulated_deform_conv2d(x, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1):
    N, C, H, W = x.shape
    Cout, _, kh, kw = weight.shape
    Ho, Wo = offset.shape[2:]
    ys = torch.arange(Ho, dtype=x.dtype).view(1, Ho, 1) * stride - padding
    xs = torch.arange(Wo, dtype=x.dtype).view(1, 1, Wo) * stride - padding
    out = 0
    for t in range(kh * kw):
        i, j = t // kw, t % kw
        h = ys + i * dilation + offset[:, 2 * t]
        w = xs + j * dilation + offset[:, 2 * t + 1]
        grid = torch.stack((2 * w / max(W - 1, 1) - 1, 2 * h / max(H - 1, 1) - 1), -1)
        s = F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
        s = s * mask[:, t:t + 1]
        out = out + torch.einsum("oc,nchw->nohw", weight[:, :, i, j], s)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out
