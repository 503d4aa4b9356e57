"""CPU oracle for multi-scale deformable attention -- TEST INFRASTRUCTURE.

The reference reaches this op through mmcv-full==1.4.0 (README.md:103), which is neither vendored
under /root/reference nor installable here; its call sites are spatial_cross_attention.py:392-394,
temporal_self_attention.py:249-252, vidar_decoder.py:507-509:
    out[b,q,h*C+c] = sum_{l,p} w[b,q,h,l,p] * bilinear(value_l[b,:,h,c], loc[b,q,h,l,p])
    pixel = loc * (W_l, H_l) - 0.5, zero padding outside  (grid_sample, align_corners=False).
PINNED (round 5) against the copy of this arithmetic the reference DOES hold in-tree, DCNv3
(projects/mmdet3d_plugin/bevformer/backbones/ops_dcnv3): one MSDA level = `dcnv3_core_pytorch`
(functions/dcnv3_func.py:147-190) with group = heads, kh*kw = points, mask = weights, and the CUDA
kernels `dcnv3_im2col_gpu_kernel` / `dcnv3_col2im_gpu_kernel_gm` with their bilinear device functions
(src/cuda/dcnv3_im2col_cuda.cuh:33-276, :776-839) -- Deformable-DETR's kernel under another name,
with the same `loc > -1 && loc < size` admission test and the same grad_loc / grad_w formulas.
tests/test_oracle_msda.py checks both formulations below against golden vectors made by the first
(tests/golden/make_msda_golden.py, forward + the three gradients) and against the second compiled
for the host (oracle/build_ref.py -> oracle/_ref/ref_dcnv3.so).
Two formulations: `msda_gather` (explicit 4-corner gather with the CUDA kernels' admission test, any
float dtype, differentiable by autograd) and `msda_grid_sample` (per-level F.grid_sample, the form of
the reference's CPU branch).  They differ only on the measure-zero set pixel == -1 exactly, where
grid_sample's autograd still returns a location gradient and the CUDA kernels return none."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def msda_gather(value, shapes, loc, w):
    """value [B,Nv,H,C], shapes [[h,w],...], loc [B,Nq,H,L,P,2], w [B,Nq,H,L,P] -> [B,Nq,H*C]"""
    B, Nv, H, C = value.shape
    _, Nq, _, L, P, _ = loc.shape
    out = value.new_zeros(B, Nq, H, C)
    start = 0
    bi = torch.arange(B).view(B, 1, 1, 1)
    hi = torch.arange(H).view(1, 1, H, 1)
    for l, (Hl, Wl) in enumerate([(int(a), int(b)) for a, b in shapes]):
        v = value[:, start:start + Hl * Wl]                      # [B, Hl*Wl, H, C]
        start += Hl * Wl
        x = loc[:, :, :, l, :, 0] * Wl - 0.5                     # [B,Nq,H,P]
        y = loc[:, :, :, l, :, 1] * Hl - 0.5
        x0 = torch.floor(x); y0 = torch.floor(y)
        lx = x - x0; ly = y - y0
        inside = (x > -1) & (y > -1) & (x < Wl) & (y < Hl)       # dcnv3_im2col_cuda.cuh:262-263
        acc = 0
        for dy, wy in ((0, 1 - ly), (1, ly)):
            for dx, wx in ((0, 1 - lx), (1, lx)):
                xi = (x0 + dx).long(); yi = (y0 + dy).long()
                ok = (xi >= 0) & (xi < Wl) & (yi >= 0) & (yi < Hl)
                idx = (yi.clamp(0, Hl - 1) * Wl + xi.clamp(0, Wl - 1))
                g = v[bi, idx, hi]                                # [B,Nq,H,P,C]
                acc = acc + g * (wy * wx * (ok & inside)).unsqueeze(-1)
        out = out + (acc * w[:, :, :, l].unsqueeze(-1)).sum(3)
    return out.reshape(B, Nq, H * C)


def msda_grid_sample(value, shapes, loc, w):
    B, Nv, H, C = value.shape
    _, Nq, _, L, P, _ = loc.shape
    start = 0
    cols = []
    for l, (Hl, Wl) in enumerate([(int(a), int(b)) for a, b in shapes]):
        v = value[:, start:start + Hl * Wl].permute(0, 2, 3, 1).reshape(B * H, C, Hl, Wl)
        start += Hl * Wl
        g = (2 * loc[:, :, :, l] - 1).permute(0, 2, 1, 3, 4).reshape(B * H, Nq, P, 2)
        cols.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    s = torch.stack(cols, 3)                                      # [B*H, C, Nq, L, P]
    ww = w.permute(0, 2, 1, 3, 4).reshape(B * H, 1, Nq, L, P)
    o = (s * ww).sum((3, 4))                                      # [B*H, C, Nq]
    return o.view(B, H, C, Nq).permute(0, 3, 1, 2).reshape(B, Nq, H * C)


def level_start_index(shapes):
    sizes = torch.tensor([int(h) * int(w) for h, w in shapes])
    return torch.cat([sizes.new_zeros(1), sizes.cumsum(0)[:-1]])


def make_case(seed, B, shapes, Nq, H=8, C=32, P=4, dtype=torch.float32, spread=0.05):
    """Synthetic operands per SURVEY §8d: locations = reference point +- offsets, clipped to
    [-0.1, 1.1] so that zero padding is exercised; softmaxed weights."""
    g = torch.Generator().manual_seed(seed)
    L = len(shapes)
    Nv = sum(int(h) * int(w) for h, w in shapes)
    value = torch.randn(B, Nv, H, C, generator=g, dtype=dtype)
    ref = torch.rand(B, Nq, 1, 1, 1, 2, generator=g, dtype=dtype) * 1.2 - 0.1
    loc = (ref + (torch.rand(B, Nq, H, L, P, 2, generator=g, dtype=dtype) * 2 - 1) * spread).clamp(-0.1, 1.1)
    w = torch.softmax(torch.randn(B, Nq, H, L * P, generator=g, dtype=dtype), -1).view(B, Nq, H, L, P)
    return value, torch.tensor(shapes, dtype=torch.int64), loc.contiguous(), w.contiguous()
