"""Golden vectors that PIN the multi-scale deformable attention oracle to reference-held code.

The reference reaches MSDA through mmcv (not vendored), but it carries the same arithmetic in-tree
as DCNv3: `dcnv3_core_pytorch` (projects/mmdet3d_plugin/bevformer/backbones/ops_dcnv3/functions/
dcnv3_func.py:147-190) = per group `F.grid_sample(bilinear, zeros, align_corners=False)` at
`2*loc-1`, times a per-point mask, summed over the kh*kw points -- one MSDA level with
heads = group, points = kh*kw, weights = mask.  It is also the reference's OWN test oracle for its
CUDA kernels (ops_dcnv3/test.py:33-61).

A case = L levels evaluated by the reference function one level at a time; the multi-level MSDA
output is their sum.  The wanted sampling locations are reached by solving the reference's own
location expression (dcnv3_func.py:166-167) for `offset`; the locations that are STORED are the ones
that expression then yields.  Gradients come from autograd through the reference function:
d/d loc = d/d offset * spatial_norm / offset_scale (chain rule of :166-167).

Run in the build container:   python tests/golden/make_msda_golden.py
"""
import sys
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).parent
sys.path.insert(0, str(HERE))
import ref_import  # noqa: E402

sys.modules.setdefault("DCNv3", types.ModuleType("DCNv3"))      # the compiled extension; unused here
F = ref_import.load_file("ref_dcnv3_func", ref_import.PLUGIN / "bevformer/backbones/ops_dcnv3/functions/dcnv3_func.py")

HEADS, CH = 8, 32


def level_case(g, H, W, kh, kw, B, Nq_hw, kind):
    """One level: returns the reference call's operands (fp64) and the locations it samples."""
    Ho, Wo = Nq_hw
    assert Ho == H - kh + 1 and Wo == W - kw + 1                 # dcnv3_func.py:93-94, pad 0, stride 1
    P = kh * kw
    value = torch.randn(B, H, W, HEADS * CH, generator=g).double()
    # wanted locations in [0,1] units: reference points +- spread, some outside, some on pixel centres
    ref = torch.rand(B, Ho * Wo, 1, 1, 2, generator=g) * 1.3 - 0.15
    want = ref + (torch.rand(B, Ho * Wo, HEADS, P, 2, generator=g) * 2 - 1) * 0.08
    if kind == "kinks":                                           # a third of the samples exactly on pixel centres / far outside
        sel = torch.rand(B, Ho * Wo, HEADS, P, 1, generator=g)
        # (pixel -1 exactly is kept out: there grid_sample's autograd and the CUDA kernels differ, see
        #  tests/test_oracle_msda.py::test_pixel_minus_one_follows_the_cuda_kernels)
        centre = (torch.floor(want * torch.tensor([W, H])).clamp(min=0) + 0.5) / torch.tensor([W, H])
        want = torch.where(sel < 0.33, centre, want)
        want = torch.where(sel > 0.9, want + 2.0, want)
    want = want.float().double()
    r = F._get_reference_points(value.shape, "cpu", kh, kw, 1, 1, 0, 0, 1, 1)
    grid = F._generate_dilation_grids(value.shape, kh, kw, 1, 1, HEADS, "cpu")
    norm = torch.tensor([W, H]).reshape(1, 1, 1, 2).repeat(1, 1, 1, HEADS * P)
    base = (r + grid * 1.0).repeat(B, 1, 1, 1, 1).flatten(3, 4)            # [B,Ho,Wo,HEADS*P*2]
    offset = ((want.reshape(B, Ho, Wo, HEADS * P * 2) - base) * norm).float().double()
    loc = (base + offset * 1.0 / norm).reshape(B, Ho * Wo, HEADS, P, 2)     # what :166-167 computes
    mask = torch.softmax(torch.randn(B, Ho * Wo, HEADS, P, generator=g), -1).double().reshape(B, Ho, Wo, HEADS * P)
    return value, offset, mask, loc, norm


CASES = {
    # name: (B, kh, kw, [(H, W)...] with equal (H-kh+1)*(W-kw+1), kind)
    "tsa_L1_P4": (2, 2, 2, [(9, 13)], "plain"),                  # TemporalSelfAttention / Prediction family
    "sca_L4_P8": (1, 2, 4, [(11, 9), (7, 13), (4, 23), (3, 33)], "plain"),   # SpatialCrossAttention family
    "kinks_L1_P4": (1, 2, 2, [(8, 16)], "kinks"),                # EXACT pixel-centre kinks (power-of-two level: every step of :166-190 is exact) + far-outside samples
}

for name, (B, kh, kw, levels, kind) in CASES.items():
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    P = kh * kw
    per = []
    for (H, W) in levels:
        per.append(level_case(g, H, W, kh, kw, B, (H - kh + 1, W - kw + 1), kind))
    Nq = (levels[0][0] - kh + 1) * (levels[0][1] - kw + 1)
    assert all((H - kh + 1) * (W - kw + 1) == Nq for H, W in levels)
    leaves = []
    out = 0
    for (value, offset, mask, loc, norm) in per:
        value.requires_grad_(True); offset.requires_grad_(True); mask.requires_grad_(True)
        o = F.dcnv3_core_pytorch(value, offset, mask, kh, kw, 1, 1, 0, 0, 1, 1, HEADS, CH, 1.0)
        out = out + o.reshape(B, Nq, HEADS * CH)
        leaves += [value, offset, mask]
    gout = torch.randn(B, Nq, HEADS * CH, generator=g).double()
    grads = torch.autograd.grad((out * gout).sum(), leaves)
    L = len(levels)
    value = torch.cat([p[0].detach().reshape(B, -1, HEADS, CH) for p in per], 1)
    loc = torch.stack([p[3] for p in per], 3)                               # [B,Nq,HEADS,L,P,2]
    w = torch.stack([p[2].detach().reshape(B, Nq, HEADS, P) for p in per], 3)
    g_value = torch.cat([grads[3 * l].reshape(B, -1, HEADS, CH) for l in range(L)], 1)
    g_loc = torch.stack([(grads[3 * l + 1] * per[l][4]).reshape(B, Nq, HEADS, P, 2) for l in range(L)], 3)
    g_w = torch.stack([grads[3 * l + 2].reshape(B, Nq, HEADS, P) for l in range(L)], 3)
    # point order of the reference: p = i*kh + j (w-offset outer) -- irrelevant to MSDA (a sum over p)
    np.savez_compressed(HERE / f"msda_{name}.npz", shapes=np.array(levels, np.int64),
                        value=value.numpy().astype(np.float32), loc=loc.numpy(), w=w.numpy().astype(np.float32),
                        gout=gout.numpy().astype(np.float32), out=out.detach().numpy(),
                        grad_value=g_value.numpy(), grad_loc=g_loc.numpy(), grad_w=g_w.numpy())
    print(name, "Nq", Nq, "L", L, "P", P, "out", tuple(out.shape), float(out.abs().mean()),
          "inside frac", float(((loc > 0) & (loc < 1)).all(-1).double().mean()))
