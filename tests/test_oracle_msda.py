"""CPU: the two independent MSDA oracle formulations agree (forward and autograd backward)."""
import pytest
import torch

from oracle import msda as M

CASES = [
    (1, [(8, 8)], 50, 4),                       # single level (TSA / prediction shape family)
    (2, [(12, 20), (6, 10), (3, 5), (2, 3)], 77, 8),   # 4 FPN levels x 8 points (SCA family)
    (1, [(1, 1)], 3, 1),
]


@pytest.mark.parametrize("B,shapes,Nq,P", CASES)
def test_gather_equals_grid_sample(B, shapes, Nq, P):
    value, sh, loc, w = M.make_case(0, B, shapes, Nq, P=P, dtype=torch.float64)
    value.requires_grad_(True); loc.requires_grad_(True); w.requires_grad_(True)
    a = M.msda_gather(value, sh, loc, w)
    ga = torch.autograd.grad(a.square().sum(), [value, loc, w])
    b = M.msda_grid_sample(value, sh, loc, w)
    gb = torch.autograd.grad(b.square().sum(), [value, loc, w])
    torch.testing.assert_close(a, b, rtol=1e-10, atol=1e-10)
    # the clamp of make_case at -0.1 puts samples on pixel == -1 exactly (-0.1 * 5 - 0.5) for the 5-wide level:
    # there the two forms differ by design (test_pixel_minus_one_follows_the_cuda_kernels)
    size = torch.tensor([[int(w_), int(h_)] for h_, w_ in shapes], dtype=torch.float64).view(1, 1, 1, -1, 1, 2)
    edge = ((loc.detach() * size - 0.5) == -1).any(-1, keepdim=True)
    for i, (x, y) in enumerate(zip(ga, gb)):
        if i == 1:
            x, y = x.masked_fill(edge, 0), y.masked_fill(edge, 0)
        torch.testing.assert_close(x, y, rtol=1e-8, atol=1e-8)


def test_weights_linearity_and_padding():
    value, sh, loc, w = M.make_case(1, 1, [(5, 7)], 20, P=4, dtype=torch.float64)
    out = M.msda_gather(value, sh, loc, w)
    torch.testing.assert_close(M.msda_gather(value, sh, loc, 2 * w), 2 * out)
    far = torch.full_like(loc, 3.0)              # every sample outside -> zeros
    assert float(M.msda_gather(value, sh, far, w).abs().sum()) == 0.0


# ---- pins against reference-held code ---------------------------------------------------------------
# (1) golden vectors produced by the reference's own `dcnv3_core_pytorch` (ops_dcnv3/functions/dcnv3_func.py:147-190,
#     the in-tree restatement of this sampling arithmetic and the reference's own test oracle, ops_dcnv3/test.py),
#     tests/golden/make_msda_golden.py;  (2) the reference's CUDA kernels `dcnv3_im2col_gpu_kernel` /
#     `dcnv3_col2im_gpu_kernel_gm` (ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh:33-276, :776-839) compiled for the host.
import numpy as np
from pathlib import Path

GOLD = Path(__file__).parent / "golden"
GOLDEN = ["tsa_L1_P4", "sca_L4_P8", "kinks_L1_P4"]


def _golden(name):
    d = np.load(GOLD / f"msda_{name}.npz")
    t = lambda k: torch.from_numpy(d[k]).double()
    return d, t


@pytest.mark.parametrize("name", GOLDEN)
@pytest.mark.parametrize("fn", ["msda_gather", "msda_grid_sample"])
def test_oracle_matches_reference_dcnv3_core_pytorch_golden(name, fn):
    d, t = _golden(name)
    value, loc, w = t("value").requires_grad_(True), t("loc").requires_grad_(True), t("w").requires_grad_(True)
    out = getattr(M, fn)(value, torch.from_numpy(d["shapes"]), loc, w)
    torch.testing.assert_close(out, t("out"), rtol=1e-11, atol=1e-11)
    gv, gl, gw = torch.autograd.grad((out * t("gout")).sum(), [value, loc, w])
    torch.testing.assert_close(gv, t("grad_value"), rtol=1e-10, atol=1e-11)
    torch.testing.assert_close(gw, t("grad_w"), rtol=1e-10, atol=1e-11)
    torch.testing.assert_close(gl, t("grad_loc"), rtol=1e-9, atol=1e-9)


def _as_dcnv3_level(value_l, loc_l, w_l, H, W):
    """One MSDA level as the operands of the reference's dcnv3 kernels with a kh x kw = 1 x P kernel window,
    stride 1, pad 0, dilation 1, offset_scale 1 and a 1 x Nq output row: the kernel computes
    loc_w = p0_w_ + (i + offset_w), loc_h = p0_h_ + (0 + offset_h) with p0_w_ = output column (cuh:245-260), so
    offset = wanted pixel coordinate - column - i.  Pixel coordinate of a [0,1] location: loc * size - 0.5."""
    B, Nq, G, P, _ = loc_l.shape
    x = loc_l[..., 0] * W - 0.5
    y = loc_l[..., 1] * H - 0.5
    col = torch.arange(Nq, dtype=torch.float64).view(1, Nq, 1, 1)
    p0w = (P - 1) // 2                                  # ((dilation_w*(kernel_w-1))>>1), cuh:231; times offset_scale = 1 at :245
    i = torch.arange(P, dtype=torch.float64).view(1, 1, 1, P)
    off = torch.stack([x - (col + p0w - p0w) - i, y], -1)            # (w, h) pairs, point p = i (kh = 1)
    inp = value_l.reshape(B, H, W, -1).contiguous()
    return inp, off.reshape(B, 1, Nq, G * P * 2).contiguous(), w_l.reshape(B, 1, Nq, G * P).contiguous()


@pytest.mark.parametrize("name", GOLDEN)
def test_oracle_matches_reference_dcnv3_kernels_host_build(name, ref_modules):
    ref = ref_modules("ref_dcnv3")
    d, t = _golden(name)
    value, loc, w, gout = t("value"), t("loc"), t("w"), t("gout")
    shapes = [(int(a), int(b)) for a, b in d["shapes"]]
    B, Nq, G, L, P, _ = loc.shape
    C = value.shape[-1]
    v = value.clone().requires_grad_(True); lo = loc.clone().requires_grad_(True); ww = w.clone().requires_grad_(True)
    o = M.msda_gather(v, torch.tensor(shapes), lo, ww)
    gv, gl, gw = torch.autograd.grad((o * gout).sum(), [v, lo, ww])
    out = 0
    start = 0
    for l, (H, W) in enumerate(shapes):
        inp, off, mask = _as_dcnv3_level(value[:, start:start + H * W], loc[:, :, :, l], w[:, :, :, l], H, W)
        # the kernel window is 1 x P here, so it needs a (fictitious) output row of Nq columns: height_out = 1
        out_l = ref.im2col(inp, off, mask, 1, P, 1, 0, 1, G, C, 1.0)
        out = out + out_l.reshape(B, Nq, G * C)
        gi, go, gm = ref.col2im(gout.reshape(B, 1, Nq, G * C).contiguous(), inp, off, mask, 1, P, 1, 0, 1, G, C, 1.0)
        torch.testing.assert_close(gv[:, start:start + H * W], gi.reshape(B, H * W, G, C), rtol=1e-10, atol=1e-11)
        torch.testing.assert_close(gw[:, :, :, l], gm.reshape(B, Nq, G, P), rtol=1e-10, atol=1e-11)
        go = go.reshape(B, Nq, G, P, 2) * torch.tensor([W, H], dtype=torch.float64)   # d/d loc = d/d pixel * size
        torch.testing.assert_close(gl[:, :, :, l], go, rtol=1e-9, atol=1e-9)
        start += H * W
    torch.testing.assert_close(o, out, rtol=1e-11, atol=1e-11)
    torch.testing.assert_close(out, t("out"), rtol=1e-11, atol=1e-11)      # the two reference forms agree too


def test_pixel_minus_one_follows_the_cuda_kernels(ref_modules):
    """Pixel coordinate == -1 exactly: the value is 0 either way; the reference's CUDA kernels skip the sample
    (`loc > -1`, dcnv3_im2col_cuda.cuh:262-263, :825-826) so its location gradient is 0, while grid_sample's
    autograd differentiates the zero-weight corner.  `msda_gather` (what the HIP kernels are compared with) follows
    the kernels."""
    ref = ref_modules("ref_dcnv3")
    H, W, G, C, P = 4, 8, 2, 4, 2
    g = torch.Generator().manual_seed(5)
    value = torch.randn(1, H * W, G, C, generator=g, dtype=torch.float64)
    loc = torch.rand(1, 6, G, 1, P, 2, generator=g, dtype=torch.float64)
    loc[0, :3, :, 0, :, 0] = -0.5 / W                                   # x == -1 exactly (power-of-two W)
    loc[0, 3:, :, 0, :, 1] = -0.5 / H                                   # y == -1
    w = torch.rand(1, 6, G, 1, P, generator=g, dtype=torch.float64)
    gout = torch.randn(1, 6, G * C, generator=g, dtype=torch.float64)
    lo = loc.clone().requires_grad_(True)
    o = M.msda_gather(value, torch.tensor([[H, W]]), lo, w)
    gl, = torch.autograd.grad((o * gout).sum(), [lo])
    assert float(o.abs().max()) == 0.0 and float(gl.abs().max()) == 0.0
    inp, off, mask = _as_dcnv3_level(value, loc[:, :, :, 0], w[:, :, :, 0], H, W)
    assert float(ref.im2col(inp, off, mask, 1, P, 1, 0, 1, G, C, 1.0).abs().max()) == 0.0
    _, go, _ = ref.col2im(gout.reshape(1, 1, 6, G * C), inp, off, mask, 1, P, 1, 0, 1, G, C, 1.0)
    assert float(go.abs().max()) == 0.0
    lo2 = loc.clone().requires_grad_(True)
    g2, = torch.autograd.grad((M.msda_grid_sample(value, torch.tensor([[H, W]]), lo2, w) * gout).sum(), [lo2])
    assert float(g2.abs().max()) > 0.0                                   # the grid_sample form does not
