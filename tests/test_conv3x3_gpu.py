"""csrc/conv3x3_mfma.hip (the DCNv2 `conv_offset`: 3x3, stride 1, pad 1, <= 32 output channels as an implicit GEMM on the
fp32 matrix cores) against an fp64 convolution of the same inputs on the CPU: the floating-point reference of a
floating-point kernel, tolerance 1e-5 of the output scale (fp32 products are exact in the matrix core; only the order of the
C * 9 additions differs).  Through the C-ABI, and through the autograd Function the backbone uses."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def call(x, w, b):
    from vidar_amd._lib import lib, check, ptr, stream_of, workspace
    N, C, H, W = x.shape
    out = torch.full((N, w.shape[0], H, W), float("nan"), device=x.device)
    ws, ws_ptr, ws_bytes = workspace(lib().vidar_conv3x3_few_workspace_bytes, C, like=x)
    check(lib().vidar_conv3x3_few_f32(ptr(x), ptr(w), ptr(b), ptr(out), N, C, H, W, w.shape[0], ws_ptr, ws_bytes,
                                      stream_of(x)), "conv3x3_few")
    torch.cuda.synchronize()
    return out


def reference(x, w, b):
    return F.conv2d(x.double().cpu(), w.double().cpu(), b.double().cpu() if b is not None else None, padding=1)


@pytest.mark.parametrize("N,C,H,W,cout,with_bias", [
    (6, 256, 58, 100, 27, True),       # stage 3 of the backbone, current frame
    (2, 512, 29, 50, 27, True),        # stage 4
    (1, 8, 5, 7, 27, True),            # a tile that spans rows, one partial tile
    (2, 16, 9, 50, 32, False),         # full 32 rows, no bias
    (1, 8, 3, 191, 3, True),           # the widest supported row
    (3, 8, 2, 1, 5, True),             # one-pixel rows: left and right padding on the same pixel
    (1, 24, 131, 33, 1, True),         # many tiles per image, a single output
])
def test_forward_matches_fp64_convolution(N, C, H, W, cout, with_bias):
    g = torch.Generator().manual_seed(N * 1000 + C + H * W)
    x = torch.randn(N, C, H, W, generator=g).cuda()
    w = (torch.randn(cout, C, 3, 3, generator=g) / (3 * C ** 0.5)).cuda()
    b = torch.randn(cout, generator=g).cuda() if with_bias else None
    got = call(x, w, b).cpu().double()
    ref = reference(x, w, b)
    assert torch.isfinite(got).all()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 1e-5 * scale


def test_empty_batch_and_limits():
    from vidar_amd._lib import lib, ptr, stream_of
    x = torch.randn(1, 8, 4, 4, device="cuda")
    w = torch.randn(27, 8, 3, 3, device="cuda")
    out = torch.empty(1, 27, 4, 4, device="cuda")
    ws = torch.empty(8 * 9 * 32 * 4, dtype=torch.uint8, device="cuda")
    f = lib().vidar_conv3x3_few_f32
    n = ctypes.c_size_t(ws.numel())
    assert f(ptr(x), ptr(w), None, ptr(out), 0, 8, 4, 4, 27, ptr(ws), n, stream_of(x)) == 0           # nothing to do
    assert f(ptr(x), ptr(w), None, ptr(out), 1, 8, 4, 4, 33, ptr(ws), n, stream_of(x)) == -22         # > 32 outputs
    assert f(ptr(x), ptr(w), None, ptr(out), 1, 12, 4, 4, 27, ptr(ws), n, stream_of(x)) == -22        # C % 8
    assert f(ptr(x), ptr(w), None, ptr(out), 1, 8, 4, 192, 27, ptr(ws), n, stream_of(x)) == -22       # row too wide
    assert f(ptr(x), ptr(w), None, ptr(out), 1, 8, 4, 4, 27, ptr(ws), ctypes.c_size_t(16), stream_of(x)) == -22
    assert f(ptr(x), ptr(w), None, ptr(out), 1, 8, 4, 4, 27, None, n, stream_of(x)) == -22
    g = lib().vidar_conv3x3_few_workspace_bytes
    g.restype = ctypes.c_size_t
    assert g(256) == 256 * 9 * 32 * 4 and g(0) == 0


def test_function_gradients_are_the_library_convolution_s():
    from vidar_amd.plugin.backbones import _Conv3x3Few
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(2, 16, 11, 13, generator=g).cuda()
    w0 = (torch.randn(27, 16, 3, 3, generator=g) / 12).cuda()
    b0 = torch.randn(27, generator=g).cuda()
    go = torch.randn(2, 27, 11, 13, generator=g).cuda()
    grads = []
    for own in (True, False):
        x, w, b = (t.clone().requires_grad_(True) for t in (x0, w0, b0))
        y = _Conv3x3Few.apply(x, w, b) if own else F.conv2d(x, w, b, padding=1)
        y.backward(go)
        grads.append((y.detach(), x.grad, w.grad, b.grad))
    for a, r in zip(*grads):
        assert float((a - r).abs().max()) <= 2e-5 * float(r.abs().max())


def test_deform_conv_pack_uses_it_and_agrees_with_the_library_path(monkeypatch):
    from vidar_amd._lib import TIMER
    from vidar_amd.plugin import backbones as B
    torch.manual_seed(3)
    m = B.ModulatedDeformConv2dPack(16, 16, 3, padding=1, bias=False).cuda()
    torch.nn.init.normal_(m.conv_offset.weight, std=0.05)
    torch.nn.init.normal_(m.conv_offset.bias, std=0.05)
    x = torch.randn(2, 16, 9, 10, device="cuda")
    outs = []
    for own in (True, False):
        monkeypatch.setattr(B, "_CONV_OFFSET_OWN", own)
        TIMER.enabled = True
        TIMER.reset()
        try:
            outs.append(m(x).detach())
            assert ("conv3x3_few" in TIMER.records) == own
        finally:
            TIMER.enabled = False
            TIMER.reset()
    assert float((outs[0] - outs[1]).abs().max()) <= 1e-4 * float(outs[1].abs().max())
