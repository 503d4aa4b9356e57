"""A stride-2 caffe-style bottleneck reads the same strided subsample of its input twice (conv1 and the downsample
convolution): `Bottleneck` gathers it once.  Same values, same gradients -- bit for bit -- as the two-copy form."""
import pytest
import torch

from vidar_amd.plugin import backbones as B


def _run(share, monkeypatch):
    monkeypatch.setattr(B, "_SHARE_SUBSAMPLE", share)
    torch.manual_seed(0)
    down = torch.nn.Sequential(B.Conv1x1(16, 32, 1, stride=2, bias=False), B.FrozenBN(32, False))
    blk = B.Bottleneck(16, 8, stride=2, downsample=down, style="caffe")
    for m in blk.modules():
        if isinstance(m, B.FrozenBN):
            m.running_mean.normal_(); m.running_var.uniform_(0.5, 2); m.weight.data.normal_(); m.bias.data.normal_()
    x = torch.randn(2, 16, 9, 11, requires_grad=True)
    shares = blk._shares_subsample(x)
    y = blk(x)
    y.square().sum().backward()
    return y.detach(), x.grad, [p.grad for p in blk.parameters() if p.grad is not None], shares


def test_shared_subsample_is_bit_identical(monkeypatch):
    monkeypatch.setattr(B.Conv1x1, "any_device", True)        # the GEMM form of the 1x1 convolutions, on the CPU
    a = _run(True, monkeypatch)
    b = _run(False, monkeypatch)
    assert a[3] and not b[3]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert len(a[2]) == len(b[2]) == 4
    for p, q in zip(a[2], b[2]):
        assert torch.equal(p, q)


def test_no_sharing_without_a_strided_conv1(monkeypatch):
    monkeypatch.setattr(B.Conv1x1, "any_device", True)
    blk = B.Bottleneck(32, 8, stride=1, downsample=None, style="caffe")
    assert not blk._shares_subsample(torch.randn(1, 32, 4, 4))
    down = torch.nn.Sequential(B.Conv1x1(16, 32, 1, stride=2, bias=False), B.FrozenBN(32, False))
    blk = B.Bottleneck(16, 8, stride=2, downsample=down, style="pytorch")      # stride on conv2: conv1 reads all of x
    assert not blk._shares_subsample(torch.randn(1, 16, 4, 4))


def test_identity_shortcut_gradient_accumulates_inside_the_convolutions_backward():
    """Bottleneck with an identity shortcut: handing the block input through the first 1x1 convolution's Function
    (backbones._Conv1x1Identity: grad_x = W^T grad_out + grad_identity in one GEMM) gives the values and gradients of
    the plain autograd graph (conv1 and the shortcut as two consumers of x)"""
    import torch
    from vidar_amd.plugin import backbones as B
    torch.manual_seed(0)
    B.Conv1x1.any_device = True
    try:
        blk = B.Bottleneck(64, 16, style="pytorch").train()
        assert blk.downsample is None
        for p in blk.parameters():
            p.requires_grad_(True)
        x = torch.randn(2, 64, 6, 5, requires_grad=True)
        outs = {}
        for on in (True, False):
            B._SHORTCUT_ACCUMULATE = on
            y = blk(x)
            g = torch.autograd.grad((y * torch.arange(y.numel()).view_as(y).float().sin()).sum(), [x] + list(blk.parameters()),
                                    allow_unused=True)
            outs[on] = (y.detach(), g)
        B._SHORTCUT_ACCUMULATE = True
        assert B.conv1x1_with_identity_ok(blk.conv1, x)                # the form really ran in the first pass
        torch.testing.assert_close(outs[True][0], outs[False][0], rtol=0, atol=0)
        for a, b in zip(outs[True][1], outs[False][1]):
            if b is None:
                assert a is None
            else:
                torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
    finally:
        B.Conv1x1.any_device = False
        B._SHORTCUT_ACCUMULATE = True
