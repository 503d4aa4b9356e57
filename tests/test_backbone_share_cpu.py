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
