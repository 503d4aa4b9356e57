"""GPU: LayerNorm(dropout(x) + residual) fused kernels (csrc/norm_fuse.hip) against torch's three ops:
p = 0 exactly the same function (values + all four gradients); p > 0: the kept set is a deterministic
function of the seed, has the right rate, forward and backward use the same mask."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _setup(rows, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, 256, generator=g).cuda().requires_grad_(True)
    r = torch.randn(rows, 256, generator=g).cuda().requires_grad_(True)
    norm = nn.LayerNorm(256).cuda()
    norm.weight.data.uniform_(0.5, 1.5, generator=None); norm.bias.data.normal_()
    gy = torch.randn(rows, 256, generator=g).cuda()
    return x, r, norm, gy


@pytest.mark.parametrize("rows", [1, 7, 64, 40000])
def test_without_dropout_equals_layer_norm_of_the_sum(rows):
    from vidar_amd.plugin.bricks import drop_add_layernorm
    x, r, norm, gy = _setup(rows)
    y = drop_add_layernorm(x, r, norm, 0.1, training=False)
    ref = norm(x + r)
    torch.testing.assert_close(y, ref, rtol=1e-5, atol=1e-5)
    a = torch.autograd.grad(y, [x, r, norm.weight, norm.bias], gy)
    b = torch.autograd.grad(ref, [x, r, norm.weight, norm.bias], gy)
    for u, v, nm in zip(a, b, ["x", "residual", "gamma", "beta"]):
        torch.testing.assert_close(u, v, rtol=2e-4, atol=2e-5 * max(1.0, float(v.abs().max())), msg=lambda m: nm + m)


def test_dropout_mask_is_consistent_between_forward_and_backward():
    from vidar_amd.plugin import bricks
    x, r, norm, gy = _setup(5000)
    torch.manual_seed(7)
    state = torch.get_rng_state()
    y = bricks.drop_add_layernorm(x.view(1, 5000, 256), r.view(1, 5000, 256), norm, 0.3, training=True).view(5000, 256)
    gx, gr = torch.autograd.grad(y, [x, r], gy)
    keep = gx != 0
    rate = float(keep.float().mean())
    assert abs(rate - 0.7) < 0.01, rate
    ref = norm(x * keep / 0.7 + r)                     # the same kept set reproduces the forward ...
    torch.testing.assert_close(y, ref, rtol=1e-5, atol=1e-5)
    bx, br = torch.autograd.grad(ref, [x, r], gy)      # ... and the backward
    torch.testing.assert_close(gx, bx, rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(gr, br, rtol=2e-4, atol=2e-5)
    torch.manual_seed(7)                               # reproducible under manual_seed
    y2 = bricks.drop_add_layernorm(x, r, norm, 0.3, training=True)
    assert torch.equal(y2, y)
    y3 = bricks.drop_add_layernorm(x, r, norm, 0.3, training=True)   # next call: another mask
    assert not torch.equal(y3, y)
    # the mask seed comes out of torch's generator: restoring the RNG state (what torch.utils.checkpoint does
    # around a recomputed forward) reproduces the mask
    torch.set_rng_state(state)
    assert torch.equal(bricks.drop_add_layernorm(x, r, norm, 0.3, training=True), y)


def test_other_widths_take_the_torch_path():
    from vidar_amd.plugin.bricks import can_fuse_norm, drop_add_layernorm
    norm = nn.LayerNorm(64).cuda()
    x = torch.randn(3, 64, device="cuda"); r = torch.randn(3, 64, device="cuda")
    assert not can_fuse_norm(norm, x)
    torch.testing.assert_close(drop_add_layernorm(x, r, norm, 0.5, training=False), norm(x + r))


@pytest.mark.parametrize("rows,cols", [(40000, 256), (184950, 256), (46080, 512), (40000, 64), (40000, 1024), (7, 16), (1, 4), (0, 128),
                                       (100003, 128)])
def test_colsum_matches_fp64_column_sums(rows, cols):
    """vidar_colsum_f32: the bias gradient of the Linear layers (sum over rows), against an fp64 sum"""
    import ctypes
    from vidar_amd._lib import lib, check, ptr, stream_of
    g = torch.Generator().manual_seed(rows + cols)
    x = torch.randn(rows, cols, generator=g).cuda()
    out = torch.full((cols,), float("nan"), device="cuda")
    check(lib().vidar_colsum_f32(ptr(x), ptr(out), ctypes.c_int64(rows), cols, stream_of(x)), "colsum")
    ref = x.double().sum(0)
    torch.testing.assert_close(out.double(), ref, rtol=1e-5, atol=2e-4 * max(1.0, rows ** 0.5))
    assert lib().vidar_colsum_f32(ptr(x), ptr(out), ctypes.c_int64(rows), 12, stream_of(x)) != 0    # 3 groups: not a power of two


def test_bricks_linear_gradients_equal_nn_linear():
    """plugin.bricks.Linear = nn.Linear with the column-sum kernel for the bias gradient (same parameters / keys)"""
    from vidar_amd.plugin.bricks import Linear
    torch.manual_seed(0)
    a = Linear(256, 128).cuda(); b = nn.Linear(256, 128).cuda()
    b.load_state_dict(a.state_dict())
    assert list(a.state_dict()) == list(b.state_dict())
    x = torch.randn(3, 5000, 256, device="cuda")
    xa = x.clone().requires_grad_(True); xb = x.clone().requires_grad_(True)
    gy = torch.randn(3, 5000, 128, device="cuda")
    ya, yb = a(xa), b(xb)
    assert torch.equal(ya, yb)
    ya.backward(gy); yb.backward(gy)
    torch.testing.assert_close(xa.grad, xb.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(a.weight.grad, b.weight.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(a.bias.grad, b.bias.grad, rtol=1e-4, atol=1e-3)
    odd = Linear(256, 80).cuda()                               # 20 column groups: torch's own backward
    odd(x).sum().backward()
    assert torch.isfinite(odd.bias.grad).all()


@pytest.mark.parametrize("shape,p", [((1, 40000, 512), 0.1), ((3, 17, 64), 0.5), ((2, 5, 8), 0.0)])
def test_relu_dropout_one_pass_each_way(shape, p):
    """dropout(relu(x)) of the FFN's hidden layer (csrc/norm_fuse.hip): values are relu(x) / (1 - p) or 0, the kept
    fraction is 1 - p, the mask is a function of the seed, and the backward is exactly the forward's own mask"""
    from vidar_amd.plugin.bricks import _ReluDropout
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(*shape, device="cuda", generator=g).requires_grad_(True)
    y = _ReluDropout.apply(x, p, 12345)
    r = torch.relu(x.detach())
    kept = y.detach() != 0
    assert not (kept & (r == 0)).any()
    torch.testing.assert_close(y.detach()[kept], (r / (1.0 - p))[kept], rtol=1e-6, atol=0)
    pos = r > 0
    frac = float((kept & pos).sum()) / max(1, int(pos.sum()))
    if p == 0.0:
        assert frac == 1.0
    elif x.numel() > 10000:
        assert abs(frac - (1.0 - p)) < 0.01
    assert torch.equal(_ReluDropout.apply(x, p, 12345), y) and (p == 0.0 or not torch.equal(_ReluDropout.apply(x, p, 777), y))
    gy = torch.randn(*shape, device="cuda", generator=g)
    gx, = torch.autograd.grad(y, x, gy)
    torch.testing.assert_close(gx, torch.where(kept, gy / (1.0 - p), torch.zeros_like(gy)), rtol=1e-6, atol=0)


def test_ffn_takes_the_fused_hidden_activation_in_training_and_torch_ops_in_eval():
    from vidar_amd.plugin.bricks import FFN
    torch.manual_seed(0)
    ffn = FFN(256, 512, ffn_drop=0.1).cuda()
    x = torch.randn(1, 300, 256, device="cuda", requires_grad=True)
    ffn.eval()
    ref = x + ffn.layers[1](torch.relu(ffn.layers[0][0](x)))
    torch.testing.assert_close(ffn(x), ref, rtol=1e-5, atol=1e-5)
    ffn.train()
    torch.manual_seed(1); a = ffn(x)
    torch.manual_seed(1); b = ffn(x)
    assert torch.equal(a, b)                                  # reproducible under manual_seed
    a.sum().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all() and ffn.layers[0][0].weight.grad is not None
