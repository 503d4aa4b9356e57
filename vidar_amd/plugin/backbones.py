"""Image branch of ViDAR: ResNet (caffe style, frozen BatchNorm, DCNv2 in the last stages) + FPN,
with the registry names / kwargs / parameter names of mmdet 2.14 `ResNet`, `FPN` and mmcv
`ModulatedDeformConv2dPack` (third party, not vendored in the reference; config
vidar_1_8_nusc_1future.py:88-106), so `pretrained/r101_dcn_fcos3d_pretrain.pth`-style checkpoints
map key for key.  Plain convolutions run on MIOpen through torch; the deformable sampling is the
gfx950 kernel pair of csrc/dcn.hip and the deformable convolution itself is a hipBLASLt GEMM."""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .registry import BACKBONES, NECKS
from .._lib import lib, check, ptr, stream_of, workspace, TIMER
from .. import gemm as G


def dcn_col2im(grad_cols, x, offset, mask, kh, kw, stride, pad, dil, Ho, Wo, gather=True):
    """gradients of the deformable column matrix w.r.t. input / offsets / mask (vidar_dcn_col2im_f32).
    gather=True: grad_x through the per-call reverse map (no atomics); False: atomic scatter."""
    import ctypes
    N, C, H, W = x.shape
    gx = torch.empty_like(x); goff = torch.empty_like(offset); gm = torch.empty_like(mask)
    ws, nbytes = None, 0
    if gather:
        f = lib().vidar_dcn_col2im_workspace_bytes
        f.restype = ctypes.c_size_t
        nbytes = int(f(N, H, W, Ho, Wo, kh, kw))
        ws = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=x.device) if nbytes else None
    with TIMER.span("dcn_col2im", 4 * (3 * x.numel() + 2 * offset.numel() + 2 * mask.numel() + grad_cols.numel())):
        check(lib().vidar_dcn_col2im_f32(ptr(grad_cols), ptr(x), ptr(offset), ptr(mask), ptr(gx), ptr(goff), ptr(gm),
                                         N, C, H, W, Ho, Wo, kh, kw, stride, pad, dil, ptr(ws),
                                         ctypes.c_size_t(nbytes if ws is not None else 0), stream_of(x)), "dcn_col2im")
    return gx, goff, gm


class _ModulatedDeformConv(Function):
    """gemm_mode "lib": the column product is a library bmm; "f32" / "bf16x3": csrc/gemm_mfma.hip, and the frozen
    BatchNorm + ReLU that follows the convolution in a bottleneck (scale / shift given) rides in its epilogue."""

    @staticmethod
    def forward(ctx, x, offset, mask, weight, bias, stride, pad, dil, gemm_mode="lib", scale=None, shift=None,
                relu=False):
        x, offset, mask = x.float().contiguous(), offset.float().contiguous(), mask.float().contiguous()
        N, C, H, W = x.shape
        Cout, _, kh, kw = weight.shape
        Ho = (H + 2 * pad - dil * (kh - 1) - 1) // stride + 1
        Wo = (W + 2 * pad - dil * (kw - 1) - 1) // stride + 1
        cols = torch.empty((N, C * kh * kw, Ho * Wo), device=x.device)
        with TIMER.span("dcn_im2col", 4 * (x.numel() + offset.numel() + mask.numel() + cols.numel())):
            check(lib().vidar_dcn_im2col_f32(ptr(x), ptr(offset), ptr(mask), ptr(cols), N, C, H, W, Ho,
                                             Wo, kh, kw, stride, pad, dil, stream_of(x)), "dcn_im2col")
        fused = scale is not None
        if gemm_mode == "lib":
            assert not fused
            # bmm with a broadcast (stride-0) weight: torch.matmul(2-D, 3-D) would transpose-copy `cols`
            out = torch.bmm(weight.reshape(1, Cout, -1).expand(N, -1, -1), cols)
            if bias is not None:
                out = out + bias.view(1, -1, 1)
        else:
            sh = shift
            if bias is not None:
                sh = bias.float() if not fused else (shift + bias.float() * scale)
            out = G.conv_forward(weight.reshape(Cout, -1), cols, scale, sh, None, relu, G.precision_of(gemm_mode))
        ctx.save_for_backward(x, offset, mask, weight, cols, scale, out if (fused and relu) else None)
        ctx.cfg = (stride, pad, dil, Ho, Wo, bias is not None, gemm_mode, fused, relu)
        return out.view(N, Cout, Ho, Wo)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        x, offset, mask, weight, cols, scale, y = ctx.saved_tensors
        stride, pad, dil, Ho, Wo, has_bias, gemm_mode, fused, relu = ctx.cfg
        N, C, H, W = x.shape
        Cout, _, kh, kw = weight.shape
        go = grad_out.float().contiguous().view(N, Cout, Ho * Wo)
        if fused:                                   # through the ReLU mask and the frozen BN scale
            go = _affine_act_backward(go, y if relu else go, scale, relu, False)[0]
        if gemm_mode == "lib":
            grad_weight = torch.bmm(go, cols.transpose(1, 2)).sum(0).reshape(weight.shape)
            # (the fp32 MFMA kernel for this product measured 0.39 vs 0.49 ms against the library's DEFAULT solution in
            #  kbench, but SLOWER than the TunableOp-selected one inside the step: 353.0 vs 349.1 ms; not used)
            grad_cols = torch.bmm(weight.reshape(1, Cout, -1).transpose(1, 2).expand(N, -1, -1), go)
        else:
            prec = G.precision_of(gemm_mode)
            grad_weight = G.conv_grad_weight(go, cols, prec).reshape(weight.shape) if ctx.needs_input_grad[3] else None
            grad_cols = G.conv_grad_input(weight.reshape(Cout, -1), go, prec)
        gx, goff, gm = dcn_col2im(grad_cols, x, offset, mask, kh, kw, stride, pad, dil, Ho, Wo)
        return (gx, goff, gm, grad_weight, (go.sum((0, 2)) if has_bias else None), None, None, None, None, None, None,
                None)


def modulated_deform_conv2d(x, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, bn=None, relu=False):
    """bn: a FrozenBN whose affine (+ `relu`) is folded into the GEMM epilogue when the MFMA GEMM path is on
    (vidar_amd.gemm.mode() != "lib") -- the caller must then NOT apply it again (see `Bottleneck.forward`)"""
    m = G.mode() if (x.is_cuda and G.own_kernels()) else "lib"
    scale = shift = None
    if bn is not None:
        assert m != "lib"
        scale, shift = bn._scale_shift()
    return _ModulatedDeformConv.apply(x, offset, mask, weight, bias, int(stride), int(padding), int(dilation), m,
                                      scale, shift, bool(relu))


# A/B switch of the DCNv2 `conv_offset` convolution: "own" = csrc/conv3x3_mfma.hip, "lib" (default) = the library
# convolution.  Measured on MI355X (profiles/r05_kbench_conv_offset.log): 0.260 vs 0.246 ms on [24, 256, 58, 100], 0.105 vs
# 0.094 ms on [6, 256, 58, 100], 0.193 vs 0.168 on [24, 512, 29, 50]; the step 344.6 vs 345.8 ms (inside the box-to-box
# spread) -- parity at best, not a win, so the library keeps the default (why: profiles/r05_conv3x3_ablation.log).
_CONV_OFFSET_OWN = os.environ.get("VIDAR_CONV_OFFSET", "lib") == "own"


class _Conv3x3Few(Function):
    """3x3 / stride 1 / pad 1 convolution with <= 32 output channels through csrc/conv3x3_mfma.hip (an implicit GEMM on the
    fp32 matrix cores: the library's Winograd kernel cannot fill its tiles with 27 outputs); the backward is the very
    `convolution_backward` autograd would issue for F.conv2d."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        w = weight.contiguous()
        N, C, H, W = x.shape
        Cout = w.shape[0]
        out = torch.empty((N, Cout, H, W), device=x.device, dtype=torch.float32)
        ws, ws_ptr, ws_bytes = workspace(lib().vidar_conv3x3_few_workspace_bytes, C, like=x)
        with TIMER.span("conv3x3_few", 4 * (x.numel() + out.numel())):
            check(lib().vidar_conv3x3_few_f32(ptr(x), ptr(w), ptr(bias), ptr(out), N, C, H, W, Cout, ws_ptr, ws_bytes,
                                              stream_of(x)), "conv3x3_few")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        need = [ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]]
        gx, gw, gb = torch.ops.aten.convolution_backward(g.contiguous(), x, w, [w.shape[0]] if ctx.has_bias else None,
                                                         [1, 1], [1, 1], [1, 1], False, [0, 0], 1, need)
        return gx, gw, gb


def conv3x3_few_ok(conv, x):
    """can `conv` (an nn.Conv2d) on `x` take the few-output implicit-GEMM kernel?  (include/vidar_hip.h lists the limits)"""
    return (_CONV_OFFSET_OWN and x.is_cuda and x.dtype == torch.float32 and conv.weight.dtype == torch.float32
            and not torch.is_autocast_enabled() and x.dim() == 4
            and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.padding_mode == "zeros" and conv.out_channels <= 32 and conv.in_channels % 8 == 0
            and x.shape[3] <= 191 and x.shape[2] * x.shape[3] < (1 << 30) and x.numel() > 0)


class ModulatedDeformConv2dPack(nn.Module):
    """mmcv.ops.ModulatedDeformConv2dPack: `conv_offset` (zero-init conv -> 3*K channels), offsets =
    first 2K channels, mask = sigmoid(last K); parameters `weight`, (`bias`), `conv_offset.*`."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, deform_groups=1, bias=True):
        super().__init__()
        if groups != 1 or deform_groups != 1:
            raise NotImplementedError("DCNv2 kernels are built for groups == deform_groups == 1")
        k = kernel_size
        self.stride, self.padding, self.dilation, self.k = stride, padding, dilation, k
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, k, k))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        n = in_channels * k * k
        nn.init.uniform_(self.weight, -1.0 / math.sqrt(n), 1.0 / math.sqrt(n))
        self.conv_offset = nn.Conv2d(in_channels, 3 * k * k, k, stride, padding, dilation, bias=True)
        nn.init.zeros_(self.conv_offset.weight)
        nn.init.zeros_(self.conv_offset.bias)

    def forward(self, x, bn=None, relu=False):
        if conv3x3_few_ok(self.conv_offset, x):
            out = _Conv3x3Few.apply(x, self.conv_offset.weight, self.conv_offset.bias)
        else:
            out = self.conv_offset(x)
        o1, o2, mask = torch.chunk(out, 3, dim=1)
        offset = torch.cat((o1, o2), dim=1)
        fuse = dict(bn=bn, relu=relu) if bn is not None else {}     # (the epilogue-fused form exists on the MFMA GEMM path only)
        return modulated_deform_conv2d(x, offset, torch.sigmoid(mask), self.weight, self.bias,
                                       self.stride, self.padding, self.dilation, **fuse)


class _AffineAct(Function):
    """y = act(x*scale[c] + shift[c] (+ residual)) in one pass (csrc/affine_act.hip)."""

    @staticmethod
    def forward(ctx, x, scale, shift, residual, relu):
        x = x.float().contiguous()
        N, C, H, W = x.shape
        res = residual.float().contiguous() if residual is not None else None
        y = torch.empty_like(x)
        with TIMER.span("affine_act_fwd", 4 * x.numel() * (3 if res is not None else 2)):
            check(lib().vidar_affine_act_fwd_f32(ptr(x), ptr(scale), ptr(shift), ptr(res), ptr(y), N, C,
                                                 H * W, int(relu), stream_of(x)), "affine_act_fwd")
        ctx.save_for_backward(y, scale)
        ctx.cfg = (relu, residual is not None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        y, scale = ctx.saved_tensors
        relu, has_res = ctx.cfg
        gx, gres = _affine_act_backward(gy.float().contiguous(), y, scale, relu, has_res)
        return gx, None, None, gres, None


def _affine_act_backward(gy, y, scale, relu, has_res):
    """grad of y = act(x*scale[c] + shift[c] (+ residual)) w.r.t. x and residual; gy, y: [N, C, ...] contiguous"""
    N, C = y.shape[:2]
    HW = y.numel() // (N * C)
    gx = torch.empty_like(y)
    gres = torch.empty_like(y) if has_res else None
    with TIMER.span("affine_act_bwd", 4 * y.numel() * (4 if has_res else 3)):
        check(lib().vidar_affine_act_bwd_f32(ptr(gy), ptr(y), ptr(scale), ptr(gx), ptr(gres), N, C,
                                             HW, int(relu), stream_of(y)), "affine_act_bwd")
    return gx, gres


class FrozenBN(nn.BatchNorm2d):
    """BN2d with `requires_grad=False` + `norm_eval=True` (config :93-95): always running stats.
    `forward(x, residual=None, relu=False)` fuses the affine with the residual add and the ReLU that
    follow it in a bottleneck.  A trainable affine (requires_grad=True configs) falls back to
    F.batch_norm + separate ops so that gamma/beta receive gradients."""

    def __init__(self, c, requires_grad=False):
        super().__init__(c)
        for p in self.parameters():
            p.requires_grad = requires_grad
        self._cache = None

    def _scale_shift(self):
        key = (self.weight._version, self.bias._version, self.running_mean._version,
               self.running_var._version, self.weight.device)
        if self._cache is None or self._cache[0] != key:
            with torch.no_grad():
                scale = (self.weight / torch.sqrt(self.running_var + self.eps)).float().contiguous()
                shift = (self.bias - self.running_mean * scale).float().contiguous()
            self._cache = (key, scale, shift)
        return self._cache[1], self._cache[2]

    def forward(self, x, residual=None, relu=False):
        if self.weight.requires_grad or not x.is_cuda:
            y = F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, False, 0.0,
                             self.eps)
            if residual is not None:
                y = y + residual
            return F.relu(y, inplace=True) if relu else y
        scale, shift = self._scale_shift()
        return _AffineAct.apply(x, scale, shift, residual, relu)


def stem_bn_relu_pool(x, bn):
    """max_pool2d(relu(bn(x)), 3, 2, 1) of the frozen stem in one pass (vidar_stem_bn_relu_pool_f32), or None when
    the fused kernel does not apply (gradients needed, trainable BN, CPU tensor, W % 4 != 0): the caller then takes
    the two-kernel path (bit-identical, tests/test_dcn_gpu.py; 1.3 ms of the step)."""
    if not x.is_cuda or bn.weight.requires_grad or (torch.is_grad_enabled() and x.requires_grad):
        return None
    x = x.float().contiguous()
    N, C, H, W = x.shape
    if W % 4 or N * C > 65535:
        return None
    scale, shift = bn._scale_shift()
    y = torch.empty((N, C, (H - 1) // 2 + 1, W // 2), device=x.device, dtype=torch.float32)
    with TIMER.span("stem_bn_relu_pool", 4 * (x.numel() + y.numel())):
        check(lib().vidar_stem_bn_relu_pool_f32(ptr(x), ptr(scale), ptr(shift), ptr(y), N, C, H, W, stream_of(x)),
              "stem_bn_relu_pool")
    return y


class Conv1x1(nn.Conv2d):
    """1x1 convolution as a batched GEMM  out[n] = W [Cout, Cin] x[n] [Cin, H*W]  on NCHW tensors.  Same parameter
    names / shapes as nn.Conv2d (`weight` [Cout, Cin, 1, 1]); two thirds of ResNet101's convolutions are 1x1, and as
    torch GEMMs they go through the tuned library solutions of vidar_amd/gemm_tuning.py instead of MIOpen's default
    rocBLAS pick.  A stride > 1 subsamples the input first (what a strided 1x1 convolution computes)."""
    as_gemm = os.environ.get("VIDAR_CONV1X1_GEMM", "1") != "0"
    any_device = False          # tests: take the GEMM form on the CPU too

    def gemm_form(self, x):
        return (self.as_gemm and (x.is_cuda or self.any_device) and self.kernel_size == (1, 1) and self.padding == (0, 0)
                and self.groups == 1)

    def forward(self, x, presampled=False):
        """presampled: `x` already is the strided subsample `x[:, :, ::sh, ::sw]` (Bottleneck shares one copy between conv1
        and the downsample convolution)"""
        if not self.gemm_form(x):
            assert not presampled
            return super().forward(x)
        if self.stride != (1, 1) and not presampled:
            x = x[:, :, ::self.stride[0], ::self.stride[1]].contiguous()
        N, C, H, W = x.shape
        out = torch.bmm(self.weight.view(1, self.out_channels, C).expand(N, -1, -1), x.reshape(N, C, H * W))
        if self.bias is not None:
            out = out + self.bias.view(1, -1, 1)
        return out.view(N, self.out_channels, H, W)


_AUTO_FUSE_RES = os.environ.get("VIDAR_AUTO_FUSE_RES", "1") != "0"


class _Conv1x1Identity(Function):
    """(conv1x1(x), x) of a bottleneck whose shortcut is the identity: the block input `x` feeds BOTH the first 1x1
    convolution and the residual add at the end of the block, so autograd would sum their two gradients with a separate
    elementwise pass over [N, C, H, W] (22 x 427 MB of traffic per step in stage 3 alone).  Handing `x` through this
    Function makes both gradients arrive in ONE backward call, and the data gradient of the convolution accumulates
    onto the shortcut's gradient inside the GEMM (beta = 1): grad_x = W^T grad_out + grad_identity."""

    @staticmethod
    def forward(ctx, x, weight):
        N, C, H, W = x.shape
        Cout = weight.shape[0]
        out = torch.bmm(weight.view(1, Cout, C).expand(N, -1, -1), x.reshape(N, C, H * W))
        ctx.save_for_backward(x, weight)
        return out.view(N, Cout, H, W), x.view_as(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, g_out, g_ident):
        x, weight = ctx.saved_tensors
        N, C, H, W = x.shape
        Cout = weight.shape[0]
        gx = gw = None
        if g_out is None:
            return g_ident, None
        go = g_out.contiguous().view(N, Cout, H * W)
        if ctx.needs_input_grad[0]:
            wt = weight.view(1, Cout, C).transpose(1, 2).expand(N, -1, -1)
            if g_ident is not None:
                # IN PLACE on the shortcut's gradient: it is a buffer the closing convolution's backward (or the engine's
                # own accumulation) just produced for this one consumer; out-of-place baddbmm would first copy it -- the
                # very pass this Function exists to remove (measured: fills / copies + 1.4 ms per step)
                gx = g_ident.contiguous().view(N, C, H * W).baddbmm_(wt, go).view(N, C, H, W)
            else:
                gx = torch.bmm(wt, go).view(N, C, H, W)
        if ctx.needs_input_grad[1]:
            gw = torch.bmm(go, x.reshape(N, C, H * W).transpose(1, 2)).sum(0).view_as(weight)
        return gx, gw


_SHORTCUT_ACCUMULATE = os.environ.get("VIDAR_SHORTCUT_ACCUMULATE", "1") != "0"


def conv1x1_with_identity_ok(conv, x):
    """the identity-shortcut form above: a bias-free, stride-1 1x1 convolution in its GEMM form on fp32 that needs gradients"""
    return (_SHORTCUT_ACCUMULATE and isinstance(conv, Conv1x1) and conv.gemm_form(x) and conv.stride == (1, 1)
            and conv.bias is None and x.dtype == torch.float32 and conv.weight.dtype == torch.float32
            and x.requires_grad and torch.is_grad_enabled() and not torch.is_autocast_enabled() and x.is_contiguous())


class _Conv1x1BNAct(Function):
    """act(bn(conv1x1(x)) + residual) with the frozen BatchNorm, the residual add and the ReLU in the epilogue of the
    MFMA GEMM (csrc/gemm_mfma.hip): the [N, Cout, H*W] product is written once instead of written, re-read and
    re-written by affine_act."""

    @staticmethod
    def forward(ctx, x, weight, scale, shift, residual, relu, precision):
        x = x.float().contiguous()
        N, C, H, W = x.shape
        Cout = weight.shape[0]
        res = residual.float().contiguous().view(N, Cout, H * W) if residual is not None else None
        y = G.conv_forward(weight.reshape(Cout, C), x.view(N, C, H * W), scale, shift, res, relu, precision)
        ctx.save_for_backward(x, weight, scale, y if relu else None)
        ctx.cfg = (relu, residual is not None, precision)
        return y.view(N, Cout, H, W)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight, scale, y = ctx.saved_tensors
        relu, has_res, precision = ctx.cfg
        N, C, H, W = x.shape
        Cout = weight.shape[0]
        gy = gy.float().contiguous().view(N, Cout, H * W)
        gz, gres = _affine_act_backward(gy, y if relu else gy, scale, relu, has_res)
        gx = G.conv_grad_input(weight.reshape(Cout, C), gz, precision).view(N, C, H, W) if ctx.needs_input_grad[0] else None
        gw = G.conv_grad_weight(gz, x.view(N, C, H * W), precision).view_as(weight) if ctx.needs_input_grad[1] else None
        return gx, gw, None, None, (gres.view(N, Cout, H, W) if has_res else None), None, None


def _fuses_1x1(conv, bn, x, residual, hw, m):
    """does `bn(conv(x))` (+ residual, ReLU) take the fused MFMA kernel?  hw = output pixels per image.
    "auto": only the block's closing convolution (the one with a residual) does -- there the epilogue saves a 3-tensor
    affine_act pass; the other 1x1 convolutions are faster as library GEMM + affine_act.  Beyond the kernel's
    addressing range -- H*W >= 2^22 or channels x H*W >= 2^29, gemm.operand_ok -- the library convolution runs as in
    every other mode."""
    fuse = G.own_kernels(m) or (m == "auto" and residual is not None and _AUTO_FUSE_RES)
    in_range = 0 < hw < (1 << 22) and max(conv.in_channels, conv.out_channels) * hw < (1 << 29)
    return (fuse and x.is_cuda and not bn.weight.requires_grad and conv.bias is None and conv.kernel_size == (1, 1)
            and conv.padding == (0, 0) and conv.groups == 1 and in_range)


def conv1x1_bn_act(conv, bn, x, residual=None, relu=False, presampled=False):
    """`bn(conv(x), residual, relu)` of a bias-free 1x1 convolution and a frozen BN; fused when the MFMA GEMM path is on.
    presampled: `x` already is the convolution's strided subsample (see Conv1x1.forward)"""
    if presampled:
        return _conv1x1_bn_act_presampled(conv, bn, x, residual, relu)
    m = G.mode()
    # a strided 1x1 convolution keeps ceil(H / s) x ceil(W / s) pixels
    hw = ((x.shape[2] - 1) // conv.stride[0] + 1) * ((x.shape[3] - 1) // conv.stride[1] + 1) if x.dim() == 4 else 0
    if not _fuses_1x1(conv, bn, x, residual, hw, m):
        return bn(conv(x), residual=residual, relu=relu)
    if conv.stride != (1, 1):
        x = x[:, :, ::conv.stride[0], ::conv.stride[1]]
    scale, shift = bn._scale_shift()
    return _Conv1x1BNAct.apply(x, conv.weight, scale, shift, residual, bool(relu), G.precision_of(m))


def _conv1x1_bn_act_presampled(conv, bn, xs, residual, relu):
    m = G.mode()
    if not _fuses_1x1(conv, bn, xs, residual, xs.shape[2] * xs.shape[3] if xs.dim() == 4 else 0, m):
        return bn(conv(xs, presampled=True), residual=residual, relu=relu)
    scale, shift = bn._scale_shift()
    return _Conv1x1BNAct.apply(xs, conv.weight, scale, shift, residual, bool(relu), G.precision_of(m))


_SHARE_SUBSAMPLE = os.environ.get("VIDAR_SHARE_SUBSAMPLE", "1") != "0"


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, style="caffe", dcn=None, bn_grad=False):
        super().__init__()
        s1, s2 = (stride, 1) if style == "caffe" else (1, stride)
        self.conv1 = Conv1x1(inplanes, planes, 1, stride=s1, bias=False)
        self.bn1 = FrozenBN(planes, bn_grad)
        if dcn is not None:
            self.conv2 = ModulatedDeformConv2dPack(planes, planes, 3, stride=s2, padding=1, dilation=1,
                                                   deform_groups=dcn.get("deform_groups", 1), bias=False)
        else:
            self.conv2 = nn.Conv2d(planes, planes, 3, stride=s2, padding=1, bias=False)
        self.bn2 = FrozenBN(planes, bn_grad)
        self.conv3 = Conv1x1(planes, planes * 4, 1, bias=False)
        self.bn3 = FrozenBN(planes * 4, bn_grad)
        self.downsample = downsample

    def _shares_subsample(self, x):
        """caffe style puts the block's stride on conv1, and the downsample convolution has the same one: both 1x1
        convolutions read the same strided subsample of x -- gather it once (one strided pass over x and, in the backward,
        one scatter into zeros and a sum on the small tensor, instead of two of each)"""
        d = self.downsample
        return (_SHARE_SUBSAMPLE and d is not None and len(d) == 2 and isinstance(d[0], Conv1x1) and isinstance(d[1], FrozenBN)
                and self.conv1.stride != (1, 1) and d[0].stride == self.conv1.stride
                and self.conv1.gemm_form(x) and d[0].gemm_form(x))

    def forward(self, x):
        if self._shares_subsample(x):
            sh, sw = self.conv1.stride
            xs = x[:, :, ::sh, ::sw].contiguous()
            identity = conv1x1_bn_act(self.downsample[0], self.downsample[1], xs, presampled=True)
            out = conv1x1_bn_act(self.conv1, self.bn1, xs, relu=True, presampled=True)
            return self._tail(out, identity)
        if self.downsample is None:
            identity = x
            if conv1x1_with_identity_ok(self.conv1, x) and not _fuses_1x1(self.conv1, self.bn1, x, None,
                                                                            x.shape[2] * x.shape[3], G.mode()):
                out, identity = _Conv1x1Identity.apply(x, self.conv1.weight)
                return self._tail(self.bn1(out, relu=True), identity)
        elif len(self.downsample) == 2 and isinstance(self.downsample[1], FrozenBN):
            identity = conv1x1_bn_act(self.downsample[0], self.downsample[1], x)
        else:
            identity = self.downsample(x)
        out = conv1x1_bn_act(self.conv1, self.bn1, x, relu=True)
        return self._tail(out, identity)

    def _tail(self, out, identity):
        if (isinstance(self.conv2, ModulatedDeformConv2dPack) and G.own_kernels() and out.is_cuda
                and not self.bn2.weight.requires_grad):
            out = self.conv2(out, bn=self.bn2, relu=True)
        else:
            out = self.bn2(self.conv2(out), relu=True)
        return conv1x1_bn_act(self.conv3, self.bn3, out, residual=identity, relu=True)


@BACKBONES.register_module()
class ResNet(nn.Module):
    arch = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}

    def __init__(self, depth=101, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=-1,
                 norm_cfg=dict(type="BN", requires_grad=True), norm_eval=True, style="pytorch", dcn=None,
                 stage_with_dcn=(False, False, False, False), strides=(1, 2, 2, 2), in_channels=3,
                 base_channels=64, init_cfg=None, pretrained=None, **kwargs):
        super().__init__()
        if dcn is not None and dcn.get("type") != "DCNv2":
            raise NotImplementedError(dcn)
        bn_grad = norm_cfg.get("requires_grad", True)
        self.out_indices, self.frozen_stages, self.norm_eval = out_indices, frozen_stages, norm_eval
        self.conv1 = nn.Conv2d(in_channels, base_channels, 7, stride=2, padding=3, bias=False)
        self.bn1 = FrozenBN(base_channels, bn_grad)
        inplanes = base_channels
        self.res_layers = []
        for i, nblocks in enumerate(self.arch[depth][:num_stages]):
            planes = base_channels * 2 ** i
            stage_dcn = dcn if stage_with_dcn[i] else None
            down = None
            if strides[i] != 1 or inplanes != planes * 4:
                down = nn.Sequential(Conv1x1(inplanes, planes * 4, 1, stride=strides[i], bias=False),
                                     FrozenBN(planes * 4, bn_grad))
            blocks = [Bottleneck(inplanes, planes, strides[i], down, style, stage_dcn, bn_grad)]
            inplanes = planes * 4
            blocks += [Bottleneck(inplanes, planes, 1, None, style, stage_dcn, bn_grad) for _ in range(1, nblocks)]
            name = f"layer{i + 1}"
            self.add_module(name, nn.Sequential(*blocks))
            self.res_layers.append(name)
        self._freeze_stages()

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            for m in (self.conv1, self.bn1):
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            for p in getattr(self, f"layer{i}").parameters():
                p.requires_grad = False

    def forward(self, x):
        x = self.conv1(x)
        y = stem_bn_relu_pool(x, self.bn1)
        x = y if y is not None else F.max_pool2d(self.bn1(x, relu=True), 3, stride=2, padding=1)
        outs = []
        for i, name in enumerate(self.res_layers):
            x = getattr(self, name)(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)


class _ConvModule(nn.Module):
    """mmcv ConvModule without norm/activation: parameters under `.conv`."""

    def __init__(self, cin, cout, k, stride=1, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=padding)
        nn.init.xavier_uniform_(self.conv.weight)
        nn.init.zeros_(self.conv.bias)

    def forward(self, x):
        return self.conv(x)


@NECKS.register_module()
class FPN(nn.Module):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1,
                 add_extra_convs=False, relu_before_extra_convs=False, init_cfg=None, **kwargs):
        super().__init__()
        self.start_level = start_level
        self.backbone_end_level = len(in_channels) if end_level == -1 else end_level
        self.num_outs = num_outs
        self.add_extra_convs = "on_input" if add_extra_convs is True else add_extra_convs
        self.relu_before_extra_convs = relu_before_extra_convs
        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        for i in range(start_level, self.backbone_end_level):
            self.lateral_convs.append(_ConvModule(in_channels[i], out_channels, 1))
            self.fpn_convs.append(_ConvModule(out_channels, out_channels, 3, padding=1))
        extra = num_outs - (self.backbone_end_level - start_level)
        if self.add_extra_convs and extra >= 1:
            for i in range(extra):
                cin = in_channels[self.backbone_end_level - 1] if (i == 0 and self.add_extra_convs == "on_input") else out_channels
                self.fpn_convs.append(_ConvModule(cin, out_channels, 3, stride=2, padding=1))

    def forward(self, inputs):
        lat = [l(inputs[i + self.start_level]) for i, l in enumerate(self.lateral_convs)]
        for i in range(len(lat) - 1, 0, -1):
            lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode="nearest")
        outs = [self.fpn_convs[i](lat[i]) for i in range(len(lat))]
        if self.num_outs > len(outs):
            if not self.add_extra_convs:
                for _ in range(self.num_outs - len(outs)):
                    outs.append(F.max_pool2d(outs[-1], 1, stride=2))
            else:
                if self.add_extra_convs == "on_input":
                    src = inputs[self.backbone_end_level - 1]
                elif self.add_extra_convs == "on_lateral":
                    src = lat[-1]
                else:
                    src = outs[-1]
                outs.append(self.fpn_convs[len(lat)](src))
                for i in range(len(lat) + 1, self.num_outs):
                    outs.append(self.fpn_convs[i](F.relu(outs[-1]) if self.relu_before_extra_convs else outs[-1]))
        return tuple(outs)
