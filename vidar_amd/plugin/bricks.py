"""Small third-party building blocks the reference takes from mmcv / mmdet / torchvision, restated
from their documented behaviour (none of them is vendored in the reference,
so their parity is 'unpinned').  Parameter names follow the originals so released checkpoints map:
  FFN.layers.{0.0,1}.*  (mmcv.cnn.bricks.transformer.FFN)
  LearnedPositionalEncoding.{row_embed,col_embed}.weight  (mmdet)"""
from __future__ import annotations

import math

import numpy as np

import torch
import torch.nn as nn
import torch.nn.functional as F

from .registry import FEEDFORWARD_NETWORK, POSITIONAL_ENCODING
from .. import gemm as _gemm


def xavier_init(module, gain=1, bias=0, distribution="normal"):
    """[3P] mmcv.cnn.xavier_init: acts on the module ITSELF only -- called on an nn.Sequential (the
    reference does that for its can_bus_mlp, transformer.py:109, vidar_head_base.py) it is a no-op,
    which leaves those Linear biases at PyTorch's default initialisation."""
    if hasattr(module, "weight") and module.weight is not None:
        (nn.init.xavier_uniform_ if distribution == "uniform" else nn.init.xavier_normal_)(module.weight, gain=gain)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def constant_init(module, val, bias=0):
    if hasattr(module, "weight") and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def build_norm_layer(cfg, num_features):
    typ = cfg.get("type", "LN")
    if typ != "LN":
        raise NotImplementedError(f"norm type {typ}")
    return "ln", nn.LayerNorm(num_features, eps=cfg.get("eps", 1e-5))


def build_activation(cfg):
    typ = cfg.get("type", "ReLU")
    if typ == "ReLU":
        return nn.ReLU(inplace=cfg.get("inplace", False))
    if typ == "GELU":
        return nn.GELU()
    raise NotImplementedError(typ)


class _LinearColsum(torch.autograd.Function):
    """F.linear whose bias gradient is the library's column-sum kernel (vidar_colsum_f32) instead of autograd's generic
    `sum(0)`; grad_input / grad_weight are the very GEMMs autograd issues (same operand order, so the tuned library
    solutions apply)."""

    @staticmethod
    def forward(ctx, x2, weight, bias):
        # 2-D in, 2-D out: the output must not be a view made inside the Function (an in-place ReLU follows in the FFN)
        ctx.save_for_backward(x2, weight)
        return torch.addmm(bias, x2, weight.t())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        import ctypes
        from .._lib import lib, check, ptr, stream_of
        x2, w = ctx.saved_tensors
        g2 = g if g.is_contiguous() else g.contiguous()
        if g2.dtype != torch.float32 or g2.data_ptr() % 16:          # the column-sum kernel reads float4
            g2 = g2.float().clone()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = g2 @ w
        if ctx.needs_input_grad[1]:
            # "auto": the exact-fp32 split-K MFMA kernel (4 x the library's best solution for these tall contractions)
            # (operands the kernel cannot read in place -- a transposed / expanded view, a leading dimension beyond its
            #  addressing range -- take the library product like every other mode)
            own = _gemm.mode() == "auto" and _gemm.linear_grad_weight_ok(g2, x2)
            if own:
                # ... and the bias gradient with it: summed by the same kernel from the grad_out tiles it stages anyway, in
                # a fixed order (deterministic like torch's sum(0); the atomic column-sum kernel below is not)
                gw, gb = _gemm.linear_grad_weight(g2, x2, _gemm.F32, with_bias=True)
                return gx, gw, gb
            gw = (x2.t() @ g2).t()
        gb = torch.empty(g2.shape[1], dtype=torch.float32, device=g2.device)
        check(lib().vidar_colsum_f32(ptr(g2), ptr(gb), ctypes.c_int64(g2.shape[0]), int(g2.shape[1]), stream_of(g2)),
              "colsum")
        return gx, gw, gb


class Linear(nn.Linear):
    """nn.Linear (same parameters / state-dict keys) for the modules on the hot path: on CUDA fp32 tensors the bias
    gradient rides along with the weight gradient (`a_rowsum` of vidar_gemm_f32: no second pass over grad_out, summed in
    a fixed order); only when the weight gradient takes the library product does it come from `vidar_colsum_f32` (one
    HBM-rate pass whose workgroups combine with fp32 atomics: last-bit run-to-run differences)."""

    def forward(self, x, relu=False):
        """relu=True: ReLU(linear(x)) -- in the epilogue of the MFMA GEMM when that path is on"""
        n = self.out_features
        f32 = (x.is_cuda and x.dtype == torch.float32 and self.weight.dtype == torch.float32
               and (self.bias is None or self.bias.dtype == torch.float32) and not torch.is_autocast_enabled())
        if f32 and _gemm.own_kernels() and x.numel() > 0:
            # the hand-written matrix-core kernel (csrc/gemm_mfma.hip): exact fp32 ("f32") or split-bf16 ("bf16x3")
            return _gemm.linear(x, self.weight, self.bias, relu=relu)
        if (f32 and self.bias is not None and self.bias.requires_grad
                and torch.is_grad_enabled() and n % 4 == 0 and (n // 4) & (n // 4 - 1) == 0 and n <= 4096):
            y = _LinearColsum.apply(x.reshape(-1, x.shape[-1]), self.weight, self.bias)
            if relu:
                # in place on the Function's own output, BEFORE the reshaping view: an in-place op on a view of a custom
                # Function's output makes autograd rebase the graph on CopySlices, whose backward clones and copies the
                # [rows, out_features] gradient three times (12 x 0.5 GB per step on the FFN's hidden layer)
                y = F.relu(y, inplace=True)
            return y.view(*x.shape[:-1], n)
        y = F.linear(x, self.weight, self.bias)
        return F.relu(y, inplace=True) if relu else y


class _ReluDropout(torch.autograd.Function):
    """dropout(relu(x), p) as one HIP pass each way (csrc/norm_fuse.hip: vidar_relu_drop_{fwd,bwd}_f32); the backward reads
    the saved OUTPUT (which the next Linear saves anyway) instead of a mask."""

    @staticmethod
    def forward(ctx, x, p, seed):
        import ctypes
        from .._lib import lib, check, ptr, stream_of
        x = x.contiguous()
        y = torch.empty_like(x)
        check(lib().vidar_relu_drop_fwd_f32(ptr(x), ptr(y), ctypes.c_int64(x.numel()), ctypes.c_float(p),
                                            ctypes.c_uint32(seed), stream_of(x)), "relu_drop_fwd")
        ctx.save_for_backward(y)
        ctx.p = p
        return y

    @staticmethod
    def backward(ctx, gy):
        import ctypes
        from .._lib import lib, check, ptr, stream_of
        y, = ctx.saved_tensors
        gy = gy.contiguous()
        gx = torch.empty_like(gy)
        check(lib().vidar_relu_drop_bwd_f32(ptr(gy), ptr(y), ptr(gx), ctypes.c_int64(gy.numel()), ctypes.c_float(ctx.p),
                                            stream_of(gy)), "relu_drop_bwd")
        return gx, None, None


def relu_dropout(linear, x, p, training):
    """dropout(relu(linear(x)), p).  Training on CUDA fp32 with p > 0: ONE elementwise pass after the GEMM each way (the
    ReLU is not handed to the Linear then, unless its GEMM epilogue does it for free); otherwise the torch ops."""
    p = float(p) if training else 0.0
    fused = (p > 0.0 and x.is_cuda and x.dtype == torch.float32 and linear.weight.dtype == torch.float32
             and not torch.is_autocast_enabled() and linear.out_features % 4 == 0 and x.numel() > 0)
    if not fused:
        return F.dropout(linear(x, relu=True), p, training)
    h = linear(x, relu=_gemm.own_kernels())            # (ReLU in the MFMA GEMM's epilogue costs nothing; it is idempotent)
    seed = int(torch.randint(0, 1 << 31, (1,), dtype=torch.int64).item()) * 2 + 1     # CPU generator: no device sync
    return _ReluDropout.apply(h, p, seed)


@FEEDFORWARD_NETWORK.register_module()
class FFN(nn.Module):
    """Linear -> act -> drop -> Linear -> drop, plus identity (x when none is given)."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2,
                 act_cfg=dict(type="ReLU", inplace=True), ffn_drop=0., dropout_layer=None,
                 add_identity=True, init_cfg=None, **kwargs):
        super().__init__()
        assert num_fcs >= 2
        self.embed_dims = embed_dims
        layers, in_ch = [], embed_dims
        for _ in range(num_fcs - 1):
            layers.append(nn.Sequential(Linear(in_ch, feedforward_channels), build_activation(act_cfg),
                                        nn.Dropout(ffn_drop)))
            in_ch = feedforward_channels
        layers.append(Linear(feedforward_channels, embed_dims))
        layers.append(nn.Dropout(ffn_drop))
        self.layers = nn.Sequential(*layers)
        self.dropout_layer = nn.Identity()
        self.add_identity = add_identity

    def _body(self, x):
        """layers[:-1]; a (Linear, ReLU, Dropout) stage hands its ReLU to the Linear (GEMM epilogue on the MFMA path)"""
        for stage in self.layers[:-1]:
            if (isinstance(stage, nn.Sequential) and len(stage) == 3 and isinstance(stage[0], Linear)
                    and isinstance(stage[1], nn.ReLU) and isinstance(stage[2], nn.Dropout)):
                x = relu_dropout(stage[0], x, stage[2].p, self.training)
            elif (isinstance(stage, nn.Sequential) and len(stage) == 3 and isinstance(stage[0], Linear)
                    and isinstance(stage[1], nn.ReLU)):
                x = stage[2](stage[0](x, relu=True))
            else:
                x = stage(x)
        return x

    def forward(self, x, identity=None, fuse_norm=None):
        if fuse_norm is not None and self.add_identity and isinstance(self.layers[-1], nn.Dropout) \
                and isinstance(self.dropout_layer, nn.Identity):
            out = self._body(x)                            # the closing Dropout moves into the fused tail
            return drop_add_layernorm(out, x if identity is None else identity, fuse_norm, self.layers[-1].p,
                                      self.training)
        out = self.layers[-1](self._body(x))
        if not self.add_identity:
            return self.dropout_layer(out)
        res = (x if identity is None else identity) + self.dropout_layer(out)
        return fuse_norm(res) if fuse_norm is not None else res


@POSITIONAL_ENCODING.register_module()
class LearnedPositionalEncoding(nn.Module):
    """pos[b, :, y, x] = cat(col_embed[x], row_embed[y])  -> [bs, 2*num_feats, h, w]"""

    def __init__(self, num_feats, row_num_embed=50, col_num_embed=50, init_cfg=None):
        super().__init__()
        self.row_embed = nn.Embedding(row_num_embed, num_feats)
        self.col_embed = nn.Embedding(col_num_embed, num_feats)
        self.num_feats = num_feats
        nn.init.uniform_(self.row_embed.weight)
        nn.init.uniform_(self.col_embed.weight)

    def forward(self, mask):
        h, w = mask.shape[-2:]
        x_embed = self.col_embed(torch.arange(w, device=mask.device))
        y_embed = self.row_embed(torch.arange(h, device=mask.device))
        pos = torch.cat((x_embed.unsqueeze(0).repeat(h, 1, 1), y_embed.unsqueeze(1).repeat(1, w, 1)), -1)
        return pos.permute(2, 0, 1).unsqueeze(0).repeat(mask.shape[0], 1, 1, 1)


def rotate_nearest(img, angle_deg, center):
    """torchvision.transforms.functional.rotate(img[C,H,W], angle, center=center) with its defaults
    (nearest interpolation, zero fill, no expand), as used to align prev_bev
    (modules/transformer.py:139-151).  [3P, recalled: inverse affine about `center`, angle
    counter-clockwise, sampling grid at pixel centres, grid_sample(align_corners=False)]."""
    C, H, W = img.shape
    cx, cy = center[0] - W * 0.5, center[1] - H * 0.5
    rot = math.radians(-float(angle_deg))
    c, s = math.cos(rot), math.sin(rot)
    m = [c, s, 0.0, -s, c, 0.0]
    m[2] += m[0] * (-cx) + m[1] * (-cy) + cx
    m[5] += m[3] * (-cx) + m[4] * (-cy) + cy
    from .utils.host import to_device_async
    # the 2x3 inverse affine already divided by the half extents (host math; one tiny pinned H2D copy)
    scaled = (np.asarray(m, dtype=np.float32).reshape(2, 3).T / np.asarray([0.5 * W, 0.5 * H], dtype=np.float32))
    theta_t = to_device_async(scaled, img.device)
    xs = torch.linspace(-W * 0.5 + 0.5, W * 0.5 - 0.5, W, device=img.device)
    ys = torch.linspace(-H * 0.5 + 0.5, H * 0.5 - 0.5, H, device=img.device)
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    base = torch.stack((gx, gy, torch.ones_like(gx)), -1).view(-1, 3)
    grid = base @ theta_t
    out = F.grid_sample(img.float().unsqueeze(0), grid.view(1, H, W, 2), mode="nearest",
                        padding_mode="zeros", align_corners=False)
    return out[0].to(img.dtype)


# --------------------------------------------------------------------------------------------------
# LayerNorm(dropout(x) + residual) as one HIP pass each way (csrc/norm_fuse.hip)
# --------------------------------------------------------------------------------------------------


def can_fuse_norm(norm, x):
    return (isinstance(norm, nn.LayerNorm) and norm.elementwise_affine and norm.bias is not None and x.is_cuda
            and x.shape[-1] == 256 and tuple(norm.normalized_shape) == (256,) and x.dtype == torch.float32)


class _DropAddLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, gamma, beta, p, eps, seed):
        import ctypes
        from .._lib import lib, check, ptr, stream_of
        x, residual = x.contiguous(), residual.contiguous()
        rows = x.numel() // 256
        y = torch.empty_like(x); s = torch.empty_like(x)
        mean = torch.empty(rows, device=x.device); rstd = torch.empty(rows, device=x.device)
        check(lib().vidar_drop_add_ln_fwd_f32(ptr(x), ptr(residual), ptr(gamma), ptr(beta), ptr(y), ptr(s), ptr(mean),
                                              ptr(rstd), ctypes.c_int64(rows), 256, ctypes.c_float(p),
                                              ctypes.c_float(eps), ctypes.c_uint32(seed), stream_of(x)), "drop_add_ln_fwd")
        ctx.save_for_backward(s, gamma, mean, rstd)
        ctx.cfg = (p, seed, rows)
        return y

    @staticmethod
    def backward(ctx, gy):
        import ctypes
        from .._lib import lib, check, ptr, stream_of
        s, gamma, mean, rstd = ctx.saved_tensors
        p, seed, rows = ctx.cfg
        gy = gy.contiguous()
        gx = torch.empty_like(s); gres = torch.empty_like(s)
        dgamma = torch.empty_like(gamma); dbeta = torch.empty_like(gamma)
        f = lib().vidar_drop_add_ln_bwd_workspace_bytes
        f.restype = ctypes.c_size_t
        ws = torch.empty((int(f(ctypes.c_int64(rows))) + 3) // 4, device=s.device)
        check(lib().vidar_drop_add_ln_bwd_f32(ptr(gy), ptr(s), ptr(gamma), ptr(mean), ptr(rstd), ptr(gx), ptr(gres),
                                              ptr(dgamma), ptr(dbeta), ptr(ws), ctypes.c_int64(rows), 256, ctypes.c_float(p),
                                              ctypes.c_uint32(seed), stream_of(s)), "drop_add_ln_bwd")
        return gx, gres, dgamma, dbeta, None, None, None


def drop_add_layernorm(x, residual, norm, p, training):
    """norm(dropout(x, p) + residual).  CUDA fp32 [.., 256] tensors with an affine nn.LayerNorm take the fused HIP
    kernels (dropout mask = hash of (seed, element), seed drawn from torch's CPU generator once per call, so a
    manual_seed run is reproducible and a recomputed forward under torch.utils.checkpoint sees the same mask);
    everything else is the three torch ops."""
    p = float(p) if training else 0.0
    if can_fuse_norm(norm, x) and residual.shape == x.shape and residual.dtype == x.dtype:
        # the 32-bit mask seed is DRAWN from torch's CPU generator (no device sync): checkpoint's RNG preservation,
        # get_rng_state / set_rng_state and manual_seed all cover it, and two models in one process do not share it
        seed = int(torch.randint(0, 1 << 31, (1,), dtype=torch.int64).item()) * 2 + 1 if p > 0.0 else 0
        return _DropAddLayerNorm.apply(x, residual, norm.weight, norm.bias, p, float(norm.eps), seed)
    return norm(F.dropout(x, p, training) + residual)
