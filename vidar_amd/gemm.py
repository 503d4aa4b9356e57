"""Host side of csrc/gemm_mfma.hip: the matrix products of the hot path on the gfx950 matrix cores.

`mode()` selects what the Linear layers / 1x1 convolutions of the path run on:
    "lib"     torch.addmm / bmm -> rocBLAS / hipBLASLt fp32 (TunableOp-selected, vidar_amd/gemm_tuning.py)
    "auto"    (default) fp32 throughout, each product on whichever fp32 kernel is faster on MI355X (profiles/r04_*):
              the library for forward / grad-input products, this library's exact-fp32 MFMA kernel for the Linear
              weight gradients (contraction over 40 000 - 185 000 rows: the library's best solution runs at 22 - 44
              TFLOP/s there, the split-K slab kernel at 90 - 100) and for the bottleneck's closing 1x1 convolution
              with its frozen BatchNorm + residual + ReLU in the epilogue
    "f32"     vidar_gemm_f32(precision = VIDAR_GEMM_F32): this library's exact-fp32 MFMA kernel, with the bias / frozen
              BatchNorm / residual / ReLU that follows the product folded into its epilogue
    "bf16x3"  the same kernel with split-bf16 products (16 significand bits >= the TF32 the reference executes these
              GEMMs in, README.md:96, tools/train.py:141-144) at 3/16 of the fp32 matrix-core cost
The environment variable VIDAR_GEMM sets the default; `set_mode` / `use` switch it at run time (bench.py records the
bf16x3 step as a second, labelled record).  There is no CPU path: the functions raise on CPU tensors."""
from __future__ import annotations

import contextlib
import ctypes
import os

import torch

from ._lib import lib, check, ptr, stream_of, TIMER

F32, BF16X3 = 0, 1
K_MAJOR, MN_MAJOR = 0, 1
_MODES = ("lib", "auto", "f32", "bf16x3")
_mode = os.environ.get("VIDAR_GEMM", "auto")
if _mode not in _MODES:
    raise ValueError(f"VIDAR_GEMM={_mode!r}: expected one of {_MODES}")


def mode() -> str:
    return _mode


def set_mode(m: str) -> str:
    global _mode
    if m not in _MODES:
        raise ValueError(f"gemm mode {m!r}: expected one of {_MODES}")
    prev, _mode = _mode, m
    return prev


@contextlib.contextmanager
def use(m: str):
    prev = set_mode(m)
    try:
        yield
    finally:
        set_mode(prev)


def precision_of(m: str | None = None) -> int:
    m = _mode if m is None else m
    return BF16X3 if m == "bf16x3" else F32


def own_kernels(m: str | None = None) -> bool:
    """True when EVERY product of the path runs on csrc/gemm_mfma.hip ("f32", "bf16x3")"""
    return (_mode if m is None else m) in ("f32", "bf16x3")


def _f32c(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"vidar_gemm: {what} must be a CUDA tensor (no CPU path)")
    if t.dtype != torch.float32:
        raise TypeError(f"vidar_gemm: {what} must be float32, got {t.dtype}")
    if t.dim() and t.stride(-1) != 1:
        raise ValueError(f"vidar_gemm: the last dimension of {what} must be contiguous (strides {tuple(t.stride())})")
    return t


_LD_MAX = 1 << 22          # leading dimensions the kernel's 31-bit per-workgroup byte offsets allow (gemm_mfma.hip)
_KLD_MAX = 1 << 29         # contraction length x leading dimension of an MN-major operand


def operand_ok(t, layout, contraction) -> bool:
    """can the 2-D (or [B, r, c]) tensor `t` be handed to vidar_gemm_f32 as it is?  The kernel takes any leading
    dimension >= the row length (row-sliced views are fine) but needs a unit last stride, an un-broadcast row stride
    and offsets that fit its 31-bit addressing; everything else takes the library product instead (bricks.Linear,
    backbones.conv1x1_bn_act in "auto" mode) or is made contiguous first (the wrappers below)."""
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() >= 2 and t.stride(-1) == 1):
        return False
    ld = t.stride(-2)
    if ld < t.shape[-1] or ld >= _LD_MAX:
        return False
    return layout == K_MAJOR or contraction * ld < _KLD_MAX


def _operand(t, layout, contraction, what):
    """`t` as the kernel can read it: itself when operand_ok, otherwise a contiguous copy (a transposed / expanded view
    used to be an error here)"""
    _f32c(t, what) if t.stride(-1) == 1 else None
    if operand_ok(t, layout, contraction):
        return t
    if not t.is_cuda:
        raise RuntimeError(f"vidar_gemm: {what} must be a CUDA tensor (no CPU path)")
    if t.dtype != torch.float32:
        raise TypeError(f"vidar_gemm: {what} must be float32, got {t.dtype}")
    c = t.contiguous()
    if not operand_ok(c, layout, contraction):
        raise ValueError(f"vidar_gemm: {what} {tuple(t.shape)} exceeds the kernel's addressing range "
                         f"(leading dimension < 2^22, contraction x leading dimension < 2^29)")
    return c


def gemm_raw(A, lda, a_layout, B, ldb, b_layout, C, ldc, M, N, K, *, batch=1, sA=0, sB=0, sC=0, scale=None, shift=None,
             vec_axis=0, residual=None, ldr=0, sR=0, relu=False, precision=F32, reduce=False, a_rowsum=None, name="gemm"):
    """one call of vidar_gemm_f32 on torch tensors (pointers are taken as they are: the caller states the geometry)"""
    L = lib()
    ws, nbytes = None, 0
    if reduce:
        f = L.vidar_gemm_workspace_bytes
        f.restype = ctypes.c_size_t
        nbytes = int(f(int(M), int(N), int(K), int(batch), int(precision), 2 if a_rowsum is not None else 1))
        if nbytes:
            ws = torch.empty(nbytes // 4, dtype=torch.float32, device=C.device)
    flops = 2.0 * M * N * K * batch
    with TIMER.span(name, flops):
        check(L.vidar_gemm_f32(ptr(A), ctypes.c_int64(lda), int(a_layout), ptr(B), ctypes.c_int64(ldb), int(b_layout),
                               ptr(C), ctypes.c_int64(ldc), int(M), int(N), int(K), int(batch), ctypes.c_int64(sA),
                               ctypes.c_int64(sB), ctypes.c_int64(sC), ptr(scale), ptr(shift), int(vec_axis),
                               ptr(residual), ctypes.c_int64(ldr), ctypes.c_int64(sR), int(bool(relu)), int(precision),
                               int(bool(reduce)), ptr(a_rowsum), ptr(ws), ctypes.c_size_t(nbytes), stream_of(C)),
              "vidar_gemm_f32")
    return C


# ---- nn.Linear on [rows, K] activations -----------------------------------------------------------------------------
def linear_forward(x2, weight, bias=None, relu=False, precision=F32):
    """relu?(x2 [M,K] @ weight[N,K]^T + bias) -> [M,N]"""
    M, K = x2.shape
    N = weight.shape[0]
    x2 = _operand(x2, K_MAJOR, K, "x"); weight = _operand(weight, K_MAJOR, K, "weight")
    y = torch.empty((M, N), dtype=torch.float32, device=x2.device)
    if M == 0:
        return y
    return gemm_raw(x2, x2.stride(0), K_MAJOR, weight, weight.stride(0), K_MAJOR, y, N, M, N, K,
                    shift=None if bias is None else _f32c(bias, "bias"), vec_axis=0, relu=relu, precision=precision,
                    name="gemm_linear_fwd")


def linear_grad_input(g2, weight, precision=F32):
    """g2 [M,N] @ weight [N,K] -> [M,K]"""
    M, N = g2.shape
    K = weight.shape[1]
    g2 = _operand(g2, K_MAJOR, N, "grad_out"); weight = _operand(weight, MN_MAJOR, N, "weight")
    gx = torch.empty((M, K), dtype=torch.float32, device=g2.device)
    if M == 0:
        return gx
    return gemm_raw(g2, g2.stride(0), K_MAJOR, weight, weight.stride(0), MN_MAJOR, gx, K, M, K, N, precision=precision,
                    name="gemm_linear_dx")


def linear_grad_weight(g2, x2, precision=F32, with_bias=False):
    """g2 [M,N]^T @ x2 [M,K] -> [N,K]  (the contraction runs over the rows: split over K', slabs summed in order).
    with_bias: also the bias gradient g2.sum(0) [N], from the same pass over g2 (`a_rowsum` of vidar_gemm_f32: summed in a
    fixed order, unlike the atomic column-sum kernel) -> (grad_weight, grad_bias)"""
    M, N = g2.shape
    K = x2.shape[1]
    g2 = _operand(g2, MN_MAJOR, M, "grad_out"); x2 = _operand(x2, MN_MAJOR, M, "x")
    gw = torch.empty((N, K), dtype=torch.float32, device=g2.device)
    gb = torch.empty(N, dtype=torch.float32, device=g2.device) if with_bias else None
    if M == 0:
        return (gw.zero_(), gb.zero_()) if with_bias else gw.zero_()
    gemm_raw(g2, g2.stride(0), MN_MAJOR, x2, x2.stride(0), MN_MAJOR, gw, K, N, K, M, precision=precision,
             reduce=True, a_rowsum=gb, name="gemm_linear_dw")
    return (gw, gb) if with_bias else gw


def linear_grad_weight_ok(g2, x2) -> bool:
    """both operands of linear_grad_weight are readable in place (no copy, no addressing overflow)"""
    return g2.dim() == 2 and x2.dim() == 2 and operand_ok(g2, MN_MAJOR, g2.shape[0]) and operand_ok(x2, MN_MAJOR, x2.shape[0])


def _colsum(g2):
    n = g2.shape[1]
    if n % 4 == 0 and (n // 4) & (n // 4 - 1) == 0 and n <= 4096 and g2.data_ptr() % 16 == 0 and g2.is_contiguous():
        gb = torch.empty(n, dtype=torch.float32, device=g2.device)
        check(lib().vidar_colsum_f32(ptr(g2), ptr(gb), ctypes.c_int64(g2.shape[0]), int(n), stream_of(g2)), "colsum")
        return gb
    return g2.sum(0)


class MfmaLinear(torch.autograd.Function):
    """F.linear (+ optional ReLU) whose three products run on csrc/gemm_mfma.hip"""

    @staticmethod
    def forward(ctx, x2, weight, bias, relu, precision):
        x2 = x2 if x2.stride(-1) == 1 else x2.contiguous()
        weight = weight if weight.stride(-1) == 1 else weight.contiguous()
        y = linear_forward(x2, weight, bias, relu, precision)
        ctx.save_for_backward(x2, weight, y if relu else None)
        ctx.cfg = (relu, precision, bias is not None)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x2, weight, y = ctx.saved_tensors
        relu, precision, has_bias = ctx.cfg
        g2 = g if (g.stride(-1) == 1 and g.dtype == torch.float32) else g.float().contiguous()
        if relu:
            g2 = g2 * (y > 0)
        gx = linear_grad_input(g2, weight, precision) if ctx.needs_input_grad[0] else None
        want_b = has_bias and ctx.needs_input_grad[2]
        gw = gb = None
        if ctx.needs_input_grad[1]:                       # the bias gradient rides along with the weight gradient
            gw = linear_grad_weight(g2, x2, precision, with_bias=want_b)
            if want_b:
                gw, gb = gw
        elif want_b:
            gb = _colsum(g2)
        return gx, gw, gb, None, None


def linear(x, weight, bias=None, relu=False, m: str | None = None):
    """F.linear on any leading shape through the MFMA kernel of mode `m` (default: the current mode)"""
    n = weight.shape[0]
    y = MfmaLinear.apply(x.reshape(-1, x.shape[-1]), weight, bias, bool(relu), precision_of(m))
    return y.view(*x.shape[:-1], n)


# ---- 1x1 / im2col convolutions on NCHW activations: out[b] = W [Co,Ci] x[b] [Ci, HW] ----------------------------------
def conv_forward(w2, x3, scale=None, shift=None, residual=None, relu=False, precision=F32):
    """act((w2 [Co,Ci] @ x3 [B,Ci,HW]) * scale[co] + shift[co] + residual [B,Co,HW]) -> [B,Co,HW]"""
    _f32c(w2, "weight"); _f32c(x3, "x")
    Bn, Ci, HW = x3.shape
    Co = w2.shape[0]
    y = torch.empty((Bn, Co, HW), dtype=torch.float32, device=x3.device)
    if residual is not None:
        assert residual.shape == y.shape and residual.is_contiguous()
    return gemm_raw(w2, w2.stride(0), K_MAJOR, x3, x3.stride(1), MN_MAJOR, y, HW, Co, HW, Ci, batch=Bn, sA=0,
                    sB=x3.stride(0), sC=Co * HW, scale=scale, shift=shift, vec_axis=1, residual=residual, ldr=HW,
                    sR=Co * HW, relu=relu, precision=precision, name="gemm_conv_fwd")


def conv_grad_input(w2, g3, precision=F32):
    """w2 [Co,Ci]^T @ g3 [B,Co,HW] -> [B,Ci,HW]"""
    _f32c(w2, "weight"); _f32c(g3, "grad_out")
    Bn, Co, HW = g3.shape
    Ci = w2.shape[1]
    gx = torch.empty((Bn, Ci, HW), dtype=torch.float32, device=g3.device)
    return gemm_raw(w2, w2.stride(0), MN_MAJOR, g3, g3.stride(1), MN_MAJOR, gx, HW, Ci, HW, Co, batch=Bn, sA=0,
                    sB=g3.stride(0), sC=Ci * HW, precision=precision, name="gemm_conv_dx")


def conv_grad_weight(g3, x3, precision=F32, scale=None):
    """sum_b g3[b] [Co,HW] @ x3[b] [Ci,HW]^T -> [Co,Ci]  (optionally row-scaled by scale[co])"""
    _f32c(g3, "grad_out"); _f32c(x3, "x")
    Bn, Co, HW = g3.shape
    Ci = x3.shape[1]
    gw = torch.empty((Co, Ci), dtype=torch.float32, device=g3.device)
    return gemm_raw(g3, g3.stride(1), K_MAJOR, x3, x3.stride(1), K_MAJOR, gw, Ci, Co, Ci, HW, batch=Bn,
                    sA=g3.stride(0), sB=x3.stride(0), scale=scale, vec_axis=1, precision=precision, reduce=True,
                    name="gemm_conv_dw")
