from __future__ import annotations

import torch

from .._lib import VidarHipError  # noqa: F401


def check_input(x: torch.Tensor, name: str):
    """CHECK_INPUT of the reference (third_lib/dvxlr/dvxlr.cpp:26-32): CUDA + contiguous,
    raised as RuntimeError exactly like TORCH_CHECK."""
    if not isinstance(x, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not x.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not x.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if x.dtype != torch.float32:
        # the reference dispatches on fp32/fp64 but allocates fp32 outputs (dvxlr.cu:490-493), so
        # fp64 inputs die inside packed_accessor32 there too
        raise RuntimeError(f"{name} must be float32 (got {x.dtype})")


def ray_dims(sigma, origin, points, tindex):
    if sigma.dim() != 5 or origin.dim() != 3 or points.dim() != 3 or tindex.dim() != 2:
        raise RuntimeError("expected sigma[N,T,Z,Y,X], origin[N,T,3], points[N,M,3], tindex[N,M]")
    N, T, Z, Y, X = sigma.shape
    M = points.shape[1]
    if points.shape[0] != N or tindex.shape != (N, M) or origin.shape[0] != N or \
            origin.shape[2] != 3 or points.shape[2] < 3:
        raise RuntimeError("inconsistent ray tensor shapes")
    if points.shape[2] != 3:
        raise RuntimeError("points must be [N,M,3]")
    return N, M, T, origin.shape[1], Z, Y, X
