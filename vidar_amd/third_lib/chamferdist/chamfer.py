"""`chamferdist.chamfer` -- ChamferDistance / knn_points with the reference's signatures
(third_lib/chamfer_dist/chamferdist/chamferdist/chamfer.py:20-134, :137-288)."""
from __future__ import annotations

import warnings
from collections import namedtuple

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _C

_KNN = namedtuple("KNN", "dists idx knn")


class _knn_points(Function):
    @staticmethod
    def forward(ctx, p1, p2, lengths1, lengths2, K, version, return_sorted=True):
        idx, dists = _C.knn_points_idx(p1, p2, lengths1, lengths2, K, version)
        ctx.save_for_backward(p1, p2, lengths1, lengths2, idx)
        ctx.mark_non_differentiable(idx)
        return dists, idx

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_dists, grad_idx):
        p1, p2, lengths1, lengths2, idx = ctx.saved_tensors
        g1, g2 = _C.knn_points_backward(p1.float(), p2.float(), lengths1, lengths2, idx,
                                        grad_dists.float())
        return g1, g2, None, None, None, None, None


def knn_gather(x, idx, lengths=None):
    N, M, U = x.shape
    _, L, K = idx.shape
    out = x[:, :, None].expand(-1, -1, K, -1).gather(1, idx[:, :, :, None].expand(-1, -1, -1, U))
    if lengths is not None and bool((lengths < K).any()):
        mask = lengths[:, None] <= torch.arange(K, device=x.device)[None]
        out[mask[:, None].expand(-1, L, -1)] = 0.0
    return out


def knn_points(p1, p2, lengths1=None, lengths2=None, K=1, version=-1, return_nn=False,
               return_sorted=True):
    if p1.shape[0] != p2.shape[0]:
        raise ValueError("pts1 and pts2 must have the same batch dimension.")
    if p1.shape[2] != p2.shape[2]:
        raise ValueError("pts1 and pts2 must have the same point dimension.")
    p1 = p1.contiguous()
    p2 = p2.contiguous()
    if lengths1 is None:
        lengths1 = torch.full((p1.shape[0],), p1.shape[1], dtype=torch.int64, device=p1.device)
    if lengths2 is None:
        lengths2 = torch.full((p1.shape[0],), p2.shape[1], dtype=torch.int64, device=p1.device)
    dists, idx = _knn_points.apply(p1, p2, lengths1, lengths2, K, version, return_sorted)
    nn = knn_gather(p2, idx, lengths2) if return_nn else None
    return _KNN(dists=dists, idx=idx, knn=nn)


class ChamferDistance(torch.nn.Module):
    def forward(self, source_cloud, target_cloud, bidirectional=False, reverse=False,
                reduction="mean"):
        if not isinstance(source_cloud, torch.Tensor) or not isinstance(target_cloud, torch.Tensor):
            raise TypeError("Expected input type torch.Tensor.")
        if source_cloud.device != target_cloud.device:
            raise ValueError("Source and target clouds must be on the same device. "
                             f"Got {source_cloud.device} and {target_cloud.device}.")
        bs, ls, ds = source_cloud.shape
        bt, lt, dt = target_cloud.shape
        if bs != bt:
            raise ValueError("Source and target pointclouds must have the same batchsize.")
        if ds != dt:
            raise ValueError("Source and target pointclouds must have the same dimensionality.")
        if bidirectional and reverse:
            warnings.warn("Both bidirectional and reverse set to True. "
                          "bidirectional behavior takes precedence.")
        if reduction not in ("sum", "mean", None):
            raise ValueError('Reduction must either be "sum" or "mean" or None.')
        len_s = torch.full((bs,), ls, dtype=torch.long, device=source_cloud.device)
        len_t = torch.full((bt,), lt, dtype=torch.long, device=target_cloud.device)
        src = knn_points(source_cloud, target_cloud, lengths1=len_s, lengths2=len_t, K=1)
        fwd_d, fwd_i = src.dists[..., 0], src.idx[..., 0]
        fwd = fwd_d.sum(1)
        bwd = bwd_d = bwd_i = None
        if reverse or bidirectional:
            tgt = knn_points(target_cloud, source_cloud, lengths1=len_t, lengths2=len_s, K=1)
            bwd_d, bwd_i = tgt.dists[..., 0], tgt.idx[..., 0]
            bwd = bwd_d.sum(1)
        if reduction == "sum":
            fwd = fwd.sum()
            bwd = bwd.sum() if bwd is not None else None
        elif reduction == "mean":
            fwd = fwd.mean()
            bwd = bwd.mean() if bwd is not None else None
        if bidirectional:
            return fwd, bwd, (fwd_d, fwd_i, bwd_d, bwd_i)
        if reverse:
            return bwd, (bwd_d, bwd_i)
        return fwd, (fwd_d, fwd_i)
