"""ViDARHeadTemplate / ViDARHeadBase -- registry names, kwargs, parameter names and loss / decode
semantics of projects/mmdet3d_plugin/bevformer/dense_heads/vidar_head_base.py:31-773.

What changed is the execution plan, not the math:
  * the waypoint / grid_sample / -inf mask / cross-entropy chain (:420-509, :586-592) is one fused
    kernel per (batch item) call (ray_ops.ray_ce), likewise the dense gumbel render (:594-630,
    :754-773) and the test-time arg-max decode (:697-731);
  * GT clouds are never compacted with boolean indexing (host syncs at :441, :464-467, :636-644):
    rays carry a frame slot (-1 = ignore) and validity is decided on device;
  * the training chamfer (:654) runs on the nearest-neighbour kernel with a validity mask instead
    of a dense [N, M, 3] expansion."""
from __future__ import annotations

import numpy as np
import torch

from ..utils.host import const_tensor, to_device_async
import torch.nn as nn
import torch.nn.functional as F

from ..bricks import xavier_init
from ..losses import chamfer_distance
from ..registry import HEADS, build_positional_encoding, build_transformer
from ..utils import e2e_predictor_utils
from . import ray_ops


@HEADS.register_module()
class ViDARHeadTemplate(nn.Module):
    def __init__(self, *args, transformer=None, num_pred_fcs=2, num_pred_height=1, can_bus_norm=True,
                 can_bus_dims=(0, 1, 2, 17), bev_h=30, bev_w=30, pc_range=None, loss_weight=None,
                 positional_encoding=dict(type="SinePositionalEncoding", num_feats=128, normalize=True),
                 eval_within_grid=False, init_cfg=None, **kwargs):
        super().__init__()
        self.bev_h, self.bev_w = bev_h, bev_w
        self.pc_range = pc_range
        self.real_w = pc_range[3] - pc_range[0]
        self.real_h = pc_range[4] - pc_range[1]
        self.can_bus_norm = can_bus_norm
        self.can_bus_dims = can_bus_dims
        self.num_pred_fcs = num_pred_fcs
        self.num_pred_height = num_pred_height
        self.positional_encoding = build_positional_encoding(positional_encoding)
        self.transformer = build_transformer(transformer)
        self.embed_dims = self.transformer.embed_dims
        self.loss_weight = np.array(loss_weight)
        assert self.loss_weight.shape[-1] == 1
        self.eval_within_grid = eval_within_grid
        self._init_layers()

    def _init_layers(self):
        self.bev_embedding = nn.Embedding(self.bev_h * self.bev_w, self.embed_dims)
        self.prev_frame_embedding = nn.Parameter(torch.Tensor(1, self.embed_dims))
        self.can_bus_mlp = nn.Sequential(
            nn.Linear(len(self.can_bus_dims), self.embed_dims // 2), nn.ReLU(inplace=True),
            nn.Linear(self.embed_dims // 2, self.embed_dims), nn.ReLU(inplace=True))
        if self.can_bus_norm:
            self.can_bus_mlp.add_module("norm", nn.LayerNorm(self.embed_dims))

    def init_weights(self):
        if getattr(self, "transformer", None) is None:
            return
        self.transformer.init_weights()
        nn.init.normal_(self.prev_frame_embedding)
        xavier_init(self.can_bus_mlp, distribution="uniform", bias=0.)

    def _get_next_bev_features(self, prev_features, img_metas, target_frame_index, tgt_points,
                               ref_points, bev_h, bev_w):
        bs = prev_features.shape[0]
        dtype = prev_features.dtype
        bev_queries = self.bev_embedding.weight.to(dtype).unsqueeze(0)
        bev_mask = torch.zeros((bs, self.bev_h, self.bev_w), device=bev_queries.device).to(dtype)
        bev_pos = self.positional_encoding(bev_mask).to(dtype)
        can_bus = np.array([m["future_can_bus"][target_frame_index] for m in img_metas])[:, self.can_bus_dims]
        can_bus = to_device_async(can_bus, bev_pos.device, dtype)
        bev_queries_input = bev_queries + self.can_bus_mlp(can_bus).unsqueeze(1)
        prev_features_input = prev_features + self.prev_frame_embedding[None, :, None, :]
        return self.transformer(prev_features_input, bev_queries_input, tgt_points=tgt_points,
                                ref_points=ref_points, bev_h=bev_h, bev_w=bev_w, bev_pos=bev_pos,
                                img_metas=img_metas)

    def forward(self, prev_feats, img_metas, target_frame_index, tgt_points, ref_points, bev_h, bev_w):
        bs, num_frames, bev_grids_num, bev_dims = prev_feats.shape
        assert bev_dims == self.embed_dims
        assert bev_h * bev_w == bev_grids_num == tgt_points.shape[1]
        return self._get_next_bev_features(prev_feats, img_metas, target_frame_index, tgt_points,
                                           ref_points, bev_h, bev_w)

    # ------------------------------------------------------------------------------------------
    def _process_gt_points(self, bev_preds, gt_points, batched_origin_points, valid_frames,
                           start_idx, pred_frame_num, bev_h, bev_w, pc_range):
        """Select the GT rays of `valid_frames` (:219-276).  Static shapes: every point is kept, the
        per-ray frame slot `tindex` is -1 for points of other frames (the reference drops them and
        NaN-pads to the longest cloud of the batch instead)."""
        valid_frame_num, inter_num, bs, token_num, num_height_pred = bev_preds.shape
        max_pts = max(p.shape[0] for p in gt_points)
        pts = torch.stack([F.pad(p, (0, 0, 0, max_pts - p.shape[0]), value=float("nan"))
                           for p in gt_points])
        t = pts[..., -1] - start_idx
        keep = torch.zeros_like(t, dtype=torch.bool)
        for i in range(start_idx, pred_frame_num):
            if i in valid_frames:
                keep |= pts[..., -1] == i
        tindex = torch.where(keep, t, torch.full_like(t, -1.0))
        tindex = torch.clamp(tindex, max=valid_frame_num - 1)
        xyz = pts[..., :3].contiguous()
        if batched_origin_points is None:
            batched_origin_points = xyz.new_zeros((bs, len(valid_frames), 3))
        origin_grids = e2e_predictor_utils.coords_to_voxel_grids(
            batched_origin_points, bev_h=bev_h, bev_w=bev_w, pillar_num=num_height_pred, pc_range=pc_range)
        gt_grids = e2e_predictor_utils.coords_to_voxel_grids(
            xyz, bev_h=bev_h, bev_w=bev_w, pillar_num=num_height_pred, pc_range=pc_range)
        return origin_grids, batched_origin_points, gt_grids, xyz, tindex

    def get_rendered_pcds(self, origin, points, tindex, gt_dist, pred_dist, pc_range):
        """(:344-389) list[bs] of list[frames] of [n,3] rendered points (boolean selection: used by
        the evaluation path, where a sync per frame is harmless)."""
        bs, num_frames, _ = origin.shape
        pcds = []
        for b in range(bs):
            per_frame = []
            for t in range(num_frames):
                mask = (tindex[b] == t) & (gt_dist[b] > 0.)
                if self.eval_within_grid:
                    mask = mask & e2e_predictor_utils.get_inside_mask(points[b], pc_range)
                p = points[b][mask]
                r = p - origin[b, t].view(1, 3)
                rn = r / torch.sqrt((r ** 2).sum(1, keepdim=True))
                per_frame.append(origin[b, t].view(1, 3) + rn * pred_dist[b][mask].view(-1, 1))
            pcds.append(per_frame)
        return pcds


@HEADS.register_module()
class ViDARHeadBase(ViDARHeadTemplate):
    def __init__(self, ray_grid_num=1026, ray_grid_step=1.0, use_ce_loss=True, use_dist_loss=False,
                 use_dense_loss=True, dense_loss_weight=1.0, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.ray_grid_num = ray_grid_num
        self.ray_grid_step = ray_grid_step
        self.use_ce_loss = use_ce_loss
        self.use_dist_loss = use_dist_loss
        self.use_dense_loss = use_dense_loss
        assert self.use_ce_loss or self.use_dist_loss or self.use_dense_loss
        if use_dist_loss:
            raise NotImplementedError("use_dist_loss is off in every released config (:567-578)")
        if ray_grid_num != ray_ops.K_SAMPLES:
            raise NotImplementedError(f"ray kernels are specialised for ray_grid_num == {ray_ops.K_SAMPLES}")
        self.dense_loss_weight = dense_loss_weight
        self.gumbel_noise_fn = None      # tests inject the reference's noise here

    @staticmethod
    def _volumes(bev_preds, b, Z, H, W):
        """bev_preds [F, 1, bs, H*W, Z] -> sigma of batch item b as [F, Z, H, W] (:548-558)."""
        return bev_preds[:, 0, b].permute(0, 2, 1).contiguous().view(-1, Z, H, W)

    def _dense_rays(self, bs, F_, Z, H, W, device):
        interval = 4
        v = e2e_predictor_utils.get_bev_grids_3d(H // interval, W // interval, Z // interval, bs=1,
                                                 device=device)
        v = (v * const_tensor([W, H, Z], v.device, v.dtype)).view(-1, 3)
        pts = v.repeat(F_, 1)
        tix = torch.arange(F_, device=device, dtype=v.dtype).repeat_interleave(v.shape[0])
        return pts, tix, v.shape[0]

    def loss(self, pred_dict, gt_points, start_idx, tgt_bev_h, tgt_bev_w, tgt_pc_range,
             pred_frame_num, img_metas=None, batched_origin_points=None, loss_weight=None):
        valid_frames = pred_dict["valid_frames"]
        bev_preds = pred_dict["next_bev_preds"][:, -1:].float()
        F_, inter_num, bs, token_num, Z = bev_preds.shape
        H, W = tgt_bev_h, tgt_bev_w
        origin_grids, origin_pts, gt_grids, gt_xyz, tindex = self._process_gt_points(
            bev_preds, gt_points, batched_origin_points, valid_frames, start_idx, pred_frame_num,
            H, W, tgt_pc_range)
        loss_weight = self.loss_weight if loss_weight is None else loss_weight
        lw = const_tensor(np.asarray(loss_weight, dtype=np.float32)[:, 0], bev_preds.device)
        step = self.ray_grid_step
        loss_dict = dict()
        sigmas = [self._volumes(bev_preds, b, Z, H, W) for b in range(bs)]

        if self.use_ce_loss:
            num = bev_preds.new_zeros(())
            den = bev_preds.new_zeros(())
            for b in range(bs):
                ce, valid = ray_ops.ray_ce(sigmas[b], origin_grids[b], gt_grids[b], tindex[b], step,
                                           self.ray_grid_num)
                w = lw[tindex[b].clamp(min=0).long()] * valid
                num = num + (ce * w).sum()
                den = den + w.sum()
            loss_dict["regularization.loss"] = num / torch.clamp(den, min=1)

        if self.use_dense_loss:
            pts, tix, per_frame = self._dense_rays(bs, F_, Z, H, W, bev_preds.device)
            total = bev_preds.new_zeros(())
            size = const_tensor([W - 1, H - 1, Z - 1], gt_grids.device, gt_grids.dtype)
            for b in range(bs):
                noise = (self.gumbel_noise_fn(pts.shape[0], self.ray_grid_num) if self.gumbel_noise_fn
                         else ray_ops.gumbel_noise(pts.shape[0], self.ray_grid_num, pts.device))
                dist = ray_ops.ray_gumbel(sigmas[b], origin_grids[b], pts, tix, noise, step,
                                          self.ray_grid_num)
                inside = ((gt_grids[b] > 0) & (gt_grids[b] < size)).all(-1)
                for f in range(F_):
                    o = origin_grids[b, f].view(1, 3)
                    p = pts[f * per_frame:(f + 1) * per_frame]
                    r = p - o
                    d = dist[f * per_frame:(f + 1) * per_frame]
                    # get_rendered_pcds keeps rays with a positive rendered distance only (:392-395);
                    # a dense voxel centre that coincides with the origin (zero-length ray) drops out
                    live = d > 0
                    # NaN-free also for the dropped zero-length ray: a masked 0/0 would still poison the
                    # backward of `unit * d` (0 * NaN), so the norm is floored and d masked BEFORE the product
                    unit = r / torch.sqrt((r ** 2).sum(1, keepdim=True)).clamp_min(1e-12)
                    pred = unit * torch.where(live, d, torch.zeros_like(d)).view(-1, 1) * 0.1
                    gt = (gt_grids[b] - o) * 0.1
                    valid = inside & (tindex[b] == f)
                    ls, lt, _, _ = chamfer_distance(pred[None], gt[None], dst_valid=valid[None],
                                                    src_valid=live[None])
                    has = (valid.sum() > 0).to(ls.dtype)
                    total = total + (ls + lt) / 2. * lw[f] * has
            loss_dict["loss.dense_voxel"] = total / (float(np.sum(loss_weight)) * bs) * self.dense_loss_weight
        return loss_dict

    @torch.no_grad()
    def get_point_cloud_prediction(self, pred_dict, gt_points, start_idx, tgt_bev_h, tgt_bev_w,
                                   tgt_pc_range, img_metas=None, batched_origin_points=None):
        """arg-max decode (:663-752) -> dict(pred_pcds, gt_pcds, origin)."""
        bev_preds = pred_dict["next_bev_preds"].float()
        valid_frames = pred_dict["valid_frames"]
        F_, inter_num, bs, token_num, Z = bev_preds.shape
        H, W = tgt_bev_h, tgt_bev_w
        origin_grids, origin_pts, gt_grids, gt_xyz, tindex = self._process_gt_points(
            bev_preds, gt_points, batched_origin_points, valid_frames, start_idx, F_, H, W, tgt_pc_range)
        last = bev_preds[:, -1:]
        pred, gt = [], []
        for b in range(bs):
            p, g = ray_ops.ray_argmax(self._volumes(last, b, Z, H, W), origin_grids[b],
                                      torch.nan_to_num(gt_grids[b], nan=-1.0e6), tindex[b],
                                      self.ray_grid_step, self.ray_grid_num)
            pred.append(p); gt.append(g)
        scale = (tgt_pc_range[3] - tgt_pc_range[0]) / W
        pred = torch.stack(pred) * scale
        gt = torch.stack(gt) * scale
        return dict(
            pred_pcds=self.get_rendered_pcds(origin_pts, gt_xyz, tindex, gt, pred, tgt_pc_range),
            gt_pcds=self.get_rendered_pcds(origin_pts, gt_xyz, tindex, gt, gt, tgt_pc_range),
            origin=origin_pts)
