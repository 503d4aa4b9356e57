"""ViDARHeadV1 -- names / kwargs / parameter names / semantics of
projects/mmdet3d_plugin/bevformer/dense_heads/vidar_head_v1.py:24-250."""
from __future__ import annotations

import copy

import numpy as np
import torch

from ..utils.host import const_tensor, to_device_async
import torch.nn as nn

from ..bricks import Linear
from ..registry import HEADS
from .vidar_head_base import ViDARHeadBase


@HEADS.register_module()
class ViDARHeadV1(ViDARHeadBase):
    def __init__(self, history_queue_length, pred_history_frame_num=0, pred_future_frame_num=0,
                 per_frame_loss_weight=(1.0,), *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.history_queue_length = history_queue_length
        self.pred_history_frame_num = pred_history_frame_num
        self.pred_future_frame_num = pred_future_frame_num
        self.pred_frame_num = 1 + pred_history_frame_num + pred_future_frame_num
        self.per_frame_loss_weight = per_frame_loss_weight
        assert len(per_frame_loss_weight) == self.pred_frame_num
        branch = []
        for _ in range(self.num_pred_fcs):
            branch += [Linear(self.embed_dims, self.embed_dims), nn.LayerNorm(self.embed_dims),
                       nn.ReLU(inplace=True)]
        branch.append(Linear(self.embed_dims, self.pred_frame_num * self.num_pred_height))
        head = nn.Sequential(*branch)
        self.bev_pred_head = nn.ModuleList(
            [copy.deepcopy(head) for _ in range(self.transformer.decoder.num_layers)])

    def forward_head(self, next_bev_feats):
        """[F, inter, bs, Q, C] -> occupancy logits [F, inter, pred_frame_num, bs, Q, Z]; the
        history / future slices are residuals on the current-frame slice (:64-92)."""
        h = self.pred_history_frame_num
        outs = []
        for lvl in range(next_bev_feats.shape[1]):
            p = self.bev_pred_head[lvl](next_bev_feats[:, lvl])
            p = p.view(*p.shape[:-1], self.num_pred_height, self.pred_frame_num)
            base = p[..., h][..., None]
            p = torch.cat([p[..., :h] + base, base, p[..., h + 1:] + base], -1)
            outs.append(p.permute(0, 4, 1, 2, 3).contiguous())
        return torch.stack(outs, 1)

    def _get_reference_gt_points(self, gt_points, src_frame_idx_list, tgt_frame_idx_list, img_metas):
        """Move the GT cloud of frame src[f] into the coordinates of frame tgt[f] and relabel it as
        frame slot f (:94-148).  Static shapes: all points stay, points of frames outside
        `src_frame_idx_list` get slot -1 (the reference selects them out with boolean indexing)."""
        bs = len(gt_points)
        dev, dt = gt_points[0].device, gt_points[0].dtype
        src_to_tgt = []
        for s, t in zip(src_frame_idx_list, tgt_frame_idx_list):
            a = np.array([m["total_cur2ref_lidar_transform"][s] for m in img_metas])
            b = np.array([m["total_ref2cur_lidar_transform"][t] for m in img_metas])
            src_to_tgt.append(torch.matmul(to_device_async(a, dev, dt), to_device_async(b, dev, dt)))
        src_to_tgt = torch.stack(src_to_tgt, 1)                       # [bs, F, 4, 4] row-vector form
        origin = src_to_tgt[:, :, 3, :3].contiguous()                 # (0,0,0,1) @ M
        n_src = int(max(src_frame_idx_list)) + 2
        slot_host = np.full((n_src + 1,), -1.0, dtype=np.float32)
        for f, s in enumerate(src_frame_idx_list):
            slot_host[int(s)] = f
        slot_of = const_tensor(slot_host, dev, dt)
        out = []
        for b, p in enumerate(gt_points):
            frame = torch.nan_to_num(p[:, -1], nan=-1.0).long().clamp(min=-1, max=n_src - 1)
            slot = slot_of[frame]                                      # [-1 index hits the -1 tail]
            m = src_to_tgt[b][slot.clamp(min=0).long()]               # [P, 4, 4]
            hom = torch.cat([p[:, :3], p.new_ones(p.shape[0], 1)], 1)
            moved = (hom.unsqueeze(-1) * m).sum(1)          # row-vector x per-point 4x4, elementwise
            out.append(torch.cat([moved[:, :3], slot[:, None]], 1))
        return out, origin

    def loss(self, pred_dict, gt_points, start_idx, tgt_bev_h, tgt_bev_w, tgt_pc_range,
             pred_frame_num, img_metas=None, batched_origin_points=None):
        bev_preds = pred_dict["next_bev_preds"]
        valid_frames = np.array(pred_dict["valid_frames"])
        start_frames = valid_frames + self.history_queue_length - self.pred_history_frame_num
        tgt_frames = valid_frames + self.history_queue_length
        if not pred_dict.get("full_prev_bev_exists", True):
            frame_idx_for_loss = [self.pred_history_frame_num] * self.pred_frame_num
        else:
            frame_idx_for_loss = np.arange(0, self.pred_frame_num)
        loss_dict = dict()
        for idx, i in enumerate(frame_idx_for_loss):
            if idx != i:
                # the reference evaluates this slice and multiplies it by 0 (:215-216); skip the work
                for k in self._loss_keys():
                    loss_dict[f"frame.{idx}.{k}.loss"] = bev_preds.new_zeros(())
                continue
            cur_bev_preds = bev_preds[:, :, i, ...].contiguous()
            cur_gt, cur_origin = self._get_reference_gt_points(
                gt_points, src_frame_idx_list=start_frames + i, tgt_frame_idx_list=tgt_frames,
                img_metas=img_metas)
            if i != self.pred_history_frame_num:
                lw = np.array([[1]] + [[0]] * (len(self.loss_weight) - 1))
            else:
                lw = self.loss_weight
            cur = super().loss(dict(next_bev_preds=cur_bev_preds,
                                    valid_frames=np.arange(0, len(start_frames))),
                               cur_gt, start_idx=start_idx, tgt_bev_h=tgt_bev_h, tgt_bev_w=tgt_bev_w,
                               tgt_pc_range=tgt_pc_range, pred_frame_num=len(self.loss_weight) - 1,
                               img_metas=img_metas, batched_origin_points=cur_origin, loss_weight=lw)
            for k, v in cur.items():
                loss_dict[f"frame.{idx}.{k}.loss"] = v * self.per_frame_loss_weight[i]
        return loss_dict

    def _loss_keys(self):
        keys = []
        if self.use_ce_loss:
            keys.append("regularization.loss")
        if self.use_dense_loss:
            keys.append("loss.dense_voxel")
        return keys

    def get_point_cloud_prediction(self, pred_dict, gt_points, start_idx, tgt_bev_h, tgt_bev_w,
                                   tgt_pc_range, img_metas=None, batched_origin_points=None):
        pred_dict = dict(pred_dict)
        pred_dict["next_bev_preds"] = pred_dict["next_bev_preds"][:, :, self.pred_history_frame_num, ...].contiguous()
        valid_frames = np.array(pred_dict["valid_frames"])
        gt, origin = self._get_reference_gt_points(
            gt_points, src_frame_idx_list=valid_frames + self.history_queue_length,
            tgt_frame_idx_list=valid_frames + self.history_queue_length, img_metas=img_metas)
        return super().get_point_cloud_prediction(
            pred_dict=pred_dict, gt_points=gt, start_idx=start_idx, tgt_bev_h=tgt_bev_h,
            tgt_bev_w=tgt_bev_w, tgt_pc_range=tgt_pc_range, img_metas=img_metas,
            batched_origin_points=origin)
