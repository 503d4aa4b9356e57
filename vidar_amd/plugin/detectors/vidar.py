"""ViDAR detector -- training-step orchestration of
projects/mmdet3d_plugin/bevformer/detectors/vidar.py:28-387 (+ history-BEV loops of
detectors/bevformer.py:158-232): frozen history BEV -> (optionally back-propagated last history
frame) -> current BEV -> future decoder loop -> occupancy head -> ray / chamfer losses.

Image branch: `extract_feat` runs img_backbone + img_neck when the config provides ones this
package can build; the hot-path bench feeds FPN feature pyramids directly through `img_feats`
(list over levels of [bs, T, cams, C, h, w]) -- the backbone is the 'next' row of SURVEY §8(f)."""
from __future__ import annotations

import copy
import os

import numpy as np
import torch

from ..utils.host import to_device_async
import torch.nn as nn

from ..registry import BACKBONES, DETECTORS, NECKS, build_head
from ..utils import e2e_predictor_utils, eval_utils


@DETECTORS.register_module()
class ViDAR(nn.Module):
    def __init__(self, future_pred_head, future_pred_frame_num, test_future_frame_num,
                 point_cloud_range, bev_h, bev_w, random_drop_image_rate=0.0,
                 random_drop_prev_rate=0.0, random_drop_prev_start_idx=1,
                 random_drop_prev_end_idx=None, grid_mask_image=True, grid_mask_backbone_feat=False,
                 grid_mask_fpn_feat=False, grid_mask_prev=False, grid_mask_cfg=None,
                 supervise_all_future=True, _viz_pcd_flag=False, _viz_pcd_path="dbg/pred_pcd",
                 _submission=False, _submission_path="submission/model",
                 # BEVFormer / MVXTwoStageDetector arguments that matter on this path
                 img_backbone=None, img_neck=None, pts_bbox_head=None, use_grid_mask=False,
                 video_test_mode=False, backwarded_prev_frame_num=0, train_cfg=None, test_cfg=None,
                 pretrained=None, **kwargs):
        super().__init__()
        self.img_backbone = BACKBONES.build(img_backbone) if (
            img_backbone and img_backbone.get("type") in BACKBONES) else None
        self.img_neck = NECKS.build(img_neck) if (img_neck and img_neck.get("type") in NECKS) else None
        self.pts_bbox_head = build_head(pts_bbox_head)
        self.future_pred_head = build_head(future_pred_head)
        # GridMask augmentation (detectors/vidar.py:45-53, :88-92): training only, applied on device
        from ..utils.grid_mask import GridMask
        self.use_grid_mask = use_grid_mask
        self.grid_mask_image = grid_mask_image
        self.grid_mask_backbone_feat = grid_mask_backbone_feat
        self.grid_mask_fpn_feat = grid_mask_fpn_feat
        self.grid_mask_prev = grid_mask_prev
        self.grid_mask = GridMask(**(grid_mask_cfg or dict(use_h=True, use_w=True, rotate=1, offset=False,
                                                           ratio=0.5, mode=1, prob=0.7)))
        self.video_test_mode = video_test_mode
        self.backwarded_prev_frame_num = backwarded_prev_frame_num
        self.future_pred_frame_num = future_pred_frame_num
        self.test_future_frame_num = test_future_frame_num
        self.only_train_cur_frame = future_pred_frame_num == 0
        self.point_cloud_range = point_cloud_range
        self.bev_h, self.bev_w = bev_h, bev_w
        self.random_drop_image_rate = random_drop_image_rate
        self.random_drop_prev_rate = random_drop_prev_rate
        self.random_drop_prev_start_idx = random_drop_prev_start_idx
        self.random_drop_prev_end_idx = random_drop_prev_end_idx
        self.supervise_all_future = supervise_all_future
        self._submission = _submission
        self._submission_path = _submission_path
        if self.only_train_cur_frame:               # vidar.py:109-115
            del self.future_pred_head.transformer
            del self.future_pred_head.bev_embedding
            del self.future_pred_head.prev_frame_embedding
            del self.future_pred_head.can_bus_mlp
            del self.future_pred_head.positional_encoding
            self.future_pred_head.transformer = None

    def init_weights(self):
        self.pts_bbox_head.init_weights()
        self.future_pred_head.init_weights()

    # ---- image branch ----------------------------------------------------------------------------
    def extract_feat(self, img, img_metas=None, len_queue=None):
        if self.img_backbone is None:
            raise RuntimeError("no image backbone was built; pass FPN features through `img_feats`")
        B = img.size(0)
        if img.dim() == 5:
            Bn, N, C, H, W = img.shape
            img = img.reshape(Bn * N, C, H, W)
        if self.use_grid_mask and self.grid_mask_image:              # vidar.py:139-140
            img = self.grid_mask(img)
        feats = self.img_backbone(img)
        if self.use_grid_mask and self.grid_mask_backbone_feat:      # :145-150
            feats = [self.grid_mask(f) for f in feats]
        if self.img_neck is not None:
            feats = self.img_neck(feats)
            if self.use_grid_mask and self.grid_mask_fpn_feat:       # :156-161
                feats = [self.grid_mask(f) for f in feats]
        out = []
        for f in feats:
            BN, C, H, W = f.shape
            if len_queue is not None:
                out.append(f.view(B // len_queue, len_queue, BN // B, C, H, W))
            else:
                out.append(f.view(B, BN // B, C, H, W))
        if img_metas and img_metas[0].get("aug_param") and \
                img_metas[0]["aug_param"]["CropResizeFlipImage_param"][-1] is True:
            out = [torch.flip(x, dims=[-1]) for x in out]
        return out

    def _queue_feats(self, img, img_feats, img_metas_list, start, end, grad):
        """FPN pyramids of frames [start, end) as list over levels of [bs, n, cams, C, h, w]."""
        if img_feats is not None:
            return [f[:, start:end] for f in img_feats]
        n = end - start
        bs = img.shape[0]
        ctx = torch.enable_grad() if grad else torch.no_grad()
        with ctx:
            x = img[:, start:end].reshape(bs * n, *img.shape[2:])
            return self.extract_feat(x, [m[end - 1] for m in img_metas_list], len_queue=n)

    # ---- history BEV (bevformer.py:158-232) --------------------------------------------------------
    def obtain_history_bev(self, img, img_metas_list, img_feats=None, num_frames=None,
                           drop_prev_index=-1):
        back = self.backwarded_prev_frame_num if self.training else 0
        split = num_frames - back
        prev_bev = None
        was_training = self.training
        if was_training:
            self.eval()
        with torch.no_grad():
            # image features of ALL history frames are frozen (eval, no grad) -- also those of the
            # back-propagated frames (bevformer.py:196-205) -- so one backbone pass serves both loops
            feats = self._queue_feats(img, img_feats, img_metas_list, 0, num_frames, grad=False) \
                if num_frames > 0 else None
            for i in range(split):
                metas = [m[i] for m in img_metas_list]
                if not metas[0]["prev_bev_exists"]:
                    prev_bev = None
                prev_bev = self.pts_bbox_head([f[:, i] for f in feats], metas, prev_bev, only_bev=True)
                if i < drop_prev_index:
                    prev_bev = None
        if was_training:
            self.train()
        for i in range(split, num_frames):
            metas = [m[i] for m in img_metas_list]
            if not metas[0]["prev_bev_exists"]:
                prev_bev = None
            prev_bev = self.pts_bbox_head([f[:, i] for f in feats], metas, prev_bev, only_bev=True)
        return prev_bev

    # ---- future alignment (vidar.py:175-237) -------------------------------------------------------
    def _get_history_ref_to_previous_transform(self, tensor, num_frames, img_metas_list):
        mats = [[m[i]["ref_lidar_to_cur_lidar"] for i in range(num_frames)] for m in img_metas_list]
        return to_device_async(np.array(mats), tensor.device, tensor.dtype)

    def _align_bev_coordnates(self, frame_idx, ref_to_history_list, img_metas):
        bs, num_frame = ref_to_history_list.shape[:2]
        t = ref_to_history_list
        future2ref = to_device_async(np.array([m["future2ref_lidar_transform"][frame_idx] for m in img_metas]), t.device, t.dtype)
        ref2future = to_device_async(np.array([m["ref2future_lidar_transform"][frame_idx] for m in img_metas]), t.device, t.dtype)
        future_to_history = torch.matmul(future2ref.unsqueeze(1).repeat(1, num_frame, 1, 1), t)
        grids = e2e_predictor_utils.get_bev_grids(self.bev_h, self.bev_w, bs * num_frame, device=t.device)
        grids = grids.view(bs, num_frame, -1, 2)
        coords = e2e_predictor_utils.bev_grids_to_coordinates(grids, self.point_cloud_range)
        coords = torch.cat([coords, torch.ones_like(coords[..., :2])], -1)
        coords = torch.matmul(coords, future_to_history)[..., :2]
        aligned, _ = e2e_predictor_utils.bev_coords_to_grids(coords, self.bev_h, self.bev_w,
                                                             self.point_cloud_range)
        aligned = ((aligned + 1) / 2.).permute(0, 2, 1, 3).contiguous()
        return grids[:, -1].contiguous(), aligned, ref2future

    # ---- training step (vidar.py:240-387) ----------------------------------------------------------
    def _plan_sca(self, img_metas, num_frames, device):
        """camera projection + visible-query index of every frame of the queue in one kernel call and one
        host read (instead of one of each per encoder pass); the encoder picks the plan up from the metas."""
        enc = getattr(getattr(self.pts_bbox_head, "transformer", None), "encoder", None)
        if enc is None or not hasattr(enc, "plan_frames") or device.type != "cuda":
            return
        frames = [[m[t] for m in img_metas] for t in range(num_frames)]
        for metas, plan in zip(frames, enc.plan_frames(frames, self.bev_h, self.bev_w, device)):
            for m in metas:
                m["_sca_plan"] = plan

    def forward_train(self, points=None, img_metas=None, img=None, gt_points=None, img_feats=None,
                      **kwargs):
        num_frames = img.size(1) if img is not None else img_feats[0].size(1)
        self._plan_sca(img_metas, num_frames, (img if img is not None else img_feats[0]).device)
        if img is not None and np.random.rand() < self.random_drop_image_rate:
            img[:, -1:, ...] = torch.zeros_like(img[:, -1:, ...])
        if np.random.rand() < self.random_drop_prev_rate:
            end = self.random_drop_prev_end_idx if self.random_drop_prev_end_idx is not None else num_frames
            drop_prev_index = np.random.randint(self.random_drop_prev_start_idx, end)
        else:
            drop_prev_index = -1
        prev_img_metas = copy.deepcopy(img_metas)
        prev_bev = self.obtain_history_bev(img, prev_img_metas, img_feats, num_frames - 1,
                                           drop_prev_index=drop_prev_index)
        if self.grid_mask_prev and prev_bev is not None:             # vidar.py:289-295
            b, n, c = prev_bev.shape
            pb = prev_bev.view(b, self.bev_h, self.bev_w, c).permute(0, 3, 1, 2).contiguous()
            prev_bev = self.grid_mask(pb).view(b, c, n).permute(0, 2, 1).contiguous()
        cur_metas = [m[num_frames - 1] for m in img_metas]
        cur_feats = [f[:, 0] for f in self._queue_feats(img, img_feats, img_metas, num_frames - 1,
                                                        num_frames, grad=True)]
        if not cur_metas[0]["prev_bev_exists"]:
            prev_bev = None
        # The reference asserts bs == 1 here (vidar.py:306): it reads the history flags of sample 0
        # for the whole batch.  Larger per-GPU batches are allowed when that reading is unambiguous,
        # i.e. every sample of the batch has the same prev_bev_exists pattern.
        flags = [[bool(meta[k]["prev_bev_exists"]) for k in range(len(meta))] for meta in prev_img_metas]
        if any(f != flags[0] for f in flags[1:]):
            raise ValueError("samples of one batch must share their prev_bev_exists pattern "
                             "(the reference only supports bs=1, vidar.py:306)")
        exists = []
        for meta in prev_img_metas:
            ok = True
            for k in range(len(meta) - 1, -1, -1):
                exists.append(ok)
                ok = ok and meta[k]["prev_bev_exists"]
        prev_bev_exists_list = np.array(exists)[::-1]

        ref_bev = self.pts_bbox_head(cur_feats, cur_metas, prev_bev, only_bev=True)
        n_heads = len(self.future_pred_head.bev_pred_head)
        next_bev_feats = [ref_bev.unsqueeze(0).repeat(n_heads, 1, 1, 1).contiguous()]
        valid_frames = [0]
        prev_bev_input = ref_bev.unsqueeze(1)
        ref_metas = [[m[num_frames - 1]] for m in prev_img_metas]
        ref_to_history = self._get_history_ref_to_previous_transform(prev_bev_input, 1, ref_metas)
        if not self.only_train_cur_frame:
            if self.supervise_all_future:
                valid_frames.extend(range(1, self.future_pred_frame_num + 1))
            else:
                valid_frames.append(int(np.random.choice(np.arange(1, self.future_pred_frame_num + 1), 1)[0]))
            for k in range(1, self.future_pred_frame_num + 1):
                tgt, aligned_prev, ref2future = self._align_bev_coordnates(k, ref_to_history, cur_metas)
                with torch.set_grad_enabled(k in valid_frames):
                    pred_feat = self.future_pred_head(prev_bev_input, cur_metas, k, tgt_points=tgt,
                                                      bev_h=self.bev_h, bev_w=self.bev_w,
                                                      ref_points=aligned_prev)
                if k in valid_frames:
                    next_bev_feats.append(pred_feat)
                prev_bev_input = torch.cat([prev_bev_input, pred_feat[-1].unsqueeze(1)], 1)[:, 1:].contiguous()
                ref_to_history = torch.cat([ref_to_history, ref2future.unsqueeze(1)], 1)[:, 1:].contiguous()
        next_bev_feats = torch.stack(next_bev_feats, 0)
        next_bev_preds = self.future_pred_head.forward_head(next_bev_feats)
        pred_dict = dict(next_bev_features=next_bev_feats, next_bev_preds=next_bev_preds,
                         valid_frames=valid_frames, full_prev_bev_exists=bool(prev_bev_exists_list.all()),
                         prev_bev_exists_list=prev_bev_exists_list)
        return self.future_pred_head.loss(pred_dict, gt_points, 0, tgt_bev_h=self.bev_h,
                                          tgt_bev_w=self.bev_w, tgt_pc_range=self.point_cloud_range,
                                          pred_frame_num=self.future_pred_frame_num + 1,
                                          img_metas=cur_metas)

    # ---- evaluation (vidar.py:389-502) ---------------------------------------------------------------
    @torch.no_grad()
    def forward_test(self, img_metas, img=None, gt_points=None, img_feats=None, **kwargs):
        """history BEV over all frames -> auto-regressive future BEVs -> arg-max decode -> per-frame
        squared-L2 chamfer distance (compute_chamfer_distance_inner) and the 4d-occ ray errors
        (utils/eval_utils.py:185-225: l1_error / absrel_error)."""
        self.eval()
        num_frames = img.size(1) if img is not None else img_feats[0].size(1)
        self._plan_sca(img_metas, num_frames, (img if img is not None else img_feats[0]).device)
        prev_bev = self.obtain_history_bev(img, img_metas, img_feats, num_frames)
        prev_bev = prev_bev[:, None, ...].contiguous()
        n_heads = len(self.future_pred_head.bev_pred_head)
        next_bev_feats = [prev_bev[:, -1].unsqueeze(0).repeat(n_heads, 1, 1, 1).contiguous()]
        valid_frames = [0] + list(range(1, self.test_future_frame_num + 1))
        ref_metas = [[m[num_frames - 1]] for m in img_metas]
        ref_to_history = self._get_history_ref_to_previous_transform(prev_bev, prev_bev.shape[1], ref_metas)
        cur_metas = [m[num_frames - 1] for m in img_metas]
        for k in range(1, self.test_future_frame_num + 1):
            tgt, aligned_prev, ref2future = self._align_bev_coordnates(k, ref_to_history, cur_metas)
            nxt = self.future_pred_head(prev_bev, cur_metas, k, tgt_points=tgt, bev_h=self.bev_h,
                                        bev_w=self.bev_w, ref_points=aligned_prev)
            next_bev_feats.append(nxt)
            prev_bev = torch.cat([prev_bev, nxt[-1].unsqueeze(1)], 1)[:, 1:].contiguous()
            ref_to_history = torch.cat([ref_to_history, ref2future.unsqueeze(1)], 1)[:, 1:].contiguous()
        next_bev_feats = torch.stack(next_bev_feats, 0)
        pred_dict = dict(next_bev_features=next_bev_feats,
                         next_bev_preds=self.future_pred_head.forward_head(next_bev_feats),
                         valid_frames=valid_frames)
        decode = self.future_pred_head.get_point_cloud_prediction(
            pred_dict, gt_points, 0, tgt_bev_h=self.bev_h, tgt_bev_w=self.bev_w,
            tgt_pc_range=self.point_cloud_range, img_metas=cur_metas)
        ret = dict()
        for f in range(len(decode["pred_pcds"][0])):
            cd, l1, absrel, count = 0.0, 0.0, 0.0, 0
            for b in range(len(decode["pred_pcds"])):
                pred, gt = decode["pred_pcds"][b][f], decode["gt_pcds"][b][f]
                cd += float(e2e_predictor_utils.compute_chamfer_distance_inner(pred, gt, self.point_cloud_range))
                if pred.shape[0] > 0:
                    e1, e2 = eval_utils.compute_ray_errors(
                        pred.cpu().numpy().astype(np.float64), gt.cpu().numpy().astype(np.float64),
                        decode["origin"][b, f].cpu().numpy().astype(np.float64), pred.device)
                    l1 += float(e1); absrel += float(e2)
                if self._submission and f > 0:      # frame 0 is the current frame (vidar.py:491-494)
                    self._save_prediction(pred, cur_metas[b], f)
                count += 1
            ret[f"frame.{f}"] = dict(count=count, chamfer_distance=cd, l1_error=l1, absrel_error=absrel)
        return [ret]

    def _save_prediction(self, pred_pcd, img_meta, frame_idx):
        """One text file per (sample, future frame): `<sample_idx>_<frame>.txt`, one predicted ray
        depth per line, '%f' formatted (vidar.py:503-519)."""
        os.makedirs(self._submission_path, exist_ok=True)
        name = os.path.join(self._submission_path, f"{img_meta['sample_idx']}_{frame_idx}.txt")
        depth = torch.sqrt((pred_pcd ** 2).sum(1)).cpu().numpy()
        with open(name, "w") as f:
            f.write("".join("%f\n" % d for d in depth))

    def forward(self, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(**kwargs)
        return self.forward_test(**kwargs)
