"""Golden vectors for the training-step ORCHESTRATION from the reference's own source text:
ViDAR.forward_train + BEV alignment helpers (detectors/vidar.py:170-387), the history-BEV loops of
BEVFormer (detectors/bevformer.py:158-232) and ViDARBEVFormerHead.forward
(dense_heads/vidar_bevformer_head.py:25-61) are exec'd inside throw-away classes (their base classes
MVXTwoStageDetector / DETRHead are mmdet3d / mmdet code that cannot be imported here), wired to the
reference's real PerceptionTransformer and ViDARHeadV1 modules (ref_mmcv_functional.py).
Image features come from a fixed linear "backbone" (adaptive pooling + seeded 1x1 projection) that
the test reproduces, so both sides see identical pyramids.

Covers: frozen history loop, the back-propagated history frame (backwarded_prev_frame_num=1), the
prev_bev_exists logic, the auto-regressive future loop with BEV re-alignment, loss aggregation.
Reduced width (embed 64, 2 heads, 2 cameras, BEV 12x12, 3 image frames, 2 future frames), CPU.
    python tests/golden/make_detector_golden.py"""
import copy
import json
import sys
import types
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(HERE)); sys.path.insert(0, str(ROOT))
import ref_mmcv_functional as R  # noqa: E402
from make_transformer_golden import PC  # noqa: E402

D, HEADS, CAMS, BEV, Z = 64, 2, 2, 12, 16
SHAPES = [(12, 20), (6, 10)]
IMG_HW = (24, 40)
QUEUE, FUTURE = 2, 2                  # image history frames, predicted future frames


def bev_head_cfg():
    from make_transformer_golden import transformer_cfg
    t = transformer_cfg()
    t["num_cams"] = CAMS
    t["encoder"]["transformerlayers"]["attn_cfgs"][1]["num_cams"] = CAMS
    return dict(type="ViDARBEVFormerHead", bev_h=BEV, bev_w=BEV, num_query=9, num_classes=10, in_channels=D,
                with_box_refine=True, as_two_stage=False, transformer=t,
                bbox_coder=dict(type="NMSFreeCoder", pc_range=PC),
                positional_encoding=dict(type="LearnedPositionalEncoding", num_feats=D // 2,
                                         row_num_embed=BEV, col_num_embed=BEV))


def future_head_cfg():
    from make_head_v1_golden import head_cfg
    h = head_cfg()
    h.update(history_queue_length=QUEUE, pred_history_frame_num=1, pred_future_frame_num=1,
             per_frame_loss_weight=[0.5, 1.0, 1.2], loss_weight=[[1], [1], [0.5]])
    return h


def model_cfg():
    return dict(type="ViDAR", use_grid_mask=False, video_test_mode=True, point_cloud_range=PC, bev_h=BEV,
                bev_w=BEV, future_pred_frame_num=FUTURE, test_future_frame_num=FUTURE,
                supervise_all_future=True, random_drop_prev_rate=0.0, random_drop_image_rate=0.0,
                backwarded_prev_frame_num=1, pts_bbox_head=bev_head_cfg(), future_pred_head=future_head_cfg())


def projections():
    g = torch.Generator().manual_seed(21)
    return [torch.randn(D, 3, generator=g) for _ in SHAPES]


def pyramid(img, proj):
    """[n, 3, H, W] -> list of [n, D, h, w]: the linear stand-in for backbone + neck"""
    return [torch.einsum("dc,nchw->ndhw", p, F.adaptive_avg_pool2d(img, s)) for s, p in zip(SHAPES, proj)]


def sample():
    from vidar_amd.synthetic import make_sample
    metas, gt = make_sample(4, queue_length=QUEUE, future_frames=FUTURE + 1, rays_per_frame=50, num_cams=CAMS)
    img = torch.randn(1, QUEUE + 1, CAMS, 3, *IMG_HW, generator=torch.Generator().manual_seed(8))
    return metas, gt, img


def zero_dropout(m):
    for x in m.modules():
        if isinstance(x, nn.Dropout):
            x.p = 0.0


def build_reference():
    head_mod, _ = R.reference_heads()
    e2e, eval_utils = R.reference_eval_stack()
    ident = lambda *a, **k: (lambda f: f)
    import os
    ns = dict(torch=torch, np=np, copy=copy, os=os, mmcv=sys.modules["mmcv"], e2e_predictor_utils=e2e,
              eval_utils=eval_utils, auto_fp16=ident, force_fp32=ident)
    src = (R.PLUGIN / "bevformer/detectors/vidar.py").read_text()
    a = src.index("    def _get_history_ref_to_previous_transform(")
    b = src.index("    def _viz_pcd(self, pred_pcd, pred_ctr,  output_path, gt_pcd=None):")
    bsrc = (R.PLUGIN / "bevformer/detectors/bevformer.py").read_text()
    c = bsrc.index("    def _obtain_frozen_history_bev(")
    d = bsrc.index("    @auto_fp16(apply_to=('img', 'points'))\n    def forward_train(")
    exec("class RefDet(torch.nn.Module):\n" + bsrc[c:d] + "\n" + src[a:b], ns)
    hsrc = (R.PLUGIN / "bevformer/dense_heads/vidar_bevformer_head.py").read_text()
    e = hsrc.index("    @auto_fp16(apply_to=('mlvl_feats'))\n    def forward(self, mlvl_feats, img_metas, prev_bev=None")
    exec("class RefBEVHead(torch.nn.Module):\n" + hsrc[e:], ns)

    cfg = model_cfg()
    bh = ns["RefBEVHead"]()
    hc = cfg["pts_bbox_head"]
    bh.bev_h = bh.bev_w = BEV
    bh.real_w, bh.real_h = PC[3] - PC[0], PC[4] - PC[1]
    bh.bev_embedding = nn.Embedding(BEV * BEV, D)
    bh.positional_encoding = R.build_from_cfg(hc["positional_encoding"], R.POSITIONAL_ENCODING)
    bh.transformer = R.build_from_cfg(hc["transformer"], R.TRANSFORMER)
    bh.transformer.init_weights()
    del bh.transformer.reference_points                     # ViDARBEVFormerHead.init_weights (:20-23)
    fh = R.build_from_cfg(cfg["future_pred_head"], R.HEADS)
    fh.init_weights()

    det = ns["RefDet"]()
    det.pts_bbox_head, det.future_pred_head = bh, fh
    det.bev_h = det.bev_w = BEV
    det.point_cloud_range = PC
    det.future_pred_frame_num = FUTURE
    det.only_train_cur_frame = False
    det.supervise_all_future = True
    det.random_drop_image_rate = det.random_drop_prev_rate = 0.0
    det.random_drop_prev_start_idx, det.random_drop_prev_end_idx = 1, None
    det.grid_mask_prev = False
    det.backwarded_prev_frame_num = 1
    det.test_future_frame_num = FUTURE
    det._viz_pcd_flag = det._submission = False
    proj = projections()

    def extract_feat(img, img_metas=None, len_queue=None):
        B = img.size(0)
        x = img.reshape(-1, *img.shape[-3:])
        out = []
        for f in pyramid(x, proj):
            BN = f.shape[0]
            out.append(f.view(B // len_queue, len_queue, BN // B, *f.shape[1:]) if len_queue is not None
                       else f.view(B, BN // B, *f.shape[1:]))
        return out
    det.extract_feat = extract_feat
    return det


def perturb(model, seed=7):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(torch.randn(p.shape, generator=g) * 0.05)


def main():
    torch.manual_seed(0); np.random.seed(0)
    det = build_reference()
    perturb(det)
    zero_dropout(det)
    det.train()
    metas, gt, img = sample()

    NOISE = []
    def fake_gumbel(logits, tau=1, hard=False, eps=1e-10, dim=-1):
        gn = -torch.empty_like(logits).exponential_(generator=torch.Generator().manual_seed(50 + len(NOISE))).log()
        NOISE.append(gn)
        idx = torch.softmax(logits + gn, dim).max(dim, keepdim=True)[1]
        return torch.zeros_like(logits).scatter_(dim, idx, 1.0)
    sys.modules["refbev.dense_heads.vidar_head_base"].F.gumbel_softmax = fake_gumbel

    losses = det.forward_train(img_metas=[copy.deepcopy(metas)], img=img.clone(), gt_points=[torch.from_numpy(gt)])
    total = sum(losses.values())
    names = ["pts_bbox_head.bev_embedding.weight",
             "pts_bbox_head.transformer.encoder.layers.0.attentions.1.deformable_attention.value_proj.weight",
             "future_pred_head.transformer.decoder.layers.1.attentions.1.sampling_offsets.weight",
             "future_pred_head.bev_pred_head.1.3.weight"]
    params = dict(det.named_parameters())
    grads = torch.autograd.grad(total, [params[n] for n in names])

    # ---- forward_test with the same weights (vidar.py:389-502) -------------------------------------
    with torch.no_grad():
        res = det.forward_test([copy.deepcopy(metas)], img=img.clone(), gt_points=[torch.from_numpy(gt)])[0]
    det.train()
    test_keys = sorted(res)
    test_vals = np.array([[res[k]["count"], res[k]["chamfer_distance"], res[k]["l1_error"], res[k]["absrel_error"]]
                          for k in test_keys], np.float64)

    sd = {"sd/" + k: v.detach().numpy() for k, v in det.state_dict().items()}
    np.savez_compressed(
        HERE / "detector_small.npz", **sd, test_keys=np.array(test_keys), test_values=test_vals, cfg_json=np.array(json.dumps(model_cfg())),
        loss_names=np.array(sorted(losses)), loss_values=np.array([float(losses[k]) for k in sorted(losses)]),
        noise_seeds=np.array([50 + i for i in range(len(NOISE))]),
        noise_shapes=np.array([list(n.shape) for n in NOISE]),
        noise_sums=np.array([float(n.double().sum()) for n in NOISE]),
        grad_names=np.array(names), **{f"grad{i}": g.numpy() for i, g in enumerate(grads)})
    print("wrote detector_small.npz", {k: round(float(v), 5) for k, v in losses.items()},
          [tuple(n.shape) for n in NOISE], "keys", len(sd), "test", dict(zip(test_keys, test_vals.round(4).tolist())))


if __name__ == "__main__":
    main()
