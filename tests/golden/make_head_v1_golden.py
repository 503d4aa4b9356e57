"""Golden vectors for the future-prediction head from the reference's OWN Python:
ViDARHeadV1 (dense_heads/vidar_head_v1.py) on ViDARHeadTemplate/ViDARHeadBase (vidar_head_base.py)
with PredictionTransformer -> PredictionDecoder -> PredictionTransformerLayer
(PredictionMSDeformableAttention x2 + LatentRendering + FFN) (modules/vidar_{transformer,decoder}.py),
plus the detector's BEV alignment helpers (detectors/vidar.py:170-237) that produce its inputs.
Reduced width (embed 64, 2 heads, BEV 12x12, 2 decoder layers), CPU, functional mmcv stand-in.

Stored: full reference state_dict, inputs, `forward` output (all decoder layers), `forward_head`
output, `_get_reference_gt_points`, the complete `loss` (with the gumbel noise it drew) and the
alignment helper outputs.       python tests/golden/make_head_v1_golden.py"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(HERE)); sys.path.insert(0, str(ROOT))
import ref_mmcv_functional as R  # noqa: E402

PC = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
D, HEADS, BEV, Z = 64, 2, 12, 16
HIST, NPREV = 2, 1        # image history frames; BEV frames in the decoder's memory (the detector
                          # always keeps exactly one: vidar.py:399, :343-346)


def head_cfg():
    attn = dict(type="PredictionMSDeformableAttention", embed_dims=D, num_heads=HEADS, num_levels=1)
    return dict(
        type="ViDARHeadV1", history_queue_length=HIST, pred_history_frame_num=1, pred_future_frame_num=1,
        per_frame_loss_weight=[0.5, 1.0, 1.2], ray_grid_num=512, ray_grid_step=1.0, use_ce_loss=True,
        use_dist_loss=False, use_dense_loss=True, num_pred_fcs=1, num_pred_height=Z, can_bus_norm=True,
        can_bus_dims=[0, 1, 2, 17], bev_h=BEV, bev_w=BEV, pc_range=PC, loss_weight=[[1], [0.5]],
        positional_encoding=dict(type="LearnedPositionalEncoding", num_feats=D // 2, row_num_embed=BEV,
                                 col_num_embed=BEV),
        transformer=dict(
            type="PredictionTransformer", embed_dims=D,
            decoder=dict(
                type="PredictionDecoder", num_layers=2, return_intermediate=True, keep_idx=[1],
                transformerlayers=dict(
                    type="PredictionTransformerLayer", attn_cfgs=[dict(attn), dict(attn)],
                    ffn_cfgs=dict(type="FFN", embed_dims=D, feedforward_channels=128, num_fcs=2, ffn_drop=0.1,
                                  act_cfg=dict(type="ReLU", inplace=True)),
                    feedforward_channels=128, ffn_dropout=0.1,
                    latent_render=dict(embed_dims=D, pred_height=Z, num_pred_fcs=0, grid_step=0.5,
                                       grid_num=32, reduction=4, act="sigmoid"),
                    operation_order=("self_attn", "norm", "cross_attn", "norm", "latent_render", "ffn",
                                     "norm")))))


def perturb(model, seed=6):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(torch.randn(p.shape, generator=g) * 0.05)


def main():
    from vidar_amd.synthetic import make_sample
    head_mod, e2e = R.reference_heads()
    Helpers = R.reference_detector_helpers()
    torch.manual_seed(0); np.random.seed(0)
    head = R.build_from_cfg(head_cfg(), R.HEADS)
    head.init_weights()
    perturb(head)
    head.eval()

    metas, gt = make_sample(3, queue_length=HIST, future_frames=2, rays_per_frame=60, num_cams=1)
    ref_meta = metas[-1]
    # ---- detector alignment helpers on the reference frame meta -------------------------------
    import types
    det = types.SimpleNamespace(bev_h=BEV, bev_w=BEV, point_cloud_range=PC)
    g = torch.Generator().manual_seed(1)
    prev_feats = torch.randn(1, NPREV, BEV * BEV, D, generator=g)
    ref_to_history = Helpers._get_history_ref_to_previous_transform(det, prev_feats, NPREV, [[ref_meta]])
    tgt, aligned, ref2future = Helpers._align_bev_coordnates(det, 1, ref_to_history, [ref_meta])

    # ---- head forward (PredictionTransformer) ---------------------------------------------------
    out = head(prev_feats, [ref_meta], 1, tgt, aligned, BEV, BEV)           # [layers, bs, HW, D]
    feats = torch.stack([out, out.flip(0) * 0.5 + 0.1], 0)                 # [pred_frames=2, layers, bs, HW, D]
    preds = head.forward_head(feats)

    # ---- GT re-referencing + loss -----------------------------------------------------------------
    gt_t = [torch.from_numpy(gt)]
    src_list, tgt_list = [HIST - 1, HIST, HIST + 1], [HIST, HIST, HIST]
    al_pts, al_org = head._get_reference_gt_points(gt_t, src_list, tgt_list, [ref_meta])

    NOISE = {}
    def fake_gumbel(logits, tau=1, hard=False, eps=1e-10, dim=-1):
        gn = -torch.empty_like(logits).exponential_(generator=torch.Generator().manual_seed(5)).log()
        NOISE.setdefault("g", []).append(gn)
        idx = torch.softmax(logits + gn, dim).max(dim, keepdim=True)[1]
        return torch.zeros_like(logits).scatter_(dim, idx, 1.0)
    sys.modules["refbev.dense_heads.vidar_head_base"].F.gumbel_softmax = fake_gumbel
    preds_l = preds.detach().clone().requires_grad_(True)
    pred_dict = dict(next_bev_features=feats, next_bev_preds=preds_l, valid_frames=[0, 1])
    loss = head.loss(pred_dict, gt_t, 0, BEV, BEV, PC, 2, img_metas=[ref_meta])
    total = sum(loss.values())
    gpred, = torch.autograd.grad(total, preds_l)

    sd = {"sd/" + k: v.detach().numpy() for k, v in head.state_dict().items()}
    meta_np = {"meta/" + k: np.asarray(ref_meta[k]) for k in
               ("future_can_bus", "future2ref_lidar_transform", "ref2future_lidar_transform",
                "total_cur2ref_lidar_transform", "total_ref2cur_lidar_transform", "can_bus")}
    np.savez_compressed(
        HERE / "head_v1_small.npz", **sd, **meta_np, cfg_json=np.array(json.dumps(head_cfg())),
        hist_ref_lidar_to_cur_lidar=np.stack([ref_meta["ref_lidar_to_cur_lidar"]]),
        prev_feats=prev_feats.numpy(), ref_to_history=ref_to_history.numpy(), tgt_points=tgt.numpy(),
        ref_points=aligned.numpy(), ref2future=ref2future.numpy(), out=out.detach().numpy(),
        feats=feats.detach().numpy(), preds=preds.detach().numpy(), gt_points=gt,
        src_list=np.array(src_list), tgt_list=np.array(tgt_list), aligned_gt=al_pts[0].numpy(),
        aligned_origin=al_org.numpy(), loss_names=np.array(sorted(loss)),
        loss_values=np.array([float(loss[k]) for k in sorted(loss)]), grad_preds=gpred.numpy(),
        noise=np.stack([n.numpy() for n in NOISE["g"]]) if len({n.shape for n in NOISE["g"]}) == 1
        else np.concatenate([n.numpy().reshape(-1) for n in NOISE["g"]]),
        noise_shapes=np.array([list(n.shape) for n in NOISE["g"]]))
    print("wrote head_v1_small.npz", out.shape, preds.shape, {k: float(v) for k, v in loss.items()},
          [tuple(n.shape) for n in NOISE["g"]], "keys", len(sd))


if __name__ == "__main__":
    main()
