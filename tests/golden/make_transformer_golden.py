"""Golden vectors for the BEV encoder stack from the reference's OWN Python modules:
PerceptionTransformer.get_bev_features -> CustomBEVFormerEncoder -> BEVFormerLayerV2
(TemporalSelfAttention + SpatialCrossAttention/MSDeformableAttention3D + LatentRendering + FFN)
(projects/mmdet3d_plugin/bevformer/modules/{transformer,encoder,encoder_v2,spatial_cross_attention,
temporal_self_attention,custom_base_transformer_layer}.py), imported in place from /root/reference
with the functional mmcv stand-in of ref_mmcv_functional.py, built from a released-config-shaped
dict at reduced width (embed 64, 2 heads of 32 channels, 2 layers, 3 cameras, 2 pyramid levels,
BEV 12x12) and run on CPU (the reference's pure-PyTorch MSDA fallback).

Stored: the reference module's full state_dict (checkpoint key compatibility is part of the test),
the inputs, the BEV embedding with and without a previous BEV, and gradients w.r.t. the BEV queries
and two parameters.     Run where /root/reference exists:  python tests/golden/make_transformer_golden.py"""
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
D, HEADS, CAMS, BEV = 64, 2, 3, 12
SHAPES = [(12, 20), (6, 10)]


def transformer_cfg():
    return dict(
        type="PerceptionTransformer", rotate_prev_bev=True, use_shift=True, use_can_bus=True,
        embed_dims=D, num_cams=CAMS, num_feature_levels=len(SHAPES), rotate_center=[BEV // 2, BEV // 2],
        encoder=dict(
            type="CustomBEVFormerEncoder", keep_idx=[1], latent_rendering_lid=[1], num_layers=2,
            pc_range=PC, num_points_in_pillar=4, return_intermediate=False,
            transformerlayers=dict(
                type="BEVFormerLayerV2",
                attn_cfgs=[dict(type="TemporalSelfAttention", embed_dims=D, num_heads=HEADS, num_levels=1),
                           dict(type="SpatialCrossAttention", pc_range=PC, num_cams=CAMS, embed_dims=D,
                                deformable_attention=dict(type="MSDeformableAttention3D", embed_dims=D,
                                                          num_heads=HEADS, num_points=8,
                                                          num_levels=len(SHAPES)))],
                latent_render=dict(embed_dims=D, pred_height=16, num_pred_fcs=0, grid_step=1.0,
                                   grid_num=32, reduction=4, act="sigmoid"),
                # the reference layer asserts ffn_cfgs['embed_dims'] == embed_dims and its default says 256
                # (custom_base_transformer_layer.py:150-153): at another width ffn_cfgs must be explicit
                ffn_cfgs=dict(type="FFN", embed_dims=D, feedforward_channels=128, num_fcs=2, ffn_drop=0.1,
                              act_cfg=dict(type="ReLU", inplace=True)),
                feedforward_channels=128, ffn_dropout=0.1,
                operation_order=("self_attn", "norm", "cross_attn", "norm", "latent_render", "ffn", "norm"))))


def inputs(seed=0):
    from vidar_amd.synthetic import make_sample
    g = torch.Generator().manual_seed(seed)
    metas, _ = make_sample(seed, rays_per_frame=1, num_cams=CAMS)
    meta = metas[2]                                    # a frame with ego motion in can_bus
    feats = [torch.randn(1, CAMS, D, h, w, generator=g) for h, w in SHAPES]
    bev_queries = torch.randn(BEV * BEV, D, generator=g)
    bev_pos = torch.randn(1, D, BEV, BEV, generator=g)
    prev_bev = torch.randn(1, BEV * BEV, D, generator=g)
    return meta, feats, bev_queries, bev_pos, prev_bev


def perturb(model, seed=5):
    """init_weights leaves zero weights / on-pixel-centre offsets; move every parameter a little so
    that all code paths carry signal and no sample sits on a bilinear kink."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(torch.randn(p.shape, generator=g) * 0.05)


def main():
    mods = R.reference_modules()
    torch.manual_seed(0); np.random.seed(0)
    ref = R.build_from_cfg(transformer_cfg(), R.TRANSFORMER)
    ref.init_weights()
    perturb(ref)
    ref.eval()
    meta, feats, bev_queries, bev_pos, prev_bev = inputs()
    bev_queries.requires_grad_(True)
    kw = dict(grid_length=(102.4 / BEV, 102.4 / BEV), bev_pos=bev_pos, img_metas=[meta])
    out0 = ref.get_bev_features(feats, bev_queries, BEV, BEV, prev_bev=None, **kw)
    out1 = ref.get_bev_features(feats, bev_queries, BEV, BEV, prev_bev=prev_bev, **kw)
    w = torch.randn(out1.shape, generator=torch.Generator().manual_seed(9))
    names = ["encoder.layers.1.attentions.1.deformable_attention.sampling_offsets.weight",
             "encoder.layers.0.attentions.0.value_proj.weight"]
    params = dict(ref.named_parameters())
    grads = torch.autograd.grad((out1 * w).sum(), [bev_queries] + [params[n] for n in names])
    sd = {"sd/" + k: v.detach().numpy() for k, v in ref.state_dict().items()}
    np.savez_compressed(
        HERE / "transformer_encoder_small.npz", **sd,
        feats0=feats[0].numpy(), feats1=feats[1].numpy(), bev_queries=bev_queries.detach().numpy(),
        bev_pos=bev_pos.numpy(), prev_bev=prev_bev.numpy(), can_bus=np.asarray(meta["can_bus"]),
        lidar2global_rotation=np.asarray(meta["lidar2global_rotation"]),
        lidar2img=np.stack(meta["lidar2img"]), img_shape=np.asarray(meta["img_shape"]),
        out_no_prev=out0.detach().numpy(), out_prev=out1.detach().numpy(), grad_weight=w.numpy(),
        grad_bev_queries=grads[0].numpy(), grad_param0=grads[1].numpy(), grad_param1=grads[2].numpy(),
        grad_param_names=np.array(names), cfg_json=np.array(json.dumps(transformer_cfg())))
    print("wrote transformer_encoder_small.npz", out0.shape, float(out0.abs().mean()), float(out1.abs().mean()),
          "keys", len(sd))


if __name__ == "__main__":
    main()
