"""In-repo equivalents of the released pre-training configs named in BASELINE.json.  They are
generated from a handful of switches instead of being copies of the reference files; the released
files themselves (projects/configs/vidar_pretrain/**) also load unchanged through
vidar_amd.plugin.Config.fromfile (tests/test_plugin_cpu.py checks both give the same model).

    cfg = get_config("vidar_1_8_nusc_1future");  model = plugin.build_detector(cfg["model"])
"""
from __future__ import annotations

PC_RANGE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
DIM, FFN_DIM, LEVELS = 256, 512, 4

VARIANTS = {
    # name: future decoder frames/layers, LatentRendering step, head slices, cameras, GT future frames
    "vidar_1_8_nusc_1future": dict(future=0, dec_layers=1, lr_step=1.0, hist_pred=3, fut_pred=1,
                                   slice_w=(0.2, 0.4, 0.6, 1.0, 1.2), cams=6, future_frames=2,
                                   backward_prev=1, drop_prev=(0.1, 3), img_hw=(928, 1600),
                                   data=dict(rand_frame_interval=(-1, 1), load_frame_interval=8, voxel_size=0.5,
                                             future_test=0)),
    "vidar_1_8_nusc_3future": dict(future=3, dec_layers=3, lr_step=0.5, hist_pred=3, fut_pred=1,
                                   slice_w=(0.2, 0.4, 0.6, 1.0, 1.2), cams=6, future_frames=4,
                                   backward_prev=0, drop_prev=(0.0, None), img_hw=(928, 1600),
                                   data=dict(rand_frame_interval=(-1, 1, 2), load_frame_interval=8, voxel_size=1.0,
                                             future_test=6)),
    # nusc_1_8_subset/mem_efficient_vidar_1_8_nusc_3future.py (README.md:143-148: ~34 GB instead of ~63 GB on an A100):
    # only the last future frame is supervised, the head predicts the current frame only, LatentRendering step 1.0
    "mem_efficient_vidar_1_8_nusc_3future": dict(future=3, dec_layers=3, lr_step=1.0, hist_pred=0, fut_pred=0,
                                                 slice_w=(1.0,), cams=6, future_frames=4,
                                                 backward_prev=0, drop_prev=(0.0, None), img_hw=(928, 1600),
                                                 supervise_all_future=False,
                                                 data=dict(rand_frame_interval=(-1, 1), load_frame_interval=8,
                                                           voxel_size=1.0, future_test=6)),
    "vidar_full_nusc_1future": dict(future=0, dec_layers=1, lr_step=0.5, hist_pred=3, fut_pred=1,
                                    slice_w=(0.2, 0.4, 0.6, 1.0, 1.2), cams=6, future_frames=1,
                                    backward_prev=1, drop_prev=(0.1, 3), img_hw=(928, 1600),
                                    data=dict(rand_frame_interval=(-1, 1, 2), load_frame_interval=1, voxel_size=0.5,
                                              future_test=0)),
    "vidar_OpenScene_mini_full_3future": dict(future=3, dec_layers=3, lr_step=0.5, hist_pred=0,
                                              fut_pred=0, slice_w=(1.0,), cams=8, future_frames=3,
                                              backward_prev=0, drop_prev=(0.0, None),
                                              img_hw=(736, 1280),
                                              data=dict(rand_frame_interval=(1,), load_frame_interval=1,
                                                        voxel_size=1.0, future_test=6, img_scale=2.0 / 3.0)),
}


def _latent_render(step):
    return dict(embed_dims=DIM, pred_height=16, num_pred_fcs=0, grid_step=step, grid_num=256,
                reduction=16, act="sigmoid")


def model_config(name, bev_h=200, bev_w=200, with_backbone=False):
    v = VARIANTS[name]
    lr = _latent_render(v["lr_step"])
    pos = dict(type="LearnedPositionalEncoding", num_feats=DIM // 2, row_num_embed=bev_h,
               col_num_embed=bev_w)
    pred_attn = dict(type="PredictionMSDeformableAttention", embed_dims=DIM, num_levels=1)
    dec_layer = dict(type="PredictionTransformerLayer", attn_cfgs=[dict(pred_attn), dict(pred_attn)],
                     feedforward_channels=FFN_DIM, ffn_dropout=0.1,
                     operation_order=("self_attn", "norm", "cross_attn", "norm", "ffn", "norm"))
    decoder = dict(type="PredictionDecoder", num_layers=v["dec_layers"], return_intermediate=True,
                   transformerlayers=dec_layer)
    if v["future"] > 0:       # 3future configs carry a latent_render entry that keep_idx=((),) deletes
        dec_layer["latent_render"] = dict(lr)
        dec_layer["operation_order"] = ("self_attn", "norm", "cross_attn", "norm", "latent_render",
                                        "ffn", "norm")
        decoder["keep_idx"] = ((),)
    n_fut = v["future"]
    head = dict(type="ViDARHeadV1", history_queue_length=4, pred_history_frame_num=v["hist_pred"],
                pred_future_frame_num=v["fut_pred"], per_frame_loss_weight=v["slice_w"],
                ray_grid_num=512, ray_grid_step=1.0, use_ce_loss=True, use_dist_loss=False,
                use_dense_loss=True, num_pred_fcs=0, num_pred_height=16, can_bus_norm=True,
                can_bus_dims=(0, 1, 2, 17), bev_h=bev_h, bev_w=bev_w, pc_range=PC_RANGE,
                loss_weight=[[1]] + ([[1]] * n_fut if n_fut else [[0]]),
                positional_encoding=dict(pos),
                transformer=dict(type="PredictionTransformer", embed_dims=DIM, decoder=decoder))
    enc_layer = dict(
        type="BEVFormerLayerV2",
        attn_cfgs=[dict(type="TemporalSelfAttention", embed_dims=DIM, num_levels=1),
                   dict(type="SpatialCrossAttention", pc_range=PC_RANGE, num_cams=v["cams"],
                        deformable_attention=dict(type="MSDeformableAttention3D", embed_dims=DIM,
                                                  num_points=8, num_levels=LEVELS),
                        embed_dims=DIM)],
        latent_render=dict(lr), feedforward_channels=FFN_DIM, ffn_dropout=0.1,
        operation_order=("self_attn", "norm", "cross_attn", "norm", "latent_render", "ffn", "norm"))
    bev_head = dict(type="ViDARBEVFormerHead", bev_h=bev_h, bev_w=bev_w, num_query=900,
                    num_classes=10, in_channels=DIM, with_box_refine=True, as_two_stage=False,
                    transformer=dict(type="PerceptionTransformer", rotate_prev_bev=True,
                                     use_shift=True, use_can_bus=True, embed_dims=DIM,
                                     num_cams=v["cams"], rotate_center=[bev_w // 2, bev_h // 2],
                                     encoder=dict(type="CustomBEVFormerEncoder", keep_idx=(2,),
                                                  latent_rendering_lid=(2,), num_layers=6,
                                                  pc_range=PC_RANGE, num_points_in_pillar=4,
                                                  return_intermediate=False,
                                                  transformerlayers=enc_layer)),
                    bbox_coder=dict(type="NMSFreeCoder", pc_range=PC_RANGE),
                    positional_encoding=dict(pos))
    model = dict(type="ViDAR", use_grid_mask=True, video_test_mode=True, point_cloud_range=PC_RANGE,
                 bev_h=bev_h, bev_w=bev_w, future_pred_frame_num=n_fut, test_future_frame_num=n_fut * 2,
                 supervise_all_future=v.get("supervise_all_future", True), random_drop_prev_rate=v["drop_prev"][0],
                 random_drop_prev_end_idx=v["drop_prev"][1],
                 backwarded_prev_frame_num=v["backward_prev"], future_pred_head=head,
                 pts_bbox_head=bev_head)
    if with_backbone:
        model["img_backbone"] = dict(type="ResNet", depth=101, num_stages=4, out_indices=(1, 2, 3),
                                     frozen_stages=1, norm_cfg=dict(type="BN2d", requires_grad=False),
                                     norm_eval=True, style="caffe",
                                     dcn=dict(type="DCNv2", deform_groups=1, fallback_on_stride=False),
                                     stage_with_dcn=(False, False, True, True))
        model["img_neck"] = dict(type="FPN", in_channels=[512, 1024, 2048], out_channels=DIM,
                                 start_level=0, add_extra_convs="on_output", num_outs=4,
                                 relu_before_extra_convs=True)
    return model


def get_config(name, bev_h=200, bev_w=200, with_backbone=False):
    v = VARIANTS[name]
    h, w = v["img_hw"]
    shapes = [((h // s) + (1 if h % s else 0), (w // s) + (1 if w % s else 0)) for s in (8, 16, 32, 64)]
    return dict(name=name, model=model_config(name, bev_h, bev_w, with_backbone), queue_length=4,
                future_frames=v["future_frames"], num_cams=v["cams"], img_hw=v["img_hw"],
                fpn_shapes=shapes, optimizer=dict(lr=2e-4, weight_decay=0.01), grad_clip=35.0,
                data=dict(samples_per_gpu=1, workers_per_gpu=4, **v["data"]))


def dataset_kwargs(meta, test_mode=False, file_cfg=None, test_stride=None):
    """-> keyword arguments of vidar_amd.data.ViDARSequenceDataset for a named recipe: the temporal augmentation
    (`rand_frame_interval`), the 1/8 subset stride (`load_frame_interval`), the GT voxel size of the point sampler and
    the future length of the split -- values of the released configs (vidar_1_8_nusc_1future.py:14-24, :294,
    :338-342; vidar_1_8_nusc_3future.py:14-28, :301; vidar_full_nusc_1future.py:14-24; OpenScene/
    vidar_OpenScene_mini_full_3future.py:14-28, :292).  `file_cfg`: a loaded released config file, whose own
    `data.train` / `data.test` entries and point-sampler voxel size take precedence.
    The subset stride belongs to the TRAINING split only: the released configs pass `load_frame_interval` to
    `data.train` (vidar_1_8_nusc_1future.py:338-342) and not to `data.val` / `data.test` (:345-368), where the dataset's
    default None applies (nuscenes_vidar_dataset_template.py:66-68) -- evaluation runs over the whole val split.  A
    stride in test mode is an explicit choice: `test_stride` here (tools/test.py --eval-stride), or a `data.test` entry
    of the loaded config file that names it."""
    d = dict(meta["data"])
    kw = dict(queue_length=meta["queue_length"], future_length=d["future_test"] if test_mode else meta["future_frames"],
              rand_frame_interval=tuple(d["rand_frame_interval"]),
              load_frame_interval=test_stride if test_mode else d["load_frame_interval"],
              voxel_size=(d["voxel_size"],) * 3, test_mode=test_mode,
              dataset="nuplan" if "OpenScene" in meta["name"] else "nuscenes")
    if d.get("img_scale"):
        kw["img_scale"] = d["img_scale"]
    if file_cfg is not None:
        split = file_cfg.data.test if test_mode else file_cfg.data.train
        for key in ("queue_length", "future_length", "load_frame_interval"):
            if key in split:
                kw[key] = split[key]
        if "rand_frame_interval" in split and not test_mode:
            kw["rand_frame_interval"] = tuple(split["rand_frame_interval"])
        for step in split.get("pipeline", []):
            if step.get("type") == "CustomVoxelBasedPointSampler":
                kw["voxel_size"] = tuple(step["cur_sweep_cfg"]["voxel_size"])
    if test_mode:
        kw["rand_frame_interval"] = (1,)
    return kw
