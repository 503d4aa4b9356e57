"""GPU: SURVEY 8(f4) end to end -- a nuScenes-layout dataset ON DISK (info pkl + camera images + lidar `.bin` key frames
and sweeps, written by the test) -> vidar_amd/data/reader.py (multi-sweep merge, ego mask, voxel subsample, image
normalisation + padding, `union2one`) -> data/loader.py (sampler + collate) -> the HIP training step WITH the image
backbone, against the same step routed to the CPU oracle: loss terms and gradients.  Also the checkpoint bridge: the
model is saved in the mmcv `.pth` layout, loaded into a fresh model and must reproduce the losses."""
import numpy as np
import pytest
import torch

from test_reader_cpu import _mini_nuscenes

pytestmark = pytest.mark.gpu


def _model(num_cams):
    from vidar_amd import train as T
    from vidar_amd.configs import get_config
    cfg = get_config("vidar_1_8_nusc_1future", bev_h=24, bev_w=24, with_backbone=True)
    cfg["model"]["pts_bbox_head"]["transformer"]["num_cams"] = num_cams
    cfg["model"]["pts_bbox_head"]["transformer"]["encoder"]["transformerlayers"]["attn_cfgs"][1]["num_cams"] = num_cams
    cfg["model"]["use_grid_mask"] = False                  # (draws from numpy's generator; pinned in test_grid_mask_cpu)
    model = T.build_model(cfg)
    for m in model.modules():
        if hasattr(m, "random_drop_prev_rate"):
            m.random_drop_prev_rate = 0.0
    g = torch.Generator().manual_seed(11)                  # off the bilinear kinks (see test_step_gpu)
    for n, p in model.named_parameters():
        if n.endswith("sampling_offsets.bias"):
            p.data += torch.randn(p.shape, generator=g) * 0.3
    # the reader hands out caffe-style images (BGR minus the channel means, +-128): tame the random-init backbone so that
    # 101 layers of fp32 stay comparable between MIOpen and the host convolutions
    for m in model.img_backbone.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.fill_(0.5)
    model.train()
    model.apply(lambda m: setattr(m, "p", 0.0) if isinstance(m, torch.nn.Dropout) else None)
    noise = -torch.empty(20000, 512).exponential_(generator=torch.Generator().manual_seed(3)).log()
    model.future_pred_head.gumbel_noise_fn = lambda R, K: noise[:R].to(next(model.parameters()).device)
    return model


def test_disk_dataset_feeds_the_hip_step_and_matches_the_oracle_step(tmp_path):
    from oracle import cpu_ops
    from vidar_amd.checkpoint import load_checkpoint, save_checkpoint
    from vidar_amd.data.loader import build_dataloader
    from vidar_amd.data.reader import ViDARSequenceDataset
    ann = _mini_nuscenes(tmp_path, n_frames=9, cams=2)
    ds = ViDARSequenceDataset(ann, queue_length=4, future_length=1)
    np.random.seed(0); torch.manual_seed(0)
    dl = build_dataloader(ds, samples_per_gpu=1, workers_per_gpu=0, num_replicas=1, rank=0, seed=0)
    batch = next(iter(dl))
    assert batch["img"].shape == (1, 5, 2, 3, 64, 64) and batch["gt_points"][0].shape[1] == 5
    torch.manual_seed(1)
    model = _model(num_cams=2)
    params = [p for p in model.parameters() if p.requires_grad]
    with cpu_ops.patched():
        ref = model(return_loss=True, **batch)
        ref_grads = torch.autograd.grad(sum(ref.values()), params)
    assert len(ref) == 10 and all(torch.isfinite(v) for v in ref.values())
    model.cuda()
    dev = dict(img=batch["img"].cuda(), img_metas=batch["img_metas"], gt_points=[g.cuda() for g in batch["gt_points"]])
    out = model(return_loss=True, **dev)
    for k in ref:
        np.testing.assert_allclose(float(out[k]), float(ref[k]), rtol=5e-3, atol=1e-5, err_msg=k)
    grads = torch.autograd.grad(sum(out.values()), params)
    num = sum(float(((a.cpu() - b) ** 2).sum()) for a, b in zip(grads, ref_grads))
    den = sum(float((b ** 2).sum()) for b in ref_grads)
    assert (num / den) ** 0.5 < 1e-2, f"relative gradient error {(num / den) ** 0.5:.2e}"
    # checkpoint bridge: mmcv layout out, strict key-for-key load into a fresh model, same losses
    path = save_checkpoint(model, tmp_path / "epoch_1.pth", meta=dict(epoch=1, iter=7))
    torch.manual_seed(5)
    fresh = _model(num_cams=2)
    _, missing, unexpected = load_checkpoint(fresh, path, strict=True)
    assert not missing and not unexpected
    fresh.cuda()
    again = fresh(return_loss=True, **dev)
    for k in ref:
        assert float(again[k]) == pytest.approx(float(out[k]), rel=1e-6, abs=1e-7), k
